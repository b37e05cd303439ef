#!/usr/bin/env python
"""Generate golden fixtures for the groupby hot path by RUNNING THE UNMODIFIED
REFERENCE (h2oai/datatable) in this container.

Usage (dev container only; the GPU box has neither /root/reference nor a build):

    # one-off: build the reference with its own build backend in a scratch dir
    #   cp -r /root/reference /tmp/dt_oracle && chmod -R u+w /tmp/dt_oracle
    #   (cd /tmp/dt_oracle && python ci/ext.py build)
    DT_REFERENCE_SRC=/tmp/dt_oracle/src python tests/golden/make_golden.py

Writes tests/golden/groupby_cases.npz (+ manifest json inside the npz).
Every case stores the inputs (numpy arrays with datatable's NA sentinels:
INT*_MIN / NaN; bool as int8 with -128 = NA) and what the reference returned:

  ri       RowIndex permutation of  DT[:, f.__rowid, by(keys)]
  off      group offsets (cumsum of DT[:, count(), by(keys)])
  gk<i>    key value of each group (by-columns of the result)
  <op>.<col>  aggregates DT[:, op(f.col), by(keys)]  for op in sum/mean/min/max/count

Inputs cover (a) the reference's own golden vectors, ported as data
(tests/ijby/test-sort.py:132-250,468-512; tests/test-groups.py:70-130,318-323;
tests/test-reduce.py NA/inf cases), and (b) seeded random
cases per key stype / value stype / NA pattern / multi-key.
"""
import json
import os
import random
import sys

import numpy as np

SRC = os.environ.get("DT_REFERENCE_SRC", "/tmp/dt_oracle/src")
sys.path.insert(0, SRC)
import datatable as dt  # noqa: E402
from datatable import f, by, count, sum as dtsum, mean as dtmean, min as dtmin, max as dtmax  # noqa: E402

dt.options.progress.enabled = False

ST = {"bool8": 1, "int8": 2, "int16": 3, "int32": 4, "int64": 5, "float32": 6, "float64": 7}
NP = {1: np.int8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float32, 7: np.float64}
NA = {1: -128, 2: -128, 3: -2**15, 4: -2**31, 5: -2**63, 6: np.nan, 7: np.nan}
DT_ST = {1: dt.bool8, 2: dt.int8, 3: dt.int16, 4: dt.int32, 5: dt.int64, 6: dt.float32, 7: dt.float64}


def to_np(lst, st):
    """python list with None -> numpy array with NA sentinel"""
    return np.array([NA[st] if x is None else x for x in lst], dtype=NP[st])


def to_list(arr, st):
    """numpy sentinel array -> python list with None (what dt.Frame ingests)"""
    if st in (6, 7):
        return [None if np.isnan(x) else float(x) for x in arr]
    if st == 1:
        return [None if x == -128 else bool(x) for x in arr]
    return [None if x == NA[st] else int(x) for x in arr]


def col_out(frame, j):
    st = ST[frame.stypes[j].name]
    return to_np(frame[:, j].to_list()[0], st), st


cases = {}
manifest = []


def add_case(name, keys, vals, note=""):
    """keys/vals: list of (numpy sentinel array, stype code)"""
    n = len(keys[0][0])
    knames = ["k%d" % i for i in range(len(keys))]
    vnames = ["v%d" % i for i in range(len(vals))]
    cols = {}
    for nm, (a, st) in zip(knames + vnames, keys + vals):
        cols[nm] = dt.Frame(to_list(a, st), stype=DT_ST[st])[0] if n else dt.Frame([[]], stype=DT_ST[st])[0]
    DT = dt.Frame([cols[nm] for nm in knames + vnames], names=knames + vnames)
    DT["rowid"] = dt.Frame(np.arange(n, dtype=np.int32))
    bys = by(*[f[nm] for nm in knames])
    assert DT.stypes[:len(keys) + len(vals)] == tuple(DT_ST[st] for _, st in keys + vals), (name, DT.stypes)
    rec = {"name": name, "n": n, "key_stypes": [st for _, st in keys], "val_stypes": [st for _, st in vals],
           "note": note, "aggs": []}
    for i, (a, st) in enumerate(keys):
        cases["%s/in.k%d" % (name, i)] = a
    for i, (a, st) in enumerate(vals):
        cases["%s/in.v%d" % (name, i)] = a
    r = DT[:, f.rowid, bys]
    cases["%s/ri" % name] = np.array(r[:, -1].to_list()[0], dtype=np.int32)
    c = DT[:, count(), bys]
    cnt = np.array(c[:, -1].to_list()[0], dtype=np.int64)
    cases["%s/off" % name] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    for i in range(len(keys)):
        g, gst = col_out(c, i)
        assert gst == keys[i][1]
        cases["%s/gk%d" % (name, i)] = g
    for i, (a, st) in enumerate(vals):
        for opn, fn in (("sum", dtsum), ("mean", dtmean), ("min", dtmin), ("max", dtmax), ("count", count)):
            res = DT[:, fn(f[vnames[i]]), bys]
            o, ost = col_out(res, len(keys))
            cases["%s/%s.v%d" % (name, opn, i)] = o
            rec["aggs"].append([opn, i, ost])
    manifest.append(rec)


rng = np.random.default_rng(20250928)


def with_na(a, st, frac):
    if frac <= 0:
        return a
    a = a.copy()
    m = rng.random(len(a)) < frac
    a[m] = NA[st]
    return a


def randvals(n, st, nafrac=0.0):
    if st in (6, 7):
        a = rng.standard_normal(n).astype(NP[st]) * 100
    elif st == 1:
        a = rng.integers(0, 2, n).astype(np.int8)
    else:
        lim = {2: 100, 3: 30000, 4: 10**9, 5: 10**15}[st]
        a = rng.integers(-lim, lim, n).astype(NP[st])
    return with_na(a, st, nafrac)


# ---- (a) reference golden vectors, as data ---------------------------------
# SURVEY Appendix B / tests/test-reduce.py style mixed NA/inf case
add_case("appendixB",
         [(to_np([3, None, 1, 3, 1, None, 2, 3], 4), 4)],
         [(to_np([1.5, 2.0, None, 4.0, None, 8.0, 16.0, float("inf")], 7), 7)],
         "SURVEY Appendix B")
add_case("appendixB_2key",
         [(to_np([2, 1, 2, 1, 2, 1], 4), 4), (to_np([5, 5, 4, 5, 4, 3], 4), 4)],
         [(to_np([0, 1, 2, 3, 4, 5], 7), 7)], "SURVEY Appendix B 2-key")
add_case("int64_sum_overflow", [(to_np([1, 1], 4), 4)], [(to_np([2**62, 2**62], 5), 5)],
         "sum wraps to INT64_MIN")
add_case("float_keys_zero_nan", [(to_np([0.0, -0.0, 0.0, None, None], 7), 7)],
         [(to_np([1, 2, 3, 4, 5], 4), 4)], "-0.0 != 0.0 as keys; NaN == NA")
# tests/ijby/test-sort.py:203-214 test_int32_upper_range
for b in (32767, 1000000):
    add_case("int32_upper_range_%d" % b, [(to_np([b, b - 1, b + 1] * 1000, 4), 4)], [], "test-sort.py:203")
# :216-228 test_int32_u2range
for dc in (32765, 32766, 32767, 32768, 65533, 65534, 65535, 65536):
    a = 100000
    add_case("int32_u2range_%d" % dc, [(to_np([a + dc, a + 10, a] * 1000, 4), 4)], [], "test-sort.py:216")
# :230-244 test_int32_unsigned
tbl = sum(([t] * 100 for t in [0x00000000, 0x00000001, 0x00007FFF, 0x00008000, 0x00008001, 0x0000FFFF,
                               0x7FFF0000, 0x7FFF0001, 0x7FFF7FFF, 0x7FFF8000, 0x7FFF8001, 0x7FFFFFFF]), [])
random.seed(7)
random.shuffle(tbl)
add_case("int32_unsigned", [(to_np(tbl, 4), 4)], [], "test-sort.py:230 (shuffled)")
# :246-250 test_int32_issue220
add_case("int32_issue220", [(to_np([None] + [1000000] * 200 + [None], 4), 4)], [], "test-sort.py:246")
# :170-179 test_int32_large_stable
for n in (30, 3000, 60000):
    add_case("int32_large_stable_%d" % n, [(to_np([None, 100, 100000] * (n // 3), 4), 4)], [], "test-sort.py:170")
# :156-167 test_int32_large (prime-cycle permutation), reduced to a smaller prime pair
p1, p2 = 50021, 100003
add_case("int32_prime_cycle", [(np.array([((n + 1) * p2) % p1 for n in range(p1)], np.int32), 4)], [],
         "test-sort.py:156 with p1=50021")
# :486-500 test_int64_large0
for n in (16, 100, 1000):
    a_, b_, c_, d_ = -6654966461866573261, -6655043958000990616, 5207085498673612884, 5206891724645893889
    add_case("int64_large0_%d" % n, [(to_np([c_, d_, a_, b_] * n, 5), 5)], [], "test-sort.py:486")
# :503-512 test_int64_large_random
for seed in (1, 2, 3):
    random.seed(seed)
    m = 2**63 - 1
    add_case("int64_large_random_%d" % seed, [(to_np([random.randint(-m, m) for _ in range(1000)], 5), 5)],
             [(randvals(1000, 7), 7)], "test-sort.py:503")
# tests/test-groups.py:318-323 test_groups_large1: 251 groups x 4000
n = 251 * 200
add_case("groups_large1", [(np.array([(i * 19) % 251 for i in range(n)], np.int32), 4)],
         [(randvals(n, 7), 7)], "test-groups.py:318 shape (251 groups x 200)")
# empty / single row / single group / all-NA key
add_case("empty", [(np.array([], np.int32), 4)], [(np.array([], np.float64), 7)], "0 rows")
add_case("one_row", [(to_np([7], 4), 4)], [(to_np([1.25], 7), 7)], "1 row")
add_case("const_key", [(to_np([5] * 1000, 5), 5)], [(randvals(1000, 7, 0.1), 7)], "single group")
add_case("all_na_key", [(to_np([None] * 100, 4), 4)], [(randvals(100, 7), 7)], "all-NA key")
add_case("all_na_value", [(rng.integers(0, 5, 200).astype(np.int32), 4)], [(to_np([None] * 200, 7), 7),
                                                                         (to_np([None] * 200, 4), 4)], "all-NA values")
add_case("inf_values", [(rng.integers(0, 3, 60).astype(np.int32), 4)],
         [(to_np([float("inf"), float("-inf"), 1.0] * 20, 7), 7)], "test-reduce.py:310-322 +-inf are valid")

# ---- (b) seeded random cases -----------------------------------------------
# config-1 shape (scaled): int32 key 100 groups, f64 values
n = 20000
add_case("c1_shape", [(rng.integers(0, 100, n).astype(np.int32), 4)], [(rng.standard_normal(n), 7)], "BASELINE C1 scaled")
# config-2 shape: int64 key, 4 f64 cols
n = 20000
add_case("c2_shape", [(rng.integers(0, 500, n).astype(np.int64), 5)],
         [(rng.standard_normal(n), 7) for _ in range(4)], "BASELINE C2 scaled")
# config-3 shape: high cardinality
n = 30000
add_case("c3_shape", [(rng.integers(0, 3000, n).astype(np.int64), 5)], [(rng.standard_normal(n), 7)], "BASELINE C3 scaled")
# config-4 shape: 2-key composite int32,int32
n = 20000
add_case("c4_shape", [(rng.integers(0, 60, n).astype(np.int32), 4), (rng.integers(0, 60, n).astype(np.int32), 4)],
         [(rng.standard_normal(n), 7)], "BASELINE C4 scaled")
# every key stype with NAs, every value stype with NAs
n = 4000
for kst in (1, 2, 3, 4, 5, 6, 7):
    if kst in (6, 7):
        k = with_na(rng.integers(-50, 50, n).astype(NP[kst]) / 4, kst, 0.05)
    elif kst == 1:
        k = with_na(rng.integers(0, 2, n).astype(np.int8), 1, 0.1)
    else:
        lim = {2: 100, 3: 3000, 4: 70000, 5: 10**12}[kst]
        k = with_na(rng.integers(-lim, lim, n).astype(NP[kst]), kst, 0.05)
        if kst == 5:
            k = with_na(rng.choice(rng.integers(-2**62, 2**62, 300), n).astype(np.int64), 5, 0.05)
    add_case("keytype_%d" % kst, [(k, kst)], [(randvals(n, vst, 0.1), vst) for vst in (1, 2, 3, 4, 5, 6, 7)],
             "key stype %d with NAs x all value stypes with NAs" % kst)
# 3-key mixed types (total significant bits > 64 forces the column-by-column path)
n = 10000
add_case("three_keys_wide",
         [(rng.choice(rng.integers(-2**62, 2**62, 20), n).astype(np.int64), 5),
          (with_na(rng.integers(0, 7, n).astype(np.int32), 4, 0.1), 4),
          (with_na(rng.integers(0, 4, n).astype(np.float64) / 2, 7, 0.1), 7)],
         [(randvals(n, 7, 0.05), 7)], ">64 significant bits over 3 keys")
add_case("two_keys_bool_int8",
         [(with_na(rng.integers(0, 2, n).astype(np.int8), 1, 0.2), 1), (with_na(rng.integers(-5, 5, n).astype(np.int8), 2, 0.2), 2)],
         [(randvals(n, 4, 0.05), 4)], "bool + int8 keys")
# full-range int32 keys with INT32_MAX/INT32_MIN+1 extremes
k = rng.integers(-2**31 + 1, 2**31, 5000).astype(np.int32)
k[:3] = [2**31 - 1, -2**31 + 1, 0]
add_case("int32_full_range", [(k, 4)], [(randvals(5000, 7), 7)], "32 significant bits")
# skewed: one giant group + many singletons
n = 20000
k = np.where(rng.random(n) < 0.7, 12345, rng.integers(0, 10**6, n)).astype(np.int64)
add_case("skewed", [(k, 5)], [(randvals(n, 7), 7)], "70% of rows in one group")

cases["__manifest__"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "groupby_cases.npz")
np.savez_compressed(out, **cases)
print("wrote", out, os.path.getsize(out) // 1024, "KiB;", len(manifest), "cases; reference", dt.__version__)
