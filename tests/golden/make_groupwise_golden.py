#!/usr/bin/env python
"""Golden fixtures for the group-wise operators that share the hot path's Groupby (SURVEY.md 8(f)
row 2): sd / median / nunique, cov / corr, cumsum / cumprod / cummin / cummax (forward and reverse),
cumcount / ngroup -- produced by RUNNING THE UNMODIFIED REFERENCE in the dev container (same recipe
as make_golden.py: DT_REFERENCE_SRC=/tmp/dt_oracle/src).

Writes tests/golden/groupwise_cases.npz.  Every case stores the inputs (numpy arrays with the
reference's NA sentinels) and, per value column v<i>:

  ri, off                    grouping of the key columns (as in groupby_cases.npz)
  sd.v<i> median.v<i> nunique.v<i>         one value per group
  cumsum.v<i> cumprod.v<i> cummin.v<i> cummax.v<i>  (+ ".rev")   one value per row, grouped order
  cumcount ngroup (+ ".rev")
  cov.<i>.<j> corr.<i>.<j>   for the listed column pairs

Inputs: (a) the reference's own vectors, as data (tests/test-reduce.py:590-800,901-944,
tests/dt/test-cumsum.py:86-112, test-cumprod.py, test-cumminmax.py:97-205, test-cumcountngroup.py,
test-nunique.py); a frame evaluated without by() there is evaluated here with a constant key, which
is the same single group; (b) seeded random cases per value stype / NA pattern / group shape.
"""
import json
import math
import os
import sys

import numpy as np

SRC = os.environ.get("DT_REFERENCE_SRC", "/tmp/dt_oracle/src")
sys.path.insert(0, SRC)
import datatable as dt  # noqa: E402
from datatable import f, by  # noqa: E402

dt.options.progress.enabled = False

ST = {"bool8": 1, "int8": 2, "int16": 3, "int32": 4, "int64": 5, "float32": 6, "float64": 7}
NP = {1: np.int8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float32, 7: np.float64}
NA = {1: -128, 2: -128, 3: -2**15, 4: -2**31, 5: -2**63, 6: np.nan, 7: np.nan}
DT_ST = {1: dt.bool8, 2: dt.int8, 3: dt.int16, 4: dt.int32, 5: dt.int64, 6: dt.float32, 7: dt.float64}
inf = math.inf


def to_np(lst, st):
    return np.array([NA[st] if x is None else x for x in lst], dtype=NP[st])


def to_list(arr, st):
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


def add_case(name, keys, vals, pairs=(), note=""):
    n = len(keys[0][0])
    knames = ["k%d" % i for i in range(len(keys))]
    vnames = ["v%d" % i for i in range(len(vals))]
    cols = [dt.Frame(to_list(a, st), stype=DT_ST[st])[0] for (a, st) in keys + vals]
    DT = dt.Frame(cols, names=knames + vnames)
    DT["rowid"] = dt.Frame(np.arange(n, dtype=np.int32))
    bys = by(*[f[nm] for nm in knames])
    nk = len(keys)
    rec = {"name": name, "n": n, "key_stypes": [st for _, st in keys], "val_stypes": [st for _, st in vals],
           "note": note, "outs": {}, "pairs": [list(p) for p in pairs]}
    for i, (a, st) in enumerate(keys):
        cases["%s/in.k%d" % (name, i)] = a
    for i, (a, st) in enumerate(vals):
        cases["%s/in.v%d" % (name, i)] = a
    cases["%s/ri" % name] = np.array(DT[:, f.rowid, bys][:, -1].to_list()[0], dtype=np.int32)
    cnt = np.array(DT[:, dt.count(), bys][:, -1].to_list()[0], dtype=np.int64)
    cases["%s/off" % name] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)

    def put(key, frame):
        o, ost = col_out(frame, nk)
        cases["%s/%s" % (name, key)] = o
        rec["outs"][key] = ost

    for i, vn in enumerate(vnames):
        for opn, fn in (("sd", dt.sd), ("median", dt.median), ("nunique", dt.nunique)):
            put("%s.v%d" % (opn, i), DT[:, fn(f[vn]), bys])
        for opn, fn in (("cumsum", dt.cumsum), ("cumprod", dt.cumprod), ("cummin", dt.cummin), ("cummax", dt.cummax)):
            put("%s.v%d" % (opn, i), DT[:, fn(f[vn]), bys])
            put("%s.v%d.rev" % (opn, i), DT[:, fn(f[vn], reverse=True), bys])
    put("cumcount", DT[:, dt.cumcount(), bys])
    put("cumcount.rev", DT[:, dt.cumcount(reverse=True), bys])
    put("ngroup", DT[:, dt.ngroup(), bys])
    put("ngroup.rev", DT[:, dt.ngroup(reverse=True), bys])
    for (i, j) in pairs:
        put("cov.%d.%d" % (i, j), DT[:, dt.cov(f[vnames[i]], f[vnames[j]]), bys])
        put("corr.%d.%d" % (i, j), DT[:, dt.corr(f[vnames[i]], f[vnames[j]]), bys])
    manifest.append(rec)


rng = np.random.default_rng(20250929)


def with_na(a, st, frac):
    if frac <= 0:
        return a
    a = a.copy()
    a[rng.random(len(a)) < frac] = NA[st]
    return a


def randvals(n, st, nafrac=0.0, card=None):
    if st in (6, 7):
        a = (rng.integers(-card, card, n) * 0.25).astype(NP[st]) if card else rng.standard_normal(n).astype(NP[st]) * 100
    elif st == 1:
        a = rng.integers(0, 2, n).astype(np.int8)
    else:
        lim = card or {2: 100, 3: 30000, 4: 10**9, 5: 10**15}[st]
        a = rng.integers(-lim, lim, n).astype(NP[st])
    return with_na(a, st, nafrac)


def const_key(n):
    return (np.zeros(n, np.int32), 4)


# ---- (a) the reference's own vectors ---------------------------------------------------------
add_case("appendixB", [(to_np([3, None, 1, 3, 1, None, 2, 3], 4), 4)],
         [(to_np([1.5, 2.0, None, 4.0, None, 8.0, 16.0, inf], 7), 7), (to_np(list(range(8)), 4), 4)],
         pairs=[(0, 1), (1, 1)], note="SURVEY Appendix B frame")
add_case("median_bool_even", [const_key(4)], [(to_np([True, False, True, False], 1), 1)], note="test-reduce.py:592")
add_case("median_bool_odd", [const_key(3)], [(to_np([True, False, True], 1), 1)], note="test-reduce.py:600")
add_case("median_bygroup", [(to_np([1, 2, 1, 1, 2, 2], 4), 4)], [(to_np([0.1, 0.2, 0.5, 0.4, 0.3, 0], 7), 7)],
         note="test-reduce.py:608")
for st in (2, 3, 4, 5):
    add_case("median_int_even_st%d" % st, [const_key(10)], [(to_np([7, 11, -2, 3, 0, 12, 12, 3, 5, 91], st), st)],
             note="test-reduce.py:617")
    add_case("median_int_odd_st%d" % st, [const_key(11)], [(to_np([4, -5, 12, 11, 4, 7, 0, 23, 45, 8, 10], st), st)],
             note="test-reduce.py:627")
add_case("median_int8_no_overflow", [const_key(2)], [(to_np([111, 112], 2), 2)], note="test-reduce.py:636")
for st in (6, 7):
    add_case("median_float_st%d" % st, [const_key(5)], [(to_np([0.0, 5.5, 7.9, inf, -inf], st), st)],
             note="test-reduce.py:645")
add_case("median_all_nas", [const_key(8)], [(to_np([None] * 8, 7), 7)], note="test-reduce.py:653")
add_case("median_some_nas", [const_key(10)], [(to_np([None, 5, None, 12, None, -3, None, None, None, 4], 4), 4)],
         note="test-reduce.py:661")
add_case("median_grouped", [(to_np([0, 0, 0, 0, 1, 1, 1, 1, 1], 3), 3)],
         [(to_np([2, 6, 1, 0, -3, 4, None, None, -1], 4), 4)], note="test-reduce.py:669")
add_case("cov_simple", [const_key(5)], [(to_np(list(range(5)), 4), 4), (to_np(list(range(5, 0, -1)), 4), 4)],
         pairs=[(0, 0), (0, 1)], note="test-reduce.py:710,760,766")
add_case("cov_single_row", [const_key(1)], [(to_np([1], 4), 4), (to_np([2], 4), 4)], pairs=[(0, 1)],
         note="test-reduce.py:716,772")
add_case("cov_float32", [const_key(3)], [(to_np([1.0, 2.0, 3.0], 6), 6), (to_np([7.5, 7.0, 6.5], 6), 6)],
         pairs=[(0, 1)], note="test-reduce.py:729")
add_case("cov_bygroup", [(to_np([1, 2, 1, 2, 1, 2], 4), 4)], [(to_np([0, 5, 10, 20, 2, 8], 4), 4)], pairs=[(0, 0)],
         note="test-reduce.py:736")
add_case("corr_with_constant", [const_key(23)], [(to_np(list(range(23)), 4), 4), (to_np([2.5] * 23, 7), 7)],
         pairs=[(0, 1)], note="test-reduce.py:779")
add_case("corr_multiple", [const_key(4)],
         [(to_np([3, 5, 9, 1], 4), 4), (to_np([4, 7, 0, 0], 4), 4), (to_np([3, 2, 1, 0], 4), 4), (to_np([0, 1, 2, 3], 4), 4)],
         pairs=[(0, 0), (0, 1), (0, 2), (0, 3), (1, 3), (2, 3), (3, 3)], note="test-reduce.py:797")
add_case("sd_single_row", [const_key(1)], [(to_np([3], 4), 4), (to_np([None], 4), 4)], note="test-reduce.py:920")
add_case("sd_const_columns", [const_key(10)],
         [(to_np([1] * 10, 4), 4), (to_np([-1.1] * 10, 7), 7), (to_np([0] * 10, 4), 4), (to_np([4.3] * 10, 7), 7)],
         note="test-reduce.py:926")
add_case("sd_float_columns", [const_key(5)],
         [(to_np([1.5, 6.4, 0.0, None, 7.22], 7), 7), (to_np([2.0, -1.1, inf, 4.0, 3.2], 7), 7),
          (to_np([1.5, 9.9, None, None, None], 7), 7), (to_np([inf, -inf, None, 0.0, None], 7), 7)],
         note="test-reduce.py:932")
add_case("sd_void_per_group", [(to_np([1, 2, 1, 2, 2], 4), 4)], [(to_np([None] * 5, 4), 4)], note="test-reduce.py:907")
add_case("cumsum_small", [const_key(5)], [(to_np(list(range(5)), 4), 4), (to_np([-1, 1, None, 2, 5.5], 7), 7)],
         note="test-cumsum.py:86,93; test-cumprod.py:87")
add_case("cumsum_groupby", [(to_np([2, 1, 1, 1, 2], 4), 4)], [(to_np([1.5, -1.5, inf, 2, 3], 7), 7)],
         note="test-cumsum.py:100,107")
add_case("cum_grouped_column", [(to_np([2, 1, None, 1, 2], 4), 4)], [(to_np([2, 1, None, 1, 2], 4), 4)],
         note="test-cumsum.py:120; test-cumminmax.py:162")
add_case("cumminmax_bool", [const_key(6)], [(to_np([None, False, None, True, False, True], 1), 1)],
         note="test-cumminmax.py:97")
add_case("cumminmax_small", [const_key(5)], [(to_np(list(range(5)), 4), 4), (to_np([None, -1, None, 5.5, 3], 7), 7)],
         note="test-cumminmax.py:107")
add_case("cumminmax_groupby", [(to_np([2, 1, 1, 1, 2], 4), 4)], [(to_np([1.5, -1.5, inf, None, 3], 7), 7)],
         note="test-cumminmax.py:151")
add_case("cumminmax_groupby_reverse", [(to_np([0, 1, 0, 2, 1], 4), 4)], [(to_np([3, 14, 15, 92, 6], 4), 4)],
         note="test-cumminmax.py:194 (string key replaced by its rank)")
add_case("nunique_small", [(to_np([1, 1, 2, 2, 2, 3], 4), 4)],
         [(to_np([5, 5, None, 7, 7, None], 4), 4), (to_np([0.0, -0.0, 1.5, None, 1.5, inf], 7), 7)],
         note="-0.0 and 0.0 are one value for nunique (std::set ordering)")

# ---- (b) seeded random cases -----------------------------------------------------------------
for idx, (vst, nafrac, card) in enumerate([(1, 0.2, None), (2, 0.1, None), (3, 0.0, 50), (4, 0.3, 7), (5, 0.05, None),
                                           (6, 0.15, 12), (7, 0.1, None), (7, 0.0, 5), (5, 0.9, 3), (7, 1.0, None)]):
    n = int(rng.integers(200, 3000))
    ng = int(rng.integers(1, 60))
    keys = [(with_na(rng.integers(0, ng, n).astype(np.int32), 4, 0.05), 4)]
    vals = [(randvals(n, vst, nafrac, card), vst), (randvals(n, 7, 0.1), 7), (randvals(n, 4, 0.1, 1000), 4)]
    add_case("rand%d_st%d" % (idx, vst), keys, vals, pairs=[(0, 1), (1, 2), (0, 0)], note="random")
# many tiny groups, two keys, float32 pair
n = 2500
add_case("rand_two_keys", [(rng.integers(0, 40, n).astype(np.int32), 4), (with_na(rng.integers(-3, 3, n).astype(np.int64), 5, 0.1), 5)],
         [(randvals(n, 6, 0.1), 6), (randvals(n, 6, 0.1), 6), (randvals(n, 3, 0.0, 4), 3)], pairs=[(0, 1), (0, 2)],
         note="two keys, float32 cov")
# one huge group next to singletons
n = 4000
k = np.where(rng.random(n) < 0.9, 7, rng.integers(100, 2000, n)).astype(np.int32)
add_case("rand_skewed", [(k, 4)], [(randvals(n, 7, 0.05), 7), (randvals(n, 5, 0.05, 20), 5)], pairs=[(0, 1)],
         note="one dominant group")
# small-magnitude integers so cumprod stays in range part of the way, then wraps
n = 600
add_case("rand_cumprod_int", [(rng.integers(0, 5, n).astype(np.int32), 4)],
         [(with_na(rng.integers(-3, 4, n).astype(np.int32), 4, 0.1), 4), (with_na((rng.random(n) * 2).astype(np.float32), 6, 0.1), 6)],
         pairs=[], note="int64 cumprod wraps; float32 cumprod")

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "groupwise_cases.npz")
cases["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
np.savez_compressed(out, **cases)
print("wrote", out, len(manifest), "cases")
