#!/usr/bin/env python
"""Golden fixtures for the set functions and the natural join, produced by RUNNING
THE UNMODIFIED REFERENCE in the dev container (DT_REFERENCE_SRC=/tmp/dt_oracle/src).

Writes tests/golden/sets_join_cases.npz:
  set cases   <name>/src<i>            the source columns (numpy, reference NA sentinels)
              <name>/<op>              values of dt.<op>(*sources) for op in union/intersect/setdiff/symdiff
              (+ "unique" = dt.unique of the frame holding every source as a column, when equal lengths)
  join cases  <name>/x<k>, <name>/j<k> key columns of X and of the KEYED (sorted) J frame
              <name>/index             per X row the row of keyed J it joins to (INT32_MIN = no match),
                                       read off a row-number column carried through X[:, :, join(J)]
Inputs: the reference's own vectors (tests/test-sets.py:129-207, tests/test-join.py:33-45,62-68,169-182,
270-279) plus seeded random cases per stype combination."""
import json
import os
import sys

import numpy as np

SRC = os.environ.get("DT_REFERENCE_SRC", "/tmp/dt_oracle/src")
sys.path.insert(0, SRC)
import datatable as dt  # noqa: E402
from datatable import join  # noqa: E402

dt.options.progress.enabled = False
ST = {"bool8": 1, "int8": 2, "int16": 3, "int32": 4, "int64": 5, "float32": 6, "float64": 7}
NP = {1: np.int8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float32, 7: np.float64}
NA = {1: -128, 2: -128, 3: -2**15, 4: -2**31, 5: -2**63, 6: np.nan, 7: np.nan}
DT_ST = {1: dt.bool8, 2: dt.int8, 3: dt.int16, 4: dt.int32, 5: dt.int64, 6: dt.float32, 7: dt.float64}


def to_np(lst, st):
    return np.array([NA[st] if x is None else x for x in lst], dtype=NP[st])


def to_list(arr, st):
    if st in (6, 7):
        return [None if np.isnan(x) else float(x) for x in arr]
    if st == 1:
        return [None if x == -128 else bool(x) for x in arr]
    return [None if x == NA[st] else int(x) for x in arr]


def frame1(a, st, name="C0"):
    return dt.Frame({name: to_list(a, st)}, stype=DT_ST[st]) if len(a) else dt.Frame({name: []}, stype=DT_ST[st])


cases, manifest = {}, []
rng = np.random.default_rng(20250930)


def add_set_case(name, srcs, note=""):
    frames = [frame1(a, st) for a, st in srcs]
    rec = {"kind": "set", "name": name, "stypes": [st for _, st in srcs], "note": note, "outs": {}}
    for i, (a, st) in enumerate(srcs):
        cases["%s/src%d" % (name, i)] = a
    for op in ("union", "intersect", "setdiff", "symdiff"):
        R = getattr(dt, op)(*frames)
        st = ST[R.stypes[0].name] if R.ncols else srcs[0][1]
        cases["%s/%s" % (name, op)] = to_np(R.to_list()[0] if R.ncols else [], st)
        rec["outs"][op] = st
    if len({len(a) for a, _ in srcs}) == 1 and len({st for _, st in srcs}) == 1:
        F = dt.Frame([to_list(a, st) for a, st in srcs], stypes=[DT_ST[st] for _, st in srcs])
        R = dt.unique(F)
        st = ST[R.stypes[0].name]
        cases["%s/unique" % name] = to_np(R.to_list()[0], st)
        rec["outs"]["unique"] = st
    manifest.append(rec)


def add_join_case(name, xkeys, jkeys, note=""):
    """xkeys / jkeys: lists of (array, stype); J's key tuples must be unique"""
    nk = len(xkeys)
    names = ["K%d" % k for k in range(nk)]
    X = dt.Frame([to_list(a, st) for a, st in xkeys], names=names, stypes=[DT_ST[st] for _, st in xkeys])
    J = dt.Frame([to_list(a, st) for a, st in jkeys] + [list(range(len(jkeys[0][0])))], names=names + ["pos0"],
                 stypes=[DT_ST[st] for _, st in jkeys] + [dt.int32])
    J.key = names
    R = X[:, :, join(J)]
    pos0_sorted = J[:, "pos0"].to_list()[0]
    rank = {p: i for i, p in enumerate(pos0_sorted)}
    idx = np.array([-2**31 if p is None else rank[p] for p in R[:, "pos0"].to_list()[0]], np.int32)
    rec = {"kind": "join", "name": name, "xstypes": [st for _, st in xkeys], "jstypes": [st for _, st in jkeys], "note": note}
    for k in range(nk):
        cases["%s/x%d" % (name, k)] = xkeys[k][0]
        cases["%s/j%d" % (name, k)] = to_np(J[:, k].to_list()[0], jkeys[k][1])
    cases["%s/index" % name] = idx
    manifest.append(rec)


# ---- reference vectors -----------------------------------------------------------------------
a5, b4, c9 = [2, 5, 7, 2, 3], [3, 4, 2, 5], [0, 3, 2, 2, 2, 2, 2, 2, 0]
add_set_case("sets2", [(to_np(a5, 4), 4), (to_np(b4, 4), 4)], "test-sets.py:129,151,173,194")
add_set_case("sets3", [(to_np(c9, 4), 4), (to_np(a5, 4), 4), (to_np(b4, 4), 4)], "test-sets.py:137,159")
add_set_case("sets3b", [(to_np([2, 5, 7, 2, 3, 6, 0], 4), 4), (to_np(b4, 4), 4), (to_np(c9, 4), 4)], "test-sets.py:180,201")
add_set_case("sets1", [(to_np([3, None, 1, 3, None], 4), 4)], "single source = unique")
add_set_case("sets_na_float", [(to_np([0.0, -0.0, None, 1.5, np.inf], 7), 7), (to_np([None, 1.5, -np.inf, 0.0], 7), 7)],
             "-0.0 and 0.0 are different elements; NA is one element")
add_set_case("sets_empty_second", [(to_np([4, 4, 1], 4), 4), (to_np([], 4), 4)], "an empty source")
for idx, (st, k, nafrac) in enumerate([(4, 2, 0.1), (5, 3, 0.0), (2, 4, 0.2), (7, 2, 0.1), (6, 3, 0.1), (1, 2, 0.3), (3, 5, 0.05)]):
    srcs = []
    for _ in range(k):
        n = int(rng.integers(1, 400))
        if st in (6, 7):
            a = (rng.integers(-40, 40, n) * 0.5).astype(NP[st])
        elif st == 1:
            a = rng.integers(0, 2, n).astype(np.int8)
        else:
            a = rng.integers(-60, 60, n).astype(NP[st])
        a = a.copy()
        a[rng.random(n) < nafrac] = NA[st]
        srcs.append((a, st))
    add_set_case("setsrand%d_st%d_k%d" % (idx, st, k), srcs, "random")
n = 300
same = [(rng.integers(0, 50, n).astype(np.int32), 4) for _ in range(3)]
add_set_case("sets_equal_lengths", same, "also dt.unique of the 3-column frame")

add_join_case("join_simple", [(to_np([1, 3, 2, 1, 1, 2, 0], 4), 4)], [(to_np([0, 1, 2, 3], 4), 4)], "test-join.py:33")
add_join_case("join_missing_levels", [(to_np([1, 2, 3], 4), 4)], [(to_np([1, 2], 4), 4)], "test-join.py:62")
add_join_case("join_multi", [(to_np([1, 2, 3, 2, 3, 1, 2, 1, 1], 4), 4), (to_np([3, 4, 5, 4, 3, 3, 3, 4, 3], 4), 4)],
              [(to_np([1, 2, 1, 2], 4), 4), (to_np([3, 3, 4, 4], 4), 4)], "test-join.py:169")
add_join_case("join_issue1800", [(to_np([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5], 4), 4)], [(to_np([0, 1, 2, 3, 4], 4), 4)],
              "test-join.py:270")
add_join_case("join_na", [(to_np([None, 5, 7, None, 6], 4), 4)], [(to_np([7, None, 5], 4), 4)], "NA joins NA")
add_join_case("join_float_x_int_j", [(to_np([1.0, 1.5, None, 3.0, -2.0, 1e20, np.inf], 7), 7)], [(to_np([3, 1, -2, None], 4), 4)],
              "float X values that are not integers of J's type never match")
add_join_case("join_int64_x_int8_j", [(to_np([1, 300, -5, 2**40, None], 5), 5)], [(to_np([-5, 1, 44], 2), 2)],
              "X values outside J's integer range never match")
add_join_case("join_int_x_float32_j", [(to_np([1, 2, 16777217, None], 4), 4)], [(to_np([2.0, 1.0, 16777216.0], 6), 6)],
              "X converted to float32 before comparing")
add_join_case("join_float64_x_float32_j", [(to_np([0.1, 0.5, 2.5], 7), 7)], [(to_np([0.5, 0.1, 9.0], 6), 6)],
              "0.1 (f64) -> float32 equals 0.1f")
add_join_case("join_bool_int", [(to_np([True, False, None, True], 1), 1)], [(to_np([1, 0, 5], 4), 4)], "bool X, int J")
add_join_case("join_empty_j", [(to_np([1, 2], 4), 4)], [(to_np([], 4), 4)], "empty J: every row NA")
for idx, (xst, jst, nj) in enumerate([(4, 4, 50), (5, 4, 300), (4, 5, 1), (3, 7, 200), (7, 7, 100), (6, 5, 80), (2, 3, 30), (5, 5, 1000)]):
    nx = int(rng.integers(1, 2000))
    pool = rng.permutation(np.arange(-nj, nj))[:nj]
    jv = pool.astype(NP[jst])
    xv = rng.integers(-nj - 5, nj + 5, nx).astype(NP[xst]) if xst not in (6, 7) else \
        (rng.integers(-2 * nj, 2 * nj, nx) * 0.5).astype(NP[xst])
    if xst == 2:
        xv = rng.integers(-100, 100, nx).astype(np.int8)
    if jst == 3 or jst == 2:
        jv = np.unique(np.clip(pool, -100, 100)).astype(NP[jst])
    xv = xv.copy()
    xv[rng.random(nx) < 0.05] = NA[xst]
    add_join_case("joinrand%d_x%d_j%d" % (idx, xst, jst), [(xv, xst)], [(jv, jst)], "random")
# two-key random
nj, nx = 400, 3000
pairs = np.unique(np.stack([rng.integers(0, 30, nj), rng.integers(-10, 10, nj)], 1), axis=0)
rng.shuffle(pairs)
add_join_case("joinrand_two_keys", [(rng.integers(0, 32, nx).astype(np.int32), 4), (rng.integers(-11, 11, nx).astype(np.int64), 5)],
              [(pairs[:, 0].astype(np.int32), 4), (pairs[:, 1].astype(np.int64), 5)], "two keys")

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sets_join_cases.npz")
cases["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
np.savez_compressed(out, **cases)
print("wrote", out, len(manifest), "cases")
