#!/usr/bin/env python
"""Golden fixtures for prod() and countna(col) (small siblings of sum / count on the same kernels):
produced by RUNNING THE UNMODIFIED REFERENCE in the dev container (oracle/_ref, built by oracle/build_ref.sh).

Writes tests/golden/prod_countna_cases.npz.  Per case: in.k (key column), in.v (value column, reference NA sentinels),
ri / off (the grouping), prod / countna (one value per group as the reference returned them), and the stypes.

Inputs: (a) the reference's own vectors as data -- tests/test-reduce.py:833-895 (test_prod_simple, _bool, _grouped,
_chained_grouped), tests/dt/test-countna.py:49-66 + the srcs_bool / srcs_int / srcs_real lists of :30-46 evaluated as one
group; (b) seeded random columns of every fixed-width stype with NAs, incl. int64 products that wrap and float products
that overflow / underflow on the way (the result depends on the row order there).
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.environ.get("DT_REFERENCE_PKG", os.path.join(ROOT, "oracle", "_ref")))
import datatable as dt  # noqa: E402
from datatable import f, by  # noqa: E402

dt.options.progress.enabled = False
ST = {"bool8": 1, "int8": 2, "int16": 3, "int32": 4, "int64": 5, "float32": 6, "float64": 7}
NP = {1: np.int8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float32, 7: np.float64}
NA = {1: -128, 2: -128, 3: -2**15, 4: -2**31, 5: -2**63, 6: np.nan, 7: np.nan}
DT_ST = {1: dt.bool8, 2: dt.int8, 3: dt.int16, 4: dt.int32, 5: dt.int64, 6: dt.float32, 7: dt.float64}
inf, nan = math.inf, math.nan


def to_np(lst, st):
    return np.array([NA[st] if (x is None or (st < 6 and isinstance(x, float) and math.isnan(x))) else x for x in lst], dtype=NP[st])


def to_list(arr, st):
    if st in (6, 7):
        return [None if np.isnan(x) else float(x) for x in arr]
    if st == 1:
        return [None if x == -128 else bool(x) for x in arr]
    return [None if x == NA[st] else int(x) for x in arr]


cases, manifest = {}, []


def add(name, k, kst, v, vst, note=""):
    n = len(k)
    DT = dt.Frame([dt.Frame(to_list(k, kst), stype=DT_ST[kst])[0], dt.Frame(to_list(v, vst), stype=DT_ST[vst])[0]], names=["k", "v"])
    DT["rowid"] = dt.Frame(np.arange(n, dtype=np.int32))
    cases[name + "/in.k"], cases[name + "/in.v"] = k, v
    cases[name + "/ri"] = np.array(DT[:, f.rowid, by(f.k)][:, -1].to_list()[0], dtype=np.int32)
    cnt = np.array(DT[:, dt.count(), by(f.k)][:, -1].to_list()[0], dtype=np.int64)
    cases[name + "/off"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    R = DT[:, [dt.prod(f.v), dt.countna(f.v)], by(f.k)]
    pst = ST[R.stypes[1].name]
    cases[name + "/prod"] = to_np(R[:, 1].to_list()[0], pst)
    cases[name + "/countna"] = np.array(R[:, 2].to_list()[0], dtype=np.int64)
    assert R.stypes[2] == dt.int64
    manifest.append({"name": name, "n": n, "key_stype": kst, "val_stype": vst, "prod_stype": pst, "note": note})


def one_group(name, lst, vst, note):
    v = to_np(lst, vst)
    add(name, np.zeros(len(v), np.int32), 4, v, vst, note)


# (a) the reference's own vectors
one_group("ref_prod_simple", [1, 2, 3, 4], 4, "test-reduce.py:833-838")
one_group("ref_prod_bool", [True, False, True], 1, "test-reduce.py:855-860")
A = to_np([True, False, True, True], 1)
add("ref_prod_grouped_B", A, 1, to_np([None, None, None, 10], 4), 4, "test-reduce.py:872-878 column B")
add("ref_prod_grouped_C", A, 1, to_np([2, 3, 5, 0.1], 7), 7, "test-reduce.py:872-878 column C")
add("ref_prod_chained_grouped", to_np([None, -3, -3, None, 5], 4), 4, to_np([None, -3, -3, None, 5], 4), 4, "test-reduce.py:889-894 (inner prod)")
add("ref_countna2", to_np([1, 1, 1, 2, 2, 2], 4), 4, to_np([None, None, None, None, 3, 5], 4), 4, "dt/test-countna.py:56-60")
srcs_bool = [[False, True, False, False, True], [True, None, None, True, False], [True], [False], [None] * 10]
srcs_int = [[5, -3, 6, 3, 0], [None, -1, 0, 26, -3], [129, 38, 27, -127, 8], [385, None, None, -3, -89], [-192, 32769, 683, 94, 0],
            [None, -32788, -4, -44444, 5], [30, -284928, 59, 3, 2147483649], [2147483648, None, None, None, None], [-1, 1], [100], [0]]
srcs_real = [[9.5, 0.2, 5.4857301, -3.14159265358979], [1.1, 2.3e12, -.5, None, inf, 0.0], [3.5, 2.36, nan, 696.9, 4097],
             [3.1415926535897932], [nan]]
for i, s in enumerate(srcs_bool):
    one_group("ref_srcs_bool%d" % i, s, 1, "dt/test-countna.py:30-32,49-53")
for i, s in enumerate(srcs_int):
    big = any(x is not None and abs(x) >= 2**31 for x in s)
    one_group("ref_srcs_int%d" % i, s, 5 if big else 4, "dt/test-countna.py:33-43,49-53")
for i, s in enumerate(srcs_real):
    one_group("ref_srcs_real%d" % i, s, 7, "dt/test-countna.py:44-46,49-53")

# (b) seeded random columns
rng = np.random.default_rng(11)
for vst in (1, 2, 3, 4, 5, 6, 7):
    for shape, n, ng in (("few", 3000, 7), ("many", 3000, 900), ("single", 2500, 1)):
        k = rng.integers(0, ng, n).astype(np.int32)
        k[rng.random(n) < 0.03] = NA[4]
        if vst == 1:
            v = rng.integers(0, 2, n).astype(np.int8)
        elif vst in (6, 7):
            v = (rng.standard_normal(n) * (50.0 if shape == "single" else 1.5)).astype(NP[vst])      # "single": overflows on the way
            v[rng.random(n) < 0.01] = np.inf if vst == 7 else 0.0
        else:
            hi = {2: 100, 3: 20000, 4: 2**30, 5: 2**60}[vst]
            v = rng.integers(-hi, hi, n).astype(NP[vst])                                             # int64 products wrap
            if shape == "few":
                v = rng.integers(-3, 4, n).astype(NP[vst])
        napos = rng.random(n) < 0.08
        v[napos] = NA[vst]
        add("rand_st%d_%s" % (vst, shape), k, 4, v, vst, "seeded random")

cases["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prod_countna_cases.npz")
np.savez_compressed(out, **cases)
print("wrote", out, len(manifest), "cases, datatable", dt.__version__)
