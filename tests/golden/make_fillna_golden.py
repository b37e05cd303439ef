#!/usr/bin/env python
"""Golden fixtures for the group-wise fillna(col, reverse) (VERDICT r05 "small siblings": expr/fexpr_fillna.cc:85-117;
the sibling of cummin / cummax on the same segmented scan): produced by RUNNING THE UNMODIFIED REFERENCE in the dev
container (oracle/_ref, built by oracle/build_ref.sh).

Writes tests/golden/fillna_cases.npz.  Per case: in.k (key column), in.v (value column, reference NA sentinels), ri / off
(the grouping), fill / fill.rev (one value per row, grouped order, as the reference returned them) and the stypes.

Inputs: (a) the reference's own vectors as data -- tests/dt/test-fillna.py:93-97 (void is not a fixed-width stype: taken as
an all-NA int32 column), :106-110, :120-128, :142-150 evaluated as one group; :194-203 (grouped; the string key becomes its
rank) and :225-235 (the by-column itself); (b) seeded random columns of every fixed-width stype x NA share x group shape,
incl. groups that are NA throughout and groups whose first / last rows are NA.
"""
import json
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


def to_np(lst, st):
    return np.array([NA[st] if x is None else x for x in lst], dtype=NP[st])


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
    R = DT[:, [dt.fillna(f.v), dt.fillna(f.v, reverse=True)], by(f.k)]
    assert ST[R.stypes[1].name] == vst and ST[R.stypes[2].name] == vst, R.stypes
    cases[name + "/fill"] = to_np(R[:, 1].to_list()[0], vst)
    cases[name + "/fill.rev"] = to_np(R[:, 2].to_list()[0], vst)
    manifest.append({"name": name, "n": n, "key_stype": kst, "val_stype": vst, "note": note})


def one_group(name, lst, vst, note):
    v = to_np(lst, vst)
    add(name, np.zeros(len(v), np.int32), 4, v, vst, note)


# (a) the reference's own vectors
one_group("ref_void_as_int32", [None, None, None], 4, "dt/test-fillna.py:93-97")
one_group("ref_trivial", [1, None], 4, "dt/test-fillna.py:106-110")
one_group("ref_bool", [None, False, None, True, False, True], 1, "dt/test-fillna.py:120-128")
one_group("ref_int", [None, None, 3, None, 4, -100, None], 4, "dt/test-fillna.py:142-150")
add("ref_grouped", to_np([0, 0, 0, 1, 1, 2, 2, 2], 4), 4, to_np([15, None, 136, 93, 743, None, None, 91], 4), 4,
    "dt/test-fillna.py:194-203 (keys a / b / c as their ranks)")
kv = to_np([2, 1, None, 1, 2], 4)
add("ref_grouped_column", kv, 4, kv, 4, "dt/test-fillna.py:225-235 (the by-column filled inside its own groups)")

# (b) seeded random columns
rng = np.random.default_rng(29)
for vst in (1, 2, 3, 4, 5, 6, 7):
    for shape, n, ng in (("few", 3000, 7), ("many", 3000, 900), ("single", 2500, 1)):
        for nafrac in (0.1, 0.6):
            k = rng.integers(0, ng, n).astype(np.int32)
            k[rng.random(n) < 0.03] = NA[4]
            if vst == 1:
                v = rng.integers(0, 2, n).astype(np.int8)
            elif vst in (6, 7):
                v = (rng.standard_normal(n) * 100).astype(NP[vst])
                v[rng.random(n) < 0.01] = np.inf
                v[rng.random(n) < 0.01] = -0.0
            else:
                hi = {2: 100, 3: 20000, 4: 2**30, 5: 2**60}[vst]
                v = rng.integers(-hi, hi, n).astype(NP[vst])
            v[rng.random(n) < nafrac] = NA[vst]
            if shape == "few":
                v[k == 3] = NA[vst]                      # one group NA throughout
            add("rand_st%d_%s_na%d" % (vst, shape, int(nafrac * 100)), k, 4, v, vst, "seeded random")

cases["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fillna_cases.npz")
np.savez_compressed(out, **cases)
print("wrote", out, len(manifest), "cases, datatable", dt.__version__)
