"""fillna(col, reverse): FExpr_FillNA::fill_rowindex (src/core/expr/fexpr_fillna.cc:85-117) -- dthip_cumulate op 6
(DTHIP_FILLNA), the sibling of cummin / cummax on the same segmented scan: inside every group an NA takes the last valid
value before it (reverse: the next one), and stays NA while there is none.

CPU: the oracle restatement against tests/golden/fillna_cases.npz (what the unmodified reference returned:
tests/golden/make_fillna_golden.py), bit for bit.  GPU: dthip_cumulate through the C ABI against the same fixtures and
against the oracle on shapes around the scan's 2048-row tiles -- bit for bit (values are copied, never computed)."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, assert_same
from oracle import oracle as o

Z = np.load(os.path.join(ROOT, "tests", "golden", "fillna_cases.npz"))
MANIFEST = json.loads(bytes(Z["manifest"]).decode())
NAMES = [c["name"] for c in MANIFEST]
BY = {c["name"]: c for c in MANIFEST}


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    c = BY[name]
    k, v = Z[name + "/in.k"], Z[name + "/in.v"]
    ri, off = o.group([k], stypes=[c["key_stype"]])
    assert_same(ri, Z[name + "/ri"], "rowindex")
    assert_same(off, Z[name + "/off"], "offsets")
    assert_same(o.cumulate("fillna", v, ri, off, stype=c["val_stype"]), Z[name + "/fill"], "fillna")
    assert_same(o.cumulate("fillna", v, ri, off, reverse=True, stype=c["val_stype"]), Z[name + "/fill.rev"], "fillna, reverse")


def test_mirror_frame_statement():
    """the Python mirror of the reference's surface spells it like the reference (dt/test-fillna.py:194-203); CPU-only
    part: the expression's repr is the reference's"""
    import datatable_amd.frame as dta
    assert repr(dta.fillna(dta.f.A)) == "FExpr<fillna(f.A, reverse=False)>"
    assert repr(dta.fillna(dta.f.B, reverse=True)) == "FExpr<fillna(f.B, reverse=True)>"
    with pytest.raises(TypeError):
        dta.fillna()
    with pytest.raises(NotImplementedError):
        dta.fillna(dta.f.A, value=2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_matches_reference(ctx, name):
    c = BY[name]
    v = Z[name + "/in.v"]
    ri, off = Z[name + "/ri"], Z[name + "/off"]
    assert_same(ctx.cumulate("fillna", v, ri, off, stype=c["val_stype"]), Z[name + "/fill"], "fillna")
    assert_same(ctx.cumulate("fillna", v, ri, off, reverse=True, stype=c["val_stype"]), Z[name + "/fill.rev"], "fillna, reverse")
    vg = v[ri]                                   # the column already in grouped order (no RowIndex)
    assert_same(ctx.cumulate("fillna", vg, None, off, stype=c["val_stype"]), Z[name + "/fill"], "fillna, identity RowIndex")


@pytest.mark.gpu
@pytest.mark.parametrize("stype,dtype", [(2, np.int8), (3, np.int16), (4, np.int32), (5, np.int64), (6, np.float32), (7, np.float64)])
@pytest.mark.parametrize("n,ng", [(1, 1), (2047, 1), (2049, 3), (300_000, 1), (300_000, 299_000), (1_000_003, 5_000)])
def test_gpu_against_oracle_shapes(ctx, stype, dtype, n, ng):
    """group shapes around the 2048-row tiles of the segmented scan (runs of NAs that cross tiles and groups), NA RowIndex
    entries (a view's NA rows are NA)"""
    rng = np.random.default_rng(n * 11 + ng + stype)
    k = rng.integers(0, ng, n).astype(np.int32)
    if dtype in (np.float32, np.float64):
        v = rng.standard_normal(2 * n).astype(dtype)
        na = np.nan
    else:
        ii = np.iinfo(dtype)
        v = rng.integers(ii.min + 1, ii.max, 2 * n, dtype=np.int64).astype(dtype)
        na = ii.min
    v[rng.random(2 * n) < 0.7] = na                              # long runs of NAs
    gri, off = o.group([k])
    view = rng.integers(0, 2 * n, n).astype(np.int32)            # the grouped view reads a LONGER stored column
    view[rng.random(n) < 0.02] = np.iinfo(np.int32).min          # NA indices
    ri = view[gri]
    for rev in (False, True):
        assert_same(ctx.cumulate("fillna", v, ri, off, reverse=rev, stype=stype),
                    o.cumulate("fillna", v, ri, off, reverse=rev, stype=stype), "fillna reverse=%s" % rev)


@pytest.mark.gpu
def test_gpu_mirror_frame_statement(ctx):
    """DT[:, [fillna(f[:]), fillna(f[:], reverse=True)], by(f[-1])] of dt/test-fillna.py:194-203 on the mirror"""
    import datatable_amd.frame as dta
    DT = dta.Frame({"C0": np.array([15, -2**31, 136, 93, 743, -2**31, -2**31, 91], np.int32),
                    "C1": np.array([0, 0, 0, 1, 1, 2, 2, 2], np.int32)})
    R = DT[:, [dta.fillna(dta.f.C0), dta.fillna(dta.f.C0, reverse=True)], dta.by(dta.f.C1)]
    got = R.to_list()
    assert got[0] == [0, 0, 0, 1, 1, 2, 2, 2]
    assert got[1] == [15, 15, 136, 93, 743, None, None, 91]
    assert got[2] == [15, 136, 136, 93, 743, 91, 91, 91]


@pytest.mark.gpu
def test_out_stype_and_abi(ctx):
    from datatable_amd import _lib
    L = _lib.load()
    for st in range(1, 8):
        assert L.dthip_cumulate_out_stype(_lib.FILLNA, st) == st
