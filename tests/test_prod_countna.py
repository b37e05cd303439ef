"""prod() and countna(col): FExpr_SumProd<false> (src/core/expr/fexpr_sumprod.cc:98-110 registers `sum` and `prod` from one
template; column/sumprod.h:34-59) and CountUnary_ColumnImpl<T, true> (column/count.h:35-58) -- dthip_reduce ops 11 / 12.

CPU: the oracle restatement against tests/golden/prod_countna_cases.npz (what the unmodified reference returned:
tests/golden/make_prod_countna_golden.py), bit for bit.  GPU: dthip_reduce through the C ABI against the same fixtures --
bit for bit as well: integer products wrap mod 2^64 (associative), float products are multiplied in the reference's own row
order, never re-associated (a product that overflows / underflows on the way depends on the order)."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, assert_same
from oracle import oracle as o

Z = np.load(os.path.join(ROOT, "tests", "golden", "prod_countna_cases.npz"))
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
    assert_same(o.reduce("prod", v, ri, off, stype=c["val_stype"]), Z[name + "/prod"], "prod")
    assert_same(o.reduce("countna", v, ri, off, stype=c["val_stype"]), Z[name + "/countna"], "countna")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_matches_reference(ctx, name):
    c = BY[name]
    v = Z[name + "/in.v"]
    ri, off = Z[name + "/ri"], Z[name + "/off"]
    assert_same(ctx.reduce("prod", v, ri, off, stype=c["val_stype"]), Z[name + "/prod"], "prod")
    assert_same(ctx.reduce("countna", v, ri, off, stype=c["val_stype"]), Z[name + "/countna"], "countna")
    # the column already in grouped order (no RowIndex), as the S-red seam hands a materialised column over
    vg = v[ri]
    assert_same(ctx.reduce("prod", vg, None, off, stype=c["val_stype"]), Z[name + "/prod"], "prod, identity RowIndex")
    assert_same(ctx.reduce("countna", vg, None, off, stype=c["val_stype"]), Z[name + "/countna"], "countna, identity RowIndex")


@pytest.mark.gpu
@pytest.mark.parametrize("stype,dtype", [(4, np.int32), (5, np.int64), (6, np.float32), (7, np.float64)])
@pytest.mark.parametrize("n,ng", [(1, 1), (2047, 1), (2049, 3), (300_000, 1), (300_000, 299_000), (1_000_003, 5_000)])
def test_gpu_against_oracle_shapes(ctx, stype, dtype, n, ng):
    """group shapes around the 2048-row tiles of the segmented scan, NA RowIndex entries (a view's NA rows count as NA)"""
    rng = np.random.default_rng(n * 7 + ng + stype)
    k = rng.integers(0, ng, n).astype(np.int32)
    if dtype in (np.float32, np.float64):
        v = rng.standard_normal(2 * n).astype(dtype) * 1.2
        v[rng.random(2 * n) < 0.05] = np.nan
    else:
        v = rng.integers(-4, 5, 2 * n).astype(dtype)
        v[rng.random(2 * n) < 0.05] = np.iinfo(dtype).min
    gri, off = o.group([k])
    view = rng.integers(0, 2 * n, n).astype(np.int32)          # the grouped view reads a LONGER stored column
    view[rng.random(n) < 0.02] = np.iinfo(np.int32).min         # NA indices
    ri = view[gri]
    for op in ("prod", "countna"):
        assert_same(ctx.reduce(op, v, ri, off, stype=stype), o.reduce(op, v, ri, off, stype=stype), op)


@pytest.mark.gpu
def test_out_stypes(ctx):
    from datatable_amd import _lib
    L = _lib.load()
    for st in range(1, 8):
        assert L.dthip_reduce_out_stype(_lib.PROD, st) == L.dthip_reduce_out_stype(_lib.SUM, st)
        assert L.dthip_reduce_out_stype(_lib.COUNTNA, st) == _lib.INT64
