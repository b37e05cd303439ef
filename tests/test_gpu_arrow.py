"""SURVEY 8(f) row 4, Arrow part: Arrow-layout columns (values buffer + validity bitmap) reach HBM in that layout and the
NA sentinels are written by a kernel on the device (dthip_from_arrow, csrc/arrow.hip) -- the reference's element-by-
element CPU materialisation (arrow_fw.cc:63-72 read by _materialize_fw) never runs.  Checked against (1) the numpy
restatement of that rule (oracle.arrow_to_sentinel), (2) the unmodified reference materialising the very same pyarrow
arrays (oracle/_ref), and (3) end to end through the reference-side binding: an Arrow-backed Frame is grouped on the GPU
while its columns stay virtual."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_same
from oracle import oracle as o

pytestmark = pytest.mark.gpu

STYPES = [(o.BOOL, None), (o.INT8, np.int8), (o.INT16, np.int16), (o.INT32, np.int32), (o.INT64, np.int64),
          (o.FLOAT32, np.float32), (o.FLOAT64, np.float64)]
# around the kernel's vector widths (2 / 4 / 8 / 16 rows per 16-byte vector), the validity byte and the block size
SIZES = [1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 255, 256, 257, 4095, 4097, 100_003, 1_000_001]


def make(rng, n, stype, npdt, null_frac):
    if stype == o.BOOL:
        values = np.packbits(rng.random(n) < 0.5, bitorder="little")
    elif np.dtype(npdt).kind == "f":
        values = rng.standard_normal(n).astype(npdt)
        if n > 4:
            values[1], values[2], values[3] = np.inf, -np.inf, -0.0
    else:
        ii = np.iinfo(npdt)
        values = rng.integers(ii.min, ii.max, n, dtype=npdt, endpoint=True)      # INT*_MIN among the VALID values too
    validity = None if null_frac is None else np.packbits(rng.random(n) >= null_frac, bitorder="little")
    return values, validity


@pytest.mark.parametrize("stype,npdt", STYPES)
def test_from_arrow_matches_the_materialised_column(ctx, stype, npdt):
    rng = np.random.default_rng(100 + stype)
    for n in SIZES:
        for nf in (None, 0.0, 0.3, 1.0):
            values, validity = make(rng, n, stype, npdt, nf)
            exp = o.arrow_to_sentinel(values, validity, n, stype)
            col = ctx.from_arrow(values, validity, n, stype)
            got = ctx.download(col, n)
            if exp.dtype.kind == "f":        # bit patterns: NaN payloads of VALID rows survive, invalid rows are the quiet NaN
                assert_same(got.view("u%d" % got.itemsize), exp.view("u%d" % exp.itemsize), "stype %d n %d nulls %s" % (stype, n, nf))
            else:
                assert_same(got, exp, "stype %d n %d nulls %s" % (stype, n, nf))
            # the same buffers already in HBM (DTHIP_DEVICE: nothing is copied)
            if n in (17, 4097, 100_003):
                dv = ctx.upload(values.view(np.int8) if values.dtype != np.int8 else values, o.INT8)
                db = ctx.upload(validity.view(np.int8), o.INT8) if validity is not None else None
                col2 = ctx.from_arrow(dv.ptr, db.ptr if db is not None else None, n, stype, device=True)
                assert_same(ctx.download(col2, n).view(np.uint8), got.view(np.uint8), "device mode")


def test_from_arrow_rejects_bad_arguments(ctx):
    v = np.zeros(8, np.int64)
    with pytest.raises(ValueError):
        ctx.from_arrow(v, None, -1, o.INT64)
    with pytest.raises(NotImplementedError):
        ctx.from_arrow(v, None, 8, 12)


def _ref():
    from oracle import ref
    dt = ref.load()
    if dt is None:
        pytest.fail("oracle/_ref (the reference build) did not travel to this machine")
    pa = pytest.importorskip("pyarrow")
    return dt, pa


def test_against_the_reference_materialising_the_same_arrow_arrays(ctx):
    """pyarrow arrays -> datatable.Frame (ArrowFw / ArrowBool columns) -> the reference materialises them
    (frame_column_data_r) -> compared with dthip_from_arrow on the SAME two Arrow buffers"""
    dt, pa = _ref()
    rng = np.random.default_rng(7)
    n = 300_001
    types = {"b": pa.bool_(), "i8": pa.int8(), "i16": pa.int16(), "i32": pa.int32(), "i64": pa.int64(), "f32": pa.float32(), "f64": pa.float64()}
    arrs = {}
    for nm, t in types.items():
        mask = rng.random(n) < 0.2
        if nm == "b":
            a = pa.array(rng.random(n) < 0.5, type=t, mask=mask)
        elif nm.startswith("f"):
            a = pa.array(rng.standard_normal(n).astype(t.to_pandas_dtype()), type=t, mask=mask)
        else:
            ii = np.iinfo(t.to_pandas_dtype())
            a = pa.array(rng.integers(ii.min + 1, ii.max, n).astype(t.to_pandas_dtype()), type=t, mask=mask)
        arrs[nm] = a
    arrs["nonull"] = pa.array(rng.integers(-9, 9, n).astype(np.int64), type=pa.int64())
    F = dt.Frame(pa.table(arrs))
    assert all(dt.internal.frame_columns_virtual(F))
    for c, nm in enumerate(F.names):
        a = arrs[nm]
        bufs = a.buffers()
        st = F.stypes[c].value
        col = ctx._lib.dthip_from_arrow  # noqa: F841  (the entry point under test)
        dc = ctx.from_arrow(np.frombuffer(bufs[1], np.uint8), np.frombuffer(bufs[0], np.uint8) if bufs[0] is not None else None, n, st)
        got = ctx.download(dc, n)
        hp = dt.internal.frame_column_data_r(F, c).value            # the reference's CPU materialisation
        exp = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(n * got.itemsize,)).view(got.dtype)
        assert_same(got.view("u%d" % got.itemsize), exp.view("u%d" % got.itemsize), "column %s" % nm)


def test_arrow_backed_frame_through_the_binding_stays_virtual():
    """DT[:, {sum, count}, by(f.k)] on a Frame built from a pyarrow table: the binding uploads the Arrow buffers
    (shim.stats['arrow_uploads']), the result equals the reference's, and the Frame's columns are STILL virtual afterwards
    -- the reference's materialisation pass did not run"""
    dt, pa = _ref()
    from datatable import f, sum as dsum, count, mean
    from integration import datatable_hip_shim as shim
    rng = np.random.default_rng(11)
    n = 400_000
    k = pa.array(rng.integers(0, 5000, n).astype(np.int64), mask=rng.random(n) < 0.01)
    v = pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.1)
    w = pa.array(rng.integers(-100, 100, n).astype(np.int32), mask=rng.random(n) < 0.1)
    t = pa.table({"k": k, "v": v, "w": w})
    old = shim.options.residency
    shim.options.residency = "auto"
    try:
        DT = shim.Frame(t)
        before = shim.stats["arrow_uploads"]
        R = DT[:, {"s": dsum(f.v), "m": mean(f.w), "n": count()}, shim.by(f.k)]
        assert shim.stats["arrow_uploads"] - before == 3
        assert all(dt.internal.frame_columns_virtual(DT)), "a column was materialised on the CPU"
        E = dt.Frame(t)[:, {"s": dsum(f.v), "m": mean(f.w), "n": count()}, dt.by(f.k)]
        assert R.names == E.names and R.stypes == E.stypes and R.nrows == E.nrows
        rl, el = R.to_list(), E.to_list()
        assert rl[0] == el[0] and rl[3] == el[3]
        for a, b in zip(rl[1] + rl[2], el[1] + el[2]):
            assert (a is None and b is None) or abs(a - b) <= 1e-9 * max(1.0, abs(b)), (a, b)
        # second statement: served from the resident columns, nothing uploaded again
        R2 = DT[:, dsum(f.w), shim.by(f.k)]
        assert shim.stats["arrow_uploads"] - before == 3
        assert R2.to_list() == dt.Frame(t)[:, dsum(f.w), dt.by(f.k)].to_list()
    finally:
        shim.options.residency = old
