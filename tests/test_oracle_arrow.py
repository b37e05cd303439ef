"""CPU: the numpy restatement of the reference's Arrow element access (oracle.arrow_to_sentinel, arrow_fw.cc:63-72 /
arrow_bool.cc) is pinned to the unmodified reference (oracle/_ref) materialising real pyarrow arrays, and the binding's
Arrow route (integration/datatable_hip_shim.py::_arrow_buffers -> dthip_from_arrow) runs end to end against the
NumPy / oracle stand-in for the library: an Arrow-backed Frame is aggregated while its columns stay virtual."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as o
from oracle import ref

pa = pytest.importorskip("pyarrow")


@pytest.fixture(scope="module")
def dt():
    m = ref.load()
    if m is None:
        pytest.skip("oracle/_ref (the reference build) is not on this machine")
    return m


def test_restatement_matches_the_reference_materialisation(dt):
    rng = np.random.default_rng(3)
    types = {"b": pa.bool_(), "i8": pa.int8(), "i16": pa.int16(), "i32": pa.int32(), "i64": pa.int64(), "f32": pa.float32(), "f64": pa.float64()}
    for n in (1, 7, 8, 9, 63, 64, 65, 1000, 50_001):
        arrs = {}
        for nm, t in types.items():
            mask = rng.random(n) < 0.3
            if nm == "b":
                arrs[nm] = pa.array(rng.random(n) < 0.5, type=t, mask=mask)
            elif nm.startswith("f"):
                arrs[nm] = pa.array(rng.standard_normal(n).astype(t.to_pandas_dtype()), type=t, mask=mask)
            else:
                ii = np.iinfo(t.to_pandas_dtype())
                arrs[nm] = pa.array(rng.integers(ii.min + 1, ii.max, n).astype(t.to_pandas_dtype()), type=t, mask=mask)
        arrs["nonull"] = pa.array(rng.integers(-9, 9, n).astype(np.int32), type=pa.int32())
        F = dt.Frame(pa.table(arrs))
        assert all(dt.internal.frame_columns_virtual(F))
        for c, nm in enumerate(F.names):
            bufs = arrs[nm].buffers()
            st = F.stypes[c].value
            vals = np.frombuffer(bufs[1], np.uint8)
            if st != o.BOOL:
                vals = vals[:n * np.dtype(o._ST2NP[st]).itemsize].view(o._ST2NP[st])
            got = o.arrow_to_sentinel(vals, np.frombuffer(bufs[0], np.uint8) if bufs[0] is not None else None, n, st)
            hp = dt.internal.frame_column_data_r(F, c).value          # the reference materialises the column here
            exp = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(n * got.itemsize,)).view(got.dtype)
            assert np.array_equal(got.view("u%d" % got.itemsize), exp.view("u%d" % got.itemsize)), (nm, n)


def test_binding_takes_arrow_columns_without_materialising_them(dt, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import shim_standin
    from datatable import f, sum as dsum, count, min as dmin
    from integration import datatable_hip_shim as shim
    ctx = shim_standin.StandinCtx()
    monkeypatch.setattr(shim, "default_context", lambda: ctx)
    monkeypatch.setattr(shim.options, "residency", "auto")
    rng = np.random.default_rng(5)
    n = 20_000
    t = pa.table({"k": pa.array(rng.integers(0, 300, n).astype(np.int32), mask=rng.random(n) < 0.02),
                  "v": pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.2),
                  "flag": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1)})
    DT = shim.Frame(t)
    before = shim.stats["arrow_uploads"]
    R = DT[:, {"s": dsum(f.v), "lo": dmin(f.v), "n": count()}, shim.by(f.k)]
    assert shim.stats["arrow_uploads"] - before == 2 and ctx._lib.calls.count("from_arrow") == 2
    assert all(dt.internal.frame_columns_virtual(DT)), "the reference materialised a column"
    E = dt.Frame(t)[:, {"s": dsum(f.v), "lo": dmin(f.v), "n": count()}, dt.by(f.k)]
    assert R.names == E.names and R.stypes == E.stypes
    rl, el = R.to_list(), E.to_list()
    assert rl[0] == el[0] and rl[2] == el[2] and rl[3] == el[3]
    assert all((a is None and b is None) or abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(rl[1], el[1]))
    # a bool (bit-packed) Arrow column as the group key
    R2 = DT[:, count(), shim.by(f.flag)]
    assert R2.to_list() == dt.Frame(t)[:, count(), dt.by(f.flag)].to_list()
    assert all(dt.internal.frame_columns_virtual(DT))
