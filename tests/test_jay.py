"""CPU: the Jay reader (datatable_amd/jay.py) against .jay files written by the unmodified reference
(tests/golden/jay/*.jay, generator tests/golden/make_jay_golden.py): names, stypes, key count and every
value; the columns are views of the mapped file (no copy)."""
import json
import math
import os

import numpy as np
import pytest

from conftest import ROOT
from datatable_amd import jay

JDIR = os.path.join(ROOT, "tests", "golden", "jay")
EXP = json.load(open(os.path.join(JDIR, "expected.json")))
NA_INT = {1: -128, 2: -2**15, 4: -2**31, 8: -2**63}


def _pylist(a, st):
    if st in (6, 7):
        return [None if v != v else ("inf" if v == math.inf else "-inf" if v == -math.inf else float(v)) for v in a]
    na = NA_INT[a.dtype.itemsize]
    if st == 1:
        return [None if v == na else bool(v) for v in a]
    return [None if v == na else int(v) for v in a]


@pytest.mark.parametrize("name", [n for n in EXP if n != "with_string"])
def test_jay_matches_reference(name):
    e = EXP[name]
    cols, nkeys = jay.jay_columns(os.path.join(JDIR, name + ".jay"))
    assert [c[0] for c in cols] == e["names"]
    assert [c[1] for c in cols] == e["stypes"]
    assert nkeys == e["nkeys"]
    for (nm, st, a), want in zip(cols, e["columns"]):
        assert len(a) == 0 or (not a.flags.owndata and not a.flags.writeable)     # a view of the mapped file
        got = _pylist(a, st)
        assert got == want, nm
        if st == 7 and name == "types":
            assert math.copysign(1.0, a[2]) == -1.0                   # -0.0 survives bit for bit


def test_jay_string_columns():
    p = os.path.join(JDIR, "with_string.jay")
    with pytest.raises(NotImplementedError, match="type str32 is outside the accelerated path"):
        jay.jay_columns(p)
    cols, nkeys = jay.jay_columns(p, strings="skip")
    assert [c[0] for c in cols] == ["k", "v"] and cols[1][2].tolist() == [1.0, 2.0, 3.0]


def test_jay_bad_files(tmp_path):
    p = tmp_path / "x.jay"
    p.write_bytes(b"JAY1")
    with pytest.raises(IOError, match="Invalid Jay file of size 4"):
        jay.jay_columns(str(p))
    good = open(os.path.join(JDIR, "keyed.jay"), "rb").read()
    p.write_bytes(b"XYZ1" + good[4:])
    with pytest.raises(IOError, match="Invalid signature for a Jay file: first 4 bytes are `XYZ1`"):
        jay.jay_columns(str(p))
    p.write_bytes(b"JAY2" + good[4:])
    with pytest.raises(IOError, match="Unsupported Jay file version"):
        jay.jay_columns(str(p))
    p.write_bytes(good[:-16] + (10**9).to_bytes(8, "little") + good[-8:])
    with pytest.raises(IOError, match="Invalid meta record size"):
        jay.jay_columns(str(p))


@pytest.mark.gpu
def test_jay_to_device_and_groupby():
    """Jay -> HBM -> the hot path, never through a parsed copy"""
    from datatable_amd import _lib as L
    from datatable_amd.engine import Context
    from datatable_amd.frame import by, f, sum, count
    ctx = Context(0)
    cols, nrows = jay.to_device(os.path.join(JDIR, "big.jay"), ctx)          # dthip_malloc + dthip_memcpy_h2d, no torch
    e = EXP["big"]
    assert nrows == len(e["columns"][0]) and cols["k"].stype == L.INT32
    r = ctx.groupby_agg([cols["k"]], [cols["v"]], [("sum", 0), ("count0", None)], nrows=nrows)
    DT = jay.open_jay(os.path.join(JDIR, "big.jay"))
    R = DT[:, [sum(f.v), count()], by(f.k)]
    assert r.key(0).tolist() == R.to_list()[0]
    assert np.allclose(r.agg(0), np.array(R.to_list()[1]))
    assert r.agg(1).tolist() == R.to_list()[2]
    g = ctx.groupby([cols["k"]], nrows=nrows)
    assert np.array_equal(np.array(e["columns"][0], np.int32)[g.rowindex()[g.offsets()[:-1]]], r.key(0))
    r.free(); g.free()
    del cols
    ctx.close()
    K = jay.open_jay(os.path.join(JDIR, "keyed.jay"))
    assert K.key == ("k",)
    X = type(K)(k=[3, 4, 9])
    from datatable_amd.frame import join
    assert X[:, :, join(K)].to_list() == [[3, 4, 9], [3.5, None, 2.5], [40, None, 30]]
