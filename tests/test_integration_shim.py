"""The reference-side binding of INTEGRATION.md (integration/datatable_hip_shim.py): query
matching, pointer/stype extraction and fall-through, exercised against the REAL reference when a
build of it is importable in this container (DT_REFERENCE_SRC, default /tmp/dt_oracle/src).
No GPU needed: nothing here calls a compute entry point of libdthip."""
import os
import sys

import numpy as np
import pytest

SRC = os.environ.get("DT_REFERENCE_SRC", "/tmp/dt_oracle/src")
if os.path.isdir(SRC) and SRC not in sys.path:
    sys.path.insert(0, SRC)
dt = pytest.importorskip("datatable", reason="the reference is not importable here")


def test_match_and_fallthrough():
    from datatable import f, sum, mean, count, min as dmin
    from integration import datatable_hip_shim as shim
    DT = shim.Frame(k=np.array([3, 1, 3], np.int64), v=np.array([1.0, 2.0, 4.0]), s=["a", "b", "c"])
    assert shim.match(DT, (slice(None), [sum(f.v), mean(f["v"]), count(), dmin(f[1])], shim.by(f.k))) == \
        ([0], [("sum", 1), ("mean", 1), ("count0", None), ("min", 1)])
    assert shim.match(DT, (slice(None), sum(f.v), shim.by("k"))) == ([0], [("sum", 1)])
    # not covered -> None -> the reference evaluates it
    assert shim.match(DT, (slice(None), sum(f.v), shim.by(f.s))) is None          # string key
    assert shim.match(DT, (slice(None), sum(f.v + 1), shim.by(f.k))) is None      # computed argument
    assert shim.match(DT, (f.v > 1, sum(f.v), shim.by(f.k))) is None              # i-filter
    assert shim.match(DT, (slice(None), f.v, shim.by(f.k))) is None               # not a reducer
    R = DT[:, sum(f.v), shim.by(f.s)]                                             # falls through
    assert type(R) is dt.Frame and R.to_list() == [["a", "b", "c"], [1.0, 2.0, 4.0]]
    assert DT[f.v > 1, :].nrows == 2


def test_borrowed_pointers_are_the_numpy_buffers():
    from integration import datatable_hip_shim as shim
    a = np.arange(10, dtype=np.int64)
    DT = shim.Frame(k=a)
    assert dt.internal.frame_column_data_r(DT, 0).value == a.ctypes.data
    assert DT.stypes[0].value == 5          # == DTHIP_INT64
