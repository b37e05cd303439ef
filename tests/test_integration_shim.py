"""The reference-side binding of INTEGRATION.md (integration/datatable_hip_shim.py): query
matching, pointer/stype extraction and fall-through, exercised against the REAL reference as built
into oracle/_ref by oracle/build_ref.sh (skipped only when that build was never made).
No GPU needed: nothing here calls a compute entry point of libdthip (tests/test_shim_e2e.py does, on the GPU)."""
import numpy as np
import pytest

from oracle import ref

dt = ref.load()
if dt is None:
    pytest.skip("oracle/_ref (the reference build) is not present: run oracle/build_ref.sh", allow_module_level=True)


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


def test_match_groupwise_operators():
    """sd / median / nunique / first / last, cov / corr and the cumulative operators are recognised from
    the reference's own reprs (old-style `Expr:stdev(FExpr<f.v>; )` nodes included)"""
    import datatable
    from datatable import f
    from integration import datatable_hip_shim as shim
    DT = shim.Frame(k=np.array([3, 1, 3], np.int64), v=np.array([1.0, 2.0, 4.0]), w=np.array([1, 2, 3], np.int32),
                    s=["a", "b", "c"])
    j = [datatable.sd(f.v), datatable.median(f["w"]), datatable.nunique(f[1]), datatable.first(f.v), datatable.cov(f.v, f.w),
         datatable.corr(f.w, f.v), datatable.count()]
    assert shim.match(DT, (slice(None), j, shim.by(f.k))) == \
        ([0], [("sd", 1), ("median", 2), ("nunique", 1), ("first", 1), ("cov", (1, 2)), ("corr", (2, 1)), ("count0", None)])
    j = [datatable.cumsum(f.v), datatable.cummax(f.w, reverse=True), datatable.cumcount(), datatable.ngroup(reverse=True)]
    assert shim.match(DT, (slice(None), j, shim.by(f.k))) == \
        ([0], [("cumsum", 1, False), ("cummax", 2, True), ("cumcount", None, False), ("ngroup", None, True)])
    # a reducer next to a row-level operator, or a string argument: the reference evaluates it
    assert shim.match(DT, (slice(None), [datatable.sd(f.v), datatable.cumsum(f.v)], shim.by(f.k))) is None
    assert shim.match(DT, (slice(None), datatable.nunique(f.s), shim.by(f.k))) is None
    assert shim.match(DT, (slice(None), datatable.sd(f.v + 1), shim.by(f.k))) is None


def test_reference_thread_settings_are_safe():
    """oracle/ref.py: sort.nthreads is stored through a uint8_t cast in the reference (sort.cc:337), so 256 would become 0
    and group() would divide by it (SIGFPE on the 256-thread GPU box): the helper clamps it, and a groupby still works"""
    from datatable import f, by
    ref.set_threads(256)
    try:
        assert dt.options.sort.nthreads == 255 and dt.options.nthreads >= 1
        DT = dt.Frame(k=np.arange(100_000) % 1000, v=np.ones(100_000))
        R = DT[:, dt.sum(f.v), by(f.k)]
        assert R.nrows == 1000 and R[:, 1].to_numpy().sum() == 100_000
    finally:
        ref.set_threads(min(8, __import__("os").cpu_count() or 1))
