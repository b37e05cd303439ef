"""The reference-side binding's HOST logic end to end WITHOUT a GPU: the same `DT[...]` statements as tests/test_shim_e2e.py
(fused aggregates, row filter, rows in grouped order, sort, the two-step form of BASELINE config 5, every residency mode)
go through integration/datatable_hip_shim.py into tests/shim_standin.py -- a NumPy / oracle stand-in for the library's
entry points -- and are compared with what the unmodified reference (oracle/_ref) returns for the same statement.
What this pins down on the CPU: query matching, pointer / stype / flag plumbing, the residency cache (uploads counted),
lazy DeviceFrame results, result assembly and names.  The kernels themselves are the GPU suite's business."""
import math

import numpy as np

from datatable_amd import _lib as L
import pytest

from oracle import ref

dt = ref.load()
if dt is None:
    pytest.skip("oracle/_ref (the reference build) is not present: run oracle/build_ref.sh", allow_module_level=True)

from shim_standin import StandinCtx                                              # noqa: E402
from test_shim_e2e import assert_frames_equal, assert_rows_equal, make_frame     # noqa: E402


@pytest.fixture()
def env(monkeypatch):
    from integration import datatable_hip_shim as shim
    ctx = StandinCtx()
    monkeypatch.setattr(shim, "default_context", lambda: ctx)
    old = shim.options.residency, shim.options.f32_sum
    yield shim, ctx
    shim.options.residency, shim.options.f32_sum = old


def _as_frame(shim, r):
    return r.to_frame() if isinstance(r, shim.DeviceFrame) else r


@pytest.mark.parametrize("mode", ["off", "auto", "lazy"])
@pytest.mark.parametrize("key", ["int64", "int32", "float64", "bool"])
def test_fused_aggregates(env, mode, key):
    shim, ctx = env
    shim.options.residency = mode
    from datatable import f, sum, mean, min, max, count
    for n in (1, 37, 5000):
        DT = make_frame(shim, n, seed=n + len(key), key=key)
        j = [op(f[c]) for c in ("f8", "f4", "i8", "i4", "i2", "i1", "b") for op in (sum, mean, min, max, count)] + [count()]
        got = DT[:, j, shim.by(f.k)]
        exp = dt.Frame.__getitem__(DT, (slice(None), j, dt.by(f.k)))
        assert (type(got) is shim.DeviceFrame) == (mode == "lazy")
        # the stand-in accumulates float32 sums the way the oracle (= the reference) does: no float32 allowance needed
        assert_frames_equal(dt, _as_frame(shim, got), exp, sizes=exp[:, -1].to_list()[0])
    DT = make_frame(shim, 2000, seed=5)
    j = [sum(f.f8), sum(f.f8), mean(f["i4"]), min(f[2]), count(f.i8), count()]
    got = _as_frame(shim, DT[:, j, shim.by(f.k, f.k2)])
    assert_frames_equal(dt, got, dt.Frame.__getitem__(DT, (slice(None), j, dt.by(f.k, f.k2))))        # duplicate names mangled alike
    assert ctx._lib.calls and all(c[0] == "groupby_agg" for c in ctx._lib.calls)
    assert all(c[1] == (shim.L.HOST if mode == "off" else shim.L.DEVICE) for c in ctx._lib.calls)


@pytest.mark.parametrize("mode", ["off", "auto", "lazy"])
def test_filter_rows_and_sort_routes(env, mode):
    shim, ctx = env
    shim.options.residency = mode
    from datatable import f
    for n in (1, 37, 3000):
        DT = make_frame(shim, n, seed=300 + n)
        preds = [f.f8 > 0, f.f8 >= 0.25000001, f.f8 <= -1.5, f.i4 < 2.5, f.i8 != 0, f.i2 == 7, f.f4 > 0.1, f.i1 >= -3,
                 f.f8 == None, f.i4 != None, f.f8 < math.inf, f.i8 > 1e30]              # noqa: E711
        for p in preds:
            got = DT[p, :]
            assert_rows_equal(dt, _as_frame(shim, got), dt.Frame.__getitem__(DT, (p, slice(None))))
        got = DT[f.f8 > 0, [f.k, f["i4"], f[3]]]
        assert_rows_equal(dt, _as_frame(shim, got), dt.Frame.__getitem__(DT, (f.f8 > 0, [f.k, f["i4"], f[3]])))
        for j, bycols in ((slice(None), [f.k]), (f[:], [f.k, f.k2]), ([f.f8, f.k, f["i1"]], ["k"]), (f.b, [f.k2])):
            got = DT[:, j, shim.by(*bycols)]
            assert_rows_equal(dt, _as_frame(shim, got), dt.Frame.__getitem__(DT, (slice(None), j, dt.by(*bycols))))
        DF = make_frame(shim, n, seed=500 + n, key="float64")
        for c in (dict(cols=[f.k]), dict(cols=[f.k], reverse=True), dict(cols=[f.k], na_position="last"),
                  dict(cols=[f.i4, f.f8], reverse=[True, False]), dict(cols=[f.b, f.f4])):
            cols = c.pop("cols")
            got = DF[:, :, shim.sort(*cols, **c)]
            assert_rows_equal(dt, _as_frame(shim, got), dt.Frame.__getitem__(DF, (slice(None), slice(None), dt.sort(*cols, **c))))
        assert_rows_equal(dt, _as_frame(shim, DF.sort("k")), dt.Frame.sort(DF, "k"))
    kinds = {c[0] for c in ctx._lib.calls}
    assert {"filter_take", "groupby_rows"} <= kinds


def test_config5_two_steps_stay_resident(env):
    """V = DT[f.x > 0, :]; V[:, :, by(f.k)] under lazy residency: ONE upload per column of DT, no download until the result
    is looked at, and the same rows the reference returns"""
    shim, ctx = env
    shim.options.residency = "lazy"
    from datatable import f
    rng = np.random.default_rng(77)
    n = 40_000
    k = rng.integers(0, 3000, n).astype(np.int64); k[rng.random(n) < 0.01] = -2**63
    x = rng.standard_normal(n); x[rng.random(n) < 0.02] = np.nan
    DT = shim.Frame(k=k, x=x)
    lib = ctx._lib
    V = DT[f.x > 0, :]
    assert isinstance(V, shim.DeviceFrame) and "pending" in repr(V) and V.names == ("k", "x")     # a view, like the reference's
    R = V[:, :, shim.by(f.k)]
    assert type(R) is shim.DeviceFrame
    # round 5: the two statements ran as ONE library call (the filter fused into the grouping), the filter alone never ran
    assert [c[0] for c in lib.calls if isinstance(c, tuple)] == ["filter_groupby_rows"], lib.calls
    assert lib.uploads == 2 and lib.downloads == 0
    assert R.names == ("k", "x") and R.nrows == V.nrows == int((x > 0).sum())
    assert ("filter_take", L.DEVICE) in lib.calls              # V.nrows made the pending view evaluate its filter
    V2 = DT[f.x > 0, :]                                        # the columns are resident: nothing is uploaded again
    assert lib.uploads == 2
    # the view used for anything else first, then grouped: the ordinary two calls, same rows
    assert V2.shape == (V.nrows, 2)
    R2 = V2[:, :, shim.by(f.k)]
    shim.options.defer_filter = False
    try:
        R3 = DT[f.x > 0, :][:, :, shim.by(f.k)]                # round 4's behaviour: filter at once
    finally:
        shim.options.defer_filter = True
    exp = dt.Frame.__getitem__(dt.Frame.__getitem__(DT, (f.x > 0, slice(None))), (slice(None), slice(None), dt.by(f.k)))
    assert_rows_equal(dt, R.to_frame(), exp)
    assert lib.downloads == 2                                  # the two result columns, once
    R.to_frame()
    assert lib.downloads == 2
    assert_rows_equal(dt, R2.to_frame(), exp); assert_rows_equal(dt, R3.to_frame(), exp)
    assert V2.nrows == V.nrows
    # a mutation of DT drops its device copies: the next query uploads the NEW values
    DT[0, "x"] = 1e9
    assert not DT.is_resident
    W = DT[f.x > 1e8, :]
    assert lib.uploads == 4 and W.nrows == 1


def test_unrouted_statements_reach_the_reference(env):
    shim, ctx = env
    from datatable import f, sum
    DT = shim.Frame(k=[3, 1, 3], v=[1.0, 2.0, 4.0], s=["a", "b", "c"])
    R = DT[:, sum(f.v), shim.by(f.s)]                          # string key: not accelerated
    assert type(R) is dt.Frame and R.to_list() == [["a", "b", "c"], [1.0, 2.0, 4.0]]
    assert DT[f.s == "a", :].nrows == 1
    with pytest.raises(NotImplementedError):                   # the reference's own refusal of i + by comes through
        DT[f.v > 1, :, shim.by(f.k)]
    assert not ctx._lib.calls


@pytest.mark.parametrize("mode", ["off", "auto", "lazy"])
def test_sred_routes(env, mode):
    """the other reducers and the cumulative operators: dthip_groupby once, then one call per j item"""
    shim, ctx = env
    shim.options.residency = mode
    import datatable
    from datatable import f
    for n in (1, 29, 3000):
        DT = make_frame(shim, n, seed=100 + n)
        j = [datatable.sd(f.f8), datatable.sd(f.i4), datatable.median(f.f8), datatable.median(f.i2), datatable.nunique(f.i1),
             datatable.nunique(f.f4), datatable.first(f.f8), datatable.last(f.i8), datatable.first(f.b), datatable.last(f.b),
             datatable.cov(f.f8, f.i4), datatable.corr(f.f4, f.f8), datatable.count()]
        got = _as_frame(shim, DT[:, j, shim.by(f.k)])
        assert_frames_equal(dt, got, dt.Frame.__getitem__(DT, (slice(None), j, dt.by(f.k))))
        DT[:, "f8"] = DT[:, dt.ifelse(dt.math.isinf(f.f8), 1.0, f.f8)]
        for reverse in (False, True):
            j = [datatable.cumsum(f.f8, reverse=reverse), datatable.cumsum(f.i4, reverse=reverse),
                 datatable.cummin(f.f4, reverse=reverse), datatable.cummax(f.i8, reverse=reverse),
                 datatable.cummax(f.b, reverse=reverse), datatable.cumprod(f.i1, reverse=reverse),
                 datatable.cumcount(reverse=reverse), datatable.ngroup(reverse=reverse),
         datatable.fillna(f.f8, reverse=reverse), datatable.fillna(f.i2, reverse=reverse), datatable.fillna(f.b, reverse=reverse)]
            got = _as_frame(shim, DT[:, j, shim.by(f.k, f.k2)])
            assert_frames_equal(dt, got, dt.Frame.__getitem__(DT, (slice(None), j, dt.by(f.k, f.k2))))
    kinds = {c[0] for c in ctx._lib.calls}
    assert {"groupby", "reduce", "reduce2", "cumulate"} <= kinds


def test_sort_with_na_removed(env):
    shim, ctx = env
    from datatable import f
    DF = make_frame(shim, 3000, seed=501, key="float64")
    for c in (dict(cols=[f.i1], na_position="remove"), dict(cols=[f.k2, f.i2], na_position="remove"),
              dict(cols=[f.k], reverse=True, na_position="remove")):
        cols = c.pop("cols")
        got = DF[:, :, shim.sort(*cols, **c)]
        assert_rows_equal(dt, _as_frame(shim, got), dt.Frame.__getitem__(DF, (slice(None), slice(None), dt.sort(*cols, **c))))


@pytest.mark.parametrize("mode", ["off", "auto", "lazy"])
def test_dict_form_of_j_names_the_result(env, mode):
    """DT[:, {"total": sum(f.x), ...}, by(...)] and the dict form on the row-returning routes: evaluated like a list, the j
    columns named after the keys, duplicates mangled the reference's way"""
    shim, ctx = env
    shim.options.residency = mode
    import datatable
    from datatable import f, sum, mean, count
    DT = make_frame(shim, 2000, seed=9)
    for j in ({"total": sum(f.f8), "n": count()}, {"k": sum(f.f8)}, {"total": sum(f.f8), "f8": mean(f.f8), "total2": sum(f.i4)},
              {"s": datatable.sd(f.f8), "m": datatable.median(f.i2)}, {"c": datatable.cumsum(f.i4), "r": datatable.cumcount()}):
        got = DT[:, j, shim.by(f.k)]
        exp = dt.Frame.__getitem__(DT, (slice(None), j, dt.by(f.k)))
        assert (type(got) is shim.DeviceFrame) == (mode == "lazy") and tuple(got.names) == exp.names
        assert_frames_equal(dt, _as_frame(shim, got), exp)
    for j in ({"a": f.f8}, {"a": f.f8, "kk": f.k}, {"f8": f.i4, "i4": f.f8}):
        for extra, ref_extra in (((shim.by(f.k),), (dt.by(f.k),)), ((shim.sort(f.k2, f.i8),), (dt.sort(f.k2, f.i8),))):
            got = DT[(slice(None), j) + extra]
            exp = dt.Frame.__getitem__(DT, (slice(None), j) + ref_extra)
            assert tuple(got.names) == exp.names
            assert_rows_equal(dt, _as_frame(shim, got), exp)
        got = DT[f.f8 > 0, j]
        assert_rows_equal(dt, _as_frame(shim, got), dt.Frame.__getitem__(DT, (f.f8 > 0, j)))
    assert ctx._lib.calls
    n0 = len(ctx._lib.calls)
    R = DT[:, {"z": f.f8 + 1}, shim.by(f.k)]                   # a computed column: the reference's, with its names
    assert type(R) is dt.Frame and R.names == ("k", "z") and len(ctx._lib.calls) == n0
