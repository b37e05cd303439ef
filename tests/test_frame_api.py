"""The reference's Python surface for the path (datatable_amd.frame: Frame, f, by, sort,
sum/mean/min/max/count) against the known answers the unmodified reference produced
(tests/test-groups.py, tests/test-reduce.py) and the golden fixtures.
Every evaluation goes through the C ABI on the GPU."""
import math

import numpy as np
import pytest

from conftest import assert_same, golden_names

pytestmark = pytest.mark.gpu

inf = math.inf


@pytest.fixture(scope="module")
def dt():
    from datatable_amd import frame
    return frame


def appendix_b(dt):
    return dt.Frame(k=[3, None, 1, 3, 1, None, 2, 3], v=[1.5, 2.0, None, 4.0, None, 8.0, 16.0, inf], i=list(range(8)))


def test_appendix_b_all_reducers(dt):
    f, by = dt.f, dt.by
    DT = appendix_b(dt)
    R = DT[:, [dt.sum(f.v), dt.mean(f.v), dt.min(f.v), dt.max(f.v), dt.count(f.v), dt.count()], by(f.k)]
    assert R.names == ("k", "v", "v.0", "v.1", "v.2", "v.3", "count")
    assert R.stypes == (4, 7, 7, 7, 7, 5, 5)      # int32, f64 x4, int64, int64 (stype.h codes)
    assert R.to_list() == [[None, 1, 2, 3], [10.0, 0.0, 16.0, inf], [5.0, None, 16.0, inf], [2.0, None, 16.0, 1.5],
                           [8.0, None, 16.0, inf], [2, 0, 1, 3], [2, 2, 1, 3]]


def test_appendix_b_grouped_row_order(dt):
    f, by = dt.f, dt.by
    DT = appendix_b(dt)
    R = DT[:, f.i, by(f.k)]
    assert R.names == ("k", "i")
    assert R.to_list() == [[None, None, 1, 1, 2, 3, 3, 3], [1, 5, 2, 4, 6, 0, 3, 7]]
    A = DT[:, :, by(f.k)]
    assert A.names == ("k", "v", "i") and A.to_list()[2] == [1, 5, 2, 4, 6, 0, 3, 7]


def test_two_keys(dt):
    f, by = dt.f, dt.by
    DT = dt.Frame(a=[2, 1, 2, 1, 2, 1], b=[5, 5, 4, 5, 4, 3], v=[0, 1, 2, 3, 4, 5])
    R = DT[:, [dt.count(), dt.sum(f.v)], by(f.a, f.b)]
    assert R.names == ("a", "b", "count", "v")
    assert R.to_list() == [[1, 1, 2, 2], [3, 5, 4, 5], [1, 2, 2, 1], [5, 4, 6, 0]]
    assert DT[:, f.v, by(f.a, f.b)].to_list()[2] == [5, 1, 3, 2, 4, 0]


def test_int64_sum_wraps_to_na(dt):
    f = dt.f
    DT = dt.Frame(g=[1, 1], x=np.array([2**62, 2**62], np.int64))
    assert DT[:, dt.sum(f.x), dt.by(f.g)].to_list() == [[1], [None]]


def test_output_stypes(dt):
    f, by = dt.f, dt.by
    DT = dt.Frame(g=[0, 0, 1], a=np.array([1.5, 2.5, 3.0], np.float32), b=np.array([1, 2, 3], np.int32))
    R = DT[:, [dt.sum(f.a), dt.mean(f.a), dt.min(f.a), dt.sum(f.b), dt.mean(f.b), dt.min(f.b), dt.max(f.b)], by(f.g)]
    assert R.stypes == (4, 6, 6, 6, 5, 7, 4, 4)
    assert R.to_list()[1:] == [[4.0, 3.0], [2.0, 3.0], [1.5, 3.0], [3, 3], [1.5, 3.0], [1, 3], [2, 3]]


def test_float_keys_signed_zero_and_nan(dt):
    f = dt.f
    DT = dt.Frame(k=np.array([0.0, -0.0, 0.0, np.nan, np.nan]), v=[1, 1, 1, 1, 1])
    R = DT[:, dt.count(), dt.by(f.k)]
    keys = R.to_numpy_columns()[0]
    assert R.to_list()[1] == [2, 1, 2]
    assert np.isnan(keys[0]) and np.signbit(keys[1]) and keys[1] == 0 and not np.signbit(keys[2])


def test_filter_in_i_with_by_is_not_implemented_like_the_reference(dt):
    f = dt.f
    DT = appendix_b(dt)
    with pytest.raises(NotImplementedError, match="evaluate_iby"):
        DT[f.v > 0, dt.sum(f.v), dt.by(f.k)]


def test_filter_view_then_groupby(dt):
    f = dt.f
    DT = appendix_b(dt)
    V = DT[f.v > 1.6, :]
    assert V.nrows == 5 and V.to_list()[2] == [1, 3, 5, 6, 7]
    R = V[:, dt.sum(f.v), dt.by(f.k)]
    assert R.to_list() == [[None, 2, 3], [10.0, 16.0, inf]]
    # a view of a view composes the RowIndexes
    W = V[f.i >= 5, :]
    assert W.to_list()[2] == [5, 6, 7]
    assert W[:, dt.count(), dt.by(f.k)].to_list() == [[None, 2, 3], [1, 1, 1]]


def test_sort(dt):
    f = dt.f
    DT = appendix_b(dt)
    assert DT.sort("k").to_list()[2] == [1, 5, 2, 4, 6, 0, 3, 7]
    assert DT[:, :, dt.sort(f.k, reverse=True)].to_list()[0] == [None, None, 3, 3, 3, 2, 1, 1]
    assert DT[:, :, dt.sort(f.k, na_position="last")].to_list()[0] == [1, 1, 2, 3, 3, 3, None, None]
    assert DT[:, f.i, dt.sort(f.v)].to_list() == [[2, 4, 0, 1, 3, 5, 6, 7]]


@pytest.mark.parametrize("rev", [True, False])
@pytest.mark.parametrize("napos", ["first", "last", "remove"])
@pytest.mark.parametrize("src", [[-5, -8, None, None, 11, 2, 8, None, 4] * 1000,
                                 [-5.9, None, -8.3, 11.5576, 2.2, 8.9, None, 4.1] * 1000,
                                 [True, None, False, None, False, True] * 1000,
                                 [0, 1, None, 2**31 - 1, None, -(2**31 - 1), None] * 1000,
                                 [0, 1, None, 2**63 - 1, None, -(2**63 - 1), None] * 1000])
def test_sort_na_position(dt, rev, napos, src):
    """tests/ijby/test-sort.py:1080-1092 of the reference, same sources and same expectation"""
    def key_func(x):
        return (x is None) ^ rev ^ (napos == "first")
    if napos == "remove":
        exp = sorted([s for s in src if s is not None], reverse=rev)
    else:
        exp = sorted(src, key=lambda x: (key_func(x), x if x is not None else 0), reverse=rev)
    DT = dt.Frame(A=src)
    RES = DT[:, :, dt.sort(0, reverse=rev, na_position=napos)]
    assert RES.to_list() == [exp]


def test_na_position_value_error(dt):
    with pytest.raises(ValueError, match="na position value las is not supported"):
        dt.sort(0, reverse=True, na_position="las")


def test_reducers_without_by(dt):
    f = dt.f
    DT = appendix_b(dt)
    R = DT[:, [dt.sum(f.v), dt.count(f.v), dt.count()]]
    assert R.names == ("v", "v.0", "count") and R.to_list() == [[inf], [6], [8]]


def test_errors(dt):
    DT = appendix_b(dt)
    with pytest.raises(KeyError):
        DT[:, dt.sum(dt.f.nope), dt.by(dt.f.k)]
    with pytest.raises(ValueError):
        DT[:, dt.f[7]]
    with pytest.raises(TypeError):
        dt.sum()


@pytest.mark.parametrize("name", golden_names())
def test_golden_through_frame_api(dt, gold, name):
    """every golden case (= outputs of the unmodified reference) spelled the reference's way"""
    c = gold.by_name[name]
    keys, vals = gold.keys(name), gold.vals(name)
    knames = ["k%d" % i for i in range(len(keys))]
    vnames = ["v%d" % i for i in range(len(vals))]
    DT = dt.Frame(dict(zip(knames + vnames, keys + vals)), stypes=list(c["key_stypes"]) + list(c["val_stypes"]))
    fn = {"sum": dt.sum, "mean": dt.mean, "min": dt.min, "max": dt.max, "count": dt.count}
    j = [fn[opn](dt.f["v%d" % vi]) for opn, vi, _ in c["aggs"]] + [dt.count()]
    R = DT[:, j, dt.by(*knames)]
    cols = R.to_numpy_columns()
    for i in range(len(keys)):
        assert_same(cols[i], gold.get(name, "gk%d" % i), "group key %d" % i)
    off = gold.get(name, "off")
    assert_same(cols[-1], np.diff(off).astype(np.int64), "count()")
    for a, (opn, vi, ost) in enumerate(c["aggs"]):
        exp = gold.get(name, "%s.v%d" % (opn, vi))
        got = cols[len(keys) + a]
        assert got.dtype == exp.dtype
        if exp.dtype.kind == "f" and opn in ("sum", "mean"):
            assert np.allclose(got, exp, rtol=1e-4 if vals[vi].dtype == np.float32 else 1e-6, atol=1e-9, equal_nan=True), opn
        else:
            assert_same(got, exp, "%s(v%d)" % (opn, vi))
    G = DT[:, :, dt.by(*knames)]
    assert G.names[:len(keys)] == tuple(knames)
    assert_same(G._ri, gold.get(name, "ri"), "grouped row order")


def test_resident_frame_matches_host_frame():
    """Frame.to_device(): the fused aggregation reads the columns from HBM; same results as the host path,
    for every reducer, descending / multi-column by(), and again after the host arrays are gone"""
    import numpy as np
    from datatable_amd.frame import Frame, f, by, sum, mean, min, max, count, first, last   # noqa: A004
    rng = np.random.default_rng(3)
    n = 200_000
    cols = dict(a=rng.integers(0, 300, n).astype(np.int32), b=rng.integers(-5, 5, n).astype(np.int64),
                v=rng.standard_normal(n), w=rng.integers(-100, 100, n).astype(np.int16))
    cols["v"][rng.random(n) < 0.05] = np.nan
    H = Frame(**cols)
    D = Frame(**cols).to_device()
    for q in ("DT[:, [sum(f.v), mean(f.v), min(f.w), max(f.w), count(f.v), count()], by(f.a)]",
              "DT[:, [sum(f.w), first(f.v), last(f.w)], by(f.a, f.b)]",
              "DT[:, sum(f.v), by(-f.b)]",
              "DT[:, count(), by(f.w)]"):
        R1, R2 = eval(q, dict(globals(), DT=H, **locals())), eval(q, dict(globals(), DT=D, **locals()))
        assert R1.names == R2.names and R1.stypes == R2.stypes
        for c1, c2 in zip(R1.to_numpy_columns(), R2.to_numpy_columns()):
            if c1.dtype.kind == "f":       # float sums: LDS atomics add in arrival order, the last bits vary run to run
                assert np.allclose(c1, c2, rtol=1e-9, atol=1e-9, equal_nan=True)
            else:
                assert np.array_equal(c1, c2)
    # a view of a resident frame falls back to the host path (its rows are a gather)
    V = D[f.v > 0, :]
    assert V[:, count(), by(f.a)].to_list() == H[f.v > 0, :][:, count(), by(f.a)].to_list()


def test_resident_frame_after_key_assignment():
    """`DT.to_device(); DT.key = "b"` reorders columns (key first) and rows in place -- the resident
    copies must not survive it (they held the old column and row order): the by() query agrees with a host frame"""
    import numpy as np
    from datatable_amd.frame import Frame, f, by, sum, count   # noqa: A004
    rng = np.random.default_rng(4)
    n = 50_000
    a = rng.integers(0, 100, n).astype(np.int32)
    b = rng.permutation(n).astype(np.int64)           # unique: a valid key
    v = rng.standard_normal(n)
    D = Frame(a=a, b=b, v=v).to_device()
    D.key = "b"
    H = Frame(a=a, b=b, v=v)
    H.key = "b"
    assert D.names == H.names == ("b", "a", "v")
    R1, R2 = D[:, [sum(f.v), count()], by(f.a)], H[:, [sum(f.v), count()], by(f.a)]
    assert R1.names == R2.names and R1.stypes == R2.stypes
    for c1, c2 in zip(R1.to_numpy_columns(), R2.to_numpy_columns()):
        assert np.allclose(c1, c2, rtol=1e-9, atol=1e-9) if c1.dtype.kind == "f" else np.array_equal(c1, c2)
