"""Tests ported from the reference's own suite (h2oai/datatable tests/test-groups.py,
tests/test-reduce.py, tests/ijby/test-sort.py) -- same inputs, same expected values, spelled
the same way against datatable_amd.frame.  String columns and computed f-expressions
(f.A + f.B, prod, ...) are outside the accelerated path and are left out or replaced by an
integer column where the test is about grouping, not strings.  file:line of every original is
given.  Every evaluation runs on the GPU through the C ABI."""
import builtins
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
builtins_sum, builtins_min = builtins.sum, builtins.min


@pytest.fixture(scope="module", params=["lsd", "msd"])
def dt(request):
    """datatable_amd.frame on the calling thread's default context, once per route of the sort path: LSD radix passes, and
    the MSD levels forced onto every input (conftest.set_sort_route) -- the ported reference cases must hold on both"""
    from conftest import set_sort_route
    from datatable_amd import frame
    from datatable_amd.engine import default_context
    set_sort_route(default_context(), request.param)
    yield frame
    set_sort_route(default_context(), "default")


def assert_equals(A, B):
    assert A.names == B.names, (A.names, B.names)
    assert A.stypes == B.stypes, (A.stypes, B.stypes)
    la, lb = A.to_list(), B.to_list()
    assert len(la) == len(lb)
    for ca, cb in zip(la, lb):
        assert len(ca) == len(cb)
        for x, y in zip(ca, cb):
            if isinstance(x, float) and isinstance(y, float):
                assert x == y or (math.isnan(x) and math.isnan(y)) or abs(x - y) <= 1e-12 * max(abs(x), abs(y)), (x, y)
            else:
                assert x == y, (ca, cb)


# ---- tests/test-groups.py -------------------------------------------------------------------

def test_groups1a(dt):                                   # test-groups.py:33-36
    DT0 = dt.Frame(A=[1, 2, 1])
    DT1 = DT0[:, "A", dt.by("A")]
    assert_equals(DT1, dt.Frame([[1, 2], [1, 2]], names=["A", "A.0"]))


def test_groups1b(dt):                                   # test-groups.py:39-52 (string column -> int codes)
    DT = dt.Frame([[1, 5, 3, 2, 1, 3, 1, 1, None], [10, 20, 30, 10, None, 60, 20, 80, 40]], names=["A", "B"])
    assert_equals(DT[:, :, dt.by("A")], dt.Frame(A=[None, 1, 1, 1, 1, 2, 3, 3, 5], B=[40, 10, None, 20, 80, 10, 30, 60, 20]))
    assert_equals(DT[:, :, dt.by("B")], dt.Frame(B=[None, 10, 10, 20, 20, 30, 40, 60, 80], A=[1, 1, 2, 5, 1, 3, None, 3, 1]))


def test_groups3(dt):                                    # test-groups.py:72-93
    random.seed(3)
    n = 100000
    src = [random.getrandbits(10) for _ in range(n)]
    f0 = dt.Frame(A=src, B=list(range(n)))
    f1 = f0[:, :, dt.by("A")]
    assert f1.shape == (n, 2)
    assert f1.names == ("A", "B")
    f1A, f1B = f1.to_list()
    f1A.append(None)
    curr_value, group_start = f1A[0], 0
    for i in range(n + 1):
        if f1A[i] != curr_value:
            group = f1B[group_start:i]
            assert group == sorted(group)
            curr_value, group_start = f1A[i], i
    assert group_start == n and curr_value is None


def test_group_slice_all(dt):                            # test-groups.py:119-127 (string column C dropped)
    f = dt.f
    DT = dt.Frame([[1, 2, 3, 4, 5, 6], [3, 0, 3, 3, 1, 0]], names=["A", "B"])
    RES = dt.Frame(B=[0, 0, 1, 3, 3, 3], A=[2, 6, 5, 1, 3, 4])
    assert_equals(DT[:, :, dt.by(f.B)], RES)
    assert_equals(DT[:, f[:], dt.by(f.B)], RES)


def test_group_reduce_all_columns(dt):                   # test-groups.py:128-141
    f = dt.f
    DT = dt.Frame(id=[3, 3, 3, 3, 4, 4, 4, 4],
                  beef=[23, None, None, None, None, None, None, None],
                  eggs=[None, 33, None, None, 197, 103, None, None],
                  fork=[None, None, 10, None, None, None, 210, None],
                  veg=[17, None, None, 40, 1, 2, None, 340])
    I64 = 5
    assert_equals(DT[:, dt.sum(f[:]), dt.by(f.id)],
                  dt.Frame(dict(id=[3, 4], beef=[23, 0], eggs=[33, 300], fork=[10, 210], veg=[57, 343]),
                           stypes={"beef": I64, "eggs": I64, "fork": I64, "veg": I64}))


def test_group_reverse_flag(dt):                         # test-groups.py:144-156
    f = dt.f
    DT = dt.Frame({"A": [1, 2, 1, 2, 2, 3, 3], "B": [2, 2, 4, 4, 23, 5, 30]})
    EXPECTED = DT[:, :, dt.by(f.A), dt.sort(-f.B)]
    RES1 = DT[:, :, dt.by("A"), dt.sort("B", reverse=True)]
    RES2 = DT[:, :, dt.by(f.A), dt.sort(f.B, reverse=True)]
    assert_equals(EXPECTED, RES1)
    assert_equals(RES1, RES2)
    assert RES1.to_list() == [[1, 1, 2, 2, 2, 3, 3], [4, 2, 23, 4, 2, 30, 5]]


def test_group_negate_column(dt):                        # test-groups.py:159-176
    f = dt.f
    DT = dt.Frame({"A": [1, 2, 1, 2, 2, 3, 3], "B": [2, 2, 4, 4, 23, 5, 30]})
    EXPECTED = dt.Frame({"A": [3, 3, 2, 2, 2, 1, 1], "B": [30, 5, 23, 4, 2, 4, 2]})
    assert_equals(EXPECTED, DT[:, :, dt.by(-f.A), dt.sort(-f.B)])
    assert_equals(EXPECTED, DT[:, :, dt.by(-f.A), dt.sort(f.B, reverse=True)])


def test_group_empty_frames(dt):                         # test-groups.py:182-209
    f = dt.f
    DT = dt.Frame(A=np.zeros(0, np.int32))
    assert DT.shape == (0, 1)
    assert DT[:, :, dt.by(f.A)].shape == (0, 1)
    D1 = DT[:, dt.count(), dt.by(f.A)]
    assert D1.shape == (0, 2) and D1.stypes == (4, 5)
    DF = dt.Frame(A=np.zeros(0, np.float32))
    D2 = DF[:, dt.count(f.A), dt.by(f.A)]
    assert D2.shape == (0, 2) and D2.stypes == (6, 5)
    D3 = DF[:, dt.sum(f.A), dt.by(f.A)]
    assert D3.shape == (0, 2) and D3.stypes == (6, 6)


def test_groups_small1(dt):                              # test-groups.py:211-217
    f = dt.f
    DT0 = dt.Frame({"A": [1, 2, 1, 2, 1, 3, 1, 1], "B": [0, 1, 2, 3, 4, 5, 6, 7]})
    DT1 = DT0[:, dt.mean(f.B), dt.by(f.A)]
    assert_equals(DT1, dt.Frame(A=[1, 2, 3], B=[3.8, 2.0, 5.0]))
    assert_equals(DT0[:, dt.mean(f.B), "A"], DT1)


def test_groups_multiple(dt):                            # test-groups.py:220-225 (color -> codes blue=0, green=1, red=2)
    f = dt.f
    f0 = dt.Frame({"color": [2, 0, 1, 2, 1], "size": [5, 2, 7, 13, 0]})
    f1 = f0[:, [dt.min(f.size), dt.max(f.size)], "color"]
    assert f1.to_list() == [[0, 1, 2], [2, 0, 5], [2, 7, 13]]


def test_groups_autoexpand(dt):                          # test-groups.py:228-235
    f = dt.f
    f0 = dt.Frame({"color": [2, 0, 1, 2, 1], "size": [5, 2, 7, 13, 0]})
    f1 = f0[:, [dt.mean(f.size), f.size], f.color]
    assert f1.to_list() == [[0, 1, 1, 2, 2], [2.0, 3.5, 3.5, 9.0, 9.0], [2, 7, 0, 5, 13]]


def test_group_boolean(dt):                              # test-groups.py:244-262
    f = dt.f
    DT = dt.Frame(A=[True, None, False, False, True, True, False, True])
    assert_equals(DT[:, dt.count(), dt.by(f.A)], dt.Frame(dict(A=[None, False, True], count=[1, 3, 4]), stypes={"count": 5}))
    DT = dt.Frame(A=[True, False, False] * 500 + [None, True])
    assert_equals(DT[:, dt.count(), dt.by(f.A)], dt.Frame(dict(A=[None, False, True], count=[1, 1000, 501]), stypes={"count": 5}))
    DT = dt.Frame(A=[True] * 1234)
    assert_equals(DT[:, dt.count(), dt.by(f.A)], dt.Frame(dict(A=[True], count=[1234]), stypes={"count": 5}))


def test_group_boolean4(dt):                             # test-groups.py:265-272
    f = dt.f
    n = 43701
    DT = dt.Frame(A=list(range(2 * n)), B=[False, True] * n)
    DTR = DT[:, dt.sum(f.A), dt.by(f.B)]
    assert_equals(DTR, dt.Frame(dict(B=[False, True], A=[sum(range(0, 2 * n, 2)), sum(range(1, 2 * n, 2))]), stypes={"A": 5}))


def test_reduce_sum(dt):                                 # test-groups.py:280-286
    f = dt.f
    f0 = dt.Frame({"color": [2, 0, 1, 2, 1], "size": [5, 2, 7, 13, -1]})
    assert f0[:, dt.sum(f.size), f.color].to_list() == [[0, 1, 2], [2, 6, 18]]


def test_reduce_sum_same_column(dt):                     # test-groups.py:289-294 (issue #3110)
    f = dt.f
    f0 = dt.Frame({"ints": [0, 1, 0, 0, 1, 2]})
    f1 = f0[:, {"sum": dt.sum(f.ints)}, f.ints]
    assert_equals(f1, dt.Frame({"ints": [0, 1, 2], "sum": [0, 2, 2]}, stypes={"sum": 5}))


def test_groups_large1(dt):                              # test-groups.py:318-323
    n = 251 * 4000
    xs = [(i * 19) % 251 for i in range(n)]
    f0 = dt.Frame({"A": xs})
    f1 = f0[:, dt.count(), dt.by("A")]
    assert f1.to_list() == [list(range(251)), [4000] * 251]


# ---- tests/test-reduce.py -------------------------------------------------------------------

SRC_I = [9, 8, 2, 3, None, None, 3, 0, 5, 5, 8, None, 1]
SRC_J = [0, 1, 0, 5, 3, 8, 1, 0, 2, 5, None, 8, 1]
INT_STYPES = [2, 3, 4, 5]            # int8, int16, int32, int64
REAL_STYPES = [6, 7]


def test_count_dt_integer(dt):                           # test-reduce.py:75-81
    f = dt.f
    df_in = dt.Frame([SRC_I])
    df_reduce = df_in[:, [dt.count(f.C0), dt.count()]]
    assert df_reduce.shape == (1, 2)
    assert df_reduce.to_list() == [[10], [13]]


def test_count_dt_groupby_integer(dt):                   # test-reduce.py:84-92
    f = dt.f
    df_in = dt.Frame([SRC_I])
    df_reduce = df_in[:, [dt.count(f.C0), dt.count()], "C0"]
    assert df_reduce.shape == (8, 3)
    assert df_reduce.to_list() == [[None, 0, 1, 2, 3, 5, 8, 9], [0, 1, 1, 1, 2, 2, 2, 1], [3, 1, 1, 1, 2, 2, 2, 1]]


def test_count_2d_dt_integer(dt):                        # test-reduce.py:102-110
    f = dt.f
    df_in = dt.Frame([SRC_I, SRC_J])
    df_reduce = df_in[:, [dt.count(f.C0), dt.count(f.C1), dt.count()]]
    assert df_reduce.shape == (1, 3)
    assert df_reduce.to_list() == [[10], [12], [13]]


def test_count_2d_dt_groupby_integer(dt):                # test-reduce.py:113-123
    f = dt.f
    df_in = dt.Frame([SRC_I, SRC_J])
    df_reduce = df_in[:, [dt.count(f.C0), dt.count(f.C1), dt.count()], "C0"]
    assert df_reduce.shape == (8, 4)
    assert df_reduce.to_list() == [[None, 0, 1, 2, 3, 5, 8, 9], [0, 1, 1, 1, 2, 2, 2, 1], [3, 1, 1, 1, 2, 2, 1, 1],
                                   [3, 1, 1, 1, 2, 2, 2, 1]]


@pytest.mark.parametrize("mm", ["min", "max"])
@pytest.mark.parametrize("st", INT_STYPES)
def test_minmax_integer(dt, mm, st):                     # test-reduce.py:286-291
    src = [0, 23, 100, 99, -11, 24, -1]
    DT = dt.Frame(A=src, stypes={"A": st})
    assert DT[:, getattr(dt, mm)(dt.f.A)].to_list() == [[{"min": min, "max": max}[mm](src)]]


@pytest.mark.parametrize("mm", ["min", "max"])
@pytest.mark.parametrize("st", INT_STYPES)
def test_minmax_integer_grouped(dt, mm, st):             # test-reduce.py:294-299
    src = [3, 2, 2, 2, 2, 3, -100, 15, -100]
    DT = dt.Frame(A=src, stypes={"A": st})
    assert DT[:, getattr(dt, mm)(dt.f.A), dt.by(dt.f.A)].to_list() == [[-100, 2, 3, 15]] * 2


@pytest.mark.parametrize("mm", ["min", "max"])
def test_minmax_real(dt, mm):                            # test-reduce.py:302-306
    src = [5.6, 12.99, 1e+12, -3.4e-22, math.nan, 0.0]
    pm = {"min": min, "max": max}[mm]
    DT = dt.Frame(A=src)
    assert DT[:, getattr(dt, mm)(dt.f.A)].to_list() == [[pm(x for x in src if x == x)]]


@pytest.mark.parametrize("mm", ["min", "max"])
def test_minmax_infs(dt, mm):                            # test-reduce.py:309-314
    src = [math.nan, 1.0, 2.5, -math.inf, 3e199, math.inf]
    answer = -math.inf if mm == "min" else +math.inf
    assert dt.Frame(A=src)[:, getattr(dt, mm)(dt.f.A)].to_list() == [[answer]]


@pytest.mark.parametrize("mm", ["min", "max"])
@pytest.mark.parametrize("src", [[math.inf], [-math.inf]])
def test_minmax_infs_only(dt, mm, src):                  # test-reduce.py:317-321
    assert dt.Frame(A=src)[:, getattr(dt, mm)(dt.f.A)].to_list() == [src]


@pytest.mark.parametrize("mm", ["min", "max"])
@pytest.mark.parametrize("st", INT_STYPES + REAL_STYPES)
def test_minmax_empty(dt, mm, st):                       # test-reduce.py:324-328
    DT1 = dt.Frame(A=[], stypes={"A": st})
    assert DT1[:, getattr(dt, mm)(dt.f.A)].to_list() == [[None]]


@pytest.mark.parametrize("mm", ["min", "max"])
@pytest.mark.parametrize("st", INT_STYPES + REAL_STYPES)
def test_minmax_nas(dt, mm, st):                         # test-reduce.py:331-335
    DT2 = dt.Frame(B=[None] * 3, stypes={"B": st})
    assert DT2[:, getattr(dt, mm)(dt.f.B)].to_list() == [[None]]


def test_sum_void_like(dt):                              # test-reduce.py:405-423 (void column -> all-NA int32)
    f = dt.f
    DT = dt.Frame([[None] * 10])
    assert_equals(DT[:, dt.sum(f.C0)], dt.Frame([[0]], stypes=[5]))
    DT = dt.Frame([[None, None, None, None, None], [1, 2, 1, 2, 2]])
    assert_equals(DT[:, dt.sum(f.C0), dt.by(f.C1)], dt.Frame(dict(C1=[1, 2], C0=[0, 0]), stypes={"C0": 5}))
    R = DT[:, dt.sum(f.C0), dt.by(f.C0)]
    assert R.to_list() == [[None], [0]] and R.stypes == (4, 5)


def test_sum_simple(dt):                                 # test-reduce.py:426-432
    DT = dt.Frame(A=list(range(5)))
    assert DT[:, dt.sum(dt.f.A)].to_list() == [[10]]


def test_sum_empty_frame(dt):                            # test-reduce.py:435-447 (void column left out)
    f = dt.f
    DT = dt.Frame([[], [], [], []], names=list("ABCD"), stypes=[1, 4, 6, 7])
    assert DT.shape == (0, 4)
    DT_sum = DT[:, dt.sum(f[:])]
    assert DT_sum.shape == (1, 4)
    assert DT_sum.names == ("A", "B", "C", "D")
    assert DT_sum.stypes == (5, 5, 6, 7)
    assert DT_sum.to_list() == [[0], [0], [0], [0]]


def test_sum_grouped(dt):                                # test-reduce.py:450-457
    f = dt.f
    DT = dt.Frame(A=[True, False, True, True], B=[None, None, None, 10], C=[2, 3, 5, -5])
    DT_sum = DT[:, dt.sum(f[:]), dt.by(f.A)]
    assert_equals(DT_sum, dt.Frame(dict(A=[False, True], B=[0, 10], C=[3, 2]), stypes={"B": 5, "C": 5}))


def test_first_last_frame(dt):                           # tests/test-reduce.py first/last semantics (head_reduce_unary.cc:116-190)
    f = dt.f
    DT = dt.Frame(A=[1, 2, 1, 2, 1, 3], B=[None, 5, 7, None, 9, 11])
    R = DT[:, [dt.first(f.B), dt.last(f.B)], dt.by(f.A)]
    assert R.names == ("A", "B", "B.0")
    assert R.to_list() == [[1, 2, 3], [None, 5, 11], [9, None, 11]]
    assert DT[:, [dt.first(f.B), dt.last(f.B)]].to_list() == [[None], [11]]
    assert dt.Frame(A=[], stypes={"A": 4})[:, dt.first(f.A)].to_list() == [[None]]


# ---- tests/test-reduce.py: median / cov / corr / sd; tests/dt/test-cum*.py, test-nunique.py ---------
#   `[..]/dt.int64` of the originals is spelled stypes=[...] here.

I64, F32, F64, I32, I8, I16, B8 = 5, 6, 7, 4, 2, 3, 1


def test_median_bool_even_nrows(dt):                    # test-reduce.py:592-597
    RES = dt.Frame(A=[True, False, True, False])[:, dt.median(dt.f.A)]
    assert RES.shape == (1, 1) and RES.stypes == (F64,) and RES.to_list() == [[0.5]]


def test_median_bool_odd_nrows(dt):                     # test-reduce.py:600-605
    RES = dt.Frame(B=[True, False, True])[:, dt.median(dt.f.B)]
    assert RES.stypes == (F64,) and RES.to_list() == [[1.0]]


def test_median_bygroup(dt):                            # test-reduce.py:608-613
    DT = dt.Frame(A=[0.1, 0.2, 0.5, 0.4, 0.3, 0], B=[1, 2, 1, 1, 2, 2])
    assert DT[:, dt.median(dt.f.A), dt.by(dt.f.B)].to_list() == [[1, 2], [0.4, 0.2]]


@pytest.mark.parametrize("st", [I8, I16, I32, I64])
def test_median_int(dt, st):                            # test-reduce.py:616-633
    DT = dt.Frame(A=[7, 11, -2, 3, 0, 12, 12, 3, 5, 91], stypes=[st])
    RES = DT[:, dt.median(dt.f.A)]
    assert RES.shape == (1, 1) and RES.stypes == (F64,) and RES.to_list() == [[6.0]]
    DT = dt.Frame(A=[4, -5, 12, 11, 4, 7, 0, 23, 45, 8, 10], stypes=[st])
    assert DT[:, dt.median(dt.f.A)].to_list() == [[8.0]]


def test_median_int_no_overflow(dt):                    # test-reduce.py:636-641
    assert dt.Frame(A=[111, 112], stypes=[I8])[:, dt.median(dt.f.A)].to_list() == [[111.5]]


@pytest.mark.parametrize("st", [F32, F64])
def test_median_float(dt, st):                          # test-reduce.py:644-650
    RES = dt.Frame(W=[0.0, 5.5, 7.9, math.inf, -math.inf], stypes=[st])[:, dt.median(dt.f.W)]
    assert RES.stypes == (st,) and RES.to_list() == [[5.5]]


def test_median_all_and_some_nas(dt):                   # test-reduce.py:653-666
    RES = dt.Frame(N=[math.nan] * 8)[:, dt.median(dt.f.N)]
    assert RES.stypes == (F64,) and RES.to_list() == [[None]]
    RES = dt.Frame(S=[None, 5, None, 12, None, -3, None, None, None, 4])[:, dt.median(dt.f.S)]
    assert RES.stypes == (F64,) and RES.to_list() == [[4.5]]


def test_median_grouped(dt):                            # test-reduce.py:669-676
    DT = dt.Frame(A=[0, 0, 0, 0, 1, 1, 1, 1, 1], B=[2, 6, 1, 0, -3, 4, None, None, -1], stypes={"A": I16, "B": I32})
    RES = DT[:, dt.median(dt.f.B), dt.by(dt.f.A)]
    assert RES.shape == (2, 2) and RES.stypes == (I16, F64) and RES.to_list() == [[0, 1], [1.5, -1.0]]


def test_cov_simple_small_float32(dt):                  # test-reduce.py:710-733
    DT = dt.Frame(A=list(range(5)), B=list(range(5)))
    assert_equals(DT[:, dt.cov(dt.f.A, dt.f.B)], dt.Frame([2.5]))
    assert_equals(dt.Frame(A=[1], B=[2])[:, dt.cov(dt.f.A, dt.f.B)], dt.Frame([None], stypes=[F64]))
    DT = dt.Frame(A=[1.0, 2.0, 3.0], B=[7.5, 7.0, 6.5], stypes=[F32, F32])
    assert_equals(DT[:, dt.cov(dt.f.A, dt.f.B)], dt.Frame([-0.5], stypes=[F32]))


def test_cov_bygroup(dt):                               # test-reduce.py:736-739
    DT = dt.Frame(ID=[1, 2, 1, 2, 1, 2], A=[0, 5, 10, 20, 2, 8])
    assert_equals(DT[:, dt.cov(dt.f.A, dt.f.A), dt.by(dt.f.ID)], dt.Frame(ID=[1, 2], C0=[28.0, 63.0]))


@pytest.mark.parametrize("seed", [11, 12])
def test_cov_corr_random(dt, seed):                     # test-reduce.py:742-752,785-794
    rng = np.random.default_rng(seed)
    arr1, arr2 = rng.random(100), rng.random(100)
    DT = dt.Frame([arr1, arr2])
    assert np.isclose(np.cov(arr1, arr2)[0, 1], DT[:, dt.cov(dt.f[0], dt.f[1])].to_list()[0][0], atol=1e-12, rtol=1e-12)
    assert np.isclose(np.corrcoef(arr1, arr2)[0, 1], DT[:, dt.corr(dt.f[0], dt.f[1])].to_list()[0][0], atol=1e-12, rtol=1e-12)


def test_corr_simple_and_constant(dt):                  # test-reduce.py:760-782
    DT = dt.Frame(A=list(range(5)), B=list(range(5)))
    assert_equals(DT[:, dt.corr(dt.f.A, dt.f.B)], dt.Frame([1.0]))
    DT = dt.Frame(A=list(range(5)), B=list(range(5, 0, -1)))
    assert_equals(DT[:, dt.corr(dt.f.A, dt.f.B)], dt.Frame([-1.0]))
    assert_equals(dt.Frame(A=[1], B=[2])[:, dt.corr(dt.f.A, dt.f.B)], dt.Frame([None], stypes=[F64]))
    DT = dt.Frame(A=list(range(23)), B=[2.5] * 23)
    assert_equals(DT[:, dt.corr(dt.f.A, dt.f.B)], dt.Frame([None], stypes=[F64]))


def test_corr_multiple(dt):                             # test-reduce.py:797-806
    DT = dt.Frame(A=[3, 5, 9, 1], B=[4, 7, 0, 0], C=[3, 2, 1, 0], D=list(range(4)))
    a, b, c = -0.07168504827326534, 0.07559289460184544, 0.7207110797203374
    assert_equals(DT[:, dt.corr(dt.f.A, dt.f[:])], dt.Frame([[1.0], [a], [b], [-b]]))
    assert_equals(DT[:, dt.corr(dt.f[:], dt.f.D)], dt.Frame([[-b], [-c], [-1.0], [1.0]]))
    assert_equals(DT[:, dt.corr(dt.f[:], dt.f[:])], dt.Frame([[1.0], [1.0], [1.0], [1.0]]))


def test_sd(dt):                                        # test-reduce.py:920-941
    DT = dt.Frame([[3], [None], [1], [5], [None]])
    assert_equals(DT[:, dt.sd(dt.f[:])], dt.Frame([[None]] * 5, stypes=[F64] * 5))
    DT = dt.Frame([[1] * 10, [-1.1] * 10, [0] * 10, [4.3] * 10, [300] * 10])
    assert_equals(DT[:, dt.sd(dt.f[:])], dt.Frame([[0.0]] * 5))
    DT = dt.Frame([[1.5, 6.4, 0.0, None, 7.22], [2.0, -1.1, math.inf, 4.0, 3.2], [1.5, 9.9, None, None, math.nan],
                   [math.inf, -math.inf, None, 0.0, math.nan]])
    assert_equals(DT[:, dt.sd(dt.f[:])],
                  dt.Frame([[3.5676696409094086], [None], [5.939696961966999], [None]], stypes=[F64] * 4))


def test_sd_per_group(dt):                              # test-reduce.py:907-911 (void column -> all-NA int column)
    DT = dt.Frame([[None, None, None, None, None], [1, 2, 1, 2, 2]], stypes=[I32, I32])
    assert_equals(DT[:, dt.sd(dt.f.C0), dt.by(dt.f.C1)], dt.Frame(C1=[1, 2], C0=[None, None], stypes={"C0": F64}))


def test_cumsum(dt):                                    # tests/dt/test-cumsum.py:78-112
    DT = dt.Frame([0], stypes=[I64])
    assert_equals(DT[:, dt.cumsum(dt.f[:])], DT)
    DT = dt.Frame([list(range(5)), [-1, 1, None, 2, 5.5]])
    assert_equals(DT[:, dt.cumsum(dt.f[:])], dt.Frame([[0, 1, 3, 6, 10], [-1, 0, 0, 2, 7.5]], stypes=[I64, F64]))
    assert_equals(DT[:, dt.cumsum(dt.f[:], reverse=True)], dt.Frame([[10, 10, 9, 7, 4], [7.5, 8.5, 7.5, 7.5, 5.5]], stypes=[I64, F64]))
    DT = dt.Frame([[2, 1, 1, 1, 2], [1.5, -1.5, math.inf, 2, 3]])
    assert_equals(DT[:, dt.cumsum(dt.f[:]), dt.by(dt.f[0])], dt.Frame([[1, 1, 1, 2, 2], [-1.5, math.inf, math.inf, 1.5, 4.5]]))
    assert_equals(DT[:, dt.cumsum(dt.f[:], reverse=True), dt.by(dt.f[0])],
                  dt.Frame([[1, 1, 1, 2, 2], [math.inf, math.inf, 2.0, 4.5, 3.0]]))


def test_cumsum_grouped_column(dt):                     # tests/dt/test-cumsum.py:120-124
    DT = dt.Frame([2, 1, None, 1, 2])
    assert_equals(DT[:, dt.cumsum(dt.f[0]), dt.by(dt.f[0])],
                  dt.Frame([[None, 1, 1, 2, 2], [0, 1, 2, 2, 4]], stypes=[I32, I64]))


def test_cumprod(dt):                                   # tests/dt/test-cumprod.py:87-112
    DT = dt.Frame([list(range(5)), [-1, 1, None, 2, 5.5]])
    assert_equals(DT[:, dt.cumprod(dt.f[:])], dt.Frame([[0, 0, 0, 0, 0], [-1, -1, -1, -2, -11]], stypes=[I64, F64]))
    DT = dt.Frame([[2, 1, 1, 1, 2], [1.5, -1.5, math.inf, 2, 3]])
    assert_equals(DT[:, dt.cumprod(dt.f[:]), dt.by(dt.f[0])], dt.Frame([[1, 1, 1, 2, 2], [-1.5, -math.inf, -math.inf, 1.5, 4.5]]))
    assert_equals(DT[:, dt.cumprod(dt.f[:], reverse=True), dt.by(dt.f[0])],
                  dt.Frame([[1, 1, 1, 2, 2], [-math.inf, math.inf, 2.0, 4.5, 3.0]]))


def test_cumminmax(dt):                                 # tests/dt/test-cumminmax.py:97-159
    DT = dt.Frame([None, False, None, True, False, True])
    assert_equals(DT[:, [dt.cummin(dt.f[:]), dt.cummax(dt.f[:])]],
                  dt.Frame([[None, False, False, False, False, False], [None, False, False, True, True, True]]))
    DT = dt.Frame([list(range(5)), [None, -1, None, 5.5, 3]])
    assert_equals(DT[:, [dt.cummin(dt.f[:]), dt.cummax(dt.f[:])]],
                  dt.Frame([[0, 0, 0, 0, 0], [None, -1, -1, -1, -1], [0, 1, 2, 3, 4], [None, -1, -1, 5.5, 5.5]],
                           stypes=[I32, F64, I32, F64]))
    DT = dt.Frame([[2, 1, 1, 1, 2], [1.5, -1.5, math.inf, None, 3]])
    assert_equals(DT[:, [dt.cummin(dt.f[:]), dt.cummax(dt.f[:])], dt.by(dt.f[0])],
                  dt.Frame([[1, 1, 1, 2, 2], [-1.5, -1.5, -1.5, 1.5, 1.5], [-1.5, math.inf, math.inf, 1.5, 3]],
                           names=["C0", "C1", "C2"]))


def test_cumminmax_grouped_column_and_reverse(dt):      # tests/dt/test-cumminmax.py:162-170,194-203
    DT = dt.Frame([2, 1, None, 1, 2])
    assert_equals(DT[:, [dt.cummin(dt.f[0]), dt.cummax(dt.f[0])], dt.by(dt.f[0])],
                  dt.Frame([[None, 1, 1, 2, 2]] * 3))
    DT = dt.Frame([[3, 14, 15, 92, 6], [0, 1, 0, 2, 1]])           # string key of the original -> its rank
    assert_equals(DT[:, [dt.cummin(dt.f[0], reverse=True), dt.cummax(dt.f[0], True)], dt.by(dt.f[1])],
                  dt.Frame([[0, 0, 1, 1, 2], [3, 15, 6, 6, 92], [15, 15, 14, 6, 92]], names=["C1", "C0", "C2"]))


def test_cumcount_ngroup(dt):                           # tests/dt/test-cumcountngroup.py:76-100
    DT = dt.Frame([0], stypes=[I64])
    assert_equals(DT[:, [dt.cumcount(True), dt.cumcount(False), dt.ngroup(True), dt.ngroup(False)]],
                  dt.Frame([[0]] * 4, stypes=[I64] * 4))
    DT = dt.Frame([0, 0, 0, 1, 1, 0])                                # 'a'/'b' of the original -> 0/1
    assert_equals(DT[:, [dt.cumcount(False), dt.cumcount(True), dt.ngroup(True), dt.ngroup(False)]],
                  dt.Frame([list(range(6)), list(range(5, -1, -1)), [0] * 6, [0] * 6], stypes=[I64] * 4))
    assert_equals(DT[:, [dt.cumcount(False), dt.ngroup(True)], dt.by(dt.f[0])],
                  dt.Frame([[0, 0, 0, 0, 1, 1], [0, 1, 2, 3, 0, 1], [1, 1, 1, 1, 0, 0]], names=["C0", "C1", "C2"],
                           stypes=[I32, I64, I64]))


def test_nunique_with_by(dt):                           # tests/dt/test-nunique.py:69-93
    DT = dt.Frame(G=[1, 1, 1, 2, 2, 2], V=[None, None, None, None, 3, 5], N=[None] * 6, stypes={"N": I32})
    RES = DT[:, {"V1": dt.nunique(dt.f.V), "V3": dt.nunique(dt.f.N)}, dt.by(dt.f.G)]
    assert_equals(RES, dt.Frame(G=[1, 2], V1=[0, 2], V3=[0, 0], stypes={"V1": I64, "V3": I64}))
    DT = dt.Frame([1, None, 1, 2, None, None])
    assert_equals(DT[:, {"nunique": dt.nunique(dt.f[0])}, dt.by(dt.f[0])],
                  dt.Frame(C0=[None, 1, 2], nunique=[0, 1, 1], stypes={"nunique": I64}))


# ---- tests/test-sets.py, tests/test-keys.py, tests/test-join.py ----------------
# string columns of the originals are replaced by integer codes that sort the same way

def _set_fns(dt):
    return [dt.union, dt.intersect, dt.setdiff, dt.symdiff]


def test_setfns_basic(dt):                              # test-sets.py:38-100
    for fn in _set_fns(dt):
        assert fn().shape == (0, 0)
        res = fn(dt.Frame([1, 2, 3, 1]))
        assert res.shape == (3, 1) and res.to_list() == [[1, 2, 3]]
        dt0, dt1, dt2 = dt.Frame([1, 2, 3, 4, 5]), dt.Frame([3, 5, 7, 9]), dt.Frame([2, 7, 11])
        r1, r2 = fn(dt0, dt1, dt2), fn([dt0, dt1, dt2])
        assert r1.names == r2.names and r1.stypes == r2.stypes and r1.to_list() == r2.to_list()
        a, b = dt.Frame(A=[2, 3, 5]), dt.Frame(B=list(range(4)))
        assert fn(a, b).names == ("A",) and fn(b, a).names == ("B",)
        d1, d2 = dt.Frame([2, 5, 7, 2, 3]), dt.Frame([3, 4, 2, 5])
        assert fn(d1, d2).to_list() == fn(d1, dt.Frame(), d2).to_list()
        assert fn(dt.Frame(), dt.Frame()).shape == (0, 0)


def test_union_intersect_setdiff_symdiff(dt):           # test-sets.py:129-207
    dt1, dt2, dt3 = dt.Frame([2, 5, 7, 2, 3]), dt.Frame([3, 4, 2, 5]), dt.Frame([0, 3, 2, 2, 2, 2, 2, 2, 0])
    assert dt.union(dt1, dt2).to_list() == [[2, 3, 4, 5, 7]]
    assert dt.union(dt3, dt1, dt2).to_list() == [[0, 2, 3, 4, 5, 7]]
    assert dt.intersect(dt1, dt2).to_list() == [[2, 3, 5]]
    assert dt.intersect(dt3, dt1, dt2).to_list() == [[2, 3]]
    assert dt.setdiff(dt1, dt2).to_list() == [[7]]
    assert dt.symdiff(dt1, dt2).to_list() == [[4, 7]]
    dt1 = dt.Frame([2, 5, 7, 2, 3, 6, 0])
    assert dt.setdiff(dt1, dt2, dt3).to_list() == [[6, 7]]
    assert dt.symdiff(dt1, dt2, dt3).to_list() == [[2, 3, 4, 6, 7]]
    with pytest.raises(ValueError, match="Only single-column Frames are allowed"):
        dt.union(dt.Frame(A=[1], B=[2]))
    with pytest.raises(TypeError, match=r"union\(\) expects a list or sequence of Frames"):
        dt.union("a")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_setfns_random(dt, seed):                       # test-sets.py:217-260
    rng = random.Random(seed)
    srcs = [[rng.randint(1, 20) for _ in range(rng.randint(1, 30))] for _ in range(rng.randint(2, 5))]
    frames = [dt.Frame(s) for s in srcs]
    sets = [set(s) for s in srcs]
    assert dt.union(frames).to_list() == [sorted(set.union(*sets))]
    assert dt.intersect(frames).to_list() == [sorted(set.intersection(*sets))]
    assert dt.setdiff(frames).to_list() == [sorted(sets[0].difference(*sets[1:]))]
    sym = set()
    for x in sets:
        sym = sym.symmetric_difference(x)
    assert dt.symdiff(frames).to_list() == [sorted(sym)]


def test_unique(dt):                                    # tests/test-sets.py via dt.unique; set_funcs.cc:180-193
    assert dt.unique(dt.Frame(A=[3, None, 1, 3, None])).to_dict() == {"A": [None, 1, 3]}
    res = dt.unique(dt.Frame(A=[1, 2, 2], B=[2.5, 1.0, None]))            # all columns, up-cast to float64
    assert res.names == ("C0",) and res.stypes == (F64,) and res.to_list() == [[None, 1.0, 2.0, 2.5]]


def test_keys_simple_and_multi(dt):                     # test-keys.py:31-55,78-87
    dt0 = dt.Frame([[2, 4, 3, 0, 1], [1, 5, 15, 12, 8], [3.6, 9.78, 2.01, -4.23, 5.3819]], names=["name", "sex", "avg"])
    assert dt0.key == tuple()
    dt0.key = "name"
    assert dt0.key == ("name",) and dt0.shape == (5, 3) and dt0.names == ("name", "sex", "avg")
    assert dt0.to_list() == [[0, 1, 2, 3, 4], [12, 8, 1, 15, 5], [-4.23, 5.3819, 3.6, 2.01, 9.78]]
    dt0.key = "sex"
    assert dt0.key == ("sex",) and dt0.names == ("sex", "name", "avg")
    assert dt0.to_list() == [[1, 5, 8, 12, 15], [2, 4, 1, 0, 3], [3.6, 9.78, 5.3819, -4.23, 2.01]]
    dt0.key = None
    assert dt0.key == tuple()
    dt0 = dt.Frame(D=list(range(6)), A=[3, 7, 5, 2, 2, 3], B=[1, 2, 2, 3, 4, 4])
    dt0.key = ["A", "B"]
    assert dt0.key == ("A", "B") and dt0.names == ("A", "B", "D")
    assert dt0.to_list() == [[2, 2, 3, 3, 5, 7], [3, 4, 1, 4, 2, 2], [3, 4, 0, 5, 2, 1]]


def test_key_invalid(dt):                               # test-keys.py:90-133
    dt0 = dt.Frame(A=list(range(5)), B=[3] * 5)
    with pytest.raises(TypeError, match="Key should be a column name, or a list/tuple of column names"):
        dt0.key = 0
    with pytest.raises(TypeError, match="instead element 1 was a <class 'NoneType'>"):
        dt0.key = ["A", None]
    dt0 = dt.Frame([[3, 4, 2, 0, 1], [7, 9, 2, 2, 7], [3, 4, 5, 3, 4]], names=["name", "A", "B"])
    with pytest.raises(ValueError, match="the values are not unique"):
        dt0.key = "A"
    dt0.key = ["A", "B"]
    assert dt0.key == ("A", "B") and dt0.names == ("A", "B", "name")
    assert dt0.to_list() == [[2, 2, 7, 7, 9], [3, 5, 3, 4, 4], [0, 2, 3, 1, 4]]
    with pytest.raises(ValueError, match="the values are not unique"):
        dt0.key = "B"
    assert dt0.key == ("A", "B") and dt0.names == ("A", "B", "name")      # untouched by the failed assignment
    with pytest.raises(ValueError, match="Column A is specified multiple times within the key"):
        dt.Frame(A=list(range(5))).key = ("A", "A")


def test_join_simple(dt):                               # test-join.py:33-45 (string columns -> codes)
    d0 = dt.Frame([[1, 3, 2, 1, 1, 2, 0], [10, 11, 12, 13, 14, 15, 16]], names=("A", "B"))
    d1 = dt.Frame([list(range(4)), [100, 101, 102, 103]], names=("A", "V"))
    d1.key = "A"
    res = d0[:, :, dt.join(d1)]
    assert res.shape == (7, 3) and res.names == ("A", "B", "V")
    assert res.to_list() == [[1, 3, 2, 1, 1, 2, 0], [10, 11, 12, 13, 14, 15, 16], [101, 103, 102, 101, 101, 102, 100]]


def test_join_missing_levels_and_errors(dt):            # test-join.py:62-85
    d0 = dt.Frame(A=[1, 2, 3])
    d1 = dt.Frame(A=[1, 2], K=[True, False])
    d1.key = "A"
    assert d0[:, :, dt.join(d1)].to_list() == [[1, 2, 3], [True, False, None]]
    with pytest.raises(ValueError, match="The join frame is not keyed"):
        d0[:, :, dt.join(dt.Frame(A=[1, 2]))]
    d2 = dt.Frame(Z=[1, 2], K=[5, 6])
    d2.key = "Z"
    with pytest.raises(ValueError, match="Key column `Z` does not exist in the left Frame"):
        d0[:, :, dt.join(d2)]


def test_join_multi(dt):                                # test-join.py:169-182
    fr1 = dt.Frame(A=[1, 2, 1, 2], B=[3, 3, 4, 4], C=[70, 71, 72, 73])
    fr1.key = ("A", "B")
    fr2 = dt.Frame([[1, 2, 3, 2, 3, 1, 2, 1, 1], [3, 4, 5, 4, 3, 3, 3, 4, 3]], names=("A", "B"))
    res = fr2[:, :, dt.join(fr1)]
    assert res.names == ("A", "B", "C")
    assert res.to_list() == [[1, 2, 3, 2, 3, 1, 2, 1, 1], [3, 4, 5, 4, 3, 3, 3, 4, 3], [70, 73, None, 73, None, 70, 71, 72, 70]]


@pytest.mark.parametrize("seed", [5, 6])
def test_join_random(dt, seed):                         # test-join.py:102-138
    rng = random.Random(seed)
    ndata, nkeys = rng.randint(1, 2000), rng.randint(1, 200)
    keys = rng.sample(range(-500, 500), nkeys)
    vals = [rng.random() for _ in keys]
    J = dt.Frame(K=keys, V=vals)
    J.key = "K"
    data = [rng.randint(-520, 520) for _ in range(ndata)]
    X = dt.Frame(K=data, I=list(range(ndata)))
    R = X[:, :, dt.join(J)]
    look = dict(zip(keys, vals))
    assert R.names == ("K", "I", "V")
    assert R.to_list() == [data, list(range(ndata)), [look.get(k) for k in data]]


def test_join_view_and_issue1800(dt):                   # test-join.py:247-256,270-279
    x = dt.Frame(A=[1, 2, 3, 1, 2, 3], B=[3, 6, 2, 4, 3, 1], C=[0, 1, 0, 0, 1, 0])
    a = x[dt.f.A == 1, ["A", "B", "C"]]
    r = dt.Frame(C=[0, 9], BB=[2, 1000])
    r.key = "C"
    res = a[:, :, dt.join(r)]
    assert res.shape == (2, 4) and res.names == ("A", "B", "C", "BB")
    assert res.to_list() == [[1, 1], [3, 4], [0, 0], [2, 2]]
    X1 = dt.Frame(A=list(range(5)), B=[0.1, 0.2, 0.3, 0.4, 0.5])
    X1.key = "A"
    X2 = dt.Frame(A=[0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5])
    assert X2[:, :, dt.join(X1)].to_dict() == {"A": [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5],
                                                "B": [0.1, 0.1, 0.2, 0.2, 0.3, 0.3, 0.4, 0.4, 0.5, 0.5, None, None]}


# ---- tests/ijby/test-sort.py:125-260,462-515,845-935: int32 / int64 / multi-column sorts -----------------

def test_int32_small(dt):                                # test-sort.py:130-138
    d0 = dt.Frame([17, 2, 96, 245, 847569, 34, -45, None, 1])
    assert d0.stypes == (I32,)
    d1 = d0.sort(0)
    assert d1.stypes == d0.stypes
    assert d1.to_list() == [[None, -45, 1, 2, 17, 34, 96, 245, 847569]]


def test_int32_small_stable(dt):                         # test-sort.py:141-151
    d0 = dt.Frame([[5, 3, 5, None, 1000000, None, 3, None], [1, 5, 10, 20, 50, 100, 200, 500]], names=["A", "B"])
    assert d0.sort("A").to_list() == [[None, None, None, 3, 3, 5, 5, 1000000], [20, 100, 500, 5, 200, 1, 10, 50]]


def test_int32_large(dt):                                # test-sort.py:154-165
    p1, p2 = 1000003, 2000003
    src = ((np.arange(p1, dtype=np.int64) + 1) * p2 % p1).astype(np.int32)
    d0 = dt.Frame([src])
    assert d0.stypes == (I32,)
    assert np.array_equal(d0.sort(0).to_numpy_columns()[0], np.arange(p1, dtype=np.int32))


@pytest.mark.parametrize("n", [30, 300, 3000, 30000, 60000, 120000])
def test_int32_large_stable(dt, n):                      # test-sort.py:168-177
    src = [None, 100, 100000] * (n // 3)
    d0 = dt.Frame([src, list(range(n))], names=["A", "B"])
    assert d0.stypes[0] == I32
    d1 = d0[:, "B", dt.sort("A")]
    assert d1.to_list() == [list(range(0, n, 3)) + list(range(1, n, 3)) + list(range(2, n, 3))]


@pytest.mark.parametrize("n", [5, 100, 500, 2500, 32767, 32768, 32769, 200000])
def test_int32_constant(dt, n):                          # test-sort.py:180-188
    tbl0 = [[100000] * n, list(range(n))]
    d0 = dt.Frame(tbl0)
    d1 = d0.sort(0)
    assert d1.stypes == d0.stypes
    assert d1.to_list() == tbl0


def test_int32_reverse_list(dt):                         # test-sort.py:191-198
    step = 10
    d0 = dt.Frame([np.arange(1000000, 0, -step, dtype=np.int32)])
    d1 = d0.sort(0)
    assert d1.stypes == d0.stypes
    assert np.array_equal(d1.to_numpy_columns()[0], np.arange(step, 1000000 + step, step, dtype=np.int32))


@pytest.mark.parametrize("b", [32767, 1000000])
def test_int32_upper_range(dt, b):                       # test-sort.py:201-210
    d0 = dt.Frame([b, b - 1, b + 1] * 1000)
    assert d0.stypes[0] == I32
    assert d0.sort(0).to_list() == [[b - 1] * 1000 + [b] * 1000 + [b + 1] * 1000]


@pytest.mark.parametrize("dc", [32765, 32766, 32767, 32768, 65533, 65534, 65535, 65536])
def test_int32_u2range(dt, dc):                          # test-sort.py:213-225
    a = 100000
    b, c = a + 10, a + dc
    d0 = dt.Frame([c, b, a] * 1000)
    assert d0.stypes[0] == I32
    assert d0.sort(0).to_list() == [[a] * 1000 + [b] * 1000 + [c] * 1000]


def test_int32_unsigned(dt):                             # test-sort.py:228-241
    tbl = builtins_sum(([t] * 100 for t in [0x00000000, 0x00000001, 0x00007FFF, 0x00008000, 0x00008001, 0x0000FFFF,
                                             0x7FFF0000, 0x7FFF0001, 0x7FFF7FFF, 0x7FFF8000, 0x7FFF8001, 0x7FFFFFFF]), [])
    d0 = dt.Frame(tbl)
    assert d0.stypes == (I32,)
    assert d0.sort(0).to_list() == [tbl]


def test_int32_issue220(dt):                             # test-sort.py:244-248
    d0 = dt.Frame([None] + [1000000] * 200 + [None])
    assert d0.sort(0).to_list() == [[None, None] + [1000000] * 200]


def test_int64_small(dt):                                # test-sort.py:467-474
    d0 = dt.Frame([10**(i * 101 % 13) for i in range(13)] + [None])
    assert d0.stypes == (I64,)
    d1 = d0.sort(0)
    assert d1.stypes == d0.stypes
    assert d1.to_list() == [[None] + [10**i for i in range(13)]]


def test_int64_small_stable(dt):                         # test-sort.py:477-483
    d0 = dt.Frame([[0, None, -1000, 11**11] * 3, list(range(12))])
    assert d0.stypes == (I64, I32)
    assert d0[:, 1, dt.sort(0)].to_list() == [[1, 5, 9, 2, 6, 10, 0, 4, 8, 3, 7, 11]]


@pytest.mark.parametrize("n", [16, 20, 30, 40, 50, 100, 500, 1000])
def test_int64_large0(dt, n):                            # test-sort.py:486-500
    a, b = -6654966461866573261, -6655043958000990616
    c, d = 5207085498673612884, 5206891724645893889
    d0 = dt.Frame([c, d, a, b] * n)
    d1 = d0.sort(0)
    assert b < a < d < c
    assert d0.to_list() == [[c, d, a, b] * n]
    assert d1.to_list() == [[b] * n + [a] * n + [d] * n + [c] * n]


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_int64_large_random(dt, seed):                   # test-sort.py:503-511
    random.seed(seed)
    m = 2**63 - 1
    tbl = [random.randint(-m, m) for i in range(1000)]
    d0 = dt.Frame(tbl)
    assert d0.stypes == (I64,)
    assert d0.sort(0).to_list() == [sorted(tbl)]


def test_int32_small_multi(dt):                          # test-sort.py:848-860
    src = [[1, 3, 2, 7, 2, 1, 1, 7, 2, 1, 1, 7], [5, 1, 9, 4, 1, 0, 3, 2, 7, 5, 8, 1]]
    d0 = dt.Frame(src, names=["A", "B"])
    d1 = d0.sort("A", "B")
    order = sorted(range(len(src[0])), key=lambda i: (src[0][i], src[1][i]))
    assert d1.names == d0.names
    assert d1.to_list() == [[src[0][i] for i in order], [src[1][i] for i in order]]


@pytest.mark.parametrize("seed", [7, 8])
def test_bool8_2cols_multi(dt, seed):                    # test-sort.py:863-876
    random.seed(seed)
    n = int(random.expovariate(0.001) + 200)
    data = [[random.choice([True, False]) for _ in range(n)] for j in range(2)]
    n0 = builtins_sum(data[0])
    n10 = builtins_sum(data[1][i] for i in range(n) if data[0][i] is False)
    n11 = builtins_sum(data[1][i] for i in range(n) if data[0][i] is True)
    d1 = dt.Frame(data).sort(0, 1)
    assert d1.to_list() == [[False] * (n - n0) + [True] * n0,
                            [False] * (n - n0 - n10) + [True] * n10 + [False] * (n0 - n11) + [True] * n11]


@pytest.mark.parametrize("seed", [9])
def test_bool8_manycols_multi(dt, seed):                 # test-sort.py:879-890 (rows compared as tuples instead of CSV lines)
    random.seed(seed)
    nrows = int(random.expovariate(0.005) + 100)
    ncols = builtins_min(int(random.expovariate(0.01) + 2), 8)          # the C ABI takes up to 8 key columns
    data = [[random.choice([True, False]) for _ in range(nrows)] for j in range(ncols)]
    d1 = dt.Frame(data).sort(*list(range(ncols)))
    assert list(zip(*d1.to_list())) == sorted(zip(*data))


@pytest.mark.parametrize("seed", [10])
def test_multisort_bool_real(dt, seed):                  # test-sort.py:893-906
    random.seed(seed)
    n = int(random.expovariate(0.001) + 200)
    col0 = [random.choice([True, False]) for _ in range(n)]
    col1 = [random.randint(1, 10) / 97 for _ in range(n)]
    d1 = dt.Frame([col0, col1]).sort(0, 1)
    n0 = builtins_sum(col0)
    assert d1.to_list() == [[False] * (n - n0) + [True] * n0,
                            sorted([col1[i] for i in range(n) if col0[i] is False]) +
                            sorted([col1[i] for i in range(n) if col0[i] is True])]


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_sort_random_multi(dt, seed):                    # test-sort.py:909-935 (string columns -> int codes)
    fns = [lambda: random.choice([True, False]), lambda: random.randint(-10, 10),
           lambda: random.choice([-1.95, 0.1, 0.3, 1.1, 7.3, -2.99, 9.13, 4.555]), lambda: random.randint(0, 11)]
    random.seed(seed)
    n = int(random.expovariate(0.01) + 2)
    data = [[random.choice(fns)() for _ in range(n)] for _ in range(5)]
    for j in range(5):        # one generator per column, as in the original
        fn = random.choice(fns)
        data[j] = [fn() for _ in range(n)]
    data.append([random.random() for _ in range(n)])
    order = sorted(list(range(n)), key=lambda x: (data[1][x], data[2][x], data[3][x], x))
    d0 = dt.Frame(data, names=list("ABCDEF"))
    assert d0.sort("B", "C", "D").to_list() == [[col[j] for j in order] for col in data]


# ---- tests/test-groups.py: the remaining tests of :33-130, :211-430 that stay inside the path ---------------

@pytest.mark.parametrize("seed", [21])
def test_groups4(dt, seed):                              # test-groups.py:94-115 (hex strings -> their integer values)
    random.seed(seed)
    n = 1000
    src = [random.getrandbits(8) for _ in range(n)]
    f1 = dt.Frame(A=src, B=list(range(n)))[:, :, dt.by("A")]
    assert f1.shape == (n, 2) and f1.names == ("A", "B")
    f1A, f1B = f1.to_list()
    f1A.append(None)
    curr_value, group_start = f1A[0], 0
    for i in range(n + 1):
        if f1A[i] != curr_value:
            group = f1B[group_start:i]
            assert group == sorted(group)
            curr_value, group_start = f1A[i], i
    assert group_start == n and curr_value is None


def test_group_boolean2_3(dt):                           # test-groups.py:250-261
    DT = dt.Frame(A=[True, False, False] * 500 + [None, True])
    assert_equals(DT[:, dt.count(), dt.by(dt.f.A)], dt.Frame(A=[None, False, True], count=[1, 1000, 501], stypes={"count": I64}))
    DT = dt.Frame(A=[True] * 1234)
    assert_equals(DT[:, dt.count(), dt.by(dt.f.A)], dt.Frame(A=[True], count=[1234], stypes={"count": I64}))


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_groupby_large_random_integers(dt, seed):        # test-groups.py:339-355 (nunique1 -> nunique)
    random.seed(seed)
    ngrps1 = random.choice([1, 1, 2, 2, 2, 3, 4, 5])
    n0 = 1 << random.choice([1, 1, 2, 2, 2, 3, 3, 3, 4, 5, 6, 7])
    chunks = ([random.sample(range(n0), random.randint(1, n0))] +
              [random.sample([0] * 100 + list(range(256)), random.randint(1, 20)) for i in range(ngrps1)])
    n = builtins_min(int(random.expovariate(0.0001)) + 10, 200000)
    sample = [builtins_sum(random.choice(chunks[i]) << (8 * i) for i in range(len(chunks))) for _ in range(n)]
    nuniques = len(set(sample))
    f0 = dt.Frame(A=sample)
    assert f0[:, dt.nunique(dt.f.A)].to_list() == [[nuniques]]
    assert dt.unique(f0).nrows == nuniques
    assert f0[:, dt.count(), dt.by(dt.f.A)].nrows == nuniques


def test_groupby_multi(dt):                              # test-groups.py:376-381
    DT = dt.Frame(A=[1, 2, 3] * 3, B=[1, 2] * 4 + [1], C=list(range(9)))
    assert DT[:, dt.sum(dt.f.C), dt.by("A", "B")].to_list() == [[1, 1, 2, 2, 3, 3], [1, 2, 1, 2, 1, 2], [6, 3, 4, 8, 10, 5]]


@pytest.mark.parametrize("seed", [41])
def test_groupby_multi_large(dt, seed):                  # test-groups.py:391-416 (letters -> their index)
    random.seed(seed)
    n = builtins_min(100 + int(random.expovariate(0.0001)), 100000)
    col0 = [random.choice([True, False]) for _ in range(n)]
    col1 = [random.randint(-10, 10) for _ in range(n)]
    col2 = [random.randint(0, 13) for _ in range(n)]
    col3 = [random.random() for _ in range(n)]
    rows = sorted((col0[i], col1[i], col2[i], col3[i]) for i in range(n))
    grouped, lastkey, sumval = [], rows[0][:3], 0
    for i in range(n):
        if rows[i][:3] != lastkey:
            grouped.append(lastkey + (sumval,))
            lastkey, sumval = rows[i][:3], 0
        sumval += rows[i][3]
    grouped.append(lastkey + (sumval,))
    DT1 = dt.Frame([col0, col1, col2, col3], names=["A", "B", "C", "D"])[:, dt.sum(dt.f.D), dt.by(dt.f.A, dt.f.B, dt.f.C)]
    want = [list(c) for c in zip(*grouped)]
    got = DT1.to_list()
    assert got[:3] == want[:3]
    assert np.allclose(got[3], want[3], rtol=1e-9)


def test_groupby_on_view(dt):                            # test-groups.py:419-430 (column C: 'b','d' -> 1, 3)
    DT = dt.Frame(A=[1, 2, 3, 1, 2, 3], B=[3, 6, 2, 4, 3, 1], C=[1, 3, 1, 1, 3, 1])
    V = DT[dt.f.A != 1, :]
    assert_equals(V, dt.Frame(A=[2, 3, 2, 3], B=[6, 2, 3, 1], C=[3, 1, 3, 1]))
    assert_equals(V[:, dt.max(dt.f.B), dt.by(dt.f.C)], dt.Frame(C=[1, 3], B=[2, 6]))


def test_int_row_with_by(dt):                            # test-groups.py:486-495
    DT = dt.Frame(A=[1, 2, 3, 1, 2, 1], B=range(6))
    by, f = dt.by, dt.f
    I32 = DT.stypes[1]
    assert_equals(DT[0, :, by(f.A)], dt.Frame(A=[1, 2, 3], B=[0, 1, 2]))
    assert_equals(DT[1, :, by(f.A)], dt.Frame(A=[1, 2], B=[3, 4]))
    assert_equals(DT[2, :, by(f.A)], dt.Frame(A=[1], B=[5]))
    R = DT[3, :, by(f.A)]
    assert R.shape == (0, 2) and R.names == ("A", "B") and R.stypes == (I32, I32)
    assert_equals(DT[-1, :, by(f.A)], dt.Frame(A=[1, 2, 3], B=[5, 4, 2]))
    assert_equals(DT[-2, :, by(f.A)], dt.Frame(A=[1, 2], B=[3, 1]))
    assert_equals(DT[-3, :, by(f.A)], dt.Frame(A=[1], B=[0]))
    R = DT[-4, :, by(f.A)]
    assert R.shape == (0, 2) and R.stypes == (I32, I32)


# ---- tests/test-reduce.py:518-537 and tests/munging/test-dt-rows.py:405-416,555-566,690-710 -------------------

def test_mean_simple_and_empty(dt):                      # test-reduce.py:518-537
    DT_mean = dt.Frame(A=list(range(5)))[:, dt.mean(dt.f.A)]
    assert DT_mean.stypes == (F64,) and DT_mean.to_list() == [[2.0]]
    DT = dt.Frame([[]] * 4, names=list("ABCD"), stypes=(B8, I32, F32, F64))
    assert DT.shape == (0, 4)
    DT_mean = DT[:, dt.mean(dt.f[:])]
    assert DT_mean.shape == (1, 4) and DT_mean.names == ("A", "B", "C", "D")
    assert DT_mean.stypes == (F64, F64, F32, F64) and DT_mean.to_list() == [[None]] * 4


def _dt0(dt):
    T, F = True, False
    return dt.Frame([[F, T, T, None, F, F, T, None, T, T], [7, -11, 9, 10000, None, 0, 0, -1, 1, None],
                     [5, 1, 1.3, 0.1, 1e5, 0, -2.6, -14, math.nan, 2]], names=["colA", "colB", "colC"])


def test_rows_bool_column(dt):                           # test-dt-rows.py:407-416,419-422
    dt0 = _dt0(dt)
    col = dt.Frame([True, False, True, True, None, False, None, True, True, False])
    dt1 = dt0[col, :]
    assert dt1.shape == (5, 3) and dt1.names == ("colA", "colB", "colC")
    assert dt1.stypes == (B8, I32, F64)
    assert dt1.to_list()[1] == [7, 9, 10000, -1, 1]
    with pytest.raises(ValueError, match="i selector has 20 rows, but applied to a Frame with 10 rows"):
        dt0[dt.Frame([bool(i % 2) for i in range(20)]), :]


def test_rows_bool_numpy_array(dt):                      # test-dt-rows.py:557-566
    arr = np.array([True, False, True, True, False, False, True, False, False, True])
    dt1 = _dt0(dt)[arr, :]
    assert dt1.shape == (5, 3) and dt1.names == ("colA", "colB", "colC")
    assert dt1.to_list()[1] == [7, 9, 10000, 0, None]


def test_rows_compare_to_scalar(dt):                     # test-dt-rows.py:691-710
    df1 = dt.Frame([[0, 1, 2, 3, 4, 5, 6, None, 7, None, 9], [3, 2, 1, 3, 4, 0, 2, None, None, 8, 9.0]], names=["A", "B"])
    assert df1.stypes == (I32, F64)
    r = df1[dt.f.A > 3, :]
    assert r.names == df1.names and r.to_list() == [[4, 5, 6, 7, 9], [4, 0, 2, None, 9]]
    assert df1[dt.f.A < 3, :].to_list() == [[0, 1, 2], [3, 2, 1]]
    assert df1[dt.f.A == None, :].to_list() == [[None, None], [None, 8]]       # noqa: E711
    assert df1[dt.f.B != None, :].to_list()[0] == [0, 1, 2, 3, 4, 5, 6, None, 9]   # noqa: E711
