"""Tests ported from the reference's own suite (h2oai/datatable tests/test-groups.py,
tests/test-reduce.py, tests/ijby/test-sort.py) -- same inputs, same expected values, spelled
the same way against datatable_amd.frame.  String columns and computed f-expressions
(f.A + f.B, prod, ...) are outside the accelerated path and are left out or replaced by an
integer column where the test is about grouping, not strings.  file:line of every original is
given.  Every evaluation runs on the GPU through the C ABI."""
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dt():
    from datatable_amd import frame
    return frame


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
