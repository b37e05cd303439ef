"""GPU parity tests for the set functions and the natural-join index, through the
C ABI (dthip_setop, dthip_join_index), against tests/golden/sets_join_cases.npz (outputs of the unmodified
reference) and against the CPU oracle on seeded inputs.  Everything here is integer / index work: bit-exact."""
import numpy as np
import pytest

from conftest import assert_same, sets_join_golden
from oracle import oracle as o

pytestmark = pytest.mark.gpu
GS = sets_join_golden()
OPS = ("union", "intersect", "setdiff", "symdiff")


@pytest.mark.parametrize("name", GS.names("set"))
def test_golden_setops(ctx, name):
    c = GS.by_name[name]
    srcs = [GS.get(name, "src%d" % i) for i in range(len(c["stypes"]))]
    if len(set(c["stypes"])) != 1:
        pytest.skip("mixed stypes are up-cast by the Frame layer")
    st = c["stypes"][0]
    stacked = np.concatenate(srcs)
    for op in OPS:
        assert_same(stacked[ctx.setop(op, srcs, stype=st)], GS.get(name, op), "%s/%s" % (name, op))
    if "unique" in c["outs"]:
        assert_same(stacked[ctx.setop("union", [stacked], stype=st)], GS.get(name, "unique"), name + "/unique")


@pytest.mark.parametrize("name", GS.names("join"))
def test_golden_join(ctx, name):
    c = GS.by_name[name]
    nk = len(c["xstypes"])
    x = [GS.get(name, "x%d" % k) for k in range(nk)]
    j = [GS.get(name, "j%d" % k) for k in range(nk)]
    assert_same(ctx.join_index(x, j, xstypes=c["xstypes"], jstypes=c["jstypes"]), GS.get(name, "index"), name)


@pytest.mark.parametrize("k", [1, 2, 3, 5])
@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float64, np.int8])
def test_setops_vs_oracle(ctx, k, dtype):
    rng = np.random.default_rng(500 + k)
    srcs = []
    for i in range(k):
        n = int(rng.integers(1, 200_000))
        if np.dtype(dtype).kind == "f":
            a = (rng.integers(-5000, 5000, n) * 0.25).astype(dtype)
            a[rng.random(n) < 0.01] = np.nan
        else:
            lim = 100 if dtype == np.int8 else 20_000
            a = rng.integers(-lim, lim, n).astype(dtype)
            a[rng.random(n) < 0.01] = np.iinfo(dtype).min
        srcs.append(a)
    for op in OPS:
        assert_same(ctx.setop(op, srcs), o.setop(op, srcs), "%s k=%d" % (op, k))


def test_setops_edge_cases(ctx):
    e = np.zeros(0, np.int32)
    a = np.array([5, 5, 1], np.int32)
    for op in OPS:
        assert ctx.setop(op, [e]).tolist() == []
        assert_same(ctx.setop(op, [a, e]), o.setop(op, [a, e]), op)
        assert_same(ctx.setop(op, [e, a]), o.setop(op, [e, a]), op)
        assert_same(ctx.setop(op, [a, a, a]), o.setop(op, [a, a, a]), op)


@pytest.mark.parametrize("xdt,jdt", [(np.int32, np.int32), (np.int64, np.int32), (np.int32, np.int64), (np.float64, np.int64),
                                     (np.int16, np.float64), (np.float64, np.float32), (np.int64, np.int8)])
@pytest.mark.parametrize("table", [1, 0])
def test_join_vs_oracle(ctx, xdt, jdt, table):
    """table=1: dense single integer keys go through the direct key->row table; 0: always the binary search"""
    ctx.set_option("join_table", table)
    try:
        _join_vs_oracle(ctx, xdt, jdt)
    finally:
        ctx.set_option("join_table", 1)


def _join_vs_oracle(ctx, xdt, jdt):
    rng = np.random.default_rng(900)
    nj, nx = 50_000, 400_000
    if np.dtype(jdt) == np.int8:
        jv = np.arange(-100, 100, 3).astype(jdt)
    else:
        jv = np.sort(rng.choice(np.arange(-200_000, 200_000), nj, replace=False)).astype(jdt)
        jv = np.unique(jv)                                # float32 rounding can merge neighbours
    if np.dtype(xdt).kind == "f":
        xv = (rng.integers(-500_000, 500_000, nx) * 0.5).astype(xdt)
        xv[rng.random(nx) < 0.02] = np.nan
    else:
        lim = 30_000 if xdt == np.int16 else 250_000
        xv = rng.integers(-lim, lim, nx).astype(xdt)
        xv[rng.random(nx) < 0.02] = np.iinfo(xdt).min
    assert_same(ctx.join_index([xv], [jv]), o.join_index([xv], [jv]), "join %s->%s" % (xdt, jdt))
    if np.dtype(jdt).kind == "i":
        # a keyed frame whose first key is NA: NA rows of X join row 0
        jn = np.concatenate([[np.iinfo(jdt).min], jv]).astype(jdt)
        assert_same(ctx.join_index([xv], [jn]), o.join_index([xv], [jn]), "join with NA key %s->%s" % (xdt, jdt))


def test_join_two_keys_and_na_key(ctx):
    rng = np.random.default_rng(901)
    pairs = np.unique(np.stack([rng.integers(0, 300, 20_000), rng.integers(-50, 50, 20_000)], 1), axis=0)
    j0, j1 = pairs[:, 0].astype(np.int32), pairs[:, 1].astype(np.int64)
    # keyed J: sorted by (j0, j1) with NA first -- put one NA key row in front
    j0 = np.concatenate([[np.iinfo(np.int32).min], j0]).astype(np.int32)
    j1 = np.concatenate([[7], j1]).astype(np.int64)
    x0 = rng.integers(-2, 302, 300_000).astype(np.int32)
    x1 = rng.integers(-52, 52, 300_000).astype(np.int64)
    x0[:100] = np.iinfo(np.int32).min
    x1[:100] = 7
    got = ctx.join_index([x0, x1], [j0, j1])
    assert_same(got, o.join_index([x0, x1], [j0, j1]), "two keys")
    assert (got[:100] == 0).all()
    hit = got >= 0
    assert (j0[got[hit]] == x0[hit]).all() and (j1[got[hit]] == x1[hit]).all()


def test_join_large_property(ctx):
    """1e7 X rows against 1e6 keys: every hit points at an equal key, every miss is absent from J"""
    rng = np.random.default_rng(902)
    jv = np.unique(rng.integers(0, 4_000_000, 1_000_000)).astype(np.int64)
    xv = rng.integers(0, 4_000_000, 10_000_000).astype(np.int64)
    got = ctx.join_index([xv], [jv])
    hit = got >= 0
    assert (jv[got[hit]] == xv[hit]).all()
    assert not np.isin(xv[~hit][:200_000], jv).any()
    assert hit.sum() == np.isin(xv, jv).sum()
