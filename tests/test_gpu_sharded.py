"""Multi-GPU path INSIDE libdthip.so (csrc/comm.hip) on ONE GPU: G logical shards (dthip_comm_init_local: the same
phases as the RCCL path, the exchange being device-to-device copies) against the single-shard result of the same
rows, and the RCCL transport itself with a 1-rank communicator (ncclAllGather + ncclSend/ncclRecv to self).

Bit-exact: group keys, counts, min/max, integer sums, the RowIndex (global row ids) and every column of the
rows-in-grouped-order form.  Float sums / means: the merge adds <= G partial sums in another order -> 1e-12."""
import numpy as np
import pytest

from conftest import assert_same

pytestmark = pytest.mark.gpu

OPS = [("sum", 0), ("mean", 0), ("min", 0), ("max", 0), ("count", 0), ("sum", 1), ("mean", 1), ("min", 1), ("max", 1), ("count0", None)]


def shard(arrs, world, uneven=False):
    n = len(arrs[0])
    if uneven:
        cuts = [0] + sorted(np.random.default_rng(n).integers(0, n + 1, world - 1).tolist()) + [n]
    else:
        cuts = [r * n // world for r in range(world + 1)]
    return [[a[cuts[r]:cuts[r + 1]] for a in arrs] for r in range(world)], cuts


def make(n, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        k = rng.integers(-5000, 5000, n).astype(np.int64)
    elif kind == "skew":                      # most rows in a few small keys, a long sparse tail
        k = (rng.random(n) ** 8 * 1e9).astype(np.int64)
    elif kind == "wide":                      # 63-bit keys from a pool
        pool = rng.integers(-2**62, 2**62, max(n // 50, 3))
        k = pool[rng.integers(0, len(pool), n)].astype(np.int64)
    elif kind == "int32":
        k = rng.integers(0, 300, n).astype(np.int32)
    elif kind == "float":
        k = (rng.integers(-200, 200, n) / 8).astype(np.float64)
        k[rng.random(n) < 0.01] = -0.0
    if n:
        na = rng.random(n) < 0.02
        if k.dtype.kind == "f":
            k[na] = np.nan
        else:
            k[na] = np.iinfo(k.dtype).min
    v = rng.standard_normal(n)
    v[rng.random(n) < 0.05] = np.nan
    w = rng.integers(-10**6, 10**6, n).astype(np.int64)
    w[rng.random(n) < 0.05] = np.iinfo(np.int64).min
    return k, v, w


def concat(results, getter):
    return np.concatenate([getter(r) for r in results])


def check_agg(ctx, comm, keys, vals, aggs, uneven=False, na_last=False):
    single = ctx.groupby_agg(keys, vals, aggs, na_last=na_last)
    ksh, _ = shard(keys, comm.world, uneven)
    vsh, _ = shard(vals, comm.world, uneven)
    res = comm.groupby_agg(ksh, vsh, aggs, na_last=na_last)
    assert sum(r.ngroups for r in res) == single.ngroups
    for i in range(len(keys)):
        assert_same(concat(res, lambda r: r.key(i)), single.key(i), "group key %d" % i)
    for a, (op, c) in enumerate(aggs):
        got, exp = concat(res, lambda r: r.agg(a)), single.agg(a)
        if exp.dtype.kind == "f" and op in ("sum", "mean"):
            assert got.dtype == exp.dtype
            assert np.array_equal(np.isnan(got), np.isnan(exp)), "%s NA pattern" % op
            m = ~np.isnan(exp)
            assert np.allclose(got[m], exp[m], rtol=1e-12 if exp.dtype == np.float64 else 1e-6, atol=1e-12), op
        else:
            assert_same(got, exp, "%s(%s)" % (op, c))
    sizes = [r.ngroups for r in res]
    for r in res:
        r.free()
    single.free()
    return sizes


@pytest.fixture(scope="module")
def comms():
    from datatable_amd.engine import LocalComm
    cs = {g: LocalComm(g, device=0) for g in (1, 2, 4, 8)}
    yield cs
    for c in cs.values():
        c.close()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("kind", ["uniform", "skew", "wide", "int32", "float"])
@pytest.mark.parametrize("n", [0, 1, 37, 100_000])
def test_logical_shards_agg(ctx, comms, world, kind, n):
    k, v, w = make(n, 1000 + n + world, kind)
    check_agg(ctx, comms[world], [k], [v, w], OPS, uneven=(n > 37 and world > 1))


@pytest.mark.parametrize("world", [2, 8])
def test_logical_shards_agg_large_two_keys_na_last(ctx, comms, world):
    n = 3_000_000
    rng = np.random.default_rng(7)
    a = rng.integers(0, 2000, n).astype(np.int32)
    b = rng.integers(0, 500, n).astype(np.int32)
    a[rng.random(n) < 0.01] = -2**31
    v = rng.standard_normal(n)
    w = rng.integers(-100, 100, n).astype(np.int64)
    check_agg(ctx, comms[world], [a, b], [v, w], OPS)
    check_agg(ctx, comms[world], [a, b], [v, w], OPS, na_last=True)


def test_splitters_balance_skewed_keys(ctx, comms):
    """the histogram splitters give every destination about the same number of partial groups even when an even split
    of [min, max] would send nearly everything to rank 0"""
    n = 2_000_000
    k, v, w = make(n, 5, "skew")
    k = np.abs(k)
    sizes = check_agg(ctx, comms[8], [k], [v, w], [("sum", 0), ("count0", None)])
    even_split_rank0 = np.unique(k[k <= k.max() // 8]).size / np.unique(k).size
    assert even_split_rank0 > 0.5                       # what the old even-width split would have done
    assert max(sizes) <= 1.6 * (sum(sizes) / 8), sizes      # groups; the partials received per rank are balanced tighter


def test_partial_sums_that_look_like_na(ctx, comms):
    """a shard whose partial float sum is NaN (+inf and -inf met) or whose int64 partial wrapped to INT64_MIN must merge
    as a VALUE (DTHIP_FLAG_NONA) -- same answer as the single shard"""
    k = np.array([1, 1, 1, 1, 2, 2, 2, 2], np.int64)
    v = np.array([np.inf, -np.inf, 1.0, 2.0, 1.0, 2.0, 3.0, 4.0])
    w = np.array([-2**62, -2**62, 5, 6, 1, 2, 3, 4], np.int64)
    check_agg(ctx, comms[2], [k], [v, w], [("sum", 0), ("mean", 0), ("sum", 1), ("count", 0), ("count0", None)])


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("kind", ["uniform", "skew", "float"])
def test_logical_shards_rows(ctx, comms, world, kind):
    n = 200_000
    k, v, w = make(n, 77 + world, kind)
    single = ctx.groupby_rows([k], [k, v, w], want_rowindex=True)
    (ksh, cuts) = shard([k], world, uneven=True)
    csh, _ = shard([k, v, w], world, uneven=True)
    res = comms[world].groupby_rows(ksh, csh, cuts[:-1])
    assert sum(r.ngroups for r in res) == single.ngroups and sum(r.nrows for r in res) == n
    for c in range(3):
        assert_same(concat(res, lambda r: r.col(c)), single.col(c), "column %d in grouped order" % c)
    assert_same(concat(res, lambda r: r.col(3)).astype(np.int32), single.rowindex(), "global row ids == the RowIndex")
    off = [0]
    for r in res:
        o = r.offsets().astype(np.int64)
        off += (o[1:] + off[-1]).tolist() if r.ngroups else []
    assert_same(np.array(off, np.int32), single.offsets(), "offsets")
    for r in res:
        r.free()
    single.free()


def test_rccl_transport_one_rank(ctx):
    """the RCCL code path itself on this box's single GPU: communicator of 1 rank, ncclAllGather of the control words,
    ncclSend / ncclRecv to self inside one group for the data"""
    from datatable_amd.engine import Context, comm_unique_id
    c = Context(0)
    c.comm_init(0, 1, comm_unique_id())
    assert c.comm_rank == 0 and c.comm_world == 1
    k, v, w = make(300_000, 3, "skew")
    aggs = [("sum", 0), ("mean", 0), ("min", 1), ("max", 1), ("count0", None)]
    got = c.sharded_groupby_agg([k], [v, w], aggs)
    exp = ctx.groupby_agg([k], [v, w], aggs)
    assert_same(got.key(0), exp.key(0), "keys")
    for a in range(len(aggs)):
        if aggs[a][0] in ("sum", "mean"):
            assert np.allclose(got.agg(a), exp.agg(a), rtol=1e-12, atol=1e-12, equal_nan=True)
        else:
            assert_same(got.agg(a), exp.agg(a), aggs[a][0])
    got.free()
    r = c.sharded_groupby_rows([k], [v], 0)
    e = ctx.groupby_rows([k], [v], want_rowindex=True)
    assert_same(r.col(0), e.col(0), "rows"); assert_same(r.col(1).astype(np.int32), e.rowindex(), "row ids")
    r.free(); e.free(); exp.free()
    c.close()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("na_last", [False, True])
def test_logical_shards_rows_two_keys_small_columns(ctx, comms, world, na_last):
    """rows in grouped order with a composite key (the range partition is on the FIRST key only), NA groups first / last,
    and payload columns of 1, 2 and 4 bytes next to the 8-byte ones"""
    rng = np.random.default_rng(900 + world)
    n = 150_000
    a = rng.integers(-40, 40, n).astype(np.int32)
    a[rng.random(n) < 0.03] = -2**31
    b = rng.integers(0, 9, n).astype(np.int64)
    c1 = rng.integers(-100, 100, n).astype(np.int8)
    c2 = rng.integers(-3000, 3000, n).astype(np.int16)
    c4 = rng.standard_normal(n).astype(np.float32)
    c8 = rng.standard_normal(n)
    cols = [c1, c2, c4, c8, a]
    single = ctx.groupby_rows([a, b], cols, want_rowindex=True, na_last=na_last)
    ksh, cuts = shard([a, b], world, uneven=True)
    csh, _ = shard(cols, world, uneven=True)
    res = comms[world].groupby_rows(ksh, csh, cuts[:-1], na_last=na_last)
    assert sum(r.ngroups for r in res) == single.ngroups
    for c in range(len(cols)):
        assert_same(concat(res, lambda r: r.col(c)), single.col(c), "column %d" % c)
    assert_same(concat(res, lambda r: r.col(len(cols))).astype(np.int32), single.rowindex(), "global row ids")
    for r in res:
        r.free()
    single.free()
