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


# ---- G shards against the ORACLE (oracle/dt_oracle.c, pinned to reference goldens), not against the library itself ----
def oracle_agg(keys, vals, aggs, na_last=False):
    from oracle import oracle as o
    ri, off = o.group(keys, na_last=na_last)
    first = ri[off[:-1]]
    gk = [k[first] for k in keys]
    cols = [np.diff(off).astype(np.int64) if c is None else o.reduce(op, vals[c], ri, off) for op, c in aggs]
    scale = [None if c is None or vals[c].dtype.kind != "f" else
             np.add.reduceat(np.abs(np.nan_to_num(vals[c][ri].astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)), off[:-1]) if len(ri) else np.zeros(0)
             for op, c in aggs]
    return gk, cols, scale, ri, off


def check_agg_oracle(comm, keys, vals, aggs, uneven=True, na_last=False):
    from conftest import assert_close
    gk, cols, scale, _, _ = oracle_agg(keys, vals, aggs, na_last)
    ksh, _ = shard(keys, comm.world, uneven)
    vsh, _ = shard(vals, comm.world, uneven)
    res = comm.groupby_agg(ksh, vsh, aggs, na_last=na_last)
    for i in range(len(keys)):
        assert_same(concat(res, lambda r: r.key(i)), gk[i], "group key %d" % i)
    for a, (op, c) in enumerate(aggs):
        got = concat(res, lambda r: r.agg(a))
        if cols[a].dtype == np.float32 and op in ("sum", "mean"):
            # the documented float64-accumulation deviation (include/dthip.h): the reference's float32 row-by-row sum
            # carries its own rounding, eps32 * sum|v| of the group
            assert got.dtype == np.float32 and np.array_equal(np.isnan(got), np.isnan(cols[a]))
            m = ~np.isnan(cols[a])
            cnt = np.maximum(np.diff(oracle_agg(keys, vals, [("count0", None)], na_last)[4]), 1) if op == "mean" else 1.0
            tol = 1e-4 * np.abs(cols[a].astype(np.float64)) + 4e-7 * scale[a] / cnt + 1e-30
            assert np.all(np.abs(got.astype(np.float64) - cols[a].astype(np.float64))[m] <= tol[m]), "%s(%s) float32" % (op, c)
        elif cols[a].dtype.kind == "f" and op in ("sum", "mean"):
            assert_close(got, cols[a], scale=scale[a], rel=1e-6, what="%s(%s)" % (op, c))
        else:
            assert_same(got, cols[a], "%s(%s)" % (op, c))
    for r in res:
        r.free()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("kind", ["uniform", "skew", "wide", "float"])
def test_shards_agg_vs_oracle(comms, world, kind):
    k, v, w = make(150_000, 4000 + world, kind)
    check_agg_oracle(comms[world], [k], [v, w], OPS)
    check_agg_oracle(comms[world], [k], [v, w], OPS, na_last=True)


@pytest.mark.parametrize("world", [2, 8])
def test_shards_agg_two_keys_vs_oracle(comms, world):
    rng = np.random.default_rng(world)
    n = 400_000
    a = rng.integers(0, 700, n).astype(np.int32)
    b = rng.integers(0, 300, n).astype(np.int32)
    a[rng.random(n) < 0.01] = -2**31
    b[rng.random(n) < 0.01] = -2**31
    v = rng.standard_normal(n)
    check_agg_oracle(comms[world], [a, b], [v], [("count0", None), ("sum", 0), ("mean", 0), ("min", 0), ("max", 0)])


@pytest.mark.parametrize("world", [2, 8])
def test_shards_agg_float32_values_vs_oracle(comms, ctx, world):
    """float32 value columns: partial sums cross the exchange in float64 and are rounded ONCE after the merge, so the
    sharded result equals the single-GPU one to the last bit of float32 in nearly every group (and both are within the
    documented 1e-4 of the reference's float32 row-by-row accumulation); min / max / count bit-exact"""
    rng = np.random.default_rng(50 + world)
    n = 300_000
    k = rng.integers(0, 2000, n).astype(np.int64)
    v = (rng.standard_normal(n) * 1000).astype(np.float32)
    v[rng.random(n) < 0.03] = np.nan
    aggs = [("sum", 0), ("mean", 0), ("min", 0), ("max", 0), ("count", 0)]
    check_agg_oracle(comms[world], [k], [v], aggs)
    single = ctx.groupby_agg([k], [v], aggs)
    ksh, _ = shard([k], world); vsh, _ = shard([v], world)
    res = comms[world].groupby_agg(ksh, vsh, aggs)
    for a in range(len(aggs)):
        got, exp = concat(res, lambda r: r.agg(a)), single.agg(a)
        assert got.dtype == exp.dtype
        if a < 2:       # float32 sum / mean: float64 partials, one rounding -> at most 1 ulp of float32 from the single-GPU result
            assert np.allclose(got, exp, rtol=2.4e-7, atol=1e-30, equal_nan=True), aggs[a]
        else:
            assert_same(got, exp, aggs[a][0])
    for r in res:
        r.free()
    single.free()


@pytest.mark.parametrize("world", [2, 8])
def test_int64_max_next_to_na_last(comms, world):
    """INT64_MAX keys and NA keys both have the all-ones image when NAs sort last: they travel to the last rank
    together and come out as two groups, NA after INT64_MAX (csrc/comm.hip key_image)"""
    rng = np.random.default_rng(3)
    n = 60_000
    k = rng.integers(-1000, 1000, n).astype(np.int64)
    k[rng.random(n) < 0.05] = np.iinfo(np.int64).max
    k[rng.random(n) < 0.05] = np.iinfo(np.int64).min
    v = rng.standard_normal(n)
    for na_last in (True, False):
        check_agg_oracle(comms[world], [k], [v], [("sum", 0), ("count0", None), ("max", 0)], na_last=na_last)
    # rows in grouped order through the same images
    from oracle import oracle as o
    ri, off = o.group([k], na_last=True)
    ksh, cuts = shard([k], world, uneven=True)
    csh, _ = shard([k, v], world, uneven=True)
    res = comms[world].groupby_rows(ksh, csh, cuts[:-1], na_last=True)
    assert_same(concat(res, lambda r: r.col(2)).astype(np.int32), ri, "global row ids == the oracle's RowIndex")
    assert_same(concat(res, lambda r: r.col(0)), k[ri], "keys in grouped order")
    assert sum(r.ngroups for r in res) == len(off) - 1
    for r in res:
        r.free()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("kind", ["uniform", "skew", "float"])
def test_shards_rows_vs_oracle(comms, world, kind):
    from oracle import oracle as o
    n = 120_000
    k, v, w = make(n, 500 + world, kind)
    ri, off = o.group([k])
    ksh, cuts = shard([k], world, uneven=True)
    csh, _ = shard([k, v, w], world, uneven=True)
    res = comms[world].groupby_rows(ksh, csh, cuts[:-1])
    assert_same(concat(res, lambda r: r.col(3)).astype(np.int32), ri, "global row ids == the oracle's RowIndex")
    assert_same(concat(res, lambda r: r.col(1)), v[ri], "value column in grouped order")
    goff = [0]
    for r in res:
        oo = r.offsets().astype(np.int64)
        goff += (oo[1:] + goff[-1]).tolist() if r.ngroups else []
    assert_same(np.array(goff, np.int32), off, "offsets")
    for r in res:
        r.free()


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("na_last", [False, True])
def test_shards_rows_heavy_na_group(comms, world, na_last):
    """round 6: an NA group heavier than a fair share (30 % of the rows) -- bit-exact against the oracle, the
    NA group on ONE shard (a share above the receive bound: exact buffers + the status round), nobody starved and the other
    shards splitting the rest evenly (split_plan.hpp sample_bounds_q)"""
    from oracle import oracle as o
    rng = np.random.default_rng(40 + world)
    n = 240_000
    k = rng.integers(0, 50_000, n).astype(np.int64)
    k[rng.random(n) < 0.30] = np.iinfo(np.int64).min
    v = rng.standard_normal(n)
    ri, off = o.group([k], na_last=na_last)
    ksh, cuts = shard([k], world, uneven=True)
    csh, _ = shard([k, v], world, uneven=True)
    res = comms[world].groupby_rows(ksh, csh, cuts[:-1], na_last=na_last)
    assert_same(concat(res, lambda r: r.col(2)).astype(np.int32), ri, "global row ids == the oracle's RowIndex")
    assert_same(concat(res, lambda r: r.col(1)), v[ri], "value column in grouped order")
    rows = [r.nrows for r in res]
    nna = int((k == np.iinfo(np.int64).min).sum())
    fair = (n - nna) / (world - 1)
    owner = 0 if not na_last else world - 1
    assert nna <= rows[owner] <= nna + 0.5 * fair, rows
    assert min(rows) > 0 and max(rows[r_] for r_ in range(world) if r_ != owner) <= 1.5 * fair, rows
    for r in res:
        r.free()


def test_sharded_errors_reach_every_rank(comms):
    """a query the distributed path refuses (first() needs the row order of a whole group) or bad arguments come back
    as ONE error for the call, nothing is left half-done, and the communicator keeps working"""
    k, v, w = make(5000, 1, "uniform")
    ksh, _ = shard([k], 4); vsh, _ = shard([v, w], 4)
    with pytest.raises(NotImplementedError):
        comms[4].groupby_agg(ksh, vsh, [("first", 0)])
    with pytest.raises(ValueError):
        comms[4].groupby_agg(ksh, vsh, [("sum", 7)])
    check_agg_oracle(comms[4], [k], [v, w], [("sum", 0), ("count0", None)])
