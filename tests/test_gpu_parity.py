"""GPU parity tests: the HIP path, called through the C ABI (libdthip.so), against
  (1) the committed golden fixtures = outputs of the unmodified reference, and
  (2) the CPU oracle on seeded inputs at sizes it finishes in seconds.
Bit-exact for RowIndex, offsets, group keys, counts, integer sums, min/max;
float sums/means within 1e-6 relative (BASELINE.json's tolerance; the GPU
re-associates the sum, the reference adds left to right)."""
import numpy as np
import pytest

from conftest import assert_close, assert_same, golden_names
from oracle import oracle as o

pytestmark = pytest.mark.gpu

OPS = ("sum", "mean", "min", "max", "count")
# fused aggregation is checked on every path of dthip_groupby_agg: 0 = the library's own choice,
# 1 = sort + segmented reduce, 2 = bucketed (sort-free) aggregation wherever its preconditions hold
AGG_PATHS = (0, 1, 2)


def group_abs_scale(v, ri, off):
    """sum of |v| per group, for the absolute part of the float tolerance"""
    if len(off) < 2:
        return np.zeros(0)
    a = np.abs(np.nan_to_num(v.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0))[ri]
    return np.add.reduceat(a, off[:-1].astype(np.int64)) if len(a) else np.zeros(len(off) - 1)


def check_agg(got, exp, opn, v, ri, off, what):
    if exp.dtype.kind == "f" and opn in ("sum", "mean"):
        sc = group_abs_scale(v, ri, off)
        if v.dtype == np.float32 and opn == "sum":
            # the reference accumulates float32 sums in float32 (sumprod.h:48-55); the GPU in float64
            assert_close(got, exp, scale=sc * 1e6, rel=1e-4, what=what)
        else:
            assert_close(got, exp, scale=sc, rel=1e-6, what=what)
    else:
        assert_same(got, exp, what)


# ---- golden fixtures (reference outputs) -----------------------------------------------

@pytest.mark.parametrize("name", golden_names())
def test_golden_groupby(routed_ctx, gold, name):
    """the reference's RowIndex / offsets / group keys, on BOTH routes of the sort path (LSD passes; MSD levels forced)"""
    ctx = routed_ctx
    c = gold.by_name[name]
    keys = gold.keys(name)
    r = ctx.groupby(keys, stypes=c["key_stypes"])
    assert r.ngroups == len(gold.get(name, "off")) - 1
    assert_same(r.offsets(), gold.get(name, "off"), "offsets")
    assert_same(r.rowindex(), gold.get(name, "ri"), "rowindex")
    for i in range(len(keys)):
        assert_same(r.group_keys(keys[i], c["key_stypes"][i]), gold.get(name, "gk%d" % i), "group key %d" % i)
    r.free()


@pytest.mark.parametrize("name", golden_names())
def test_golden_fused_agg(routed_ctx, gold, name):
    ctx = routed_ctx
    c = gold.by_name[name]
    keys, vals = gold.keys(name), gold.vals(name)
    aggs = [(opn, vi) for opn, vi, _ in c["aggs"]] + [("count0", None)]
    ri, off = gold.get(name, "ri"), gold.get(name, "off")
    for path in AGG_PATHS:
        ctx.set_option("agg_path", path)
        try:
            r = ctx.groupby_agg(keys, vals, aggs, key_stypes=c["key_stypes"], value_stypes=c["val_stypes"])
        finally:
            ctx.set_option("agg_path", 0)
        tag = " [agg_path=%d]" % path
        assert_same(r.offsets(), off, "offsets" + tag)
        for i in range(len(keys)):
            assert_same(r.key(i), gold.get(name, "gk%d" % i), "group key %d%s" % (i, tag))
        for a, (opn, vi, ost) in enumerate(c["aggs"]):
            check_agg(r.agg(a), gold.get(name, "%s.v%d" % (opn, vi)), opn, vals[vi], ri, off, "%s(v%d)%s" % (opn, vi, tag))
        assert_same(r.agg(len(c["aggs"])), np.diff(off).astype(np.int64), "count()" + tag)
        r.free()


@pytest.mark.parametrize("name", ["appendixB", "keytype_4", "keytype_7", "c2_shape", "skewed", "all_na_value"])
def test_golden_reduce_through_rowindex(ctx, gold, name):
    """S-red seam: reducers over a caller-supplied RowIndex + offsets"""
    c = gold.by_name[name]
    vals = gold.vals(name)
    ri, off = gold.get(name, "ri"), gold.get(name, "off")
    for opn, vi, ost in c["aggs"]:
        got = ctx.reduce(opn, vals[vi], ri, off, stype=c["val_stypes"][vi])
        check_agg(got, gold.get(name, "%s.v%d" % (opn, vi)), opn, vals[vi], ri, off, "%s(v%d)" % (opn, vi))
    assert_same(ctx.reduce("count0", None, None, off), np.diff(off).astype(np.int64), "count()")


# ---- seeded random inputs vs the oracle ----------------------------------------------------

def _vs_oracle(ctx, keys, vals, aggs=OPS, key_stypes=None, check_ri=True):
    ri, off = o.group(keys, stypes=key_stypes)
    if check_ri:
        r = ctx.groupby(keys, stypes=key_stypes)
        assert_same(r.offsets(), off, "offsets")
        assert_same(r.rowindex(), ri, "rowindex")
        r.free()
    alist = [(opn, vi) for vi in range(len(vals)) for opn in aggs] + [("count0", None)]
    expected = [o.reduce(opn, vals[vi], ri, off) for opn, vi in alist[:-1]]
    gkeys = [(k.view(np.int8) if k.dtype == np.bool_ else k)[ri[off[:-1]]] for k in keys]
    for path in AGG_PATHS:
        ctx.set_option("agg_path", path)
        try:
            r = ctx.groupby_agg(keys, vals, alist, key_stypes=key_stypes)
        finally:
            ctx.set_option("agg_path", 0)
        tag = " [agg_path=%d]" % path
        assert_same(r.offsets(), off, "fused offsets" + tag)
        for i in range(len(keys)):
            assert_same(r.key(i), gkeys[i], "fused group key %d%s" % (i, tag))
        for a, (opn, vi) in enumerate(alist[:-1]):
            check_agg(r.agg(a), expected[a], opn, vals[vi], ri, off, "%s(v%d)%s" % (opn, vi, tag))
        assert_same(r.agg(len(alist) - 1), np.diff(off).astype(np.int64), "count()" + tag)
        r.free()
    # sparse key ranges: the hash combiner (LDS hash tables -> partial groups -> merge), forced on
    # (takes effect where the bucketed path does not apply; several value columns: one pass of hash tables per column)
    for hm in (2, 3):
        ctx.set_option("hash_mode", hm)
        try:
            r = ctx.groupby_agg(keys, vals, alist, key_stypes=key_stypes)
        finally:
            ctx.set_option("hash_mode", 0)
        tag = " [hash_mode=%d]" % hm
        assert_same(r.offsets(), off, "fused offsets" + tag)
        for i in range(len(keys)):
            assert_same(r.key(i), gkeys[i], "fused group key %d%s" % (i, tag))
        for a, (opn, vi) in enumerate(alist[:-1]):
            check_agg(r.agg(a), expected[a], opn, vals[vi], ri, off, "%s(v%d)%s" % (opn, vi, tag))
        assert_same(r.agg(len(alist) - 1), np.diff(off).astype(np.int64), "count()" + tag)
        r.free()
    # the same without group sizes in the result (option agg_offsets=0, no count()): the bucketed
    # path then tracks key presence only and uses larger tables / fewer buckets
    ctx.set_option("agg_path", 2)
    ctx.set_option("agg_offsets", 0)
    try:
        r = ctx.groupby_agg(keys, vals, alist[:-1], key_stypes=key_stypes)
    finally:
        ctx.set_option("agg_path", 0)
        ctx.set_option("agg_offsets", 1)
    assert r.ngroups == len(off) - 1
    for i in range(len(keys)):
        assert_same(r.key(i), gkeys[i], "group key %d [no offsets]" % i)
    for a, (opn, vi) in enumerate(alist[:-1]):
        check_agg(r.agg(a), expected[a], opn, vals[vi], ri, off, "%s(v%d) [no offsets]" % (opn, vi))
    r.free()


def test_config1_full_size(ctx):
    """BASELINE config 1 at full size: 1e6 rows, int32 key (100 groups), sum(float64)"""
    rng = np.random.default_rng(1234 + 1)
    n = 1_000_000
    k = rng.integers(0, 100, n).astype(np.int32)
    v = rng.standard_normal(n)
    _vs_oracle(ctx, [k], [v])


def test_config2_shape(ctx):
    """BASELINE config 2 shape (1e6 of the 1e8 rows): int64 key, 4 float64 columns, sum+mean+min+max"""
    rng = np.random.default_rng(1234 + 2)
    n = 1_000_000
    k = rng.integers(0, 100_000, n).astype(np.int64)
    vals = [rng.standard_normal(n) for _ in range(4)]
    _vs_oracle(ctx, [k], vals, aggs=("sum", "mean", "min", "max"))


def test_config3_shape(ctx):
    """BASELINE config 3 shape scaled: 4e6 rows, int64 key with 4e4 groups (100 rows/group), 24-bit-like key"""
    rng = np.random.default_rng(1234 + 3)
    n = 4_000_000
    k = rng.integers(0, 10_000_000, n).astype(np.int64)      # 24 significant bits -> three 8-bit passes
    v = rng.standard_normal(n)
    _vs_oracle(ctx, [k], [v], aggs=("sum",))


def test_config3_hard_keys(ctx):
    """full-range int64 keys drawn from a pool: the 64-significant-bit path (8 passes, 64-bit radix keys)"""
    rng = np.random.default_rng(99)
    n = 1_000_000
    pool = rng.integers(-2**62, 2**62, 10_000)
    k = rng.choice(pool, n).astype(np.int64)
    v = rng.standard_normal(n)
    _vs_oracle(ctx, [k], [v], aggs=("sum", "count"))


def test_config4_shape(ctx):
    """BASELINE config 4 shape scaled: 2-key composite (int32,int32), count + sum"""
    rng = np.random.default_rng(1234 + 4)
    n = 2_000_000
    a = rng.integers(0, 3163, n).astype(np.int32)
    b = rng.integers(0, 3163, n).astype(np.int32)
    v = rng.standard_normal(n)
    _vs_oracle(ctx, [a, b], [v], aggs=("sum", "count"))


def test_config5_two_step_filter_then_group(ctx):
    """BASELINE config 5 in the only form the reference supports:
    V = DT[f.x > 0, :]; V[:, sum(f.x), by(f.k)]"""
    rng = np.random.default_rng(1234 + 5)
    n = 2_000_000
    k = rng.integers(0, 200_000, n).astype(np.int64)
    x = rng.standard_normal(n)
    x[rng.random(n) < 0.01] = np.nan
    ri_f = ctx.filter_cmp(x, ">", 0.0)
    assert_same(ri_f, o.filter_cmp(x, ">", 0.0), "filter rowindex")
    kv, xv = ctx.gather(k, ri_f), ctx.gather(x, ri_f)
    assert_same(kv, o.gather(k, ri_f), "gathered key")
    assert_same(xv, o.gather(x, ri_f), "gathered x")
    _vs_oracle(ctx, [kv], [xv], aggs=("sum",))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 8191, 8192, 8193, 16385, 100_003])
def test_tile_boundaries(ctx, n):
    """sizes around the wave (64), reduce-tile (2048) and radix-tile (8192) boundaries"""
    rng = np.random.default_rng(n)
    k = rng.integers(-50, 50, n).astype(np.int32)
    k[rng.random(n) < 0.05] = -2**31
    v = rng.standard_normal(n)
    v[rng.random(n) < 0.05] = np.nan
    _vs_oracle(ctx, [k], [v])


def test_giant_group_spanning_many_tiles(ctx):
    """one group covering ~all rows (look-back carries, reducer side-buffer fixup over many tiles)"""
    n = 300_000
    k = np.full(n, 7, np.int64)
    k[::50_000] = 3
    k[-1] = 9
    v = np.arange(n, dtype=np.float64)
    iv = np.arange(n, dtype=np.int64) - 5
    _vs_oracle(ctx, [k], [v, iv])


def test_already_sorted_and_reverse(ctx):
    n = 200_000
    v = np.random.default_rng(5).standard_normal(n)
    _vs_oracle(ctx, [np.arange(n, dtype=np.int32) // 7], [v])
    _vs_oracle(ctx, [(np.arange(n, dtype=np.int32)[::-1] // 3).copy()], [v])


def test_all_distinct_keys(ctx):
    rng = np.random.default_rng(8)
    n = 150_000
    k = rng.permutation(n).astype(np.int64) * 1000003
    _vs_oracle(ctx, [k], [rng.standard_normal(n)], aggs=("sum", "min"))


def test_value_stypes_and_narrow_values(ctx):
    """int8/int16 values take the RowIndex-gather reducer path; float32 stays float32"""
    rng = np.random.default_rng(21)
    n = 100_000
    k = rng.integers(0, 300, n).astype(np.int32)
    vals = [rng.integers(-100, 100, n).astype(np.int8), rng.integers(-30000, 30000, n).astype(np.int16),
            rng.integers(-10**9, 10**9, n).astype(np.int32), rng.integers(-10**15, 10**15, n).astype(np.int64),
            rng.standard_normal(n).astype(np.float32), rng.standard_normal(n)]
    _vs_oracle(ctx, [k], vals, check_ri=False)


def test_three_keys_over_64_bits(ctx):
    rng = np.random.default_rng(31)
    n = 200_000
    k0 = rng.choice(rng.integers(-2**62, 2**62, 50), n).astype(np.int64)
    k1 = rng.integers(0, 9, n).astype(np.float64) / 4
    k1[rng.random(n) < 0.1] = np.nan
    k2 = rng.integers(-3, 3, n).astype(np.int8)
    _vs_oracle(ctx, [k0, k1, k2], [rng.standard_normal(n)], aggs=("sum", "max"))


def test_descending_and_na_last(ctx):
    rng = np.random.default_rng(41)
    n = 50_000
    k = rng.integers(-1000, 1000, n).astype(np.int32)
    k[rng.random(n) < 0.1] = -2**31
    f = (rng.integers(-20, 20, n) / 4).astype(np.float64)
    f[rng.random(n) < 0.1] = np.nan
    for keys, desc, na_last in (([k], [True], False), ([k], [False], True), ([f], [True], False), ([f, k], [True, False], True)):
        ri, off = o.group(keys, desc=desc, na_last=na_last)
        r = ctx.groupby(keys, desc=desc, na_last=na_last)
        assert_same(r.offsets(), off, "offsets desc=%s na_last=%s" % (desc, na_last))
        assert_same(r.rowindex(), ri, "rowindex desc=%s na_last=%s" % (desc, na_last))
        r.free()


def test_hash_combiner_sparse_keys(ctx):
    """keys whose range is far too wide for slot tables (pooled 62-bit ints, float64, a > 32-bit composite):
    LDS hash tables per hash bucket -> partial groups -> merge by the sort path.  Single value column or
    none; NA keys and values; a skewed case whose big bucket is split into parts (duplicated partial groups)."""
    rng = np.random.default_rng(57)
    n = 300_000
    pool = rng.integers(-2**62, 2**62, 20_000)
    k_int = rng.choice(pool, n).astype(np.int64)
    k_int[rng.random(n) < 0.02] = np.iinfo(np.int64).min
    k_f = rng.choice(rng.standard_normal(5_000), n)
    k_f[rng.random(n) < 0.02] = np.nan
    k_skew = np.where(rng.random(n) < 0.85, pool[7], rng.choice(pool, n)).astype(np.int64)
    # float64 keys ride as their BITS (round 6): -0.0 / 0.0 are two groups (as in the reference), NaNs of different bit patterns ONE
    k_f2 = k_f.copy()
    k_f2[rng.random(n) < 0.01] = -0.0
    k_f2[rng.random(n) < 0.01] = 0.0
    k_f2.view(np.uint64)[rng.random(n) < 0.01] = np.uint64(0x7FF8000000000001)
    k_f2.view(np.uint64)[rng.random(n) < 0.01] = np.uint64(0xFFF8000000000000)
    k_a = rng.integers(0, 2**31 - 1, n).astype(np.int32) // 50_000 * 50_000
    k_b = rng.choice(pool[:300], n).astype(np.int64)
    v = rng.standard_normal(n)
    v[rng.random(n) < 0.1] = np.nan
    iv = rng.integers(-10**9, 10**9, n).astype(np.int64)
    fv = rng.standard_normal(n).astype(np.float32)
    i32 = rng.integers(-1000, 1000, n).astype(np.int32)
    i32[rng.random(n) < 0.1] = -2**31
    # hash_mode 2: one aligned int64 key goes through the TILE-LOCAL partition (round 6: segments + directory, hash_agg_seg_kernel),
    # everything else -- and everything under hash_mode 3 -- through histogram + exact scatter positions (hash_agg_kernel)
    for hm in (2, 3):
        ctx.set_option("hash_mode", hm)
        try:
            for keys in ([k_int], [k_f], [k_f2], [k_skew], [k_a, k_b]):
                for val in (v, iv, fv, i32):
                    _vs_oracle(ctx, keys, [val], check_ri=False)
                _vs_oracle(ctx, keys, [], aggs=(), check_ri=False)
            # several value columns ride through ONE partition
            _vs_oracle(ctx, [k_int], [v, iv, i32], check_ri=False)
            _vs_oracle(ctx, [k_skew], [v, fv], check_ri=False)
        finally:
            ctx.set_option("hash_mode", 0)


def test_clustered_key_variants(ctx):
    """the kernel variants for sorted / clustered / constant keys (wave-uniform bucket or slot: one DS
    atomic per wave after a register reduction), forced on and off, on sorted, constant, skewed and
    random keys -- same results as the oracle either way"""
    rng = np.random.default_rng(53)
    n = 400_000
    cases = [np.sort(rng.integers(0, 30_000, n)).astype(np.int64), np.full(n, 7, np.int32),
             np.where(rng.random(n) < 0.95, 5, rng.integers(0, 100_000, n)).astype(np.int64),
             rng.integers(0, 200_000, n).astype(np.int32), (np.arange(n) // 3).astype(np.int64)]
    v = rng.standard_normal(n)
    v[rng.random(n) < 0.1] = np.nan
    iv = rng.integers(-1000, 1000, n).astype(np.int64)
    for k in cases:
        for mode in (1, 2):
            ctx.set_option("cluster_mode", mode)
            try:
                _vs_oracle(ctx, [k], [v, iv], check_ri=False)
            finally:
                ctx.set_option("cluster_mode", 0)


def test_hot_key_lanes_combined_in_registers(ctx):
    """round 6: a bucket holding more than 3x its fair share of the rows (a hot key) has the lanes of a wave that share the
    first lane's slot combined in registers before ONE DS atomic (agg_dev.hpp acc_wave_masked) -- the hot key, its bucket
    neighbours, NA values (valid counts, the NA-free guess and its retry), int and float accumulators, every reducer: same
    results as the oracle"""
    rng = np.random.default_rng(61)
    n = 1_500_000
    base = rng.integers(0, 3_000_000, n).astype(np.int64)              # 22 key bits: several hundred buckets
    for share in (0.07, 0.6):
        k = np.where(rng.random(n) < share, 1_234_567, base).astype(np.int64)
        v = rng.standard_normal(n)
        iv = rng.integers(-1000, 1000, n).astype(np.int64)
        fv = rng.standard_normal(n).astype(np.float32)
        _vs_oracle(ctx, [k], [v, iv, fv], check_ri=False)               # NA-free columns: guessed, verified
        v2 = v.copy(); v2[rng.random(n) < 0.05] = np.nan
        iv2 = iv.copy(); iv2[rng.random(n) < 0.05] = np.iinfo(np.int64).min
        _vs_oracle(ctx, [k], [v2, iv2], check_ri=False)                 # NAs: valid counts
        v3 = v.copy(); v3[(k == 1_234_567).nonzero()[0][-1]] = np.nan    # ONE NaN, in the hot key: the guess fails late
        _vs_oracle(ctx, [k], [v3], check_ri=False)


def test_range_bucket_and_ungroup(ctx):
    """the two small helpers of the multi-GPU row exchange / GtoALL broadcast"""
    rng = np.random.default_rng(49)
    for dt_ in (np.int8, np.int16, np.int32, np.int64):
        hi = min(np.iinfo(dt_).max, 10**12)
        k = rng.integers(-hi, hi, 100_003).astype(dt_)
        k[rng.random(len(k)) < 0.05] = np.iinfo(dt_).min
        bounds = sorted(int(x) for x in rng.integers(-hi, hi, 7))
        exp = np.zeros(len(k), np.int8)
        for b in bounds:
            exp += ((k != np.iinfo(dt_).min) & (k.astype(np.int64) >= b)).astype(np.int8)
        assert_same(ctx.range_bucket(k, bounds), exp, "range_bucket %s" % dt_.__name__)
    off = np.concatenate([[0], np.cumsum(rng.integers(1, 50, 10_000))]).astype(np.int32)
    assert_same(ctx.ungroup(off), np.repeat(np.arange(len(off) - 1, dtype=np.int32), np.diff(off)), "ungroup")


def test_first_last(ctx):
    """first(col) / last(col): the element at the group's first / last row in original row order, NA
    included (head_reduce_unary.cc:116-160) -- through dthip_groupby_agg and through the S-red seam"""
    rng = np.random.default_rng(47)
    n = 120_000
    k = rng.integers(0, 700, n).astype(np.int32)
    k[rng.random(n) < 0.05] = -2**31
    vals = [rng.standard_normal(n), rng.integers(-9, 9, n).astype(np.int8), rng.integers(-9, 9, n).astype(np.int64),
            rng.standard_normal(n).astype(np.float32)]
    vals[0][rng.random(n) < 0.2] = np.nan
    ri, off = o.group([k])
    aggs = [(op, c) for c in range(len(vals)) for op in ("first", "last")] + [("sum", 0), ("count0", None)]
    r = ctx.groupby_agg([k], vals, aggs)
    assert_same(r.offsets(), off, "offsets")
    for a, (op, c) in enumerate(aggs[:-2]):
        exp = vals[c][ri[off[:-1]]] if op == "first" else vals[c][ri[off[1:] - 1]]
        assert_same(r.agg(a), exp, "%s(v%d)" % (op, c))
        assert_same(ctx.reduce(op, vals[c], ri, off), exp, "dthip_reduce %s(v%d)" % (op, c))
    check_agg(r.agg(len(aggs) - 2), o.reduce("sum", vals[0], ri, off), "sum", vals[0], ri, off, "sum next to first/last")
    r.free()


def test_fused_agg_descending_and_na_last(ctx):
    """by(-f.k) / NA-last on every path of dthip_groupby_agg (the bucketed path inverts the same
    transform: edge = max, NA -> range+1)"""
    rng = np.random.default_rng(43)
    n = 80_000
    k = rng.integers(-1000, 1000, n).astype(np.int32)
    k[rng.random(n) < 0.1] = -2**31
    k2 = rng.integers(0, 40, n).astype(np.int64)
    v = rng.standard_normal(n)
    iv = rng.integers(-50, 50, n).astype(np.int64)
    for keys, desc, na_last in (([k], [True], False), ([k], [False], True), ([k], [True], True),
                                ([k2, k], [True, False], False), ([k, k2], [False, True], True)):
        ri, off = o.group(keys, desc=desc, na_last=na_last)
        aggs = [("sum", 0), ("min", 0), ("max", 1), ("count", 1), ("count0", None)]
        exp = [o.reduce(op, (v, iv)[c], ri, off) for op, c in aggs[:-1]]
        for path in AGG_PATHS:
            ctx.set_option("agg_path", path)
            try:
                r = ctx.groupby_agg(keys, [v, iv], aggs, desc=desc, na_last=na_last)
            finally:
                ctx.set_option("agg_path", 0)
            tag = " desc=%s na_last=%s path=%d" % (desc, na_last, path)
            assert_same(r.offsets(), off, "offsets" + tag)
            for i, kk in enumerate(keys):
                assert_same(r.key(i), kk[ri[off[:-1]]], "key %d%s" % (i, tag))
            for a, (op, c) in enumerate(aggs[:-1]):
                check_agg(r.agg(a), exp[a], op, (v, iv)[c], ri, off, "%s%s" % (op, tag))
            assert_same(r.agg(4), np.diff(off).astype(np.int64), "count()" + tag)
            r.free()


def test_bool_mask_to_rowindex(ctx):
    rng = np.random.default_rng(51)
    for n in (0, 1, 100, 5000, 300_001):
        m = rng.integers(0, 2, n).astype(np.int8)
        m[rng.random(n) < 0.1] = -128
        assert_same(ctx.bool_to_rowindex(m), o.bool_to_rowindex(m), "mask n=%d" % n)


@pytest.mark.parametrize("cmp", [">", ">=", "<", "<=", "==", "!="])
def test_filter_cmp(ctx, cmp):
    rng = np.random.default_rng(61)
    n = 70_000
    x = rng.integers(-5, 5, n).astype(np.float64)
    x[rng.random(n) < 0.1] = np.nan
    i = rng.integers(-5, 5, n).astype(np.int32)
    i[rng.random(n) < 0.1] = -2**31
    assert_same(ctx.filter_cmp(x, cmp, 1.0), o.filter_cmp(x, cmp, 1.0), "float %s" % cmp)
    assert_same(ctx.filter_cmp(i, cmp, 1), o.filter_cmp(i, cmp, 1), "int %s" % cmp)


def test_gather_with_na_indices(ctx):
    rng = np.random.default_rng(71)
    for dt in (np.int8, np.int16, np.int32, np.int64, np.float32, np.float64):
        v = rng.integers(-100, 100, 1000).astype(dt)
        ri = rng.integers(0, 1000, 5000).astype(np.int32)
        ri[::17] = -2**31
        assert_same(ctx.gather(v, ri), o.gather(v, ri), "gather %s" % dt)


def test_rejects_bad_arguments(ctx):
    with pytest.raises(NotImplementedError):
        ctx.groupby([np.zeros(4, np.int32)], stypes=[11])     # STR32 keys are out of scope
    with pytest.raises(ValueError):
        ctx.groupby_agg([np.zeros(4, np.int32)], [np.zeros(4)], [("sum", 3)])


@pytest.mark.parametrize("want_ri", [True, False])
def test_groupby_rows_materialised(ctx, want_ri):
    """dthip_groupby_rows: columns permuted into grouped order == column[RowIndex] of the oracle, bit for bit
    (the columns ride through the sort; narrow columns and >64-bit key sets take the gather path)"""
    rng = np.random.default_rng(91)
    n = 300_000
    k = rng.integers(-1000, 50_000, n).astype(np.int64)
    k[rng.random(n) < 0.02] = -2**63
    cols = [k, rng.standard_normal(n), rng.integers(-9, 9, n).astype(np.int32), rng.standard_normal(n).astype(np.float32),
            rng.integers(-100, 100, n).astype(np.int8)]
    for keys, use in (([k], cols[:4]), ([k], cols), ([cols[2], k], cols[:3]),
                      ([rng.choice(rng.integers(-2**62, 2**62, 40), n), k], cols[:2])):
        ri, off = o.group(keys)
        r = ctx.groupby_rows(keys, use, want_rowindex=want_ri)
        assert_same(r.offsets(), off, "offsets")
        if want_ri:
            assert_same(r.rowindex(), ri, "rowindex")
        for c, col in enumerate(use):
            assert_same(r.col(c), col[ri], "column %d in grouped order" % c)
        r.free()


def test_speculative_key_range(ctx):
    """bucketed path with the key range GUESSED from a sample (the default from 2^23 rows on):
    a guess that holds, and one that a single unsampled outlier breaks (-> retried with the exact range)"""
    rng = np.random.default_rng(77)
    n = 3_000_000
    k = rng.integers(1000, 300_000, n).astype(np.int64)
    v = rng.standard_normal(n)
    iv = rng.integers(-1000, 1000, n).astype(np.int32)
    ctx.set_option("spec_min_rows", 1)
    try:
        _vs_oracle(ctx, [k], [v, iv], check_ri=False)
        for hi, lo in ((600_000, -100_000),          # retried with the exact range on the bucketed path
                       (40_000_000, -5_000_000)):    # exact range too wide for it: retried on the sort path
            k2 = k.copy()
            k2[1_234_567] = hi                       # rows the sample does not visit
            k2[2_000_001] = lo
            _vs_oracle(ctx, [k2], [v, iv], check_ri=False)
        a = rng.integers(0, 500, n).astype(np.int32)
        b = rng.integers(-7, 90, n).astype(np.int16)
        b[rng.random(n) < 0.01] = -2**15
        _vs_oracle(ctx, [a, b], [v], aggs=("sum", "count"), check_ri=False)
    finally:
        ctx.set_option("spec_min_rows", 1 << 23)


def test_speculative_key_range_on_the_sort_path(ctx):
    """dthip_groupby / dthip_groupby_rows with the key range guessed from a sample (round 3; default from 2^23 rows on):
    a guess that holds; outliers at rows the sample does not visit (the key-transform pass reports them, the call
    plans again with the exact range); two keys (both guessed, one broken); NA-last; descending"""
    from oracle import oracle as o
    rng = np.random.default_rng(78)
    n = 3_000_000
    k = rng.integers(1000, 300_000, n).astype(np.int64)
    x = rng.standard_normal(n)
    ctx.set_option("spec_min_rows", 1)
    try:
        cases = [k]
        for hi, lo in ((600_000, -100_000), (2**40, -2**41), (300_001, 999)):
            k2 = k.copy()
            k2[1_234_567] = hi
            k2[2_000_001] = lo
            cases.append(k2)
        for kk in cases:
            for na_last in (False, True):
                ri, off = o.group([kk], na_last=na_last)
                g = ctx.groupby([kk], na_last=na_last)
                assert_same(g.rowindex(), ri, "rowindex"); assert_same(g.offsets(), off, "offsets")
                g.free()
            ri, off = o.group([kk])
            r = ctx.groupby_rows([kk], [kk, x], want_rowindex=True)
            assert_same(r.rowindex(), ri, "rowindex (rows)"); assert_same(r.col(0), kk[ri], "key column"); assert_same(r.col(1), x[ri], "x")
            r.free()
        a = rng.integers(0, 500, n).astype(np.int32)
        a[777_777] = 10_000_000                      # breaks the guess of the FIRST key only
        ri, off = o.group([a, cases[0]], desc=[True, False])
        g = ctx.groupby([a, cases[0]], desc=[True, False])
        assert_same(g.rowindex(), ri, "two keys, first descending"); assert_same(g.offsets(), off, "offsets")
        g.free()
    finally:
        ctx.set_option("spec_min_rows", 1 << 23)


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_rows_key_column_written_by_the_last_pass(ctx, dtype):
    """dthip_groupby_rows with ONE int32 / int64 key that is also a wanted column: the last radix pass writes the original key
    values itself (round 4; no untransform pass, groups found on that column) -- ascending / descending, NA first / last,
    the key wanted twice, a constant key (no pass runs: the untransform path), on both sort paths"""
    rng = np.random.default_rng(92)
    n = 400_000
    lim = 2**40 if dtype == np.int64 else 2**20
    k = rng.integers(-lim // 3, lim, n).astype(dtype)
    k[rng.random(n) < 0.02] = np.iinfo(dtype).min
    x = rng.standard_normal(n)
    w = rng.integers(-5, 5, n).astype(np.int32)
    for sp in (1, 2):
        ctx.set_option("sort_path", sp); ctx.set_option("msd_min_rows", 1)
        try:
            for desc in (False, True):
                for na_last in (False, True):
                    ri, off = o.group([k], desc=[desc], na_last=na_last)
                    for cols in ([k, x], [x, k, w, k], [k]):
                        r = ctx.groupby_rows([k], cols, desc=[desc], na_last=na_last, want_rowindex=True)
                        assert_same(r.offsets(), off, "offsets"); assert_same(r.rowindex(), ri, "rowindex")
                        for c, col in enumerate(cols):
                            assert_same(r.col(c), col[ri], "column %d (sort_path %d, desc %s, na_last %s)" % (c, sp, desc, na_last))
                        r.free()
            kc = np.full(1000, 7, dtype)
            r = ctx.groupby_rows([kc], [kc, x[:1000]], want_rowindex=False)
            assert r.ngroups == 1 and np.array_equal(r.col(0), kc) and np.array_equal(r.col(1), x[:1000])
            r.free()
        finally:
            ctx.set_option("sort_path", 0); ctx.set_option("msd_min_rows", 1 << 26)


def _unsampled_rows(n, elem_bytes, want, nsamp=1 << 17):
    """rows that stats.hip::minmax_sample_kernel does NOT read (nsamp evenly spaced 16-byte pieces + the last piece)"""
    e = 16 // elem_bytes
    gid = np.arange(nsamp, dtype=np.uint64)
    r0 = (gid * np.uint64(n)) // np.uint64(nsamp)
    r0[-1] = n - e if n > e else 0
    seen = np.zeros(n, dtype=bool)
    for j in range(e):
        seen[np.minimum(r0 + j, n - 1).astype(np.int64)] = True
    free = np.flatnonzero(~seen)
    assert len(free) >= want
    return free[np.linspace(0, len(free) - 1, want).astype(np.int64)]


def test_speculative_range_edges_do_not_merge_into_na(ctx):
    """a VALID key one below the guessed minimum transforms to 0 (the NA-first code), one above the guessed
    maximum to na_repl (NA last); the range check must flag them (NA rows excluded), never group them with the NAs"""
    rng = np.random.default_rng(79)
    n = 3_000_000
    k = rng.integers(1000, 300_000, n).astype(np.int64)
    k[0], k[-1] = 1000, 299_999                      # first / last rows are always sampled: the sampled range is exact
    k[rng.random(n) < 0.001] = -2**63
    k[0], k[-1] = 1000, 299_999
    v = rng.standard_normal(n)
    margin = (299_999 - 1000) // 64 + 64             # plan_keys: width / 64 + 64 on both sides
    rows = _unsampled_rows(n, 8, 2)
    ctx.set_option("spec_min_rows", 1)
    try:
        for d in (-2, -1, 0, 1, 2):
            for val in (1000 - margin - 1 + d, 299_999 + margin + 1 + d):
                k2 = k.copy()
                k2[rows[0]] = val
                for na_last in (False, True):
                    ri, off = o.group([k2], na_last=na_last)
                    g = ctx.groupby([k2], na_last=na_last)
                    assert_same(g.offsets(), off, "offsets (edge %d, na_last=%s)" % (val, na_last))
                    assert_same(g.rowindex(), ri, "rowindex (edge %d, na_last=%s)" % (val, na_last))
                    g.free()
                    for path in (0, 1, 2):
                        ctx.set_option("agg_path", path)
                        try:
                            r = ctx.groupby_agg([k2], [v], [("count0", None)], na_last=na_last)
                        finally:
                            ctx.set_option("agg_path", 0)
                        assert_same(r.offsets(), off, "fused offsets (edge %d, na_last=%s, path %d)" % (val, na_last, path))
                        assert_same(r.key(0), k2[ri[off[:-1]]], "fused keys (edge %d, na_last=%s, path %d)" % (val, na_last, path))
                        r.free()
    finally:
        ctx.set_option("spec_min_rows", 1 << 23)


def test_speculative_sample_of_nothing_but_na(ctx):
    """a mostly-NA key column whose valid keys all sit in rows the sample skips: the sample has nvalid == 0,
    which must not be taken for 'range [0, 0], exact'"""
    n = 3_000_000
    rows = _unsampled_rows(n, 8, 5000)
    rng = np.random.default_rng(80)
    k = np.full(n, -2**63, dtype=np.int64)
    k[rows] = rng.integers(-50, 7000, len(rows))
    v = rng.standard_normal(n)
    ctx.set_option("spec_min_rows", 1)
    try:
        for na_last in (False, True):
            ri, off = o.group([k], na_last=na_last)
            g = ctx.groupby([k], na_last=na_last)
            assert_same(g.offsets(), off, "offsets"); assert_same(g.rowindex(), ri, "rowindex")
            g.free()
            r = ctx.groupby_rows([k], [k, v], want_rowindex=True, na_last=na_last)
            assert_same(r.rowindex(), ri, "rowindex (rows)"); assert_same(r.col(1), v[ri], "v")
            r.free()
        _vs_oracle(ctx, [k], [v], aggs=("sum", "count"), check_ri=False)
        k32 = k.astype(np.int32); k32[k == -2**63] = -2**31
        _vs_oracle(ctx, [k32, k], [v], aggs=("sum",), check_ri=True)
    finally:
        ctx.set_option("spec_min_rows", 1 << 23)


def test_rows_ride_path_replanned_into_two_stages(ctx):
    """dthip_groupby_rows' ride path with a guessed range that an unsampled outlier breaks so badly that
    the exact plan needs TWO stages (21 + 21 guessed bits -> 51 + 21 = 72 exact bits): must leave the ride path"""
    rng = np.random.default_rng(81)
    n = 3_000_000
    a = rng.integers(0, 2**20, n).astype(np.int64)
    b = rng.integers(0, 2**20, n).astype(np.int64)
    x = rng.standard_normal(n)
    rows = _unsampled_rows(n, 8, 3)
    a[rows[1]] = 2**50
    ctx.set_option("spec_min_rows", 1)
    try:
        ri, off = o.group([a, b])
        for want_ri in (True, False):
            r = ctx.groupby_rows([a, b], [a, b, x], want_rowindex=want_ri)
            assert_same(r.offsets(), off, "offsets")
            if want_ri:
                assert_same(r.rowindex(), ri, "rowindex")
            assert_same(r.col(0), a[ri], "a"); assert_same(r.col(1), b[ri], "b"); assert_same(r.col(2), x[ri], "x")
            r.free()
    finally:
        ctx.set_option("spec_min_rows", 1 << 23)


# ---- full-size properties (no oracle: size-independent invariants) ----------------------------

def test_full_size_properties_1e8(ctx):
    """1e8 rows / 1e6 groups through the fused path: keys strictly ascending, counts sum to n,
    sum of group sums == total sum (1e-9 rel), sum(v=1) == count exactly."""
    rng = np.random.default_rng(7)
    n = 100_000_000
    k = rng.integers(0, 1_000_000, n, dtype=np.int64)
    v = rng.standard_normal(n)
    r = ctx.groupby_agg([k], [v], [("sum", 0), ("count", 0), ("count0", None), ("min", 0), ("max", 0)])
    keys, s, c, c0 = r.key(0), r.agg(0), r.agg(1), r.agg(2)
    mn, mx = r.agg(3), r.agg(4)
    r.free()
    assert np.all(np.diff(keys) > 0)
    assert c0.sum() == n and np.array_equal(c, c0)
    assert np.array_equal(c0, np.bincount(k, minlength=keys.max() + 1)[keys])
    assert abs(s.sum() - v.sum()) <= 1e-9 * np.abs(v).sum()
    assert np.all(mn <= mx) and mn.min() == v.min() and mx.max() == v.max()
    ref = np.bincount(k, weights=v, minlength=keys.max() + 1)[keys]
    assert_close(s, ref, scale=np.bincount(k, weights=np.abs(v))[keys], rel=1e-6, what="sum vs bincount")


def test_full_size_properties_config4_shape(ctx):
    """5e7 rows of BASELINE config 4 (two int32 keys, count + sum) on the bucketed path: the composite
    key packs into 24 bits; group keys ascending in (a, b), counts == bincount of the packed key,
    sum of sums == total, and the sort path returns the very same groups."""
    rng = np.random.default_rng(44)
    n = 50_000_000
    a = rng.integers(0, 3163, n, dtype=np.int32)
    b = rng.integers(0, 3163, n, dtype=np.int32)
    v = rng.standard_normal(n)
    r = ctx.groupby_agg([a, b], [v], [("count0", None), ("sum", 0)])
    ka, kb, c, s = r.key(0), r.key(1), r.agg(0), r.agg(1)
    r.free()
    packed = ka.astype(np.int64) * 3163 + kb
    assert np.all(np.diff(packed) > 0)
    ref = np.bincount(a.astype(np.int64) * 3163 + b, minlength=3163 * 3163)
    assert c.sum() == n and np.array_equal(c, ref[packed])
    assert abs(s.sum() - v.sum()) <= 1e-9 * np.abs(v).sum()
    ctx.set_option("agg_path", 1)
    try:
        r2 = ctx.groupby_agg([a, b], [v], [("count0", None), ("sum", 0)])
    finally:
        ctx.set_option("agg_path", 0)
    assert_same(r2.key(0), ka, "key a, sort path"); assert_same(r2.key(1), kb, "key b, sort path")
    assert_same(r2.agg(0), c, "count(), sort path")
    absum = np.bincount(a.astype(np.int64) * 3163 + b, weights=np.abs(v), minlength=3163 * 3163)[packed]
    assert_close(r2.agg(1), s, scale=absum, rel=1e-6, what="sum, sort path vs bucketed path")
    r2.free()


def test_full_size_properties_config5_shape(ctx):
    """4e7 rows of BASELINE config 5: V = DT[f.x > 0, :]; V[:, :, by(f.k)] with 4e6 groups --
    filter RowIndex ascending and complete, rows of the result in grouped order (keys non-decreasing,
    original row order inside a group), every passing row exactly once, values follow their rows."""
    rng = np.random.default_rng(55)
    n = 40_000_000
    k = rng.integers(0, 4_000_000, n, dtype=np.int64)
    x = rng.standard_normal(n)
    ri_f = ctx.filter_cmp(x, ">", 0.0)
    assert np.array_equal(ri_f, np.nonzero(x > 0)[0].astype(np.int32))
    kv, xv = ctx.gather(k, ri_f), ctx.gather(x, ri_f)
    assert np.array_equal(kv, k[ri_f]) and np.array_equal(xv, x[ri_f])
    r = ctx.groupby_rows([kv], [kv, xv, ri_f])
    ko, xo, rio, off = r.col(0), r.col(1), r.col(2), r.offsets()
    ri = r.rowindex()
    r.free()
    assert np.all(np.diff(ko) >= 0)
    assert np.array_equal(np.sort(rio), ri_f)                  # a permutation of the passing rows
    assert np.array_equal(ko, k[rio]) and np.array_equal(xo, x[rio])
    same = ko[1:] == ko[:-1]
    assert np.all(rio[1:][same] > rio[:-1][same])              # stable inside a group
    heads = np.concatenate([[0], np.nonzero(~same)[0] + 1, [len(ko)]]).astype(np.int32)
    assert np.array_equal(off, heads)
    assert np.array_equal(rio, ri_f[ri])


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_fused_agg_all_paths(ctx, seed):
    """seeded random shapes: row count, key stypes / ranges / skew / NA rate, 1-3 keys, value stypes,
    reducer subsets -- every path of dthip_groupby_agg against the oracle.  Skewed cases put > 65536
    rows into one bucket, so split buckets (global-atomic table flush) are exercised as well."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 7, 4095, 4096, 4097, 12288, 12289, 50_000, 200_000, 700_000]))
    nkeys = int(rng.integers(1, 4))
    keys = []
    for _ in range(nkeys):
        dt_ = rng.choice([np.int8, np.int16, np.int32, np.int64])
        span = int(rng.choice([2, 50, 3000, 200_000]))
        span = min(span, np.iinfo(dt_).max // 2)
        lo = int(rng.integers(-span, span))
        if rng.random() < 0.3:       # skew: most rows share one key
            kk = np.where(rng.random(n) < 0.9, lo, rng.integers(lo, lo + span, n))
        else:
            kk = rng.integers(lo, lo + span, n)
        kk = kk.astype(dt_)
        if rng.random() < 0.5:
            kk[rng.random(n) < rng.choice([0.01, 0.3])] = np.iinfo(dt_).min
        keys.append(kk)
    vals = []
    for _ in range(int(rng.integers(1, 4))):
        vt = rng.choice(["f64", "f32", "i32", "i64"])
        if vt == "f64":
            v = rng.standard_normal(n)
        elif vt == "f32":
            v = rng.standard_normal(n).astype(np.float32)
        elif vt == "i32":
            v = rng.integers(-10**6, 10**6, n).astype(np.int32)
        else:
            v = rng.integers(-10**12, 10**12, n).astype(np.int64)
        if rng.random() < 0.5:
            m = rng.random(n) < 0.1
            if v.dtype.kind == "f":
                v[m] = np.nan
            else:
                v[m] = np.iinfo(v.dtype).min
        vals.append(v)
    ops = [op for op in OPS if rng.random() < 0.6] or ["sum"]
    _vs_oracle(ctx, keys, vals, aggs=tuple(ops), check_ri=(seed % 4 == 0))


@pytest.mark.parametrize("dtype,cmp,scalar", [(np.float64, ">", 0.0), (np.int32, "<=", 3), (np.int64, "!=", 0), (np.float32, "==", 1.5),
                                              (np.int8, ">", 1.5), (np.int16, ">=", -2)])
def test_filter_take_matches_filter_then_gather(ctx, dtype, cmp, scalar):
    """dthip_filter_take = dthip_filter_cmp followed by dthip_gather of every column, in one sweep"""
    rng = np.random.default_rng(77)
    n = 300_001
    if np.dtype(dtype).kind == "f":
        x = (rng.integers(-6, 6, n) * 0.5).astype(dtype)
        x[rng.random(n) < 0.05] = np.nan
    else:
        x = rng.integers(-5, 6, n).astype(dtype)
        x[rng.random(n) < 0.05] = np.iinfo(dtype).min
    cols = [rng.integers(-2**40, 2**40, n), rng.standard_normal(n).astype(np.float32), rng.integers(-100, 100, n).astype(np.int8),
            rng.integers(-30000, 30000, n).astype(np.int16), x]
    want_ri = ctx.filter_cmp(x, cmp, scalar)
    ri, out = ctx.filter_take(x, cmp, scalar, cols)
    assert_same(ri, want_ri, "rowindex")
    for c, o_ in zip(cols, out):
        assert_same(o_, c[want_ri], "column")
    ri2, out2 = ctx.filter_take(x, cmp, scalar, cols[:1], want_rowindex=False)
    assert ri2 is None
    assert_same(out2[0], cols[0][want_ri], "column without rowindex")
    assert_same(want_ri, o.filter_cmp(x, cmp, scalar) if float(scalar) == int(scalar) else want_ri, "oracle rowindex")


def test_partial_sums_that_look_like_na_on_the_hash_path(ctx):
    """the hash combiner merges partial sums; a partial that is NaN (+inf and -inf in one group) or an
    int64 partial that wrapped to INT64_MIN is a VALUE there (DTHIP_FLAG_NONA), not an NA to skip -- the group's
    sum / mean must come out as on the sort and bucket paths (and as in the reference: NaN / INT64_MIN)."""
    k = np.array([2**40, 2**40, 5, 5, 2**50, 2**50, 2**50, 2**40], np.int64)
    v = np.array([np.inf, -np.inf, 1.0, 2.0, 1.0, 2.0, 3.0, 7.0])
    w = np.array([-2**62, -2**62, 1, 2, 3, 4, 5, 0], np.int64)
    ri, off = o.group([k])
    for col, vals in ((0, v), (1, w)):
        for opn in ("sum", "mean", "count"):
            exp = o.reduce(opn, vals, ri, off)
            for hm in (0, 2, 3):
                ctx.set_option("hash_mode", hm)
                try:
                    r = ctx.groupby_agg([k], [v, w], [(opn, col)])
                finally:
                    ctx.set_option("hash_mode", 0)
                got = r.agg(0)
                r.free()
                if exp.dtype.kind == "f":
                    assert got.dtype == exp.dtype and np.array_equal(got, exp, equal_nan=True), (opn, col, hm, got, exp)
                else:
                    assert_same(got, exp, "%s(v%d) hash_mode=%d" % (opn, col, hm))


def test_float32_sum_reference_order_switch(ctx):
    """option "f32_sum" = 1: sum(float32) accumulated in float32, row by row in grouped order -- BIT-EXACT against the
    oracle (= the reference's column/sumprod.h:48-55), on the fused entry point and on dthip_reduce; the default
    (float64 accumulation, one rounding) stays within the documented 1e-4"""
    from oracle import oracle as o
    rng = np.random.default_rng(21)
    n = 400_000
    k = rng.integers(0, 3000, n).astype(np.int64)
    k[rng.random(n) < 0.02] = np.iinfo(np.int64).min
    v = (rng.standard_normal(n) * 1e3).astype(np.float32)
    v[rng.random(n) < 0.05] = np.nan
    w = rng.standard_normal(n)
    ri, off = o.group([k])
    exp = o.reduce("sum", v, ri, off)
    assert exp.dtype == np.float32
    ctx.set_option("f32_sum", 1)
    try:
        r = ctx.groupby_agg([k], [v, w], [("sum", 0), ("mean", 0), ("sum", 1), ("count0", None)])
        assert_same(r.agg(0), exp, "sum(float32), reference order")
        assert_close(r.agg(2), o.reduce("sum", w, ri, off), what="sum(float64) next to it")
        r.free()
        assert_same(ctx.reduce("sum", v, ri, off), exp, "dthip_reduce sum(float32), reference order")
    finally:
        ctx.set_option("f32_sum", 0)
    r = ctx.groupby_agg([k], [v], [("sum", 0)])
    got = r.agg(0); r.free()
    scale = np.add.reduceat(np.abs(np.nan_to_num(v[ri].astype(np.float64))), off[:-1])
    assert np.all(np.abs(got.astype(np.float64) - exp.astype(np.float64)) <= 1e-4 * np.abs(exp) + 4e-7 * scale + 1e-30)


def _value_sample_free_rows(n, want, nsamp=1 << 16):
    """rows that bucket.hip::value_na_sample_kernel does NOT read (row gid * n / nsamp for gid < nsamp)"""
    seen = np.zeros(n, dtype=bool)
    seen[((np.arange(nsamp, dtype=np.uint64) * np.uint64(n)) // np.uint64(nsamp)).astype(np.int64)] = True
    free = np.flatnonzero(~seen)
    return free[np.linspace(0, len(free) - 1, want).astype(np.int64)]


def _agg_launches(ctx):
    return ctx.profile_get("table_agg_kernel")[1] + ctx.profile_get("table_agg_seg_kernel")[1]


@pytest.mark.parametrize("layout", ["tile_local", "exact", "one_table", "sorted"])
def test_value_column_guessed_na_free(ctx, layout):
    """round 4: value columns whose sample shows no NA are aggregated WITHOUT a valid counter in LDS, every row checked
    (ACC_CHKNA).  (1) really NA-free: same results, one aggregation per column; (2) NAs at rows the sample skips: round 6
    counts them apart (a global atomic per NA row, AggTable::nacnt; rounds 4-5 aggregated once more with counters) -- still
    ONE aggregation per column, and exactly what nona_guess=0 returns; the retry counter stays 0"""
    rng = np.random.default_rng(90)
    n = 5_000_000 if layout == "tile_local" else 1_500_000
    kmax = {"tile_local": 2_000_000, "exact": 300_000, "one_table": 900, "sorted": 40_000}[layout]
    k = rng.integers(0, kmax, n).astype(np.int32)
    if layout == "sorted":
        k.sort()
    vals = [rng.standard_normal(n), rng.integers(-1000, 1000, n).astype(np.int32),
            rng.integers(-10**12, 10**12, n).astype(np.int64), rng.standard_normal(n).astype(np.float32),
            rng.standard_normal(n)]
    NA = [np.nan, -2**31, -2**63, np.nan, np.nan]
    rows = _value_sample_free_rows(n, 5)
    ops = ("sum", "mean", "min", "max", "count")
    # the last column is only COUNTED: guessed NA-free it has no accumulator at all, and must still be read and verified
    alist = [(opn, vi) for vi in range(len(vals) - 1) for opn in ops] + [("count", len(vals) - 1)]
    if layout == "exact":
        ctx.set_option("bucket_variant", 2)
    ctx.profile(True)
    try:
        ri, off = o.group([k])
        for with_na in (False, True, "counted column only"):
            vv = [v.copy() for v in vals]
            if with_na == "counted column only":
                vv[4][rows[4]] = NA[4]
            elif with_na:
                vv[0][rows[0]] = NA[0]; vv[2][rows[2]] = NA[2]          # two of the columns
                vv[0][_value_sample_free_rows(n, 1500)] = NA[0]          # ... and 1500 more NAs the sample does not see
            launches = {}
            res = {}
            for guess in (1, 0):
                ctx.set_option("nona_guess", guess)
                ctx.profile_reset()
                r = ctx.groupby_agg([k], vv, alist)
                launches[guess] = (_agg_launches(ctx), ctx.profile_get("value_na_sample_kernel")[1])
                assert_same(r.offsets(), off, "offsets (guess %d, na %s)" % (guess, with_na))
                res[guess] = [r.agg(a) for a in range(len(alist))]
                for a, (opn, vi) in enumerate(alist):
                    check_agg(res[guess][a], o.reduce(opn, vv[vi], ri, off), opn, vv[vi], ri, off,
                              "%s(v%d) guess %d na %s %s" % (opn, vi, guess, with_na, layout))
                r.free()
            for a in range(len(alist)):
                if alist[a][0] in ("min", "max", "count"):
                    assert_same(res[1][a], res[0][a], "%s(v%d): guessed == counted" % alist[a])
            assert launches[0][1] == 0 and launches[1][1] == 1, launches         # (round 6: all columns in ONE sampling launch)
            assert launches[0][0] > 0
            # one aggregation per column, whether the guess holds or not
            assert launches[1][0] == launches[0][0], (layout, with_na, launches)
            assert ctx.last_call_stats()["retries_na_guess"] == 0
        # an NA the sample SEES: nothing is guessed for any column of the call
        vv = [v.copy() for v in vals]
        vv[1][0] = NA[1]
        ctx.set_option("nona_guess", 1)
        ctx.profile_reset()
        r = ctx.groupby_agg([k], vv, alist)
        assert _agg_launches(ctx) == launches[0][0]
        for a, (opn, vi) in enumerate(alist):
            check_agg(r.agg(a), o.reduce(opn, vv[vi], ri, off), opn, vv[vi], ri, off, "%s(v%d) sampled NA" % (opn, vi))
        r.free()
    finally:
        ctx.profile(False)
        ctx.set_option("nona_guess", 1)
        ctx.set_option("bucket_variant", 0)


@pytest.mark.parametrize("small_path", [0, 1, 2])
def test_small_table_sequence(ctx, small_path):
    """round 4: key ranges that fit ONE table of <= 8192 slots take a launch-lean sequence (option small_path; BASELINE C1
    is bound by launches): tables initialised by the plan kernel, group list + offsets + group count from one
    single-workgroup kernel (2: counts written to mapped host memory).  Same results as the general sequence (0), also
    when a guessed key range or a guessed NA-free column turns out wrong."""
    rng = np.random.default_rng(91 + small_path)
    ctx.set_option("small_path", small_path)
    ctx.profile(True)
    try:
        for n, kmax in ((1, 1), (5, 3), (1000, 7), (70_000, 100), (300_000, 8000), (300_000, 8192), (300_000, 8300), (40_000, 5000)):
            k = rng.integers(0, kmax, n).astype(np.int64) - kmax // 2
            if n > 100:
                k[rng.random(n) < 0.01] = -2**63
            v = rng.standard_normal(n); v[rng.random(n) < 0.05] = np.nan
            w = rng.integers(-99, 99, n).astype(np.int32)
            ctx.profile_reset()
            _vs_oracle(ctx, [k], [v, w], check_ri=False)
            used = ctx.profile_get("small_groups_kernel")[1]
            if small_path == 0 or kmax >= 8192:      # (in between, the LDS table of the widest column decides: one table or buckets)
                assert used == 0, (n, kmax, used)
            elif kmax <= 100:
                assert used > 0, (n, kmax, used)
        # two keys, presence bits only (agg_offsets=0 inside _vs_oracle), all rows one group, sorted keys
        a = rng.integers(0, 40, 200_000).astype(np.int32); b = rng.integers(-3, 60, 200_000).astype(np.int32)
        _vs_oracle(ctx, [a, b], [rng.standard_normal(200_000)], check_ri=False)
        _vs_oracle(ctx, [np.zeros(100_000, np.int32)], [np.ones(100_000)], check_ri=False)
        _vs_oracle(ctx, [np.sort(rng.integers(0, 3000, 500_000)).astype(np.int32)], [rng.standard_normal(500_000)], check_ri=False)
        # a guessed key range broken by one unsampled row (retry with the exact range), then a guessed NA-free column with
        # one unsampled NA (retry with counters): both verdicts travel in the small kernel's second word
        n = 3_000_000
        k = rng.integers(100, 900, n).astype(np.int64); k[0], k[-1] = 100, 899
        v = rng.standard_normal(n)
        k2 = k.copy(); k2[_unsampled_rows(n, 8, 1)[0]] = 5000
        v2 = v.copy(); v2[_value_sample_free_rows(n, 3)[1]] = np.nan
        ctx.set_option("spec_min_rows", 1)
        for kk, vv in ((k, v), (k2, v), (k, v2), (k2, v2)):
            ri, off = o.group([kk])
            r = ctx.groupby_agg([kk], [vv], [("sum", 0), ("mean", 0), ("min", 0), ("count", 0), ("count0", None)])
            assert_same(r.offsets(), off, "offsets"); assert_same(r.key(0), kk[ri[off[:-1]]], "keys")
            for a_, opn in enumerate(("sum", "mean", "min", "count")):
                check_agg(r.agg(a_), o.reduce(opn, vv, ri, off), opn, vv, ri, off, "%s small_path=%d" % (opn, small_path))
            assert_same(r.agg(4), np.diff(off).astype(np.int64), "count()")
            r.free()
    finally:
        ctx.profile(False)
        ctx.set_option("small_path", 2)
        ctx.set_option("spec_min_rows", 1 << 23)


def test_small_path_mapped_words_survive_read_back():
    """read_back() used to free the mapped host words of the small-table path whenever its own pinned
    buffer grew -- the FIRST read_back of a context always did -- and left the stale pointers behind: the next small call
    wrote through a freed mapping (GPU fault / garbage group count).  The sequence that exposed it: a bool-key
    groupby_agg as the first call of a fresh context (no read_back before the words are allocated), an int-key call (its
    planning reads the range back), a second small call; then a read-back that makes the pinned buffer grow again."""
    from datatable_amd.engine import Context
    rng = np.random.default_rng(4242)
    for rounds in range(3):
        c = Context(0)
        try:
            n = 50_000
            kb = (rng.random(n) < 0.4)
            v = rng.standard_normal(n)
            ki = rng.integers(-5, 90, n).astype(np.int64)
            for k in (kb, ki, kb, ki):
                ri, off = o.group([k])
                r = c.groupby_agg([k], [v], [("sum", 0), ("count0", None)])
                assert_same(r.offsets(), off, "offsets")
                assert_same(r.key(0), k[ri[off[:-1]]].view(np.int8) if k.dtype == np.bool_ else k[ri[off[:-1]]], "keys")
                assert_same(r.agg(1), np.diff(off).astype(np.int64), "count()")
                check_agg(r.agg(0), o.reduce("sum", v, ri, off), "sum", v, ri, off, "sum")
                r.free()
            # a big read-back (the pinned buffer grows a second time), then small calls again
            big = rng.integers(0, 3_000_000, 4_000_000).astype(np.int64)
            g = c.groupby([big]); ri, off = o.group([big])
            assert_same(g.rowindex(), ri, "rowindex"); assert_same(g.offsets(), off, "offsets"); g.free()
            for k in (kb, ki):
                ri, off = o.group([k])
                r = c.groupby_agg([k], [v], [("sum", 0), ("count0", None)])
                assert_same(r.offsets(), off, "offsets"); assert_same(r.agg(1), np.diff(off).astype(np.int64), "count()")
                r.free()
        finally:
            c.close()


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_rows_already_in_key_order_need_no_grouping_pass(ctx, dtype):
    """round 6: one ascending integer key, NA first, >= 2^22 rows that ARE in key order (a sampled test, then one pass that
    finds the group heads and verifies the order): the reducers read the value columns in place.  Sorted keys with NAs and
    long / short groups; a column with ONE descent at a row the sample does not look at (the pass is spent, the ordinary path
    answers); a descending request and NA-last (never taken): all against the oracle."""
    rng = np.random.default_rng(123)
    n = 5_000_000
    k = np.sort(rng.integers(-50_000, 900_000, n)).astype(dtype)
    k[:1234] = np.iinfo(dtype).min                                   # NA keys lead an ascending, NA-first order
    v = rng.standard_normal(n); v[rng.random(n) < 0.03] = np.nan
    w = rng.integers(-1000, 1000, n).astype(np.int32)
    f4 = rng.standard_normal(n).astype(np.float32)
    alist = [(opn, vi) for vi in range(3) for opn in ("sum", "mean", "min", "max", "count")] + [("count0", None)]
    vals = (v, w, f4)

    def run(keys, na_last=False, desc=False):
        ri, off = o.group([keys], desc=[desc], na_last=na_last)
        r = ctx.groupby_agg([keys], list(vals), alist, desc=[desc], na_last=na_last)
        st = ctx.last_call_stats()
        assert_same(r.offsets(), off, "offsets")
        assert_same(r.key(0), keys[ri[off[:-1]]], "group keys")
        for a, (opn, vi) in enumerate(alist[:-1]):
            check_agg(r.agg(a), o.reduce(opn, vals[vi], ri, off), opn, vals[vi], ri, off, "%s(v%d)" % (opn, vi))
        assert_same(r.agg(len(alist) - 1), np.diff(off).astype(np.int64), "count()")
        r.free()
        return st

    st = run(k)
    assert st["path"] == "presorted" and st["retries_key_range"] == 0, st
    k2 = k.copy()
    k2[3_333_333], k2[3_333_334] = k2[3_333_334] + 7, k2[3_333_333]   # one descent between two sampled strata
    st = run(k2)
    assert st["path"] != "presorted", st
    assert run(k, na_last=True)["path"] != "presorted"
    assert run(k, desc=True)["path"] != "presorted"
    assert run(np.ascontiguousarray(k[::-1]))["path"] != "presorted"
