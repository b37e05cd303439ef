"""The tile-local layout of the bucketed aggregation (no histogram pass: bucket.hip `bucket_partition_kernel` with a
directory, `dir_transpose / dir_totals / seg_plan`, `table_agg_seg_kernel`), forced on inputs of every size with
option bucket_variant = 3 (by default it is chosen from 2^22 rows on when >= 1024 buckets are evenly filled), against
the oracle: keys, offsets, counts, min/max, integer sums bit-exact, float sums / means <= 1e-6."""
import numpy as np
import pytest

from test_gpu_parity import _vs_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture()
def tl(ctx):
    ctx.set_option("bucket_variant", 3)
    yield ctx
    ctx.set_option("bucket_variant", 0)


@pytest.mark.parametrize("n", [4096, 12287, 12288, 12289, 24576, 100_003, 1_000_000])
def test_sizes_around_the_partition_tile(tl, n):
    rng = np.random.default_rng(n)
    k = rng.integers(0, 3_000_000, n).astype(np.int64)           # 22 significant bits -> 256+ buckets
    k[rng.random(n) < 0.01] = np.iinfo(np.int64).min
    v = rng.standard_normal(n)
    v[rng.random(n) < 0.05] = np.nan
    w = rng.integers(-1000, 1000, n).astype(np.int32)
    _vs_oracle(tl, [k], [v, w], check_ri=False)


def test_every_reducer_two_keys_and_no_values(tl):
    rng = np.random.default_rng(5)
    n = 500_000
    a = rng.integers(0, 3163, n).astype(np.int32)
    b = rng.integers(0, 3163, n).astype(np.int32)
    b[rng.random(n) < 0.01] = -2**31
    v = rng.standard_normal(n)
    i8 = rng.integers(-10**15, 10**15, n).astype(np.int64)
    f4 = rng.standard_normal(n).astype(np.float32)
    _vs_oracle(tl, [a, b], [v, i8, f4], aggs=("sum", "mean", "min", "max", "count"), check_ri=False)
    r = tl.groupby_agg([a, b], [], [("count0", None)])
    from oracle import oracle as o
    ri, off = o.group([a, b])
    assert np.array_equal(r.offsets(), off) and np.array_equal(r.agg(0), np.diff(off))
    r.free()


def test_hot_bucket_long_segments_and_split_parts(tl):
    """one key holds a third of the rows: its bucket is split into tile-range parts (global atomics merge) and streamed
    in long-segment mode; the other buckets take the 16-lanes-per-segment path"""
    rng = np.random.default_rng(6)
    n = 3_000_000
    k = rng.integers(0, 2_000_000, n).astype(np.int64)
    k[rng.random(n) < 0.33] = 777_777
    v = rng.standard_normal(n)
    _vs_oracle(tl, [k], [v], aggs=("sum", "count", "min"), check_ri=False)


def test_guessed_key_range_verified_by_the_partition_pass(tl):
    """with no histogram pass in front of it, the partition kernel itself reports keys outside a sampled range"""
    rng = np.random.default_rng(77)
    n = 3_000_000
    k = rng.integers(1000, 3_000_000, n).astype(np.int64)
    v = rng.standard_normal(n)
    tl.set_option("spec_min_rows", 1)
    try:
        _vs_oracle(tl, [k], [v], aggs=("sum",), check_ri=False)
        k[1_234_567] = 4_100_000          # rows the sample does not visit: the retry uses the exact range
        k[2_000_001] = -90_000
        _vs_oracle(tl, [k], [v], aggs=("sum",), check_ri=False)
    finally:
        tl.set_option("spec_min_rows", 1 << 23)


def test_default_choice_at_6e6_rows(ctx):
    """from 2^22 rows on the layout is picked by the sampled bucket histogram: even keys take it; round 6: so do even keys
    with a FEW hot buckets among them (one hot key, a hot key + the NA group: their parts form seg_plan_kernel's list A,
    dealt over all XCDs); a smoothly skewed key keeps the exact-position layout; all must agree with the oracle"""
    rng = np.random.default_rng(8)
    n = 6_000_000
    k = rng.integers(0, 10_000_000, n).astype(np.int64)
    v = rng.standard_normal(n)
    w = rng.integers(-50, 50, n).astype(np.int32)
    _vs_oracle(ctx, [k], [v], aggs=("sum",), check_ri=False)
    k[rng.random(n) < 0.2] = 5
    _vs_oracle(ctx, [k], [v], aggs=("sum",), check_ri=False)
    assert ctx.last_call_stats()["path"] == "bucketed"
    k[rng.random(n) < 0.1] = np.iinfo(np.int64).min                  # + a heavy NA group
    v[rng.random(n) < 0.01] = np.nan
    _vs_oracle(ctx, [k], [v, w], aggs=("sum", "mean", "min", "max", "count"), check_ri=False)
    ks = (rng.random(n) ** 6 * 1e7).astype(np.int64)                  # smooth skew: hundreds of uneven buckets
    _vs_oracle(ctx, [ks], [v, w], aggs=("sum", "count"), check_ri=False)


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
@pytest.mark.parametrize("na_last,desc", [(False, False), (True, False), (False, True), (True, True)])
def test_outlier_keys_are_listed_not_retried(tl, dtype, na_last, desc):
    """round 6: rows whose key lies outside the GUESSED range go to a list instead of making the query start over; their
    groups are spliced in front of / behind the range's groups (agg.hip splice_outlier_groups).  Outliers on both sides,
    duplicates among them, NA keys present (first / last), ascending / descending, every reducer + count(): against the
    oracle, and dthip_last_call_stats must show NO repeated sweep"""
    from conftest import assert_same
    from test_gpu_parity import check_agg, _unsampled_rows
    from oracle import oracle as o
    rng = np.random.default_rng(91 + int(na_last) * 2 + int(desc))
    n = 3_000_000
    k = rng.integers(1000, 3_000_000, n).astype(dtype)
    k[0], k[-1] = 1000, 2_999_999                                  # (first / last rows are always sampled)
    k[rng.random(n) < 0.001] = np.iinfo(dtype).min                 # NA keys
    k[0], k[-1] = 1000, 2_999_999
    rows = _unsampled_rows(n, np.dtype(dtype).itemsize, 9)
    k[rows[:3]] = 5_000_000                                        # above: one group of three rows
    k[rows[3]] = 5_000_001
    k[rows[4:7]] = [-70_000, -70_000, 12]                          # below: two groups
    k[rows[7]] = np.iinfo(dtype).max                               # the far ends of the type
    k[rows[8]] = np.iinfo(dtype).min + 1
    v = rng.standard_normal(n); v[rng.random(n) < 0.02] = np.nan
    w = rng.integers(-1000, 1000, n).astype(np.int32)
    alist = [(opn, vi) for vi in range(2) for opn in ("sum", "mean", "min", "max", "count")] + [("count0", None)]
    ri, off = o.group([k], desc=[desc], na_last=na_last)
    tl.set_option("spec_min_rows", 1)
    try:
        r = tl.groupby_agg([k], [v, w], alist, desc=[desc], na_last=na_last)
        st = tl.last_call_stats()
    finally:
        tl.set_option("spec_min_rows", 1 << 23)
    assert st["path"] == "bucketed" and st["retries_key_range"] == 0 and st["outlier_rows_listed"] == 8, st      # (key 12 lies inside the 1/64 margin of the guess)
    assert_same(r.offsets(), off, "offsets")
    assert_same(r.key(0), k[ri[off[:-1]]], "group keys")
    for a, (opn, vi) in enumerate(alist[:-1]):
        check_agg(r.agg(a), o.reduce(opn, (v, w)[vi], ri, off), opn, (v, w)[vi], ri, off, "%s(v%d)" % (opn, vi))
    assert_same(r.agg(len(alist) - 1), np.diff(off).astype(np.int64), "count()")
    r.free()


def test_more_outliers_than_the_list_holds_still_retry(tl):
    """a guess that is simply wrong (a tenth of the rows outside it) overflows the 65536-entry list: the query runs again with
    the exact range, as before"""
    from test_gpu_parity import _unsampled_rows
    rng = np.random.default_rng(93)
    n = 3_000_000
    k = rng.integers(1000, 2_000_000, n).astype(np.int64)
    rows = _unsampled_rows(n, 8, 300_000)
    k[rows] = rng.integers(2_500_000, 3_000_000, len(rows))
    v = rng.standard_normal(n)
    tl.set_option("spec_min_rows", 1)
    try:
        _vs_oracle(tl, [k], [v], aggs=("sum", "count"), check_ri=False)
        r = tl.groupby_agg([k], [v], [("sum", 0)]); st = tl.last_call_stats(); r.free()
    finally:
        tl.set_option("spec_min_rows", 1 << 23)
    assert st["retries_key_range"] == 1, st
