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
    """from 2^22 rows on the layout is picked by the sampled bucket histogram: even keys take it, a hot key does not;
    both must agree with the oracle"""
    rng = np.random.default_rng(8)
    n = 6_000_000
    k = rng.integers(0, 10_000_000, n).astype(np.int64)
    v = rng.standard_normal(n)
    _vs_oracle(ctx, [k], [v], aggs=("sum",), check_ri=False)
    k[rng.random(n) < 0.2] = 5
    _vs_oracle(ctx, [k], [v], aggs=("sum",), check_ri=False)
