"""The MSD levels of the sort path (round 4: two stable scatter levels + every final bucket ordered in LDS, plan.hip
sort_stage / radix.hip) against the oracle, bit for bit, on inputs small enough for the oracle -- the levels are forced on
with `sort_path` = 2, `msd_min_rows` = 1 and small `msd_bucket_rows` so that a few hundred thousand rows already make
hundreds of buckets per level, ragged tiles, empty final buckets and buckets of one row.  The same cases run on the LSD
passes (`sort_path` = 1): both paths must return the reference's stable permutation (sort.cc:1206-1353 contract: stable,
NA first)."""
import numpy as np
import pytest

from conftest import assert_same
from oracle import oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture()
def msd(ctx):
    ctx.set_option("sort_path", 2)
    ctx.set_option("msd_min_rows", 1)
    yield ctx
    ctx.set_option("sort_path", 0)
    ctx.set_option("msd_min_rows", 1 << 26)
    ctx.set_option("msd_bucket_rows", 2048)


def _ran_msd(ctx, fn):
    ctx.profile_reset(); ctx.profile(True)
    try:
        r = fn()
    finally:
        ctx.profile(False)
    return r, ctx.profile_get("msd_scan_kernel")[1]


def _check(ctx, keys, cols, expect_msd=True, **kw):
    ri, off = o.group(keys, **kw)
    for path in (2, 1):
        ctx.set_option("sort_path", path)
        g, launches = _ran_msd(ctx, lambda: ctx.groupby(keys, **kw))
        if path == 2 and expect_msd is not None:
            assert (launches > 0) == expect_msd, "MSD levels %s" % ("did not run" if expect_msd else "ran unexpectedly")
        if path == 1:
            assert launches == 0
        assert_same(g.offsets(), off, "offsets [sort_path=%d]" % path)
        assert_same(g.rowindex(), ri, "rowindex [sort_path=%d]" % path)
        g.free()
        if cols:
            for want_ri in (True, False):
                r = ctx.groupby_rows(keys, cols, want_rowindex=want_ri, **{k: v for k, v in kw.items() if k in ("desc", "na_last")})
                assert_same(r.offsets(), off, "rows offsets [sort_path=%d]" % path)
                if want_ri:
                    assert_same(r.rowindex(), ri, "rows rowindex [sort_path=%d]" % path)
                for c, col in enumerate(cols):
                    assert_same(r.col(c), col[ri], "column %d in grouped order [sort_path=%d]" % (c, path))
                r.free()
    ctx.set_option("sort_path", 2)


@pytest.mark.parametrize("n,hi,bucket_rows", [(300_000, 60_000, 2048), (300_000, 100_000, 256), (1_000_000, 2_000_000, 512),
                                              (3_000_000, 30_000_000, 1024), (2_500_001, 4_000_000, 64), (700_000, 130_000, 4096)])
def test_msd_single_key_vs_oracle(msd, n, hi, bucket_rows):
    rng = np.random.default_rng(n + hi)
    msd.set_option("msd_bucket_rows", bucket_rows)
    k = rng.integers(-hi // 7, hi, n).astype(np.int64)
    k[rng.random(n) < 0.01] = -2**63
    x = rng.standard_normal(n)
    w = rng.integers(-2**31 + 1, 2**31 - 1, n).astype(np.int32)
    _check(msd, [k], [k, x, w], expect_msd=None)


def test_msd_levels_really_run(msd):
    rng = np.random.default_rng(5)
    n = 1_000_000
    msd.set_option("msd_bucket_rows", 512)          # S = 11 bits of scatter, 20-bit keys: 9 bits left for the final level
    k = rng.integers(0, 2**20 - 5, n).astype(np.int64)
    _check(msd, [k], [rng.standard_normal(n)], expect_msd=True)
    _check(msd, [k], [], expect_msd=True, na_last=True)
    _check(msd, [k.astype(np.int32)], [rng.standard_normal(n).astype(np.float32)], expect_msd=True, desc=[True])


def test_msd_two_keys_packed(msd):
    rng = np.random.default_rng(6)
    n = 1_200_000
    msd.set_option("msd_bucket_rows", 1024)
    a = rng.integers(0, 600, n).astype(np.int32)
    b = rng.integers(-9, 500, n).astype(np.int16)
    b[rng.random(n) < 0.02] = -2**15
    a[rng.random(n) < 0.02] = -2**31
    x = rng.standard_normal(n)
    _check(msd, [a, b], [x, a], expect_msd=None)
    _check(msd, [a, b], [x], expect_msd=None, desc=[False, True], na_last=True)


def test_msd_falls_back_when_a_final_bucket_overflows(msd):
    """heavy duplicates: few distinct keys over a wide range put > 8192 rows into one final bucket -- the levels give up
    after their histograms and the LSD passes start over from the untouched keys"""
    rng = np.random.default_rng(7)
    n = 1_500_000
    msd.set_option("msd_bucket_rows", 512)
    x = rng.standard_normal(n)
    # the two scatter digits CORRELATED (level-1 digit == level-2 digit): both marginal histograms are flat, so the cheap
    # predictor lets the levels start, and the level-2 histogram then finds 64 cells of n / 64 rows: the real fallback
    d = rng.integers(0, 63, n).astype(np.int64)
    k = (d << 15) | (d << 9) | rng.integers(0, 512, n).astype(np.int64)
    k[0], k[1] = 0, 2**21 - 2                      # pins the key range: 21 bits = 6 + 6 + 9
    _check(msd, [k], [x], expect_msd=True)         # the histogram kernels ran; the result is the LSD path's
    # few distinct keys over a wide range: 37 of them may still hide behind 27 x 29 non-empty marginal bins (the levels
    # start and give up after the level-2 histogram); 5 of them cannot (<= 25 cells for 1.5e6 rows: LSD at once)
    vals = rng.integers(0, 2**20, 37).astype(np.int64)
    _check(msd, [vals[rng.integers(0, len(vals), n)]], [x], expect_msd=None)
    _check(msd, [vals[rng.integers(0, 5, n)]], [x], expect_msd=False)
    # clustered: sorted keys with long runs
    k2 = np.sort(rng.integers(0, 2**20, n).astype(np.int64))
    _check(msd, [k2], [x], expect_msd=None)
    # skew: one hot key among uniform ones
    k3 = rng.integers(0, 2**20, n).astype(np.int64)
    k3[rng.random(n) < 0.3] = 777_777
    _check(msd, [k3], [x], expect_msd=None)


def test_msd_not_taken_when_it_does_not_apply(msd):
    rng = np.random.default_rng(8)
    n = 400_000
    x = rng.standard_normal(n)
    for k in (rng.integers(0, 100, n).astype(np.int64),            # fewer key bits than scatter bits
              rng.integers(-2**40, 2**40, n).astype(np.int64),     # 64-bit packed keys
              rng.standard_normal(n)):                             # float64 keys
        _check(msd, [k], [x], expect_msd=False)
