"""dthip_filter_groupby_rows -- `V = DT[f.x <cmp> c, :]; V[:, cols, by(key)]` (BASELINE config 5's two statements) in one
call -- against the oracle's filter -> gather -> group (rowindex_array.cc:130-170, :258-269, sort.cc:1411-1495), bit for
bit: offsets, every requested column in grouped order and the COMPOSED RowIndex.  The fused route (csrc/tlsort.hip: filter +
key transform + first sort level in one sweep with a tile-local layout, the second level gathering the segments) is forced
onto small inputs with sort_path = 2 / msd_min_rows = 1; the same cases run on the two-call route (filter_rows_fused = 0)."""
import numpy as np
import pytest

from conftest import assert_same
from oracle import oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fused(ctx):
    ctx.set_option("sort_path", 2); ctx.set_option("msd_min_rows", 1); ctx.set_option("msd_bucket_rows", 2048)
    ctx.set_option("filter_rows_fused", 1)
    yield ctx
    ctx.set_option("sort_path", 0); ctx.set_option("msd_min_rows", 1 << 26); ctx.set_option("msd_bucket_rows", 2048)
    ctx.set_option("filter_rows_fused", 1); ctx.set_option("spec_min_rows", 1 << 23)


def expected(x, cmp, scalar, keys, cols, **kw):
    if scalar is None:                                 # f.x == None: the NA rows; f.x != None: the valid ones
        na = np.isnan(x) if x.dtype.kind == "f" else (x == np.iinfo(x.dtype).min)
        fri = np.flatnonzero(na if cmp == "==" else ~na).astype(np.int32)
    else:
        fri = o.filter_cmp(x, cmp, scalar)
    if len(fri) == 0:
        return fri, np.zeros(1, np.int32), [c[:0] for c in cols]
    p, off = o.group([k[fri] for k in keys], **kw)
    comp = fri[p]                                      # RowIndex composition ab*bc (rowindex_array.cc:258-269)
    return comp, off, [c[comp] for c in cols]


def run(ctx, x, cmp, scalar, keys, cols, expect_fused, want_rowindex=True, expect_tl2=None, stats=None, **kw):
    """the query on three routes -- fused with a tile-local second level (tl_level2 = 2), fused with a scattering second
    level (0), the two calls (filter_rows_fused = 0) -- against the oracle.  expect_fused: the scatter form must run to the
    end (True) / must not (False) / may (None); expect_tl2: the same for the tile-local form"""
    comp, off, ecols = expected(x, cmp, scalar, keys, cols, **kw)
    ekw = {k: v for k, v in kw.items() if k in ("desc", "na_last")}
    for route, tl2, want in (("fused, tile-local level 2", 2, expect_tl2), ("fused, scatter level 2", 0, expect_fused), ("two calls", None, False)):
        ctx.set_option("filter_rows_fused", 0 if tl2 is None else 1)
        ctx.set_option("tl_level2", tl2 or 0)
        ctx.profile_reset(); ctx.profile(True)
        try:
            r = ctx.filter_groupby_rows(x, cmp, scalar, keys, cols, want_rowindex=want_rowindex, **ekw)
        finally:
            ctx.profile(False)
        # the fused route ran TO THE END: its first level and the final level, and no fall-back to the filter's take kernel
        ran = ctx.profile_get("tl_level1_kernel")[1] > 0 and ctx.profile_get("msd_final_kernel")[1] > 0 and \
            ctx.profile_get("compact_take_kernel")[1] == 0
        tag = " [%s]" % route
        if stats is not None and ran:
            stats[route] = stats.get(route, 0) + 1
        if want is not None:
            assert ran == want, "the fused route %s%s" % ("did not run" if want else "ran unexpectedly", tag)
        if tl2 == 2 and ran:
            assert ctx.profile_get("tl_gather_hist_kernel")[1] == 0 and ctx.profile_get("tl_final_starts_kernel")[1] > 0, tag
        assert r.nrows == len(comp), "rows" + tag
        assert_same(r.offsets(), off, "offsets" + tag)
        if want_rowindex:
            assert_same(r.rowindex(), comp, "composed rowindex" + tag)
        for c in range(len(cols)):
            got = r.col(c)
            exp = ecols[c].view(np.int8) if ecols[c].dtype == np.bool_ else ecols[c]
            if exp.dtype.kind == "f":
                assert_same(got.view("u%d" % got.itemsize), exp.view("u%d" % exp.itemsize), "column %d%s" % (c, tag))
            else:
                assert_same(got, exp, "column %d%s" % (c, tag))
        r.free()
    ctx.set_option("filter_rows_fused", 1); ctx.set_option("tl_level2", 1)


# (the levels need  significant key bits - scatter bits <= 9  with scatter bits ~ log2(passing rows / msd_bucket_rows): the
# shapes below are chosen so that the fused route applies; wider keys take the two-call route, tested further down)
@pytest.mark.parametrize("n,hi,bucket_rows,tl2", [(1, 5, 64, None), (100, 50, 64, None), (5000, 3000, 64, None), (8191, 20_000, 64, None),
                                                  (8192, 20_000, 64, None), (8193, 20_000, 64, None), (100_003, 60_000, 512, None),
                                                  (1_000_003, 2_000_000, 64, None), (3_000_000, 10_000_000, 64, None),
                                                  (2_500_001, 300_000, 64, None), (3_000_000, 500_000, 2048, True),
                                                  (2_000_000, 100_000, 1024, True), (100_003, 30_000, 1024, True),
                                                  (1_000_003, 100_000, 2048, True), (600_000, 60_000, 2048, True),
                                                  (5_000_000, 1_000_000, 2048, True)])
def test_config5_shape(fused, n, hi, bucket_rows, tl2):
    """int64 key, float64 predicate column riding along, the key column rebuilt by the final level, the composed RowIndex"""
    rng = np.random.default_rng(n)
    fused.set_option("msd_bucket_rows", bucket_rows)
    k = rng.integers(0, hi, n).astype(np.int64)
    x = rng.standard_normal(n)
    if n > 1000:
        x[rng.random(n) < 0.01] = np.nan
        k[rng.integers(0, n, 7)] = -2**63           # (a FEW NA keys: thousands of them are one big group, see the test below)
    run(fused, x, ">", 0.0, [k], [k, x], expect_fused=None if n < 5000 else True, expect_tl2=tl2)


@pytest.mark.parametrize("cmp,scalar", [(">", 0.25), (">=", -0.5), ("<", 0.0), ("<=", 1.0), ("==", 0.0), ("!=", 0.0), ("==", None), ("!=", None)])
def test_predicates_float64_and_int64(fused, cmp, scalar):
    rng = np.random.default_rng(5)
    n = 400_000
    fused.set_option("msd_bucket_rows", 256)
    k = rng.integers(-2000, 2000, n).astype(np.int32)
    x = np.round(rng.standard_normal(n), 1); x[rng.random(n) < 0.05] = np.nan
    xi = rng.integers(-3, 4, n).astype(np.int64); xi[rng.random(n) < 0.05] = -2**63
    w = rng.integers(-10**6, 10**6, n).astype(np.int32)
    run(fused, x, cmp, scalar, [k], [x, k], expect_fused=True)
    run(fused, xi, cmp, None if scalar is None else int(scalar), [k], [w], expect_fused=True, want_rowindex=False)


@pytest.mark.parametrize("bucket_rows", [256, 2048])           # 2048: the final buckets fill windows -> tile-local level 2 too
def test_column_sets_and_orders(fused, bucket_rows):
    rng = np.random.default_rng(6)
    n = 600_000
    fused.set_option("msd_bucket_rows", bucket_rows)
    T = True if bucket_rows == 2048 else None
    k = rng.integers(0, 100_000, n).astype(np.int64); k[rng.integers(0, n, 40)] = -2**63
    x = rng.standard_normal(n)
    y = rng.standard_normal(n)
    w = rng.integers(-99, 99, n).astype(np.int32)
    f32 = rng.standard_normal(n).astype(np.float32)
    for cols, ri, fusedp in (([k, x], True, True), ([x], True, True), ([], True, True), ([k], True, True), ([x, k, x, k], True, True),
                             ([w], True, False), ([w], False, True), ([w, x], False, True), ([x, y], False, True), ([f32, x], False, True),
                             ([x, y], True, False), ([w, f32], False, False), ([x, y, w], False, False), ([k], False, False)):
        run(fused, x, ">", 0.0, [k], cols, expect_fused=fusedp, want_rowindex=ri, expect_tl2=(T if fusedp else False))
    run(fused, x, "<", 0.3, [k], [x, k], expect_fused=True, expect_tl2=T, na_last=True)
    run(fused, x, "<", 0.3, [k], [x, k], expect_fused=True, expect_tl2=T, desc=[True])
    run(fused, y, ">", 0.0, [k], [x], expect_fused=True, expect_tl2=T)           # the predicate column does not ride


def test_queries_outside_the_fused_route(fused):
    rng = np.random.default_rng(8)
    n = 300_000
    a = rng.integers(0, 300, n).astype(np.int32); b = rng.integers(0, 40, n).astype(np.int32)
    x = rng.standard_normal(n)
    kf = np.round(rng.standard_normal(n), 2)
    x32 = x.astype(np.float32)
    wide = rng.integers(-2**50, 2**50, n).astype(np.int64)
    run(fused, x, ">", 0.0, [a, b], [x, a], expect_fused=False)                 # two keys
    run(fused, x, ">", 0.0, [kf], [x], expect_fused=False)                       # float key
    run(fused, x32, ">", 0.0, [a], [a], expect_fused=False)                      # 4-byte predicate column
    run(fused, x, ">", 0.0, [wide], [x], expect_fused=False)                     # key wider than 32 bits
    fused.set_option("sort_path", 1)
    run(fused, x, ">", 0.0, [a], [x], expect_fused=False)                        # LSD passes only
    fused.set_option("sort_path", 2)
    run(fused, x, ">", 5.0e9, [a], [x, a], expect_fused=None)                    # nothing passes


@pytest.mark.parametrize("bucket_rows", [64, 2048])
def test_clustered_sorted_and_duplicate_keys(fused, bucket_rows):
    """long segments (a tile's rows all in one or two level-1 buckets: the whole-wave branch of the segment loader), a
    constant top digit, and keys with so many duplicates that a final bucket outgrows a tile (the call falls back)"""
    rng = np.random.default_rng(9)
    n = 2_000_000
    fused.set_option("msd_bucket_rows", bucket_rows)
    T = True if bucket_rows == 2048 else None
    x = rng.standard_normal(n)
    ks = np.sort(rng.integers(0, 5_000_000 if bucket_rows == 64 else 250_000, n)).astype(np.int64)
    run(fused, x, ">", 0.0, [ks], [ks, x], expect_fused=True, expect_tl2=T)
    run(fused, x, ">", 0.0, [ks[::-1].copy()], [x], expect_fused=True, expect_tl2=T)
    kc = (np.arange(n) // 50_000 * 1000 + rng.integers(0, 1000, n)).astype(np.int64)      # clustered in blocks of 50000 rows
    run(fused, x, ">", -0.5, [kc], [kc, x], expect_fused=True, expect_tl2=T)
    klow = rng.integers(0, 2000, n).astype(np.int64) + 10_000_000; klow[0] = 0          # top digits constant but for one row
    run(fused, x, ">", 0.0, [klow], [klow, x], expect_fused=None)
    kdup = (rng.integers(0, 12, n) * 100_000_000 // 12).astype(np.int64)                  # 12 distinct keys over a wide range
    run(fused, x, ">", 0.0, [kdup], [kdup, x], expect_fused=None)
    kone = np.full(n, 7, np.int64)
    run(fused, x, ">", 0.0, [kone], [kone, x], expect_fused=None)


def test_a_big_na_group_falls_back(fused):
    """1 % NA keys are ONE group of thousands of rows: the final bucket holding it outgrows a tile, the fused route gives up
    after its levels' plan and the call takes the two-call route -- same results"""
    rng = np.random.default_rng(12)
    n = 2_000_000
    fused.set_option("msd_bucket_rows", 64)
    k = rng.integers(0, 300_000, n).astype(np.int64); k[rng.random(n) < 0.02] = -2**63
    x = rng.standard_normal(n)
    run(fused, x, ">", 0.0, [k], [k, x], expect_fused=False)
    run(fused, x, ">", 0.0, [k], [k, x], expect_fused=False, na_last=True)


def test_guessed_key_range_violated_by_a_passing_row(fused):
    """the key range is guessed from a sample at spec_min_rows and above; a PASSING row outside it makes level 1 raise its
    flag and the call plans again with the exact range; a failing row outside it does not matter"""
    from test_gpu_parity import _unsampled_rows
    rng = np.random.default_rng(10)
    n = 3_000_000
    fused.set_option("spec_min_rows", 1); fused.set_option("msd_bucket_rows", 64)
    k = rng.integers(1000, 900_000, n).astype(np.int64); k[0], k[-1] = 1000, 899_999
    x = rng.standard_normal(n)
    r1, r2 = _unsampled_rows(n, 8, 2)
    k2 = k.copy(); k2[r1] = 1_500_000; x2 = x.copy(); x2[r1] = 1.0          # passes: retry
    run(fused, x2, ">", 0.0, [k2], [k2, x2], expect_fused=True)
    k3 = k.copy(); k3[r2] = -77; x3 = x.copy(); x3[r2] = -1.0                 # fails the predicate: no retry needed
    run(fused, x3, ">", 0.0, [k3], [k3, x3], expect_fused=True)


def test_device_resident_columns(fused):
    rng = np.random.default_rng(11)
    n = 1_500_000
    fused.set_option("msd_bucket_rows", 64)
    k = rng.integers(0, 700_000, n).astype(np.int64)
    x = rng.standard_normal(n)
    comp, off, (ek, ex) = expected(x, ">", 0.0, [k], [k, x])
    dk, dx = fused.upload(k), fused.upload(x)
    r = fused.filter_groupby_rows(dx, ">", 0.0, [dk], [dk, dx], nrows=n)
    assert_same(r.offsets(), off, "offsets"); assert_same(r.rowindex(), comp, "composed rowindex")
    assert_same(r.col(0), ek, "key column"); assert_same(r.col(1), ex, "x column")
    r.free()
