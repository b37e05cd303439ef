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


def run(ctx, x, cmp, scalar, keys, cols, expect_fused, want_rowindex=True, **kw):
    comp, off, ecols = expected(x, cmp, scalar, keys, cols, **kw)
    ekw = {k: v for k, v in kw.items() if k in ("desc", "na_last")}
    for route in (1, 0):
        ctx.set_option("filter_rows_fused", route)
        ctx.profile_reset(); ctx.profile(True)
        try:
            r = ctx.filter_groupby_rows(x, cmp, scalar, keys, cols, want_rowindex=want_rowindex, **ekw)
        finally:
            ctx.profile(False)
        ran = ctx.profile_get("tl_level1_kernel")[1] > 0 and ctx.profile_get("tl_level2_kernel")[1] > 0
        tag = " [%s route]" % ("fused" if route else "two-call")
        if route == 1 and expect_fused is not None:
            assert ran == expect_fused, "the fused route %s" % ("did not run" if expect_fused else "ran unexpectedly")
        if route == 0:
            assert not ran
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
    ctx.set_option("filter_rows_fused", 1)


# (the levels need  significant key bits - scatter bits <= 9  with scatter bits ~ log2(passing rows / msd_bucket_rows): the
# shapes below are chosen so that the fused route applies; wider keys take the two-call route, tested further down)
@pytest.mark.parametrize("n,hi,bucket_rows", [(1, 5, 64), (100, 50, 64), (5000, 3000, 64), (8191, 20_000, 64), (8192, 20_000, 64),
                                              (8193, 20_000, 64), (100_003, 60_000, 512), (1_000_003, 2_000_000, 64),
                                              (3_000_000, 10_000_000, 64), (2_500_001, 300_000, 64), (3_000_000, 500_000, 2048),
                                              (2_000_000, 100_000, 1024)])
def test_config5_shape(fused, n, hi, bucket_rows):
    """int64 key, float64 predicate column riding along, the key column rebuilt by the final level, the composed RowIndex"""
    rng = np.random.default_rng(n)
    fused.set_option("msd_bucket_rows", bucket_rows)
    k = rng.integers(0, hi, n).astype(np.int64)
    x = rng.standard_normal(n)
    if n > 1000:
        x[rng.random(n) < 0.01] = np.nan
        k[rng.random(n) < 0.01] = -2**63
    run(fused, x, ">", 0.0, [k], [k, x], expect_fused=None if n < 5000 else True)


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
    run(fused, xi, cmp, None if scalar is None else int(scalar), [k], [w], expect_fused=True)


def test_column_sets_and_orders(fused):
    rng = np.random.default_rng(6)
    n = 600_000
    fused.set_option("msd_bucket_rows", 256)
    k = rng.integers(0, 200_000, n).astype(np.int64); k[rng.random(n) < 0.02] = -2**63
    x = rng.standard_normal(n)
    y = rng.standard_normal(n)
    w = rng.integers(-99, 99, n).astype(np.int32)
    f32 = rng.standard_normal(n).astype(np.float32)
    for cols, ri, fusedp in (([k, x], True, True), ([x], True, True), ([], True, True), ([k], True, True), ([x, k, x, k], True, True),
                             ([w], True, False), ([w], False, True), ([w, x], False, True), ([x, y], False, True), ([f32, x], False, True),
                             ([x, y], True, False), ([w, f32], False, False), ([x, y, w], False, False), ([k], False, False)):
        run(fused, x, ">", 0.0, [k], cols, expect_fused=fusedp, want_rowindex=ri)
    run(fused, x, "<", 0.3, [k], [x, k], expect_fused=True, na_last=True)
    run(fused, x, "<", 0.3, [k], [x, k], expect_fused=True, desc=[True])
    run(fused, y, ">", 0.0, [k], [x], expect_fused=True)                         # the predicate column does not ride


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


def test_clustered_sorted_and_duplicate_keys(fused):
    """long segments (a tile's rows all in one or two level-1 buckets: the whole-wave branch of the segment loader), a
    constant top digit, and keys with so many duplicates that a final bucket outgrows a tile (the call falls back)"""
    rng = np.random.default_rng(9)
    n = 2_000_000
    fused.set_option("msd_bucket_rows", 64)
    x = rng.standard_normal(n)
    ks = np.sort(rng.integers(0, 5_000_000, n)).astype(np.int64)
    run(fused, x, ">", 0.0, [ks], [ks, x], expect_fused=True)
    run(fused, x, ">", 0.0, [ks[::-1].copy()], [x], expect_fused=True)
    kc = (np.arange(n) // 50_000 * 1000 + rng.integers(0, 1000, n)).astype(np.int64)      # clustered in blocks of 50000 rows
    run(fused, x, ">", -0.5, [kc], [kc, x], expect_fused=True)
    klow = rng.integers(0, 2000, n).astype(np.int64) + 10_000_000; klow[0] = 0          # top digits constant but for one row
    run(fused, x, ">", 0.0, [klow], [klow, x], expect_fused=None)
    kdup = (rng.integers(0, 12, n) * 100_000_000 // 12).astype(np.int64)                  # 12 distinct keys over a wide range
    run(fused, x, ">", 0.0, [kdup], [kdup, x], expect_fused=None)
    kone = np.full(n, 7, np.int64)
    run(fused, x, ">", 0.0, [kone], [kone, x], expect_fused=None)


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
