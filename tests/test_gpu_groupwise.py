"""GPU parity tests for the group-wise operators sharing the hot path's Groupby,
called through the C ABI: dthip_reduce(sd | median | nunique), dthip_reduce2(cov | corr),
dthip_cumulate(cumsum | cumprod | cummin | cummax | cumcount | ngroup, forward and reverse), against
  (1) tests/golden/groupwise_cases.npz = outputs of the unmodified reference, and
  (2) the CPU oracle (oracle/dt_oracle_groupwise.c) on seeded inputs it finishes in seconds.
Bit-exact: median, nunique, cummin/cummax, cumcount/ngroup, integer cumsum/cumprod (they wrap like the
reference's int64 arithmetic).  Floating point: sd/cov/corr and float cumsum/cumprod are re-associated
(Chan's pairwise merge instead of the sequential Welford loop; scan instead of a running sum):
1e-6 relative for float64, 5e-5 for float32 outputs, NA / inf patterns identical."""
import numpy as np
import pytest

from conftest import assert_same, groupwise_golden
from oracle import oracle as o

pytestmark = pytest.mark.gpu

GW = groupwise_golden()
EXACT = ("median", "nunique", "cummin", "cummax", "cumcount", "ngroup")


def close(got, want, what, scale=1.0):
    assert got.dtype == want.dtype, "%s dtype %s != %s" % (what, got.dtype, want.dtype)
    assert got.shape == want.shape, "%s shape" % what
    g, w = got.astype(np.float64), want.astype(np.float64)
    assert np.array_equal(np.isnan(g), np.isnan(w)), "%s: NA pattern differs" % what
    m = ~np.isnan(w)
    inf = np.isinf(w) & m
    assert np.array_equal(g[inf], w[inf]), "%s: infinities differ" % what
    m &= ~np.isinf(w)
    rel = 5e-5 if want.dtype == np.float32 else 1e-6
    err = np.abs(g[m] - w[m])
    tol = rel * np.abs(w[m]) + rel * 1e-3 * scale
    bad = np.nonzero(err > tol)[0]
    assert len(bad) == 0, "%s: %d values off, first %r vs %r" % (what, len(bad), g[m][bad[0]], w[m][bad[0]])


def vscale(*cols):
    s = 1.0
    for v in cols:
        f = np.abs(np.nan_to_num(v.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0))
        if v.dtype.kind in "iu":
            f = f[v != np.iinfo(v.dtype).min]
        s *= float(f.max()) if len(f) else 1.0
    return s


def gpu_out(ctx, c_stypes, key, ri, off, vals):
    parts = key.split(".")
    op, rev = parts[0], parts[-1] == "rev"
    if op in ("cumcount", "ngroup"):
        return ctx.cumulate(op, None, ri, off, reverse=rev)
    if op in ("cov", "corr"):
        i, j = int(parts[1]), int(parts[2])
        return ctx.reduce2(op, vals[i], vals[j], ri, off, stypes=(c_stypes[i], c_stypes[j]))
    vi = int(parts[1][1:])
    if op in ("sd", "median", "nunique"):
        return ctx.reduce(op, vals[vi], ri, off, stype=c_stypes[vi])
    return ctx.cumulate(op, vals[vi], ri, off, reverse=rev, stype=c_stypes[vi])


def compare(got, want, key, vals, what):
    parts = key.split(".")
    op = parts[0]
    if op in EXACT or want.dtype.kind in "iu":
        assert_same(got, want, what)
    elif op in ("cov", "corr"):
        i, j = int(parts[1]), int(parts[2])
        close(got, want, what, scale=vscale(vals[i], vals[j]) if op == "cov" else 1.0)
    else:
        vi = int(parts[1][1:])
        close(got, want, what, scale=vscale(vals[vi]) * (len(vals[vi]) if op == "cumsum" else 1.0))


@pytest.mark.parametrize("name", GW.names())
def test_golden_groupwise(ctx, name):
    c = GW.by_name[name]
    keys, vals = GW.keys(name), GW.vals(name)
    r = ctx.groupby(keys, stypes=c["key_stypes"])
    ri, off = r.rowindex(), r.offsets()
    r.free()
    assert_same(ri, GW.get(name, "ri"), "rowindex")
    assert_same(off, GW.get(name, "off"), "offsets")
    for key in c["outs"]:
        if key.startswith("cumprod") and GW.get(name, key).dtype.kind == "f":
            continue          # checked below with a product-aware tolerance
        compare(gpu_out(ctx, c["val_stypes"], key, ri, off, vals), GW.get(name, key), key, vals, "%s/%s" % (name, key))


@pytest.mark.parametrize("name", [n for n in GW.names() if any(k.startswith("cumprod") for k in GW.by_name[n]["outs"])])
def test_golden_float_cumprod(ctx, name):
    c = GW.by_name[name]
    vals = GW.vals(name)
    ri, off = GW.get(name, "ri"), GW.get(name, "off")
    for key in c["outs"]:
        want = GW.get(name, key)
        if not (key.startswith("cumprod") and want.dtype.kind == "f"):
            continue
        got = gpu_out(ctx, c["val_stypes"], key, ri, off, vals)
        assert got.dtype == want.dtype
        g, w = got.astype(np.float64), want.astype(np.float64)
        assert np.array_equal(np.isnan(g), np.isnan(w)), key
        # a long product re-associated: relative error grows with the number of factors; values next
        # to the overflow / underflow thresholds may land on either side, so they are left out
        big = np.abs(w) > (1e300 if want.dtype == np.float64 else 1e36)
        tiny = np.abs(w) < (1e-300 if want.dtype == np.float64 else 1e-36)
        m = ~np.isnan(w) & ~big & ~tiny & ~np.isinf(g)
        rel = 1e-4 if want.dtype == np.float32 else 1e-9
        assert np.all(np.abs(g[m] - w[m]) <= rel * np.abs(w[m])), key


def _inputs(seed, n, ng, vst, nafrac, shape="uniform"):
    rng = np.random.default_rng(seed)
    if shape == "uniform":
        k = rng.integers(0, ng, n).astype(np.int32)
    elif shape == "skew":
        k = np.where(rng.random(n) < 0.8, 3, rng.integers(0, ng, n)).astype(np.int32)
    else:   # "single"
        k = np.zeros(n, np.int32)
    if vst == 7:
        v = rng.standard_normal(n) * 50 + 10
        v[rng.random(n) < nafrac] = np.nan
    elif vst == 6:
        v = (rng.standard_normal(n) * 5).astype(np.float32)
        v[rng.random(n) < nafrac] = np.nan
    else:
        lim = {2: 100, 3: 1000, 4: 10**6, 5: 10**12}[vst]
        v = rng.integers(-lim, lim, n).astype({2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64}[vst])
        v[rng.random(n) < nafrac] = np.iinfo(v.dtype).min
    w = rng.standard_normal(n) * 3 + 0.01 * np.nan_to_num(v.astype(np.float64), nan=0.0)
    w[rng.random(n) < nafrac] = np.nan
    return k, v, w


@pytest.mark.parametrize("shape,n,ng", [("uniform", 300_000, 1000), ("uniform", 200_000, 150_000), ("skew", 250_000, 5000),
                                        ("single", 100_000, 1), ("uniform", 5000, 4999), ("uniform", 2049, 3)])
@pytest.mark.parametrize("vst", [7, 5, 4, 6, 2])
def test_groupwise_vs_oracle(ctx, shape, n, ng, vst):
    k, v, w = _inputs(100 + vst + n, n, ng, vst, 0.07, shape)
    r = ctx.groupby([k])
    ri, off = r.rowindex(), r.offsets()
    r.free()
    ori, ooff = o.group([k])
    assert_same(ri, ori, "rowindex")
    assert_same(off, ooff, "offsets")
    for op in ("sd", "median", "nunique"):
        got, want = ctx.reduce(op, v, ri, off), o.reducex(op, v, ri, off)
        if op == "sd":
            close(got, want, op, scale=vscale(v))
        else:
            assert_same(got, want, op)
            if vst in (6, 7):
                # float columns take the sorted-rows path by default; the distinct-pairs path must agree
                ctx.set_option("median_pairs", 1)
                try:
                    assert_same(ctx.reduce(op, v, ri, off), want, op + " (pairs)")
                finally:
                    ctx.set_option("median_pairs", 0)
    for op in ("cov", "corr"):
        close(ctx.reduce2(op, v, w, ri, off), o.reduce2(op, v, w, ri, off), op, scale=vscale(v, w) if op == "cov" else 1.0)
    for op in ("cumsum", "cummin", "cummax"):
        for rev in (False, True):
            got, want = ctx.cumulate(op, v, ri, off, reverse=rev), o.cumulate(op, v, ri, off, reverse=rev)
            if op == "cumsum" and want.dtype.kind == "f":
                # the reference's float32 running sum is itself only float32-accurate over a long group
                close(got, want, "%s rev=%s" % (op, rev), scale=vscale(v) * (n if vst == 7 else n * 100))
            else:
                assert_same(got, want, "%s rev=%s" % (op, rev))
    if vst in (2, 4):
        # integer products wrap identically however the scan associates them
        assert_same(ctx.cumulate("cumprod", v, ri, off), o.cumulate("cumprod", v, ri, off), "cumprod")
    for op in ("cumcount", "ngroup"):
        for rev in (False, True):
            assert_same(ctx.cumulate(op, None, ri, off, reverse=rev), o.cumulate(op, None, ri, off, reverse=rev), op)


def test_groupwise_identity_rowindex(ctx):
    """rowindex = NULL: the column is already in grouped order"""
    rng = np.random.default_rng(9)
    n = 70_000
    sizes = rng.integers(1, 40, 4000)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    off = off[off <= n]
    off[-1] = n
    v = rng.standard_normal(n)
    v[rng.random(n) < 0.1] = np.nan
    vi = rng.integers(-9, 9, n).astype(np.int64)
    close(ctx.reduce("sd", v, None, off), o.reducex("sd", v, None, off), "sd")
    assert_same(ctx.reduce("median", v, None, off), o.reducex("median", v, None, off), "median")
    assert_same(ctx.reduce("nunique", vi, None, off), o.reducex("nunique", vi, None, off), "nunique")
    assert_same(ctx.cumulate("cumsum", vi, None, off, reverse=True), o.cumulate("cumsum", vi, None, off, reverse=True), "cumsum")
    assert_same(ctx.cumulate("cummax", v, None, off), o.cumulate("cummax", v, None, off), "cummax")
    close(ctx.reduce2("corr", v, vi, None, off), o.reduce2("corr", v, vi, None, off), "corr")


def test_groupwise_na_rowindex_entries(ctx):
    """negative RowIndex entries read as NA values (column/view.cc:140-145)"""
    v = np.array([1.0, 2.0, 4.0, 8.0, 16.0])
    ri = np.array([0, -2**31, 2, 3, -2**31, 4], np.int32)
    off = np.array([0, 3, 6], np.int32)
    assert_same(ctx.cumulate("cumsum", v, ri, off), o.cumulate("cumsum", v, ri, off), "cumsum")
    assert_same(ctx.cumulate("cummin", v, ri, off, reverse=True), o.cumulate("cummin", v, ri, off, reverse=True), "cummin")
    assert_same(ctx.reduce("median", v, ri, off), o.reducex("median", v, ri, off), "median")
    assert_same(ctx.reduce("nunique", v, ri, off), o.reducex("nunique", v, ri, off), "nunique")
    close(ctx.reduce("sd", v, ri, off), o.reducex("sd", v, ri, off), "sd")


def test_groupwise_large(ctx):
    """2e7 rows, 1e5 groups and one group: every position is scanned exactly once whatever the group
    sizes; properties that do not need the oracle at this size"""
    rng = np.random.default_rng(77)
    n = 20_000_000
    k = rng.integers(0, 100_000, n).astype(np.int32)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    r = ctx.groupby([k])
    ri, off = r.rowindex(), r.offsets()
    r.free()
    cs = ctx.cumulate("cumsum", v, ri, off)
    vg = v[ri]
    total = np.cumsum(vg)
    base = np.repeat(np.concatenate([[0], total[off[1:-1] - 1]]), np.diff(off))
    assert np.array_equal(cs, total - base)
    ng = ctx.cumulate("ngroup", None, ri, off)
    assert np.array_equal(ng, np.repeat(np.arange(len(off) - 1), np.diff(off)))
    # sd / cov per group at a size where the tile-carry scan runs several chunks (> 4096 tiles)
    vf = vg.astype(np.float64)
    cnt = np.diff(off).astype(np.float64)
    s1 = np.add.reduceat(vf, off[:-1].astype(np.int64))
    s2 = np.add.reduceat(vf * vf, off[:-1].astype(np.int64))
    var = (s2 - s1 * s1 / cnt) / (cnt - 1)
    sdg = ctx.reduce("sd", v, ri, off)
    assert np.allclose(sdg, np.sqrt(var), rtol=1e-9)
    w = (np.arange(n) % 1000).astype(np.int64)
    wf = w[ri].astype(np.float64)
    sw = np.add.reduceat(wf, off[:-1].astype(np.int64))
    svw = np.add.reduceat(vf * wf, off[:-1].astype(np.int64))
    covg = ctx.reduce2("cov", v, w, ri, off)
    assert np.allclose(covg, (svw - s1 * sw / cnt) / (cnt - 1), rtol=1e-7, atol=1e-6)
    one = np.array([0, n], np.int32)
    cm = ctx.cumulate("cummax", v, None, one)
    assert np.array_equal(cm, np.maximum.accumulate(v))
    sd = ctx.reduce("sd", v, None, one)
    assert abs(sd[0] - np.std(v.astype(np.float64), ddof=1)) < 1e-6 * sd[0]
    nu = ctx.reduce("nunique", v, None, one)
    assert nu[0] == len(np.unique(v))
    med = ctx.reduce("median", v, None, one)
    assert med[0] == np.median(v)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_cov_corr_with_infinities(ctx, dt):
    """groups holding +-inf among their valid pairs: the reference's sequential update ends in +-inf or NaN
    depending on where the infinity sits; those groups are re-evaluated sequentially on the GPU"""
    rng = np.random.default_rng(41)
    n = 60_000
    k = rng.integers(0, 3000, n).astype(np.int32)
    a = (rng.standard_normal(n) * 4).astype(dt)
    b = (rng.standard_normal(n) * 4).astype(dt)
    for col in (a, b):
        m = rng.random(n)
        col[m < 0.002] = np.inf
        col[(m >= 0.002) & (m < 0.004)] = -np.inf
        col[(m >= 0.004) & (m < 0.03)] = np.nan
    ri, off = o.group([k])
    assert np.isinf(a).sum() > 50 and np.isinf(b).sum() > 50
    for op in ("cov", "corr"):
        want = o.reduce2(op, a, b, ri, off)
        got = ctx.reduce2(op, a, b, ri, off)
        close(got, want, op, scale=16.0)
        assert np.isinf(want).any() or op == "corr"
    # the last row of a group is the infinity: cov stays +-inf there
    x = np.array([1.0, 2.0, 3.0, 1.0, 2.0, np.inf], dt)
    y = np.array([2.0, 1.0, np.inf, 5.0, 7.0, 1.0], dt)
    off2 = np.array([0, 3, 6], np.int32)
    for op in ("cov", "corr"):
        assert_same(ctx.reduce2(op, x, y, None, off2), o.reduce2(op, x, y, None, off2), op)
