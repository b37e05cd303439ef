"""The reference-side binding of INTEGRATION.md (integration/datatable_hip_shim.py): query
matching, pointer/stype extraction and fall-through, exercised against the REAL reference as built
into oracle/_ref by oracle/build_ref.sh (skipped only when that build was never made).
No GPU needed: nothing here calls a compute entry point of libdthip (tests/test_shim_e2e.py does, on the GPU)."""
import numpy as np
import pytest

from oracle import ref

dt = ref.load()
if dt is None:
    pytest.skip("oracle/_ref (the reference build) is not present: run oracle/build_ref.sh", allow_module_level=True)


def test_match_and_fallthrough():
    from datatable import f, sum, mean, count, min as dmin
    from integration import datatable_hip_shim as shim
    DT = shim.Frame(k=np.array([3, 1, 3], np.int64), v=np.array([1.0, 2.0, 4.0]), s=["a", "b", "c"])
    assert shim.match(DT, (slice(None), [sum(f.v), mean(f["v"]), count(), dmin(f[1])], shim.by(f.k))) == \
        ([0], [("sum", 1), ("mean", 1), ("count0", None), ("min", 1)])
    assert shim.match(DT, (slice(None), sum(f.v), shim.by("k"))) == ([0], [("sum", 1)])
    # not covered -> None -> the reference evaluates it
    assert shim.match(DT, (slice(None), sum(f.v), shim.by(f.s))) is None          # string key
    assert shim.match(DT, (slice(None), sum(f.v + 1), shim.by(f.k))) is None      # computed argument
    assert shim.match(DT, (f.v > 1, sum(f.v), shim.by(f.k))) is None              # i-filter
    assert shim.match(DT, (slice(None), f.v, shim.by(f.k))) is None               # not a reducer
    R = DT[:, sum(f.v), shim.by(f.s)]                                             # falls through
    assert type(R) is dt.Frame and R.to_list() == [["a", "b", "c"], [1.0, 2.0, 4.0]]
    assert DT[f.v > 1, :].nrows == 2


def test_borrowed_pointers_are_the_numpy_buffers():
    from integration import datatable_hip_shim as shim
    a = np.arange(10, dtype=np.int64)
    DT = shim.Frame(k=a)
    assert dt.internal.frame_column_data_r(DT, 0).value == a.ctypes.data
    assert DT.stypes[0].value == 5          # == DTHIP_INT64


def test_match_groupwise_operators():
    """sd / median / nunique / first / last, cov / corr and the cumulative operators are recognised from
    the reference's own reprs (old-style `Expr:stdev(FExpr<f.v>; )` nodes included)"""
    import datatable
    from datatable import f
    from integration import datatable_hip_shim as shim
    DT = shim.Frame(k=np.array([3, 1, 3], np.int64), v=np.array([1.0, 2.0, 4.0]), w=np.array([1, 2, 3], np.int32),
                    s=["a", "b", "c"])
    j = [datatable.sd(f.v), datatable.median(f["w"]), datatable.nunique(f[1]), datatable.first(f.v), datatable.cov(f.v, f.w),
         datatable.corr(f.w, f.v), datatable.count()]
    assert shim.match(DT, (slice(None), j, shim.by(f.k))) == \
        ([0], [("sd", 1), ("median", 2), ("nunique", 1), ("first", 1), ("cov", (1, 2)), ("corr", (2, 1)), ("count0", None)])
    j = [datatable.cumsum(f.v), datatable.cummax(f.w, reverse=True), datatable.cumcount(), datatable.ngroup(reverse=True)]
    assert shim.match(DT, (slice(None), j, shim.by(f.k))) == \
        ([0], [("cumsum", 1, False), ("cummax", 2, True), ("cumcount", None, False), ("ngroup", None, True)])
    # fillna(col, reverse) rides the same scan (fexpr_fillna.cc:85-117); fillna(col, value=...) is an ifelse: the reference's
    j = [datatable.fillna(f.v), datatable.fillna(f.w, reverse=True)]
    assert shim.match(DT, (slice(None), j, shim.by(f.k))) == ([0], [("fillna", 1, False), ("fillna", 2, True)])
    assert shim.match(DT, (slice(None), datatable.fillna(f.v, value=0.0), shim.by(f.k))) is None
    # a reducer next to a row-level operator, or a string argument: the reference evaluates it
    assert shim.match(DT, (slice(None), [datatable.sd(f.v), datatable.cumsum(f.v)], shim.by(f.k))) is None
    assert shim.match(DT, (slice(None), datatable.nunique(f.s), shim.by(f.k))) is None
    assert shim.match(DT, (slice(None), datatable.sd(f.v + 1), shim.by(f.k))) is None


def test_reference_thread_settings_are_safe():
    """oracle/ref.py: sort.nthreads is stored through a uint8_t cast in the reference (sort.cc:337), so 256 would become 0
    and group() would divide by it (SIGFPE on the 256-thread GPU box): the helper clamps it, and a groupby still works"""
    from datatable import f, by
    ref.set_threads(256)
    try:
        assert dt.options.sort.nthreads == 255 and dt.options.nthreads >= 1
        DT = dt.Frame(k=np.arange(100_000) % 1000, v=np.ones(100_000))
        R = DT[:, dt.sum(f.v), by(f.k)]
        assert R.nrows == 1000 and R[:, 1].to_numpy().sum() == 100_000
    finally:
        ref.set_threads(min(8, __import__("os").cpu_count() or 1))


def _canonical_mask(vals, na, th):
    """rows a canonical predicate of shim._threshold selects (what dthip_filter_take evaluates): NA rows never pass,
    except for != (and the degenerate 'all')"""
    kind = th[0]
    if kind == "none":
        return np.zeros(len(vals), bool)
    if kind == "all":
        return np.ones(len(vals), bool)
    if kind == "isna":
        return na.copy()
    if kind == "notna":
        return ~na
    t = th[1]
    with np.errstate(invalid="ignore"):
        m = {"ge": vals >= t, "le": vals <= t, "eq": vals == t, "ne": vals != t}[kind]
    return (m | na) if kind == "ne" else (m & ~na)


@pytest.mark.parametrize("stype", ["int8", "int32", "int64", "float32", "float64"])
def test_filter_thresholds_recovered_exactly(stype):
    """`f.col <cmp> scalar`: an FExpr prints float scalars with six decimals, so the shim recovers the exact threshold by
    probing the reference's own evaluation of the expression; the canonical predicate it hands to dthip_filter_take must
    select exactly the rows the reference selects -- for scalars the repr cannot carry, int columns against fractions,
    float32 columns against float64 scalars, out-of-range and infinite scalars, None"""
    import operator
    from datatable import f
    from integration import datatable_hip_shim as shim
    rng = np.random.default_rng(len(stype))
    npdt = np.dtype(stype)
    n = 400
    if npdt.kind == "f":
        vals = np.concatenate([rng.standard_normal(n - 8) * 3, [0.0, -0.0, 1.5, 1.5000001, np.inf, -np.inf, 1e-300 if stype == "float64" else 1e-30, 2.0]]).astype(npdt)
        na = rng.random(n) < 0.1
        vals[na] = np.nan
        scalars = [0, 1.5, 1.5000001, 1.4999999999, -0.0, 2, 1e-300, -1e308, 1e308, float("inf"), float("-inf"), 0.1, 1 / 3, 1e-7, None, True]
    else:
        info = np.iinfo(npdt)
        vals = rng.integers(max(info.min + 1, -50), min(info.max, 50), n).astype(npdt)
        vals[:4] = [info.max, info.min + 1, 0, -1]
        na = rng.random(n) < 0.1
        vals[na] = info.min
        scalars = [0, 3, -7, 2.5, -0.5, 3.0, 1e12, -1e12, info.max, info.min + 1, 2**40, 1e300, float("inf"), None, False]
    DT = shim.Frame(a=np.arange(n, dtype=np.int32), x=vals, b=rng.standard_normal(n))
    cmp = {">": operator.gt, ">=": operator.ge, "<": operator.lt, "<=": operator.le, "==": operator.eq, "!=": operator.ne}
    routed = 0
    for sc in scalars:
        for op, fn in cmp.items():
            if sc is None and op not in ("==", "!="):
                continue
            for col in (f.x, f["x"], f[1]):
                expr = fn(col, sc)
                exp = dt.Frame.__getitem__(DT, (expr, "a")).to_numpy().ravel()         # the reference's own answer
                plan = shim.match_filter(DT, (expr, slice(None)))
                if plan is None:
                    continue
                ci, th, cols = plan
                assert ci == 1 and cols == [0, 1, 2]
                got = np.nonzero(_canonical_mask(vals, na, th))[0]
                assert np.array_equal(got, exp), (stype, op, sc, th, len(got), len(exp))
                routed += 1
    # nearly every predicate is taken (== / != against a float scalar that the six-decimal repr cannot carry goes to the reference)
    assert routed >= (0.8 if stype.startswith("float") else 0.9) * 3 * (6 * (len(scalars) - 1) + 2), routed


def test_match_rows_sort_and_filter_forms():
    from datatable import f, sum
    from integration import datatable_hip_shim as shim
    DT = shim.Frame(k=np.array([3, 1, 3], np.int64), v=np.array([1.0, 2.0, 4.0]), w=np.array([1, 2, 3], np.int32))
    S = shim.Frame(k=np.array([3, 1, 3], np.int64), s=["a", "b", "c"])
    assert shim.match_rows(DT, (slice(None), slice(None), shim.by(f.k))) == ([0], [1, 2])       # `:` leaves the by-column out
    assert shim.match_rows(DT, (slice(None), f[:], shim.by("k"))) == ([0], [1, 2])
    assert shim.match_rows(DT, (slice(None), [f.k, f.w], shim.by(f.k))) == ([0], [0, 2])          # listed explicitly: kept
    assert shim.match_rows(DT, (slice(None), [f.k, "w"], shim.by(f.k))) is None                   # mixed selector types
    assert shim.match_rows(DT, (slice(None), f.v, shim.by(f.k, f.w))) == ([0, 2], [1])
    assert shim.match_rows(DT, (slice(None), sum(f.v), shim.by(f.k))) is None                     # a reducer: match()'s
    assert shim.match_rows(DT, (f.v > 1, slice(None), shim.by(f.k))) is None                      # i + by: the reference refuses it
    assert shim.match_rows(S, (slice(None), slice(None), shim.by(f.k))) is None                   # a string column rides along
    assert shim.match_sort(DT, (slice(None), slice(None), shim.sort(f.k))) == ([0], [False], 0, [0, 1, 2])
    assert shim.match_sort(DT, (slice(None), [f.v], shim.sort(f.k, f.w, reverse=[True, False], na_position="last"))) == \
        ([0, 2], [True, False], 1, [1])
    assert shim.match_sort(DT, (slice(None), slice(None), shim.sort(f.k, na_position="remove")))[2] == 2
    assert shim.match_sort(DT, (slice(None), slice(None), shim.sort(f.k + 1))) is None            # computed sort key
    assert shim.match_filter(DT, (f.v > 1, slice(None)))[:2] == (1, ("ge", float(np.nextafter(1.0, 2.0))))
    assert shim.match_filter(DT, (f.v > 1, [f.k, f.w]))[2] == [0, 2]
    assert shim.match_filter(DT, ((f.v > 1) & (f.k > 1), slice(None))) is None                    # a compound predicate
    assert shim.match_filter(DT, (f.v > f.w, slice(None))) is None                                # column against column
    assert shim.match_filter(S, (f.k > 1, slice(None))) is None                                   # string column selected
    assert shim.match_filter(S, (f.k > 1, "k"))[2] == [0]
    # the native objects still reach the reference for everything else
    R = S[:, :, shim.sort(f.s)]
    assert R.to_list() == [[3, 1, 3], ["a", "b", "c"]]


# ---- residency layer (round 4): cache validation, invalidation by every mutating call, DeviceFrame -- with a stand-in
# for the library's memory entry points (host memory plays HBM), so the host logic runs without a GPU
class _FakeLib:
    def __init__(self):
        import ctypes as C
        self.C = C
        self.live = {}
        self.uploads = 0

    def dthip_malloc(self, h, nbytes, pp):
        buf = self.C.create_string_buffer(int(nbytes))
        p = self.C.addressof(buf)
        self.live[p] = buf
        pp._obj.value = p
        return 0

    def dthip_free(self, h, p):
        self.live.pop(p.value, None)
        return 0

    def dthip_memcpy_h2d(self, h, dst, src, n):
        self.uploads += 1
        self.C.memmove(dst.value, src.value, n)
        return 0

    def dthip_memcpy_d2h(self, h, dst, src, n):
        self.C.memmove(dst.value, src.value, n)
        return 0

    def dthip_set_option(self, h, name, v):
        return 0


class _FakeCtx:
    def __init__(self):
        self._lib, self._h = _FakeLib(), 1

    def set_option(self, name, value):
        pass


def test_resident_cache_is_validated_and_dropped_by_mutations():
    from datatable import f, update
    from integration import datatable_hip_shim as shim
    ctx = _FakeCtx()
    old = shim.options.residency
    shim.options.residency = "auto"
    try:
        DT = shim.Frame(k=np.array([3, 1, 3, 2], np.int64), v=np.array([1.0, 2.0, 4.0, 8.0]), s=["a", "b", "c", "d"])
        e = shim._resident(DT, [0, 1], ctx)
        assert ctx._lib.uploads == 2 and DT.is_resident and [x.nrows for x in e] == [4, 4]
        assert shim._resident(DT, [1, 0], ctx)[0] is e[1] and ctx._lib.uploads == 2          # served from the cache
        cols, mem = shim._columns(DT, [1], ctx)
        assert mem == shim.L.DEVICE and cols[0].data == e[1].ptr and cols[0].stype == 7
        got = np.frombuffer(ctx._lib.live[e[1].ptr], dtype=np.float64, count=4)
        assert got.tolist() == [1.0, 2.0, 4.0, 8.0]
        mutations = [
            lambda D: D.__setitem__((0, "v"), 5.0),
            lambda D: D.__delitem__((slice(None), "s")),
            lambda D: D.cbind(dt.Frame(z=[1, 2, 3, 4])),
            lambda D: D.rbind(dt.Frame(k=[9], v=[9.0], s=["z"])),
            lambda D: D.replace(1.0, 7.0),
            lambda D: D.materialize(),
            lambda D: setattr(D, "nrows", 2),
            lambda D: setattr(D, "key", "k"),
            lambda D: setattr(D, "names", ("k", "w", "s")),
            lambda D: D[:, update(v=f.v * 2)],
        ]
        for m in mutations:
            D = shim.Frame(k=np.array([3, 1, 4, 2], np.int64), v=np.array([1.0, 2.0, 4.0, 8.0]), s=["a", "b", "c", "d"])
            shim._resident(D, [0, 1], ctx)
            assert D.is_resident
            m(D)
            assert not D.is_resident, "a mutating call kept the device copies"
        # second line of defence: a column whose host buffer moved, or whose row count / stype changed, is uploaded again
        D = shim.Frame(k=np.arange(6, dtype=np.int64), v=np.arange(6.0))
        shim._resident(D, [0], ctx)
        n0 = ctx._lib.uploads
        D.__dict__["_dthip_dev"][0].host_ptr += 8                   # pretend the buffer was reallocated behind our back
        shim._resident(D, [0], ctx)
        assert ctx._lib.uploads == n0 + 1
        shim.options.residency = "off"
        assert shim._resident(D, [0], ctx) is None
        assert shim._columns(D, [0], ctx)[1] == shim.L.HOST
        # a plain reference Frame is never cached
        shim.options.residency = "auto"
        assert shim._resident(dt.Frame(k=[1, 2]), [0], ctx) is None
        D.to_device(ctx)
        assert sorted(D.__dict__["_dthip_dev"]) == [0, 1]
        D.release_device()
        assert not D.is_resident
    finally:
        shim.options.residency = old


def test_device_frame_metadata_download_and_adoption():
    from integration import datatable_hip_shim as shim
    ctx = _FakeCtx()
    src = shim.Frame(a=np.array([5, 6, 7], np.int32), b=np.array([True, False, True]), c=np.array([0.5, np.nan, 2.0]))
    dev = [shim._upload_column(ctx, src, c) for c in range(3)]
    for e in dev:
        e.host_ptr = None
    V = shim.DeviceFrame(ctx, shim._mangled(["a", "a", ""]), [dt.stype.int32, dt.stype.bool8, dt.stype.float64], 3, dev, True)
    assert V.names == ("a", "a.0", "C0") and V.shape == (3, 3) and V.nrows == 3 and V.ncols == 3 and len(V) == 3
    assert V.stypes == (dt.stype.int32, dt.stype.bool8, dt.stype.float64)
    assert "HBM" in repr(V) and V._frame is None                       # nothing downloaded so far
    assert V.to_list() == [[5, 6, 7], [True, False, True], [0.5, None, 2.0]]      # any Frame method: download once
    F = V.to_frame()
    assert type(F) is shim.Frame and F.stypes == V.stypes and F is V.to_frame()
    assert F.is_resident and sorted(F.__dict__["_dthip_dev"]) == [0, 1, 2]          # the device copies were adopted
    n0 = ctx._lib.uploads
    old = shim.options.residency
    shim.options.residency = "auto"
    try:
        assert shim._resident(F, [0, 2], ctx)[0].ptr == dev[0].ptr and ctx._lib.uploads == n0     # nothing uploaded again
    finally:
        shim.options.residency = old
    assert np.asarray(V).shape == (3, 3)
    assert V[1, "a"] == 6                                              # other forms of [] go to the Frame
    W = shim.DeviceFrame(ctx, ["x"], [dt.stype.int64], 0, [shim._DevColumn(0, 0, None, 0, 5, None)], False)
    assert type(W.to_frame()) is dt.Frame and W.to_frame().shape == (0, 1)
