"""Reference-side binding (seam S-py): what a datatable maintainer adds to route
the `DT[:, {sum|mean|min|max|count}(f.col) ..., by(cols)]` hot path to libdthip.so -- and, through
the S-red entry points, the other reducers and group-wise operators that share its Groupby
(first/last/sd/median/nunique, cov/corr, cumsum/cumprod/cummin/cummax/fillna/cumcount/ngroup).

It runs INSIDE the reference's Python process (needs `import datatable`), touches no reference
source, and forwards every other form of `DT[...]` to the reference unchanged:

    import datatable as dt
    from datatable import f, sum, mean, count
    from integration.datatable_hip_shim import Frame, by, sort  # the shim's Frame, by() and sort()
    DT = Frame(dt.fread("data.jay"))                             # or Frame(k=..., v=...)
    DT[:, [sum(f.v), count()], by(f.k)]                          # -> libdthip (MI355X): dthip_groupby_agg
    V = DT[f.x > 0, :]; V[:, :, by(f.k)]                         # -> dthip_filter_take, dthip_groupby_rows (BASELINE config 5)
    DT[:, [f.a, f.b], by(f.k)];  DT[:, :, sort(f.k)];  DT.sort("k")   # -> dthip_groupby_rows / dthip_groupby
    DT[f.s == "a", :]                                            # -> the reference, as before

How data crosses the boundary (all borrowed, zero-copy on the host side):
  * column buffers: `dt.internal.frame_column_data_r(frame, i)` -> `ctypes.c_void_p`
    (src/core/datatablemodule.cc:135-145; same pointers as DtFrame_ColumnDataR,
    src/datatable/include/datatable.h:80-99).  It materialises virtual columns first, so
    a filter view arrives as a plain buffer.
  * stype codes: `frame.stypes[i].value` == enum dthip_stype (src/core/stype.h:41-62)
  * NA sentinels are the reference's own storage (INT*_MIN / NaN): nothing to convert
  * results come back as numpy buffers wrapped by `dt.Frame` through the buffer protocol
    (src/core/py_buffers.cc:54-113); names = by-columns, then one column per reducer named
    after its input (fexpr_reduce_unary.cc:60-64), "count" for count() (fexpr_count.cc:120);
    duplicate names are mangled by the reference itself (names.cc:232-266).
  * errors: negative DTHIP_E* codes -> the Python exception the reference would raise
    (`datatable_amd._lib.check`, mirroring api.cc:34-38).

`by()` and `sort()` must be the shim's: the reference's `datatable.by` / `datatable.sort` objects are opaque from
Python (src/core/expr/py_by.cc:71-78, py_sort.cc:33-100), so the shim cannot read the columns out of them.

Row filters `DT[f.col <cmp> scalar, cols]`: an FExpr cannot be taken apart from Python either and its repr prints
float scalars with six decimals only (`FExpr<f.v >= 1.500000>`), so the column and the operator come from the repr
and the EXACT threshold is recovered by asking the reference itself: the predicate is evaluated on small probe frames
of candidate values (a bisection over the column's value domain, ~6 evaluations of <= 1024 rows), which yields the
smallest / largest element value that passes.  Whatever promotion rules the reference applies (an int column against
2.5, a float32 column against a float64 scalar) are thereby reproduced rather than re-implemented.  The result of a
row filter, of `DT[:, cols, by()]` and of a sort is a shim Frame (still a `datatable.Frame`) with materialised columns,
so the second step of the two-step form of config 5 stays on the GPU; aggregations return the base class like the
reference does.
"""
import os
import re
import threading
import warnings

import numpy as np

import datatable as dt

from datatable_amd import _lib as L
from datatable_amd.engine import ST2NP, Context, _DevBuf


class _Options:
    """shim.options
    residency  "auto" (default): the columns an accelerated query touches are uploaded to HBM ONCE and kept per Frame
                  (cache entry = column index -> device buffer, valid while the column's host buffer pointer
                  `frame_column_data_r`, the row count and the stype are unchanged; every mutating Frame call drops the
                  cache); every route then passes DTHIP_DEVICE pointers.  Results are downloaded at once and are plain
                  reference objects, exactly as before.
               "lazy": "auto" + results STAY in HBM: `DT[...]` returns a DeviceFrame (names / stypes / shape answered
                  from metadata, the next accelerated `DT[...]` on it runs on the device columns, anything else
                  downloads once and hands over to a real Frame).  The two statements of BASELINE config 5 never leave
                  the GPU this way.
               "off": host pointers in, host buffers out on every call (rounds 1-3; PCIe-bound).
               Environment: DTHIP_SHIM_RESIDENCY.
    f32_sum    True (default -- a drop-in binding returns the reference's bits): sum(float32 column) accumulates in float32 row
               by row like the reference (dthip option f32_sum = 1, column/sumprod.h:48-55: bit for bit, one thread per
               group); False: float64 accumulation, rounded once (more accurate and faster; differs from the reference by its
               own float32 rounding, ~1e-7 x sum|v| per group).  Environment: DTHIP_SHIM_F32_SUM=0 / 1."""
    residency = os.environ.get("DTHIP_SHIM_RESIDENCY", "auto")
    # residency "lazy" only: V = DT[f.x <cmp> c, cols] stays a pending view until it is used; V[:, cols, by(key)] then runs as
    # ONE fused library call (BASELINE config 5's two statements).  False: the filter runs at once (round 4's behaviour)
    defer_filter = os.environ.get("DTHIP_SHIM_DEFER_FILTER", "1") not in ("", "0")
    f32_sum = os.environ.get("DTHIP_SHIM_F32_SUM", "1") not in ("", "0")


options = _Options()
stats = {"arrow_uploads": 0, "fused_filter_rows": 0}      # columns that went to HBM in Arrow layout (dthip_from_arrow), for tests and curiosity


_tls = threading.local()


def default_context():
    """The binding's OWN context of the calling thread -- not datatable_amd.engine.default_context(): the options set in
    _context() (agg_offsets = 0, f32_sum) would otherwise change what every other user of that shared context gets
    (engine.groupby_agg(...).offsets(), torch_bridge) merely because this module was imported."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = _tls.ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return ctx


def _context(ctx=None):
    ctx = ctx or default_context()
    if not getattr(ctx, "_shim_init", False):
        # the result Frame of DT[:, sum(f.v), by(f.k)] holds keys and aggregates, no group sizes: without them the bucketed
        # aggregation tracks key presence only and uses twice the table slots per bucket (C3: 8.46 -> 8.0 ms)
        ctx.set_option("agg_offsets", 0)
        ctx._shim_init = True
    want = 1 if options.f32_sum else 0
    if getattr(ctx, "_shim_f32", None) != want:
        ctx.set_option("f32_sum", want)
        ctx._shim_f32 = want
    return ctx


class _DevColumn:
    """one column in HBM: pointer, bytes, what it mirrors (host pointer / rows / stype) and the owner of the memory
    (a dthip_malloc buffer, or the dthip_result a pointer was borrowed from)"""
    __slots__ = ("ptr", "nbytes", "host_ptr", "nrows", "stype", "keep")

    def __init__(self, ptr, nbytes, host_ptr, nrows, stype, keep):
        self.ptr, self.nbytes, self.host_ptr, self.nrows, self.stype, self.keep = int(ptr or 0), nbytes, host_ptr, nrows, stype, keep


def _dev_alloc(ctx, nbytes):
    import ctypes as C
    p = C.c_void_p()
    L.check(ctx._lib.dthip_malloc(ctx._h, max(int(nbytes), 1), C.byref(p)))
    return _DevBuf(ctx, p.value)


def _arrow_buffers(frame, c, virtual=None):
    """(validity address or 0, values address, owner) when column c can be handed over in Arrow layout WITHOUT the
    reference's materialisation pass, else None.  A column that came from an Arrow table is two buffers -- validity bitmap
    and values (ArrowFw_ColumnImpl / ArrowBool_ColumnImpl, column_from_arrow.cc:40-59) -- and is reported `virtual`;
    `frame_column_data_r` would first rewrite it element by element into a sentinel column on the CPU
    (arrow_fw.cc:63-72 read by _materialize_fw).  Its Arrow export is zero-copy (Column::to_arrow hands out the column's own
    buffers, frame/to_arrow.cc:85-118), so the two addresses go to dthip_from_arrow and the NA sentinels are written by a
    kernel on the device.  Needs pyarrow (the reference's own to_arrow() does)."""
    # `virtual`: the frame's virtual-column tuple, asked for ONCE per upload batch by _resident (it is O(ncols) itself).
    # A virtual column that is not Arrow-backed (a view, a cast, a computed column) would be materialised by to_arrow() on the
    # CPU and converted a second time on the device: there is no Python-visible mark of an Arrow-backed column, so only frames
    # whose columns are ALL virtual and that are not views of another frame's rows (what dt.Frame(pyarrow_table) gives) take
    # this route; everything else keeps frame_column_data_r
    if virtual is None:
        virtual = dt.internal.frame_columns_virtual(frame)
    if not virtual[c] or not all(virtual):
        return None
    try:
        import pyarrow  # noqa: F401
    except ImportError:
        return None
    try:
        col = dt.Frame.__getitem__(frame, (slice(None), c)).to_arrow().column(0)
    except (TypeError, ValueError, NotImplementedError, RuntimeError):      # stypes / layouts the reference's exporter refuses
        return None
    if col.num_chunks != 1:
        return None
    a = col.chunk(0)
    bufs = a.buffers()
    if a.offset != 0 or len(bufs) != 2 or bufs[1] is None or len(a) != frame.nrows:
        return None
    return (bufs[0].address if bufs[0] is not None else 0), bufs[1].address, a


def _upload_column(ctx, frame, c, virtual=None):
    import ctypes as C
    st = frame.stypes[c].value
    n = frame.nrows
    ab = _arrow_buffers(frame, c, virtual) if st in _ACCEL_STYPES and n else None
    if ab is not None:
        nbytes = n * ST2NP[st].itemsize
        buf = _dev_alloc(ctx, nbytes)
        L.check(ctx._lib.dthip_from_arrow(ctx._h, C.c_void_p(ab[1]), C.c_void_p(ab[0]) if ab[0] else None, n, st, L.HOST, C.c_void_p(buf.ptr)))
        stats["arrow_uploads"] += 1
        # (no host pointer to re-validate against: asking the reference for one would materialise the column after all;
        # every mutating Frame call drops the cache, and row count / stype are compared on every use)
        return _DevColumn(buf.ptr, nbytes, None, n, st, buf)
    hp = dt.internal.frame_column_data_r(frame, c).value or 0
    nbytes = n * ST2NP[st].itemsize
    buf = _dev_alloc(ctx, nbytes)
    if nbytes:
        L.check(ctx._lib.dthip_memcpy_h2d(ctx._h, C.c_void_p(buf.ptr), C.c_void_p(hp), nbytes))
    return _DevColumn(buf.ptr, nbytes, hp, n, st, buf)


def _resident(frame, cols, ctx):
    """device columns of `cols` of a host Frame (uploading what the cache lacks or what went stale), or None when
    residency is off / the frame is not the shim's"""
    if options.residency == "off" or not isinstance(frame, Frame):
        return None
    cache = frame.__dict__.setdefault("_dthip_dev", {})
    if frame.__dict__.get("_dthip_ctx") not in (None, ctx):
        cache.clear()
    frame.__dict__["_dthip_ctx"] = ctx
    out = []
    virtual = None
    for c in cols:
        e = cache.get(c)
        st = frame.stypes[c].value
        if e is not None and e.host_ptr is not None:
            hp = dt.internal.frame_column_data_r(frame, c).value or 0
            if hp != e.host_ptr:
                e = None
        if e is not None and (e.nrows != frame.nrows or e.stype != st):
            e = None
        if e is None:
            if virtual is None:
                virtual = dt.internal.frame_columns_virtual(frame)
            e = cache[c] = _upload_column(ctx, frame, c, virtual)
        out.append(e)
    return out


def _columns(frame, cols, ctx, flags=None):
    """(list of L.Col, memory space) for one library call: device pointers when the frame is a DeviceFrame or its columns
    are resident, else the reference's borrowed host pointers (src/datatable/include/datatable.h:80-99)"""
    flags = flags or [0] * len(cols)
    if isinstance(frame, DeviceFrame):
        return [L.Col(frame._cols[c].ptr, frame.stypes[c].value, fl) for c, fl in zip(cols, flags)], L.DEVICE
    dev = _resident(frame, cols, ctx)
    if dev is not None:
        return [L.Col(e.ptr, frame.stypes[c].value, fl) for e, c, fl in zip(dev, cols, flags)], L.DEVICE
    return [L.Col(dt.internal.frame_column_data_r(frame, c).value, frame.stypes[c].value, fl) for c, fl in zip(cols, flags)], L.HOST


def _lazy():
    return options.residency == "lazy"

# min()/max() print their argument as a one-element list: FExpr<min([f.v])>
_REDUCER = re.compile(r"^FExpr<(sum|mean|min|max|count)\(\[?(?:f\.(\w+)|f\['([^']+)'\]|f\[(\d+)\])?\]?\)>$")
_COLUMN = re.compile(r"^FExpr<(?:f\.(\w+)|f\['([^']+)'\]|f\[(\d+)\])>$")
_CREF = r"(?:f\.\w+|f\['[^']+'\]|f\[\d+\])"
# old-style expression nodes print as  Expr:stdev(FExpr<f.v>; )
_REDUCER_OLD = re.compile(r"^Expr:(stdev|median|nunique|first|last)\((FExpr<%s>); \)$" % _CREF)
_REDUCER2 = re.compile(r"^Expr:(cov|corr)\((FExpr<%s>), (FExpr<%s>); \)$" % (_CREF, _CREF))
_CUMULATIVE = re.compile(r"^FExpr<(cumsum|cumprod|cummin|cummax|fillna)\((%s), reverse=(True|False)\)>$" % _CREF)
_CUMCOUNT = re.compile(r"^FExpr<(cumcount|ngroup)\(reverse=(True|False)\)>$")
_OLD2OP = {"stdev": "sd"}
_FUSED = ("sum", "mean", "min", "max", "count", "count0")
_ACCEL_STYPES = {1, 2, 3, 4, 5, 6, 7}


class by:
    """by(f.k, "name", ...) understood by the shim; `.native()` is the reference's own object"""

    def __init__(self, *cols):
        self.cols = list(cols[0]) if len(cols) == 1 and isinstance(cols[0], (list, tuple)) else list(cols)

    def native(self):
        return dt.by(*self.cols)


class sort:
    """sort(f.k, ..., reverse=False, na_position="first") understood by the shim (py_sort.cc:33-100)"""

    def __init__(self, *cols, reverse=False, na_position="first"):
        self.cols = list(cols[0]) if len(cols) == 1 and isinstance(cols[0], (list, tuple)) else list(cols)
        self.reverse, self.na_position = reverse, na_position

    def native(self):
        return dt.sort(*self.cols, reverse=self.reverse, na_position=self.na_position)


_FILTER = re.compile(r"^FExpr<(%s) (>=|<=|==|!=|>|<) (.+)>$" % _CREF)
_INT = re.compile(r"^-?\d+$")
_FLOAT = re.compile(r"^-?(\d+\.\d+|inf)$")
_ALLCOLS = "FExpr<f[:]>"


def _colindex(frame, spec):
    """'name' | FExpr<f.name> | FExpr<f[i]> -> column index, or None if not a plain column"""
    if isinstance(spec, str):
        return frame.names.index(spec) if spec in frame.names else None
    if isinstance(spec, int):
        return spec if 0 <= spec < frame.ncols else None
    m = _COLUMN.match(repr(spec))
    if not m:
        return None
    if m.group(3) is not None:
        return _colindex(frame, int(m.group(3)))
    return _colindex(frame, m.group(1) or m.group(2))


def match(frame, item):
    """(i, j, by) -> (key indices, [(op, value index | None)]) when libdthip covers the query, else None"""
    if not (isinstance(item, tuple) and len(item) == 3 and isinstance(item[2], by)):
        return None
    i, j, b = item
    if not (i is None or i is Ellipsis or (isinstance(i, slice) and i == slice(None))):
        return None
    if frame.nrows > 2**31 - 1:
        return None
    keys = [_colindex(frame, c) for c in b.cols]
    if not keys or any(k is None for k in keys) or not _uniform(b.cols):
        return None
    aggs = []
    for expr in (j if isinstance(j, (list, tuple)) else [j]):
        r = repr(expr)
        m = _REDUCER.match(r)
        if m:
            op, ref = m.group(1), (m.group(2) or m.group(3) or m.group(4))
            if ref is None:
                if op != "count":
                    return None
                aggs.append(("count0", None))
                continue
            ci = _colindex(frame, int(ref) if m.group(4) is not None else ref)
            if ci is None:
                return None
            aggs.append((op, ci))
            continue
        m = _REDUCER_OLD.match(r)
        if m:
            ci = _colspec(frame, m.group(2))
            if ci is None:
                return None
            aggs.append((_OLD2OP.get(m.group(1), m.group(1)), ci))
            continue
        m = _REDUCER2.match(r)
        if m:
            ca, cb = _colspec(frame, m.group(2)), _colspec(frame, m.group(3))
            if ca is None or cb is None:
                return None
            aggs.append((m.group(1), (ca, cb)))
            continue
        m = _CUMULATIVE.match(r)
        if m:
            ci = _colspec(frame, "FExpr<%s>" % m.group(2))
            if ci is None:
                return None
            aggs.append((m.group(1), ci, m.group(3) == "True"))
            continue
        m = _CUMCOUNT.match(r)
        if m:
            aggs.append((m.group(1), None, m.group(2) == "True"))
            continue
        return None
    rowwise = [len(a) == 3 for a in aggs]
    if any(rowwise) and not all(rowwise):
        return None           # reducers broadcast next to row-level columns: left to the reference
    used = list(keys)
    for a in aggs:
        c = a[1]
        used += list(c) if isinstance(c, tuple) else ([] if c is None else [c])
    if any(frame.stypes[c].value not in _ACCEL_STYPES for c in used):
        return None           # strings, dates, ...: the reference handles them
    return keys, aggs


def _colspec(frame, text):
    """'FExpr<f.v>' (as printed inside an old-style Expr) -> column index"""
    m = _COLUMN.match(text)
    if not m:
        return None
    if m.group(3) is not None:
        return _colindex(frame, int(m.group(3)))
    return _colindex(frame, m.group(1) or m.group(2))


_MANGLED = {}


def _mangled(names):
    """result names as the reference would make them: duplicates mangled (names.cc:232-266), "" auto-named
    (asked of the reference itself on a zero-row Frame; remembered per tuple of names)"""
    key = tuple(names)
    hit = _MANGLED.get(key)
    if hit is None:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", dt.exceptions.DatatableWarning)
            hit = list(dt.Frame([[]] * len(names), names=list(names)).names) if names else []
        if len(_MANGLED) < 4096:
            _MANGLED[key] = hit
    return list(hit)


class _Out:
    """output columns of one query, in the memory space of the call: numpy buffers (DTHIP_HOST) or HBM (DTHIP_DEVICE)"""

    def __init__(self, ctx, mem):
        self.ctx, self.mem = ctx, mem
        self.cols, self.names, self.stypes, self.nrows = [], [], [], None

    def alloc(self, n, st):
        """-> (pointer for the library call, holder)"""
        if self.mem == L.HOST:
            a = np.empty(n, ST2NP[st])
            return a.ctypes.data, a
        buf = _dev_alloc(self.ctx, n * ST2NP[st].itemsize)
        return buf.ptr, _DevColumn(buf.ptr, n * ST2NP[st].itemsize, None, n, st, buf)

    def add(self, holder, name, st, nrows=None):
        if nrows is not None:
            if isinstance(holder, _DevColumn):
                holder.nrows, holder.nbytes = nrows, nrows * ST2NP[st].itemsize
            else:
                holder = holder[:nrows]
        self.cols.append(holder); self.names.append(name); self.stypes.append(st)

    def borrowed(self, ptr, n, st, keep, name):
        """a column that lives inside a dthip_result (kept alive by `keep`)"""
        self.add(_DevColumn(ptr, n * ST2NP[st].itemsize, None, n, st, keep), name, st)

    def _host(self, holder, st):
        import ctypes as C
        if not isinstance(holder, _DevColumn):
            return holder
        a = np.empty(holder.nrows, ST2NP[st])
        if a.nbytes:
            L.check(self.ctx._lib.dthip_memcpy_d2h(self.ctx._h, C.c_void_p(a.ctypes.data), C.c_void_p(holder.ptr), a.nbytes))
        return a

    def finish(self, frame, nkeys_bool=(), wrap=False):
        """the result object: a DeviceFrame (lazy residency, device call) or the reference's Frame built from host buffers"""
        bool_cols = [i for i, st in enumerate(self.stypes) if st == L.BOOL]
        if _lazy() and self.mem == L.DEVICE:
            n = self.cols[0].nrows if self.cols else 0
            return DeviceFrame(self.ctx, _mangled(self.names), [dt.stype(st) for st in self.stypes], n, self.cols, wrap)
        host = [self._host(h, st) for h, st in zip(self.cols, self.stypes)]
        res = _finish(host, self.names, bool_cols)
        return Frame(res) if wrap else res


class _ResultKeep:
    """owner of a dthip_result whose buffers are borrowed by DeviceFrame columns"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def free(self):
        if self.h is not None and self.ctx._h is not None:
            self.ctx._lib.dthip_result_free(self.ctx._h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _finish(cols, names, bool_cols=()):
    """numpy result buffers -> the result Frame.  bool8 columns (keys, and min/max/first/last/cummin/cummax of a
    bool8 column: the reference keeps the stype, fexpr_minmax.cc:47-68) travel as int8 with NA = -128 and are
    cast back here (int8 -> bool8 keeps NA)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", dt.exceptions.DatatableWarning)     # duplicate names are mangled, as in the reference
        res = dt.Frame(cols, names=names)                                   # "" -> auto-named C<k> by the reference
    for i in sorted(set(bool_cols)):
        res[:, i] = dt.as_type(dt.f[i], dt.bool8)
    return res


def _carr(cols):
    return (L.Col * max(len(cols), 1))(*cols)


def run_sred(frame, keys, aggs, ctx=None):
    """the S-red route: dthip_groupby once, then dthip_reduce / dthip_reduce2 (one row per group) or
    dthip_cumulate (one row per input row, grouped order) per j item -- in the memory space of the frame's columns"""
    import ctypes as C
    from datatable_amd.engine import OPS, OPS2, CUMOPS
    ctx = _context(ctx)
    lib = ctx._lib
    n = frame.nrows
    used = list(keys)
    for a in aggs:
        c = a[1]
        used += list(c) if isinstance(c, tuple) else ([] if c is None else [c])
    used = list(dict.fromkeys(used))
    lcols, mem = _columns(frame, used, ctx)
    col = dict(zip(used, lcols))
    out = _Out(ctx, mem)
    h = C.c_void_p()
    L.check(lib.dthip_groupby(ctx._h, _carr([col[k] for k in keys]), len(keys), n, L.NA_FIRST, mem, 1, C.byref(h)))
    keep = _ResultKeep(ctx, h)
    try:
        ng = lib.dthip_result_ngroups(h)
        if mem == L.HOST:
            ri, off = np.empty(n, np.int32), np.empty(ng + 1, np.int32)
            L.check(lib.dthip_result_copy_rowindex(ctx._h, h, ri.ctypes.data, L.HOST))
            L.check(lib.dthip_result_copy_offsets(ctx._h, h, off.ctypes.data, L.HOST))
            ri_p, off_p = ri.ctypes.data, off.ctypes.data
        else:
            ri_p, off_p = lib.dthip_result_rowindex(h), lib.dthip_result_offsets(h)
        rowwise = bool(aggs) and len(aggs[0]) == 3
        for k in keys:                                     # the by-columns: every row / first row of each group
            st = frame.stypes[k].value
            ptr, holder = out.alloc(n if rowwise else ng, st)
            kc = col[k]
            if rowwise:
                if n:
                    L.check(lib.dthip_gather(ctx._h, C.byref(kc), C.c_void_p(ri_p), n, mem, C.c_void_p(ptr)))
            else:
                L.check(lib.dthip_result_group_keys(ctx._h, h, C.byref(kc), mem, C.c_void_p(ptr)))
            out.add(holder, frame.names[k], st)
        for a in aggs:
            op, c = a[0], a[1]
            if rowwise:
                st = lib.dthip_cumulate_out_stype(CUMOPS[op], L.INT64 if c is None else frame.stypes[c].value)
                ptr, holder = out.alloc(n, st)
                vc = col[c] if c is not None else None
                if n:
                    L.check(lib.dthip_cumulate(ctx._h, CUMOPS[op], C.byref(vc) if vc is not None else None,
                                               C.c_void_p(ri_p) if c is not None else None, C.c_void_p(off_p), ng, n,
                                               1 if a[2] else 0, mem, C.c_void_p(ptr)))
                out.add(holder, "" if c is None else frame.names[c], st)
            elif isinstance(c, tuple):
                ca, cb = col[c[0]], col[c[1]]
                st = lib.dthip_reduce2_out_stype(ca.stype, cb.stype)
                ptr, holder = out.alloc(ng, st)
                if ng:
                    L.check(lib.dthip_reduce2(ctx._h, OPS2[op], C.byref(ca), C.byref(cb), C.c_void_p(ri_p), C.c_void_p(off_p), ng, n,
                                              mem, C.c_void_p(ptr)))
                out.add(holder, "", st)
            elif c is None:
                ptr, holder = out.alloc(ng, L.INT64)
                if ng:
                    L.check(lib.dthip_reduce(ctx._h, L.COUNT0, None, None, C.c_void_p(off_p), ng, n, mem, C.c_void_p(ptr)))
                out.add(holder, "count", L.INT64)
            else:
                vc = col[c]
                st = lib.dthip_reduce_out_stype(OPS[op], vc.stype)
                ptr, holder = out.alloc(ng, st)
                if ng:
                    L.check(lib.dthip_reduce(ctx._h, OPS[op], C.byref(vc), C.c_void_p(ri_p), C.c_void_p(off_p), ng, n, mem,
                                             C.c_void_p(ptr)))
                out.add(holder, frame.names[c], st)
        return out.finish(frame)
    finally:
        keep.free()


def run(frame, keys, aggs, ctx=None):
    """evaluate the matched query through the C ABI (dthip_groupby_agg)"""
    import ctypes as C
    if not all(len(a) == 2 and a[0] in _FUSED for a in aggs):
        return run_sred(frame, keys, aggs, ctx)
    ctx = _context(ctx)
    lib = ctx._lib
    vcols = sorted({c for _, c in aggs if c is not None})
    lcols, mem = _columns(frame, list(keys) + vcols, ctx)
    karr, varr = _carr(lcols[:len(keys)]), _carr(lcols[len(keys):])
    aarr = (L.Agg * len(aggs))(*[L.Agg({"sum": L.SUM, "mean": L.MEAN, "min": L.MIN, "max": L.MAX, "count": L.COUNT,
                                        "count0": L.COUNT0}[op], -1 if c is None else vcols.index(c)) for op, c in aggs])
    h = C.c_void_p()
    L.check(lib.dthip_groupby_agg(ctx._h, karr, len(keys), varr, len(vcols), aarr, len(aggs), frame.nrows,
                                  L.NA_FIRST, mem, C.byref(h)))
    keep = _ResultKeep(ctx, h)
    out = _Out(ctx, mem)
    lazy = _lazy() and mem == L.DEVICE
    try:
        ng = lib.dthip_result_ngroups(h)
        for i, k in enumerate(keys):
            st = frame.stypes[k].value
            if lazy:
                out.borrowed(lib.dthip_result_key(h, i), ng, st, keep, frame.names[k])
            else:
                a = np.empty(ng, ST2NP[st])
                L.check(lib.dthip_result_copy_key(ctx._h, h, i, a.ctypes.data, L.HOST))
                out.add(a, frame.names[k], st)
        for a_i, (op, c) in enumerate(aggs):
            st = lib.dthip_result_agg_stype(h, a_i)
            name = "count" if c is None else frame.names[c]
            if lazy:
                out.borrowed(lib.dthip_result_agg(h, a_i), ng, st, keep, name)
            else:
                a = np.empty(ng, ST2NP[st])
                L.check(lib.dthip_result_copy_agg(ctx._h, h, a_i, a.ctypes.data, L.HOST))
                out.add(a, name, st)
        if lazy:
            return out.finish(frame)
        out.mem = L.HOST                 # the copies above already landed in host buffers
        return out.finish(frame)
    finally:
        if not lazy:
            keep.free()


# ---- row-returning routes: DT[f.x <cmp> c, cols], DT[:, cols, by(keys)], DT[:, cols, sort(...)] ---------------------
def _uniform(items):
    """the reference refuses lists that mix names, indices and expressions ("Mixed selector types are not allowed")"""
    return len({("s" if isinstance(x, str) else "i" if isinstance(x, int) else "e") for x in items}) <= 1


def _is_all(x):
    return x is None or x is Ellipsis or (isinstance(x, slice) and x == slice(None))


def _jcols(frame, j, exclude=()):
    """j = : | f[:] | column | [columns] -> column indices (`:` / f[:] leave out `exclude`: the by-columns), or None"""
    if _is_all(j) or (not isinstance(j, (str, int, list, tuple, dict)) and repr(j) == _ALLCOLS):
        return [c for c in range(frame.ncols) if c not in exclude]
    if isinstance(j, dict):
        return None
    out = []
    items = j if isinstance(j, (list, tuple)) else [j]
    if not _uniform(items):
        return None            # the reference raises that itself
    for x in items:
        if isinstance(x, bool):
            return None
        c = _colindex(frame, x)
        if c is None:
            return None
        out.append(c)
    return out


def _accel(frame, cols):
    return frame.nrows <= 2**31 - 1 and all(frame.stypes[c].value in _ACCEL_STYPES for c in cols)


def _ord(a):
    """order-preserving integer image of a float array (-0.0 just below +0.0); integers are their own image"""
    if a.dtype.kind != "f":
        return a.astype(object)
    it = np.int64 if a.dtype == np.float64 else np.int32
    b = a.view(it).astype(object)
    top = int(np.iinfo(it).max)
    return np.array([int(x) if x >= 0 else int(x) ^ top for x in b], dtype=object)


def _unord(o, dtype):
    if dtype.kind != "f":
        return np.array([int(x) for x in o], dtype=dtype)
    it = np.int64 if dtype == np.float64 else np.int32
    top = int(np.iinfo(it).max)
    return np.array([x if x >= 0 else x ^ top for x in o], dtype=it).view(dtype)


def _threshold(frame, expr, ci, op, text):
    """The exact meaning of `f.col <op> scalar` on column ci's stype, recovered from the reference's own evaluation of
    `expr` on probe frames: ("ge" | "le", smallest / largest passing element value), ("eq" | "ne", value),
    ("isna",) / ("notna",) / ("none",) -- or None when the predicate is not one the library evaluates."""
    if text == "None":
        return ("isna",) if op == "==" else ("notna",) if op == "!=" else None
    st = frame.stypes[ci].value
    if st == L.BOOL:
        return None
    dtype = ST2NP[st]
    if text in ("True", "False"):
        approx = 1.0 if text == "True" else 0.0
    elif _INT.match(text):
        approx = int(text)
    elif _FLOAT.match(text):
        approx = float(text)
    else:
        return None
    name = frame.names[ci]
    if dtype.kind == "f":
        with np.errstate(over="ignore"):
            approx = float(dtype.type(approx)) if dtype == np.float64 or abs(float(approx)) <= 3.4e38 else (np.inf if approx > 0 else -np.inf)

    def count(vals):
        """rows of a probe frame (same names and stypes as `frame`; column ci = vals) passing the predicate"""
        m = len(vals)
        parts = []
        for c in range(frame.ncols):
            stc = frame.stypes[c]
            if c == ci:
                parts.append(dt.Frame({frame.names[c]: vals}))
            elif stc.value in ST2NP:
                parts.append(dt.Frame({frame.names[c]: np.zeros(m, np.bool_ if stc.value == L.BOOL else ST2NP[stc.value])}))
            else:
                parts.append(dt.Frame({frame.names[c]: [None] * m}, stype=stc))
        P = dt.cbind(*parts) if len(parts) > 1 else parts[0]
        return dt.Frame.__getitem__(P, (expr, name)).nrows

    isf = dtype.kind == "f"
    if isf:
        lo_d, hi_d = (-np.inf, np.inf)
        dom = _ord(np.array([lo_d, hi_d], dtype))
    else:
        info = np.iinfo(dtype)
        dom = np.array([int(info.min) + 1, int(info.max)], dtype=object)        # the minimum is the NA sentinel
    if op in ("==", "!="):
        if isf:
            c0 = dtype.type(approx)
            if not np.isfinite(c0) and not np.isinf(c0):
                return None
            probe = np.array([np.nextafter(c0, dtype.type(-np.inf)), c0, np.nextafter(c0, dtype.type(np.inf))], dtype)
        else:
            if isinstance(approx, float):
                if approx != int(approx):
                    return ("none",) if op == "==" else ("all",)               # never equal / always unequal (NA included)
                approx = int(approx)
            if approx < dom[0] or approx > dom[1]:
                return ("none",) if op == "==" else ("all",)
            probe = np.array([v for v in (approx - 1, approx, approx + 1) if dom[0] <= v <= dom[1]], dtype)
            c0 = dtype.type(approx)
        hits = [count(probe[i:i + 1]) for i in range(len(probe))]
        want = [int((v == c0) == (op == "==")) for v in probe]
        return (("eq" if op == "==" else "ne"), c0.item()) if hits == want else None
    # orderings: a step function over the ordered domain; find the step
    up = op in (">", ">=")                                                        # passing values are the large ones
    if isf:
        c0 = dtype.type(approx)
        if np.isinf(c0):
            lo = hi = int(_ord(np.array([c0], dtype))[0])
        else:
            w = dtype.type(4e-6) + abs(c0) * dtype.type(1e-6 if dtype == np.float32 else 1e-15)
            with np.errstate(over="ignore"):
                lo, hi = _ord(np.array([c0 - w, c0 + w], dtype))
            lo, hi = int(lo) - 8, int(hi) + 8
    else:
        lo, hi = int(np.floor(approx)) - 2, int(np.ceil(approx)) + 2
    lo, hi = min(max(lo, int(dom[0])), int(dom[1])), min(max(hi, int(dom[0])), int(dom[1]))
    ends = _unord([lo, hi], dtype)
    f_lo, f_hi = count(ends[:1]), count(ends[1:])
    if f_lo == f_hi:                                       # no step inside the window: look at the whole domain
        lo, hi = int(dom[0]), int(dom[1])
        ends = _unord([lo, hi], dtype)
        f_lo, f_hi = count(ends[:1]), count(ends[1:])
        if f_lo == f_hi:
            return ("notna",) if f_lo else ("none",)
    if (f_hi == 1) != up:
        return None                                         # not monotone the way the operator says: leave it alone
    while hi - lo > 1:
        m = min(1024, hi - lo + 1)
        cand = sorted({lo + (hi - lo) * i // (m - 1) for i in range(m)})
        npass = count(_unord(cand, dtype))
        # passing candidates are the top npass (up) or the bottom npass (down)
        if up:
            lo, hi = cand[len(cand) - npass - 1], cand[len(cand) - npass]
        else:
            lo, hi = cand[npass - 1], cand[npass]
    t = _unord([hi if up else lo], dtype)[0]
    return ("ge" if up else "le", t.item())


_TH_CACHE = {}


def match_filter(frame, item):
    """DT[f.col <cmp> scalar, cols] -> (predicate column, ("ge"|"le"|"eq"|..., value), selected columns) or None"""
    if not (isinstance(item, tuple) and len(item) == 2):
        return None
    i, j = item
    if isinstance(i, (int, slice, list, tuple, str, type(None), np.ndarray)) or i is Ellipsis or isinstance(i, dt.Frame):
        return None
    m = _FILTER.match(repr(i))
    if not m:
        return None
    ci = _colspec(frame, "FExpr<%s>" % m.group(1))
    cols = _jcols(frame, j)
    if ci is None or cols is None or not cols or not _accel(frame, cols + [ci]) or frame.nrows == 0:
        return None
    # the probing costs ~3 ms of reference evaluations: remembered per predicate OBJECT (the repr cannot be the key -- it
    # prints six decimals, two different scalars may print alike) and per stype of the column it was probed on
    ck = (id(i), frame.stypes[ci].value, m.group(2), m.group(3))
    hit = _TH_CACHE.get(ck)
    if hit is not None and hit[0] is i:
        th = hit[1]
    else:
        try:
            th = _threshold(frame, i, ci, m.group(2), m.group(3))
        except Exception:
            return None
        if len(_TH_CACHE) >= 64:
            _TH_CACHE.pop(next(iter(_TH_CACHE)))
        _TH_CACHE[ck] = (i, th)              # holding the object keeps its id() from being reused
    return None if th is None else (ci, th, cols)


def run_filter(frame, ci, th, cols, ctx=None, _eager=False):
    """the passing rows of `cols`, materialised in one sweep next to the predicate column (dthip_filter_take)"""
    import ctypes as C
    ctx = _context(ctx)
    lib = ctx._lib
    n = frame.nrows
    kind = th[0]
    if kind in ("none", "all"):
        sel = slice(0, 0) if kind == "none" else slice(None)
        base = frame.to_frame() if isinstance(frame, DeviceFrame) else frame
        return Frame(dt.Frame.__getitem__(base, (sel, [frame.names[c] for c in cols])))
    st = frame.stypes[ci].value
    isf = st in (L.FLOAT32, L.FLOAT64)
    code = {"ge": L.GE, "le": L.LE, "eq": L.EQ, "ne": L.NE, "isna": L.ISNA, "notna": L.NOTNA}[kind]
    cf = float(th[1]) if len(th) > 1 and isf else 0.0
    cint = int(th[1]) if len(th) > 1 and not isf else 0
    lcols, mem = _columns(frame, [ci] + list(cols), ctx)
    pcol = lcols[0]
    if _lazy() and mem == L.DEVICE and options.defer_filter and not _eager:
        # lazy residency: the result is what it is in the reference -- a VIEW (a RowIndex over the parent's columns,
        # rowindex_array.cc:130-170: nothing is copied until something reads it).  `V[:, cols, by(key)]` on it runs as ONE
        # library call (dthip_filter_groupby_rows: the filter fused into the first sort level); anything else makes the view
        # materialise first (dthip_filter_take, below), exactly as before
        owners = [frame._cols[c] for c in [ci] + list(cols)] if isinstance(frame, DeviceFrame) else _resident(frame, [ci] + list(cols), ctx)
        return _PendingFilter(ctx, [frame.names[c] for c in cols], [frame.stypes[c] for c in cols], lcols, (code, cf, cint), n, owners)
    out = _Out(ctx, mem)
    for lo in range(0, len(cols), 8):                       # dthip_filter_take takes up to 8 columns per sweep
        part = cols[lo:lo + 8]
        carr = _carr(lcols[1 + lo:1 + lo + len(part)])
        bufs = [out.alloc(n, frame.stypes[c].value) for c in part]
        optr = (C.c_void_p * len(part))(*[b[0] for b in bufs])
        k = C.c_int64(0)
        L.check(lib.dthip_filter_take(ctx._h, C.byref(pcol), code, cf, cint, carr, len(part), n, mem, None, optr, C.byref(k)))
        for c, b in zip(part, bufs):
            out.add(b[1], frame.names[c], frame.stypes[c].value, nrows=k.value)
    return out.finish(frame, wrap=True)


def match_rows(frame, item):
    """DT[:, cols, by(keys)] with plain columns in j -> (key indices, column indices) or None"""
    if not (isinstance(item, tuple) and len(item) == 3 and isinstance(item[2], by) and _is_all(item[0])):
        return None
    keys = [_colindex(frame, c) for c in item[2].cols]
    if not keys or any(k is None for k in keys) or len(set(keys)) != len(keys) or not _uniform(item[2].cols):
        return None
    cols = _jcols(frame, item[1], exclude=keys)
    if cols is None or not _accel(frame, keys + cols) or frame.nrows == 0:
        return None
    return keys, cols


def _rows_call(frame, keys, desc, cols, na_pos, ctx):
    """dthip_groupby_rows (or, for na_position='remove', dthip_groupby + dthip_gather per column): `cols` in key order"""
    import ctypes as C
    lib = ctx._lib
    n = frame.nrows
    used = list(dict.fromkeys(list(keys) + list(cols)))
    flags = {k: (L.FLAG_DESCENDING if d else 0) for k, d in zip(keys, desc)}
    lcols, mem = _columns(frame, used, ctx)
    col = dict(zip(used, lcols))
    karr = _carr([L.Col(col[k].data, col[k].stype, flags[k]) for k in keys])
    out = _Out(ctx, mem)
    h = C.c_void_p()
    if na_pos == L.NA_REMOVE:
        L.check(lib.dthip_groupby(ctx._h, karr, len(keys), n, na_pos, mem, 1, C.byref(h)))
        keep = _ResultKeep(ctx, h)
        try:
            m = lib.dthip_result_nrows(h)
            if mem == L.HOST:
                ri = np.empty(m, np.int32)
                L.check(lib.dthip_result_copy_rowindex(ctx._h, h, ri.ctypes.data, L.HOST))
                ri_p = ri.ctypes.data
            else:
                ri_p = lib.dthip_result_rowindex(h)
            for c in cols:
                st = frame.stypes[c].value
                ptr, holder = out.alloc(m, st)
                if m:
                    L.check(lib.dthip_gather(ctx._h, C.byref(col[c]), C.c_void_p(ri_p), m, mem, C.c_void_p(ptr)))
                out.add(holder, frame.names[c], st)
        finally:
            keep.free()
        return out
    carr = _carr([col[c] for c in cols])
    L.check(lib.dthip_groupby_rows(ctx._h, karr, len(keys), carr, len(cols), n, na_pos, mem, 0, C.byref(h)))
    keep = _ResultKeep(ctx, h)
    lazy = _lazy() and mem == L.DEVICE
    try:
        for i, c in enumerate(cols):
            st = frame.stypes[c].value
            if lazy:
                out.borrowed(lib.dthip_result_col(h, i), n, st, keep, frame.names[c])
            else:
                a = np.empty(n, ST2NP[st])
                L.check(lib.dthip_result_copy_col(ctx._h, h, i, a.ctypes.data, L.HOST))
                out.add(a, frame.names[c], st)
    finally:
        if not lazy:
            keep.free()
            out.mem = L.HOST
    return out


def run_rows(frame, keys, cols, ctx=None):
    """DT[:, cols, by(keys)]: the by-columns, then `cols`, every row, in grouped order (evaluate_select,
    eval_context.cc:497-508): the columns ride through the sort on the device"""
    ctx = _context(ctx)
    allc = list(keys) + list(cols)
    return _rows_call(frame, keys, [False] * len(keys), allc, L.NA_FIRST, ctx).finish(frame, wrap=True)


def match_sort(frame, item):
    """DT[:, cols, sort(...)] -> (key indices, descending flags, na_pos, column indices) or None"""
    if not (isinstance(item, tuple) and len(item) == 3 and isinstance(item[2], sort) and _is_all(item[0])):
        return None
    srt = item[2]
    keys = [_colindex(frame, c) for c in srt.cols]
    if not keys or any(k is None for k in keys) or not _uniform(srt.cols):
        return None
    rev = srt.reverse
    desc = [bool(rev)] * len(keys) if not isinstance(rev, (list, tuple)) else [bool(x) for x in rev]
    if len(desc) != len(keys) or srt.na_position not in ("first", "last", "remove"):
        return None                                              # the reference raises its own error
    cols = _jcols(frame, item[1])
    if cols is None or not cols or not _accel(frame, keys + cols) or frame.nrows == 0:
        return None
    return keys, desc, {"first": L.NA_FIRST, "last": L.NA_LAST, "remove": L.NA_REMOVE}[srt.na_position], cols


def run_sort(frame, keys, desc, na_pos, cols, ctx=None):
    ctx = _context(ctx)
    return _rows_call(frame, keys, desc, cols, na_pos, ctx).finish(frame, wrap=True)


def _dict_j(item):
    """DT[i, {"name": expr, ...}, ...] -> (the same statement with j as a list, [names]): the reference evaluates the
    values like a list and names the j columns after the keys (by-columns first; duplicates mangled, names.cc:232-266)"""
    if isinstance(item, tuple) and len(item) >= 2 and isinstance(item[1], dict) and item[1] and \
            all(isinstance(k, str) for k in item[1]):
        return (item[0], list(item[1].values())) + tuple(item[2:]), list(item[1])
    return item, None


def _renamed(res, names, nhead):
    """the j columns of an accelerated result take the names of a dict-form j; None when the shape is not by-columns + j"""
    nc = res.ncols
    if nc != nhead + len(names):
        return None
    new = _mangled(list(res.names[:nhead]) + list(names))
    if isinstance(res, DeviceFrame):
        res._names = tuple(new)
        if res._frame is not None:
            res._frame.names = new
    else:
        res.names = new
    return res


def _route(frame, item):
    """the accelerated evaluation of frame[item], or NotImplemented"""
    item, names = _dict_j(item)
    res = NotImplemented
    plan = match(frame, item)
    if plan is not None:
        res = run(frame, *plan)
    else:
        for matcher, runner in ((match_filter, run_filter), (match_rows, run_rows), (match_sort, run_sort)):
            plan = matcher(frame, item)
            if plan is not None:
                res = runner(frame, *plan)
                break
    if res is NotImplemented or names is None:
        return res
    nhead = len(item[2].cols) if len(item) == 3 and isinstance(item[2], by) else 0
    res = _renamed(res, names, nhead)
    return NotImplemented if res is None else res


def _native(item):
    return tuple(x.native() if isinstance(x, (by, sort)) else x for x in item) if isinstance(item, tuple) else item


class DeviceFrame:
    """Result of an accelerated `DT[...]` under `options.residency = "lazy"`: the columns live in HBM (inside the
    dthip_result they came from, or in dthip_malloc buffers), the metadata answers like a Frame's, the next accelerated
    `DT[...]` runs on the device columns, and ANY other use downloads the columns once (`to_frame()`) and hands over to
    the real Frame -- whose resident cache then adopts the device buffers, so nothing is uploaded again.
    Not a `datatable.Frame` subclass on purpose: the base type's C-level buffer protocol cannot be intercepted from
    Python, and it must never export memory that has not been downloaded yet."""

    def __init__(self, ctx, names, stypes, nrows, cols, wrap):
        self._ctx, self._names, self._stypes, self._nrows, self._cols, self._wrap = ctx, tuple(names), tuple(stypes), int(nrows), list(cols), wrap
        self._frame = None

    names = property(lambda self: self._names)
    stypes = property(lambda self: self._stypes)
    nrows = property(lambda self: self._nrows)
    ncols = property(lambda self: len(self._names))
    shape = property(lambda self: (self._nrows, len(self._names)))
    is_resident = property(lambda self: True)

    def to_frame(self):
        import ctypes as C
        if self._frame is None:
            host = []
            for e, st in zip(self._cols, self._stypes):
                a = np.empty(self._nrows, ST2NP[st.value])
                if a.nbytes:
                    L.check(self._ctx._lib.dthip_memcpy_d2h(self._ctx._h, C.c_void_p(a.ctypes.data), C.c_void_p(e.ptr), a.nbytes))
                host.append(a)
            res = _finish(host, list(self._names), [i for i, st in enumerate(self._stypes) if st == dt.stype.bool8])
            if self._wrap:
                res = Frame(res)
                # the device copies stay valid for the new Frame: adopt them as its resident cache
                cache = {}
                for c, e in enumerate(self._cols):
                    hp = dt.internal.frame_column_data_r(res, c).value or 0
                    cache[c] = _DevColumn(e.ptr, e.nbytes, hp, self._nrows, self._stypes[c].value, e.keep)
                res.__dict__["_dthip_dev"] = cache
                res.__dict__["_dthip_ctx"] = self._ctx
            self._frame = res
        return self._frame

    def __getitem__(self, item):
        if self._frame is None:
            r = _route(self, item)
            if r is not NotImplemented:
                return r
        return self.to_frame()[item]

    def __getattr__(self, name):                      # anything else a Frame can do: on the downloaded Frame
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.to_frame(), name)

    def __array__(self, *a, **k):
        return np.asarray(self.to_frame().to_numpy(), *a, **k)

    def __len__(self):
        return self._nrows

    def __repr__(self):
        return "<DeviceFrame [%d rows x %d cols] in HBM%s>" % (self._nrows, len(self._names), "" if self._frame is None else " (downloaded)")


class _PendingFilter(DeviceFrame):
    """`V = DT[f.x <cmp> c, cols]` under residency "lazy", not evaluated yet: the parent's device columns (the buffers, not
    the Frame: a later change of the parent cannot reach them), the predicate, and the selection.  Names / stypes / ncols
    answer from metadata; `V[:, j, by(key)]` with plain columns in j evaluates filter + grouping in one library call;
    every other use (nrows, to_frame, another query, ...) runs the filter first and behaves like any DeviceFrame."""

    def __init__(self, ctx, names, stypes, lcols, pred, nrows_in, owners):
        DeviceFrame.__init__(self, ctx, names, stypes, -1, [], True)
        self._pending = (lcols, pred, nrows_in)          # lcols[0] = predicate column, lcols[1:] = the selected columns
        self._owners = owners                            # the _DevColumn objects behind lcols: they keep the buffers alive

    def _materialise(self):
        import ctypes as C
        pend = self.__dict__.get("_pending")
        if pend is None:
            return
        lcols, (code, cf, cint), n = pend
        ctx, lib = self._ctx, self._ctx._lib
        out = _Out(ctx, L.DEVICE)
        ncol = len(lcols) - 1
        for lo in range(0, ncol, 8):
            part = lcols[1 + lo:1 + lo + 8]
            bufs = [out.alloc(n, c.stype) for c in part]
            optr = (C.c_void_p * len(part))(*[b[0] for b in bufs])
            k = C.c_int64(0)
            L.check(lib.dthip_filter_take(ctx._h, C.byref(lcols[0]), code, cf, cint, _carr(part), len(part), n, L.DEVICE, None, optr, C.byref(k)))
            for i, (c, b) in enumerate(zip(part, bufs)):
                out.add(b[1], self._names[lo + i], c.stype, nrows=k.value)
        self._cols = out.cols
        self._nrows = out.cols[0].nrows if out.cols else 0
        self._pending = None

    nrows = property(lambda self: (self._materialise(), self._nrows)[1])
    shape = property(lambda self: (self.nrows, len(self._names)))

    def to_frame(self):
        self._materialise()
        return DeviceFrame.to_frame(self)

    def __len__(self):
        return self.nrows

    def __repr__(self):
        if self.__dict__.get("_pending") is not None:
            return "<DeviceFrame [pending row filter x %d cols] in HBM>" % len(self._names)
        return DeviceFrame.__repr__(self)

    def _fused_rows(self, item):
        """V[:, j, by(keys)] on the pending view -> one dthip_filter_groupby_rows call, or NotImplemented"""
        import ctypes as C
        item, names = _dict_j(item)
        if not (isinstance(item, tuple) and len(item) == 3 and isinstance(item[2], by) and _is_all(item[0])):
            return NotImplemented
        keys = [_colindex(self, c) for c in item[2].cols]
        if not keys or any(k is None for k in keys) or len(set(keys)) != len(keys) or not _uniform(item[2].cols):
            return NotImplemented
        cols = _jcols(self, item[1], exclude=keys)
        if cols is None or not all(self._stypes[c].value in _ACCEL_STYPES for c in keys + cols):
            return NotImplemented
        lcols, (code, cf, cint), n = self._pending
        ctx, lib = self._ctx, self._ctx._lib
        allc = list(keys) + list(cols)
        karr = _carr([lcols[1 + k] for k in keys])
        carr = _carr([lcols[1 + c] for c in allc])
        h = C.c_void_p()
        L.check(lib.dthip_filter_groupby_rows(ctx._h, C.byref(lcols[0]), code, cf, cint, karr, len(keys), carr, len(allc), n,
                                              L.NA_FIRST, L.DEVICE, 0, C.byref(h)))
        keep = _ResultKeep(ctx, h)
        m = lib.dthip_result_nrows(h)
        out = _Out(ctx, L.DEVICE)
        for i, c in enumerate(allc):
            out.borrowed(lib.dthip_result_col(h, i), m, self._stypes[c].value, keep, self._names[c])
        stats["fused_filter_rows"] += 1
        res = out.finish(self, wrap=True)
        if names is not None:
            res = _renamed(res, names, len(keys))
            if res is None:
                return NotImplemented
        return res

    def __getitem__(self, item):
        if self.__dict__.get("_pending") is not None and self._frame is None:
            r = self._fused_rows(item)
            if r is not NotImplemented:
                return r
            self._materialise()
        return DeviceFrame.__getitem__(self, item)


class Frame(dt.Frame):
    """datatable.Frame whose __getitem__ sends the group-by hot path to the GPU: aggregations, the row filter and the
    rows-in-grouped-order / sort forms.  Everything else is the reference's.  With `options.residency` "auto" / "lazy"
    the columns those queries touch stay in HBM between queries (see _Options); the reference's own contract for borrowed
    column pointers applies -- valid until the next mutating call (src/datatable/include/datatable.h:80-99,113-114) -- and
    every mutating Frame method below drops the device copies."""

    def __getitem__(self, item):
        r = _route(self, item)
        if r is not NotImplemented:
            return r
        if isinstance(item, tuple) and any(isinstance(x, dt.update) for x in item):
            self.release_device()                                # DT[:, update(...)] assigns in place
        return super().__getitem__(_native(item))

    def sort(self, *cols):
        """Frame.sort(cols): ascending, NA first (src/core/sort.cc:539-558) == DT[:, :, sort(cols)]"""
        plan = match_sort(self, (slice(None), slice(None), sort(*cols)))
        if plan is not None:
            return run_sort(self, *plan)
        return super().sort(*cols)

    # ---- residency ---------------------------------------------------------------------------------------------------
    def to_device(self, ctx=None):
        """upload every fixed-width column now (otherwise columns are uploaded by the first query that touches them)"""
        if options.residency == "off":
            raise ValueError("shim.options.residency is 'off'")
        cols = [c for c in range(self.ncols) if self.stypes[c].value in _ACCEL_STYPES]
        _resident(self, cols, _context(ctx))
        return self

    def release_device(self):
        """drop the device copies (they are dropped automatically by every mutating call)"""
        self.__dict__.pop("_dthip_dev", None)

    @property
    def is_resident(self):
        return bool(self.__dict__.get("_dthip_dev"))

    # ---- every mutating entry point of the reference's Frame drops the device copies ------------------------------------
    def __setitem__(self, key, value):
        self.release_device()
        return super().__setitem__(key, value)

    def __delitem__(self, key):
        self.release_device()
        return super().__delitem__(key)

    def cbind(self, *a, **k):
        self.release_device()
        return super().cbind(*a, **k)

    def rbind(self, *a, **k):
        self.release_device()
        return super().rbind(*a, **k)

    def replace(self, *a, **k):
        self.release_device()
        return super().replace(*a, **k)

    def materialize(self, *a, **k):
        self.release_device()
        return super().materialize(*a, **k)

    def _drop_then_set(prop):                                    # noqa: N805 -- builds the hooked property setters below
        base = getattr(dt.Frame, prop)

        def getter(self):
            return base.__get__(self)

        def setter(self, value):
            self.release_device()
            base.__set__(self, value)

        def deleter(self):
            self.release_device()
            base.__delete__(self)

        return property(getter, setter, deleter, base.__doc__)

    nrows = _drop_then_set("nrows")       # resizes the columns
    key = _drop_then_set("key")           # re-orders the rows
    names = _drop_then_set("names")
    del _drop_then_set
