"""Reference-side binding (seam S-py): what a datatable maintainer adds to route
the `DT[:, {sum|mean|min|max|count}(f.col) ..., by(cols)]` hot path to libdthip.so -- and, through
the S-red entry points, the other reducers and group-wise operators that share its Groupby
(first/last/sd/median/nunique, cov/corr, cumsum/cumprod/cummin/cummax/cumcount/ngroup).

It runs INSIDE the reference's Python process (needs `import datatable`), touches no reference
source, and forwards every other form of `DT[...]` to the reference unchanged:

    import datatable as dt
    from datatable import f, sum, mean, count
    from integration.datatable_hip_shim import Frame, by        # the shim's Frame and by()
    DT = Frame(dt.fread("data.jay"))                             # or Frame(k=..., v=...)
    DT[:, [sum(f.v), count()], by(f.k)]                          # -> libdthip (MI355X)
    DT[f.v > 0, :]                                               # -> the reference, as before

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

`by()` must be the shim's: the reference's `datatable.by` object is opaque from Python
(src/core/expr/py_by.cc:71-78), so the shim cannot read the grouping columns out of it.
"""
import re
import warnings

import numpy as np

import datatable as dt

from datatable_amd import _lib as L
from datatable_amd.engine import ST2NP, default_context

# min()/max() print their argument as a one-element list: FExpr<min([f.v])>
_REDUCER = re.compile(r"^FExpr<(sum|mean|min|max|count)\(\[?(?:f\.(\w+)|f\['([^']+)'\]|f\[(\d+)\])?\]?\)>$")
_COLUMN = re.compile(r"^FExpr<(?:f\.(\w+)|f\['([^']+)'\]|f\[(\d+)\])>$")
_CREF = r"(?:f\.\w+|f\['[^']+'\]|f\[\d+\])"
# old-style expression nodes print as  Expr:stdev(FExpr<f.v>; )
_REDUCER_OLD = re.compile(r"^Expr:(stdev|median|nunique|first|last)\((FExpr<%s>); \)$" % _CREF)
_REDUCER2 = re.compile(r"^Expr:(cov|corr)\((FExpr<%s>), (FExpr<%s>); \)$" % (_CREF, _CREF))
_CUMULATIVE = re.compile(r"^FExpr<(cumsum|cumprod|cummin|cummax)\((%s), reverse=(True|False)\)>$" % _CREF)
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
    if not keys or any(k is None for k in keys):
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


def _col(frame, c):
    return L.Col(dt.internal.frame_column_data_r(frame, c).value, frame.stypes[c].value, 0)


def _finish(frame, keys, cols, names, bool_cols=()):
    """numpy result buffers -> the result Frame.  bool8 columns (keys, and min/max/first/last/cummin/cummax of a
    bool8 column: the reference keeps the stype, fexpr_minmax.cc:47-68) travel as int8 with NA = -128 and are
    cast back here (int8 -> bool8 keeps NA)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", dt.exceptions.DatatableWarning)     # duplicate names are mangled, as in the reference
        res = dt.Frame(cols, names=names)                                   # "" -> auto-named C<k> by the reference
    bools = set(bool_cols) | {i for i, k in enumerate(keys) if frame.stypes[k] == dt.stype.bool8}
    for i in sorted(bools):
        res[:, i] = dt.as_type(dt.f[i], dt.bool8)
    return res


def run_sred(frame, keys, aggs, ctx=None):
    """the S-red route: dthip_groupby once, then dthip_reduce / dthip_reduce2 (one row per group) or
    dthip_cumulate (one row per input row, grouped order) per j item -- host pointers throughout"""
    import ctypes as C
    from datatable_amd.engine import OPS, OPS2, CUMOPS
    ctx = ctx or default_context()
    lib = ctx._lib
    n = frame.nrows
    karr = (L.Col * len(keys))(*[_col(frame, k) for k in keys])
    h = C.c_void_p()
    L.check(lib.dthip_groupby(ctx._h, karr, len(keys), n, L.NA_FIRST, L.HOST, 1, C.byref(h)))
    try:
        ng = lib.dthip_result_ngroups(h)
        ri, off = np.empty(n, np.int32), np.empty(ng + 1, np.int32)
        L.check(lib.dthip_result_copy_rowindex(ctx._h, h, ri.ctypes.data, L.HOST))
        L.check(lib.dthip_result_copy_offsets(ctx._h, h, off.ctypes.data, L.HOST))
    finally:
        lib.dthip_result_free(ctx._h, h)
    rowwise = bool(aggs) and len(aggs[0]) == 3
    cols, names, bool_cols = [], [], []
    sel = ri if rowwise else ri[off[:-1]]              # the by-columns: every row / first row of each group
    for k in keys:
        out = np.empty(len(sel), ST2NP[frame.stypes[k].value])
        kc = _col(frame, k)
        L.check(lib.dthip_gather(ctx._h, C.byref(kc), sel.ctypes.data, len(sel), L.HOST, out.ctypes.data))
        cols.append(out); names.append(frame.names[k])
    for a in aggs:
        op, c = a[0], a[1]
        if rowwise:
            st = lib.dthip_cumulate_out_stype(CUMOPS[op], L.INT64 if c is None else frame.stypes[c].value)
            out = np.empty(n, ST2NP[st])
            if st == L.BOOL:
                bool_cols.append(len(cols))
            vc = _col(frame, c) if c is not None else None
            if n:
                L.check(lib.dthip_cumulate(ctx._h, CUMOPS[op], C.byref(vc) if vc is not None else None,
                                           ri.ctypes.data if c is not None else None, off.ctypes.data, ng, n,
                                           1 if a[2] else 0, L.HOST, out.ctypes.data))
            names.append("" if c is None else frame.names[c])
        elif isinstance(c, tuple):
            ca, cb = _col(frame, c[0]), _col(frame, c[1])
            out = np.empty(ng, ST2NP[lib.dthip_reduce2_out_stype(ca.stype, cb.stype)])
            if ng:
                L.check(lib.dthip_reduce2(ctx._h, OPS2[op], C.byref(ca), C.byref(cb), ri.ctypes.data, off.ctypes.data, ng, n,
                                          L.HOST, out.ctypes.data))
            names.append("")
        elif c is None:
            out = np.empty(ng, np.int64)
            if ng:
                L.check(lib.dthip_reduce(ctx._h, L.COUNT0, None, None, off.ctypes.data, ng, n, L.HOST, out.ctypes.data))
            names.append("count")
        else:
            vc = _col(frame, c)
            st = lib.dthip_reduce_out_stype(OPS[op], vc.stype)
            out = np.empty(ng, ST2NP[st])
            if st == L.BOOL:
                bool_cols.append(len(cols))
            if ng:
                L.check(lib.dthip_reduce(ctx._h, OPS[op], C.byref(vc), ri.ctypes.data, off.ctypes.data, ng, n, L.HOST,
                                         out.ctypes.data))
            names.append(frame.names[c])
        cols.append(out)
    return _finish(frame, keys, cols, names, bool_cols)


def run(frame, keys, aggs, ctx=None):
    """evaluate the matched query through the C ABI (dthip_groupby_agg, host pointers)"""
    import ctypes as C
    if not all(len(a) == 2 and a[0] in _FUSED for a in aggs):
        return run_sred(frame, keys, aggs, ctx)
    ctx = ctx or default_context()
    lib = ctx._lib
    vcols = sorted({c for _, c in aggs if c is not None})
    karr = (L.Col * len(keys))(*[L.Col(dt.internal.frame_column_data_r(frame, k).value, frame.stypes[k].value, 0) for k in keys])
    varr = (L.Col * max(len(vcols), 1))(*[L.Col(dt.internal.frame_column_data_r(frame, c).value, frame.stypes[c].value, 0) for c in vcols])
    aarr = (L.Agg * len(aggs))(*[L.Agg({"sum": L.SUM, "mean": L.MEAN, "min": L.MIN, "max": L.MAX, "count": L.COUNT,
                                        "count0": L.COUNT0}[op], -1 if c is None else vcols.index(c)) for op, c in aggs])
    h = C.c_void_p()
    L.check(lib.dthip_groupby_agg(ctx._h, karr, len(keys), varr, len(vcols), aarr, len(aggs), frame.nrows,
                                  L.NA_FIRST, L.HOST, C.byref(h)))
    try:
        ng = lib.dthip_result_ngroups(h)
        cols, names, bool_cols = [], [], []
        for i, k in enumerate(keys):
            out = np.empty(ng, ST2NP[frame.stypes[k].value])
            L.check(lib.dthip_result_copy_key(ctx._h, h, i, out.ctypes.data, L.HOST))
            cols.append(out); names.append(frame.names[k])
        for a, (op, c) in enumerate(aggs):
            st = lib.dthip_result_agg_stype(h, a)
            out = np.empty(ng, ST2NP[st])
            if st == L.BOOL:
                bool_cols.append(len(cols))
            L.check(lib.dthip_result_copy_agg(ctx._h, h, a, out.ctypes.data, L.HOST))
            cols.append(out); names.append("count" if c is None else frame.names[c])
    finally:
        lib.dthip_result_free(ctx._h, h)
    return _finish(frame, keys, cols, names, bool_cols)


class Frame(dt.Frame):
    """datatable.Frame whose __getitem__ sends the groupby-aggregate hot path to the GPU.
    Everything else -- and every Frame it returns -- is the reference's."""

    def __getitem__(self, item):
        plan = match(self, item)
        if plan is not None:
            return run(self, *plan)
        if isinstance(item, tuple):
            item = tuple(x.native() if isinstance(x, by) else x for x in item)
        return super().__getitem__(item)
