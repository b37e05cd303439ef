"""Reference-side binding (seam S-py): what a datatable maintainer adds to route
the `DT[:, {sum|mean|min|max|count}(f.col) ..., by(cols)]` hot path to libdthip.so.

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
        m = _REDUCER.match(repr(expr))
        if not m:
            return None
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
    used = keys + [c for _, c in aggs if c is not None]
    if any(frame.stypes[c].value not in _ACCEL_STYPES for c in used):
        return None           # strings, dates, ...: the reference handles them
    return keys, aggs


def run(frame, keys, aggs, ctx=None):
    """evaluate the matched query through the C ABI (dthip_groupby_agg, host pointers)"""
    import ctypes as C
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
        cols, names = [], []
        for i, k in enumerate(keys):
            out = np.empty(ng, ST2NP[frame.stypes[k].value])
            L.check(lib.dthip_result_copy_key(ctx._h, h, i, out.ctypes.data, L.HOST))
            cols.append(out); names.append(frame.names[k])
        for a, (op, c) in enumerate(aggs):
            out = np.empty(ng, ST2NP[lib.dthip_result_agg_stype(h, a)])
            L.check(lib.dthip_result_copy_agg(ctx._h, h, a, out.ctypes.data, L.HOST))
            cols.append(out); names.append("count" if c is None else frame.names[c])
    finally:
        lib.dthip_result_free(ctx._h, h)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", dt.exceptions.DatatableWarning)     # duplicate names are mangled, as in the reference
        res = dt.Frame(cols, names=names)
    for i, k in enumerate(keys):                                            # bool8 keys travel as int8
        if frame.stypes[k] == dt.stype.bool8:
            res[:, i] = dt.Frame(res[:, i].to_numpy().astype(np.bool_))
    return res


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
