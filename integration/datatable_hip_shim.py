"""Reference-side binding (seam S-py): what a datatable maintainer adds to route
the `DT[:, {sum|mean|min|max|count}(f.col) ..., by(cols)]` hot path to libdthip.so -- and, through
the S-red entry points, the other reducers and group-wise operators that share its Groupby
(first/last/sd/median/nunique, cov/corr, cumsum/cumprod/cummin/cummax/cumcount/ngroup).

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
    try:
        th = _threshold(frame, i, ci, m.group(2), m.group(3))
    except Exception:
        return None
    return None if th is None else (ci, th, cols)


def _wrap(frame, cols, names, bool_cols):
    return Frame(_finish(frame, [], cols, names, bool_cols))


def run_filter(frame, ci, th, cols, ctx=None):
    """the passing rows of `cols`, materialised in one sweep next to the predicate column (dthip_filter_take)"""
    import ctypes as C
    ctx = ctx or default_context()
    lib = ctx._lib
    n = frame.nrows
    kind = th[0]
    if kind in ("none", "all"):
        sel = slice(0, 0) if kind == "none" else slice(None)
        return Frame(dt.Frame.__getitem__(frame, (sel, [frame.names[c] for c in cols])))
    st = frame.stypes[ci].value
    isf = st in (L.FLOAT32, L.FLOAT64)
    code = {"ge": L.GE, "le": L.LE, "eq": L.EQ, "ne": L.NE, "isna": L.ISNA, "notna": L.NOTNA}[kind]
    cf = float(th[1]) if len(th) > 1 and isf else 0.0
    cint = int(th[1]) if len(th) > 1 and not isf else 0
    pcol = _col(frame, ci)
    outs, names, bool_cols = [], [], []
    npass = None
    for lo in range(0, len(cols), 8):                       # dthip_filter_take takes up to 8 columns per sweep
        part = cols[lo:lo + 8]
        carr = (L.Col * len(part))(*[_col(frame, c) for c in part])
        bufs = [np.empty(n, ST2NP[frame.stypes[c].value]) for c in part]
        optr = (C.c_void_p * len(part))(*[b.ctypes.data for b in bufs])
        k = C.c_int64(0)
        L.check(lib.dthip_filter_take(ctx._h, C.byref(pcol), code, cf, cint, carr, len(part), n, L.HOST, None, optr, C.byref(k)))
        npass = k.value
        for c, b in zip(part, bufs):
            if frame.stypes[c].value == L.BOOL:
                bool_cols.append(len(outs))
            outs.append(b[:npass]); names.append(frame.names[c])
    return _wrap(frame, outs, names, bool_cols)


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
    karr = (L.Col * len(keys))(*[L.Col(dt.internal.frame_column_data_r(frame, k).value, frame.stypes[k].value,
                                       L.FLAG_DESCENDING if d else 0) for k, d in zip(keys, desc)])
    h = C.c_void_p()
    outs = []
    if na_pos == L.NA_REMOVE:
        L.check(lib.dthip_groupby(ctx._h, karr, len(keys), n, na_pos, L.HOST, 1, C.byref(h)))
        try:
            m = lib.dthip_result_nrows(h)
            ri = np.empty(m, np.int32)
            L.check(lib.dthip_result_copy_rowindex(ctx._h, h, ri.ctypes.data, L.HOST))
        finally:
            lib.dthip_result_free(ctx._h, h)
        for c in cols:
            out = np.empty(m, ST2NP[frame.stypes[c].value])
            cc = _col(frame, c)
            if m:
                L.check(lib.dthip_gather(ctx._h, C.byref(cc), ri.ctypes.data, m, L.HOST, out.ctypes.data))
            outs.append(out)
        return outs
    carr = (L.Col * max(len(cols), 1))(*[_col(frame, c) for c in cols])
    L.check(lib.dthip_groupby_rows(ctx._h, karr, len(keys), carr, len(cols), n, na_pos, L.HOST, 0, C.byref(h)))
    try:
        for i, c in enumerate(cols):
            out = np.empty(n, ST2NP[frame.stypes[c].value])
            L.check(lib.dthip_result_copy_col(ctx._h, h, i, out.ctypes.data, L.HOST))
            outs.append(out)
    finally:
        lib.dthip_result_free(ctx._h, h)
    return outs


def run_rows(frame, keys, cols, ctx=None):
    """DT[:, cols, by(keys)]: the by-columns, then `cols`, every row, in grouped order (evaluate_select,
    eval_context.cc:497-508): the columns ride through the sort on the device"""
    ctx = ctx or default_context()
    allc = list(keys) + list(cols)
    outs = _rows_call(frame, keys, [False] * len(keys), allc, L.NA_FIRST, ctx)
    bools = [i for i, c in enumerate(allc) if frame.stypes[c].value == L.BOOL]
    return _wrap(frame, outs, [frame.names[c] for c in allc], bools)


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
    ctx = ctx or default_context()
    outs = _rows_call(frame, keys, desc, cols, na_pos, ctx)
    bools = [i for i, c in enumerate(cols) if frame.stypes[c].value == L.BOOL]
    return _wrap(frame, outs, [frame.names[c] for c in cols], bools)


class Frame(dt.Frame):
    """datatable.Frame whose __getitem__ sends the group-by hot path to the GPU: aggregations, the row filter and the
    rows-in-grouped-order / sort forms.  Everything else is the reference's."""

    def __getitem__(self, item):
        plan = match(self, item)
        if plan is not None:
            return run(self, *plan)
        for matcher, runner in ((match_filter, run_filter), (match_rows, run_rows), (match_sort, run_sort)):
            plan = matcher(self, item)
            if plan is not None:
                return runner(self, *plan)
        if isinstance(item, tuple):
            item = tuple(x.native() if isinstance(x, (by, sort)) else x for x in item)
        return super().__getitem__(item)

    def sort(self, *cols):
        """Frame.sort(cols): ascending, NA first (src/core/sort.cc:539-558) == DT[:, :, sort(cols)]"""
        plan = match_sort(self, (slice(None), slice(None), sort(*cols)))
        if plan is not None:
            return run_sort(self, *plan)
        return super().sort(*cols)
