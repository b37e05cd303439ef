"""Host-side mirror of the reference's Python surface for the `DT[i, j, by()]` path.

Same names, argument meaning and error behaviour as h2oai/datatable for the forms the
hot path covers, so that the parity tests read like the reference's own
(tests/test-groups.py, tests/test-reduce.py, tests/ijby/test-sort.py):

    from datatable_amd.frame import Frame, f, by, sum, mean, min, max, count
    DT = Frame(k=[3, None, 1, 3], v=[1.5, 2.0, None, 4.0])
    DT[:, [sum(f.v), mean(f.v), count()], by(f.k)]      # fused groupby-aggregate on the GPU
    V = DT[f.v > 1.6, :]                                # boolean filter -> RowIndex view
    V[:, sum(f.v), by(f.k)]                             # view gather -> groupby
    DT[:, :, by(f.k)]  /  DT[:, f.i, by(f.k)]           # rows in grouped order
    DT.sort("k")  /  DT[:, :, sort(f.k)]

What is evaluated where: every i/j/by evaluation below is a call into libdthip.so through
`engine.Context` (C ABI in include/dthip.h).  Nothing here computes on the CPU except
result-Frame assembly (names, dtypes), which is also Python-side bookkeeping in the reference
(src/core/expr/eval_context.cc:497-508, src/core/frame/names.cc:232-266).  Anything outside
the path raises NotImplementedError -- in a deployment the shim of INTEGRATION.md forwards
those forms to the reference instead.

Storage convention = the reference's SentinelFw columns (src/core/column/sentinel_fw.cc):
numpy arrays with NA as INT*_MIN / NaN, bool8 as int8 with -128.
"""
import builtins

import numpy as np

from . import _lib as L
from .engine import NP2ST, ST2NP, default_context

__all__ = ["Frame", "f", "by", "sort", "sum", "mean", "min", "max", "count"]

_NA_INT = {1: np.iinfo(np.int8).min, 2: np.iinfo(np.int16).min, 4: np.iinfo(np.int32).min, 8: np.iinfo(np.int64).min}


# ---- f-expressions (src/core/expr/fexpr*.cc; only what the path needs) ---------------------

class FExpr:
    pass


class ColRef(FExpr):
    """f.name / f[i]  (src/core/expr/fexpr_column.cc)"""

    def __init__(self, ref):
        self.ref = ref

    def _cmp(self, op, other):
        if isinstance(other, FExpr):
            raise NotImplementedError("column-to-column comparisons are outside the accelerated path")
        return Filter(self, op, other)

    def __gt__(self, o): return self._cmp(">", o)
    def __ge__(self, o): return self._cmp(">=", o)
    def __lt__(self, o): return self._cmp("<", o)
    def __le__(self, o): return self._cmp("<=", o)
    def __eq__(self, o): return self._cmp("==", o)      # noqa: E704
    def __ne__(self, o): return self._cmp("!=", o)      # noqa: E704
    __hash__ = None

    def __repr__(self):
        return "FExpr<f.%s>" % self.ref if isinstance(self.ref, str) else "FExpr<f[%r]>" % (self.ref,)


class Filter(FExpr):
    """f.col <cmp> scalar  ->  boolean column -> RowIndex (rowindex_array.cc:130-170)"""

    def __init__(self, col, op, scalar):
        self.col, self.op, self.scalar = col, op, scalar


class Reducer(FExpr):
    """sum/mean/min/max/count(f.col), count()  (src/core/expr/fexpr_{sumprod,mean,minmax,count}.cc)"""

    def __init__(self, op, arg):
        self.op, self.arg = op, arg

    def __repr__(self):
        return "FExpr<%s(%s)>" % (self.op, "" if self.arg is None else repr(self.arg)[6:-1])


class _Namespace:
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return ColRef(name)

    def __getitem__(self, item):
        if isinstance(item, (str, int, np.integer)):
            return ColRef(item)
        raise NotImplementedError("f[%r]: only single-column selectors are on the accelerated path" % (item,))


f = _Namespace()


class by:
    """by(f.k, ...) / by("k", ...)  (src/core/expr/py_by.cc:71-78)"""

    def __init__(self, *cols):
        if len(cols) == 1 and isinstance(cols[0], (list, tuple)):
            cols = tuple(cols[0])
        self.cols = [c if isinstance(c, ColRef) else ColRef(c) for c in cols]


class sort:
    """sort(f.k, ..., reverse=False, na_position="first")  (src/core/expr/py_sort.cc:33-100)"""

    def __init__(self, *cols, reverse=False, na_position="first"):
        if len(cols) == 1 and isinstance(cols[0], (list, tuple)):
            cols = tuple(cols[0])
        self.cols = [c if isinstance(c, ColRef) else ColRef(c) for c in cols]
        self.reverse = [bool(reverse)] * len(self.cols) if not isinstance(reverse, (list, tuple)) else list(reverse)
        if na_position not in ("first", "last", "remove"):
            raise ValueError("na position value %s is not supported" % (na_position,))     # py_sort.cc's message
        self.na_last = na_position == "last"
        self.na_remove = na_position == "remove"


def _reducer(op):
    def fn(arg=None):
        if arg is None:
            if op != "count":
                raise TypeError("%s() requires a column expression" % op)
            return Reducer("count0", None)
        if isinstance(arg, str):
            arg = ColRef(arg)
        if not isinstance(arg, ColRef):
            raise NotImplementedError("%s() of a computed expression is outside the accelerated path" % op)
        return Reducer(op, arg)
    fn.__name__ = op
    return fn


sum = _reducer("sum")        # noqa: A001  (same names as datatable's)
mean = _reducer("mean")
min = _reducer("min")        # noqa: A001
max = _reducer("max")        # noqa: A001
count = _reducer("count")


# ---- Frame ---------------------------------------------------------------------------------

def _to_column(x):
    """python list / numpy array -> (array in datatable's sentinel storage)"""
    if isinstance(x, np.ndarray):
        a = x
        if a.dtype == np.bool_:
            return a.view(np.int8).copy(), L.BOOL
        if a.dtype in NP2ST:
            return np.ascontiguousarray(a), NP2ST[a.dtype]
        raise NotImplementedError("column dtype %s" % a.dtype)
    vals = list(x)
    nonnull = [v for v in vals if v is not None]
    if all(isinstance(v, (bool, np.bool_)) for v in nonnull) and nonnull:
        return np.array([-128 if v is None else int(v) for v in vals], np.int8), L.BOOL
    if all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in nonnull):
        big = any(abs(int(v)) > 2**31 - 1 for v in nonnull)
        dt = np.int64 if big else np.int32        # datatable's default int stype is int32
        na = np.iinfo(dt).min
        return np.array([na if v is None else v for v in vals], dt), NP2ST[np.dtype(dt)]
    return np.array([np.nan if v is None else float(v) for v in vals], np.float64), L.FLOAT64


def _mangle(names):
    """duplicate-name mangling of result frames: v, v -> v, v.0 (src/core/frame/names.cc:232-266)"""
    seen, out = set(), []
    for nm in names:
        if nm not in seen:
            seen.add(nm); out.append(nm); continue
        k = 0
        while "%s.%d" % (nm, k) in seen:
            k += 1
        new = "%s.%d" % (nm, k)
        seen.add(new); out.append(new)
    return out


class Frame:
    """Columnar frame over numpy buffers in datatable's storage convention.  A Frame may be a
    VIEW: `_ri` then holds an int32 RowIndex into the parent's buffers (column/view.cc:140-196)."""

    def __init__(self, _data=None, names=None, stypes=None, **cols):
        self._cols, self._stypes, self._names = [], [], []
        self._ri = None
        self._ctx = None
        src = []
        if isinstance(_data, Frame):
            self._cols, self._stypes, self._names, self._ri = list(_data._cols), list(_data._stypes), list(_data._names), _data._ri
            return
        if isinstance(_data, dict):
            src = list(_data.items())
        elif isinstance(_data, (list, tuple)) and _data and isinstance(_data[0], (list, tuple, np.ndarray)):
            src = [(names[i] if names else "C%d" % i, c) for i, c in enumerate(_data)]
        elif isinstance(_data, (list, tuple, np.ndarray)) and len(_data):
            src = [(names[0] if names else "C0", _data)]
        src += list(cols.items())
        n = None
        for i, (nm, c) in enumerate(src):
            a, st = _to_column(c)
            if stypes is not None:
                want = stypes[i] if isinstance(stypes, (list, tuple)) else stypes.get(nm)
                if want is not None and want != st:
                    a = self._cast(a, st, want); st = want
            if n is None:
                n = len(a)
            elif len(a) != n:
                raise ValueError("column %r has %d rows, expected %d" % (nm, len(a), n))
            self._cols.append(a); self._stypes.append(st); self._names.append(str(nm))

    @staticmethod
    def _cast(a, st, want):
        dt = ST2NP[want]
        if st in (L.FLOAT32, L.FLOAT64):
            if want in (L.FLOAT32, L.FLOAT64):
                return a.astype(dt)
            out = np.where(np.isnan(a), _NA_INT[dt.itemsize], np.nan_to_num(a)).astype(dt)
            return out
        na = a == _NA_INT[a.dtype.itemsize]
        if want in (L.FLOAT32, L.FLOAT64):
            out = a.astype(dt); out[na] = np.nan
            return out
        out = a.astype(dt); out[na] = _NA_INT[dt.itemsize]
        return out

    # -- properties ---------------------------------------------------------------------
    @property
    def nrows(self):
        if self._ri is not None:
            return len(self._ri)
        return len(self._cols[0]) if self._cols else 0

    @property
    def ncols(self): return len(self._cols)

    @property
    def shape(self): return (self.nrows, self.ncols)

    @property
    def names(self): return tuple(self._names)

    @property
    def stypes(self): return tuple(self._stypes)

    def __len__(self): return self.ncols

    def _context(self):
        return default_context()

    def _index(self, ref):
        if isinstance(ref, ColRef):
            ref = ref.ref
        if isinstance(ref, (int, np.integer)):
            i = int(ref)
            if i < 0:
                i += self.ncols
            if not 0 <= i < self.ncols:
                raise ValueError("Column index %d is invalid for a Frame with %d columns" % (int(ref), self.ncols))
            return i
        if ref not in self._names:
            raise KeyError("Column %s does not exist in the Frame" % (ref,))
        return self._names.index(ref)

    def _materialized(self, i):
        """column i as a contiguous buffer (the view gather is ColumnImpl::_materialize_fw)"""
        if self._ri is None:
            return self._cols[i]
        return self._context().gather(self._cols[i], self._ri, stype=self._stypes[i])

    def materialize(self):
        if self._ri is not None:
            self._cols = [self._materialized(i) for i in range(self.ncols)]
            self._ri = None
        return self

    def to_numpy_columns(self):
        return [self._materialized(i) for i in range(self.ncols)]

    def to_dict(self):
        return {nm: self._pylist(i) for i, nm in enumerate(self._names)}

    def to_list(self):
        return [self._pylist(i) for i in range(self.ncols)]

    def _pylist(self, i):
        a, st = self._materialized(i), self._stypes[i]
        if st in (L.FLOAT32, L.FLOAT64):
            return [None if v != v else float(v) for v in a]
        na = _NA_INT[a.dtype.itemsize]
        if st == L.BOOL:
            return [None if v == na else bool(v) for v in a]
        return [None if v == na else int(v) for v in a]

    def __repr__(self):
        return "<Frame [%d rows x %d cols] %s>" % (self.nrows, self.ncols, ", ".join(self._names))

    @staticmethod
    def _from_columns(cols, stypes, names):
        fr = Frame()
        fr._cols, fr._stypes, fr._names = list(cols), list(stypes), _mangle(list(names))
        return fr

    # -- DT[i, j, by] -------------------------------------------------------------------
    def __getitem__(self, item):
        if not isinstance(item, tuple):
            raise NotImplementedError("single-selector DT[x] is outside the accelerated path")
        i = item[0]
        j = item[1] if len(item) > 1 else slice(None)
        rest = item[2:]
        byx = [r for r in rest if isinstance(r, by)]
        srt = [r for r in rest if isinstance(r, sort)]
        if len(rest) != len(byx) + len(srt) or len(byx) > 1 or len(srt) > 1:
            raise NotImplementedError("only by() and sort() modifiers are on the accelerated path")
        all_rows = i is None or i is Ellipsis or (isinstance(i, slice) and i == slice(None))
        if isinstance(i, Filter):
            if byx or srt:
                # the reference cannot do this either: src/core/expr/fexpr_func.cc:61-73
                raise NotImplementedError("FExpr_Func::evaluate_iby() not implemented yet")
            return self._filter(i)._select(j)
        if not all_rows:
            raise NotImplementedError("row selector %r is outside the accelerated path" % (i,))
        if byx:
            if srt:
                raise NotImplementedError("by() together with sort() is outside the accelerated path")
            return self._groupby(j, byx[0])
        if srt:
            return self._sorted(srt[0])._select(j)
        return self._select(j)

    def _select(self, j):
        if j is None or j is Ellipsis or (isinstance(j, slice) and j == slice(None)):
            return self
        refs = j if isinstance(j, (list, tuple)) else [j]
        if any(isinstance(r, Reducer) for r in refs):
            return self._groupby(refs, None)
        idx = [self._index(r) for r in refs]
        fr = Frame._from_columns([self._cols[k] for k in idx], [self._stypes[k] for k in idx],
                                 [self._names[k] for k in idx])
        fr._ri = self._ri
        return fr

    def _filter(self, flt):
        ci = self._index(flt.col)
        ctx = self._context()
        ri = ctx.filter_cmp(self._materialized(ci), flt.op, flt.scalar, stype=self._stypes[ci])
        fr = Frame(self)
        if self._ri is not None:
            # composition ab*bc: res[i] = ab[bc[i]] (rowindex_array.cc:258-269) = a gather of the RowIndex
            ri = ctx.gather(self._ri, ri)
        fr._ri = ri
        return fr

    def _sorted(self, s):
        ctx = self._context()
        idx = [self._index(c) for c in s.cols]
        keys = [self._materialized(k) for k in idx]
        r = ctx.groupby(keys, stypes=[self._stypes[k] for k in idx], desc=s.reverse, na_last=s.na_last,
                        na_remove=s.na_remove)
        ri = r.rowindex()
        r.free()
        fr = Frame(self)
        fr._ri = ri if self._ri is None else ctx.gather(self._ri, ri)
        return fr

    def sort(self, *cols):
        """Frame.sort(cols): ascending, NA first (src/core/sort.cc:539-558)"""
        return self._sorted(sort(*cols))

    def _groupby(self, j, byx):
        ctx = self._context()
        if self.nrows > 2**31 - 1:
            raise ValueError("nrows > 2**31-1: RowIndex and group offsets are int32")
        kidx = [self._index(c) for c in byx.cols] if byx is not None else []
        keys = [self._materialized(k) for k in kidx]
        kst = [self._stypes[k] for k in kidx]
        sel_all = j is None or j is Ellipsis or (isinstance(j, slice) and j == slice(None))
        refs = [] if sel_all else (list(j) if isinstance(j, (list, tuple)) else [j])
        if isinstance(j, dict):
            raise NotImplementedError("dict selectors are outside the accelerated path")
        reducers = [r for r in refs if isinstance(r, Reducer)]
        if reducers and len(reducers) != len(refs):
            raise NotImplementedError("mixing reducers and plain columns in j is outside the accelerated path")
        if not kidx:
            # DT[:, sum(f.v)] without by(): one group over all rows
            keys, kst = [np.zeros(self.nrows, np.int8)], [L.INT8]
        if reducers:
            vidx, aggs = [], []
            for r in reducers:
                if r.op == "count0":
                    aggs.append(("count0", None)); continue
                ci = self._index(r.arg)
                if ci not in vidx:
                    vidx.append(ci)
                aggs.append((r.op, vidx.index(ci)))
            vals = [self._materialized(c) for c in vidx]
            res = ctx.groupby_agg(keys, vals, aggs, key_stypes=kst, value_stypes=[self._stypes[c] for c in vidx])
            cols = [res.key(k) for k in range(len(kidx))]
            sts = list(kst[:len(kidx)])
            names = [self._names[k] for k in kidx]
            for a, r in enumerate(reducers):
                cols.append(res.agg(a)); sts.append(res.agg_stype(a))
                names.append("count" if r.op == "count0" else self._names[self._index(r.arg)])
            res.free()
            return Frame._from_columns(cols, sts, names)
        # rows in grouped order: by-columns first, then the selected (or all remaining) columns
        res = ctx.groupby(keys, stypes=kst)
        ri = res.rowindex()
        res.free()
        rest = [self._index(r) for r in refs] if refs else [c for c in range(self.ncols) if c not in kidx]
        order = kidx + rest
        fr = Frame._from_columns([self._cols[c] for c in order], [self._stypes[c] for c in order],
                                 [self._names[c] for c in order])
        fr._ri = ri if self._ri is None else ctx.gather(self._ri, ri)
        return fr


def _unused():   # keep flake-style tools quiet about the shadowed builtins being intentional
    return builtins.sum, builtins.min, builtins.max
