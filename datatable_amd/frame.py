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
from .engine import CUMOPS as L_CUMOPS, NP2ST, OPS as L_OPS, ST2NP, default_context

__all__ = ["Frame", "f", "by", "sort", "sum", "mean", "min", "max", "count", "first", "last",
           "sd", "median", "nunique", "cov", "corr", "cumsum", "cumprod", "cummin", "cummax", "cumcount", "ngroup", "fillna",
           "unique", "union", "intersect", "setdiff", "symdiff", "join"]

_NA_INT = {1: np.iinfo(np.int8).min, 2: np.iinfo(np.int16).min, 4: np.iinfo(np.int32).min, 8: np.iinfo(np.int64).min}


# ---- f-expressions (src/core/expr/fexpr*.cc; only what the path needs) ---------------------

class FExpr:
    pass


class ColRef(FExpr):
    """f.name / f[i]  (src/core/expr/fexpr_column.cc).  -f.name inside by()/sort() = descending."""

    def __init__(self, ref, desc=False):
        self.ref = ref
        self.desc = desc

    def __neg__(self):
        return ColRef(self.ref, not self.desc)

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


class AllCols(FExpr):
    """f[:] -- every column that is not a by() column"""

    def __repr__(self):
        return "FExpr<f[:]>"


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


class Reducer2(FExpr):
    """cov(f.a, f.b) / corr(f.a, f.b)  (src/core/expr/head_reduce_binary.cc:226-270); unnamed in the result"""

    def __init__(self, op, a, b):
        self.op, self.a, self.b = op, a, b

    def __repr__(self):
        return "FExpr<%s(%s, %s)>" % (self.op, repr(self.a)[6:-1], repr(self.b)[6:-1])


class Cumulative(FExpr):
    """cumsum / cumprod / cummin / cummax / fillna(f.col, reverse=False), cumcount(reverse) / ngroup(reverse)
    (src/core/expr/fexpr_cumsumprod.cc, fexpr_cumminmax.cc, fexpr_cumcountngroup.cc, fexpr_fillna.cc): one value per row,
    rows in grouped order"""

    def __init__(self, op, arg, reverse=False):
        self.op, self.arg, self.reverse = op, arg, bool(reverse)

    def __repr__(self):
        return "FExpr<%s(%sreverse=%s)>" % (self.op, "" if self.arg is None else repr(self.arg)[6:-1] + ", ", self.reverse)


class _Namespace:
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return ColRef(name)

    def __getitem__(self, item):
        if isinstance(item, (str, int, np.integer)):
            return ColRef(item)
        if isinstance(item, slice) and item == slice(None):
            return AllCols()
        raise NotImplementedError("f[%r]: only single-column selectors are on the accelerated path" % (item,))


f = _Namespace()


class by:
    """by(f.k, ...) / by("k", ...)  (src/core/expr/py_by.cc:71-78)"""

    def __init__(self, *cols):
        if len(cols) == 1 and isinstance(cols[0], (list, tuple)):
            cols = tuple(cols[0])
        self.cols = [c if isinstance(c, ColRef) else ColRef(c) for c in cols]
        if not self.cols:
            raise ValueError("by() needs at least one column")


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


class join:
    """join(J): natural left join on J's key columns (src/core/expr/py_join.cc:40-70)"""

    def __init__(self, frame):
        if not isinstance(frame, Frame):
            raise TypeError("The argument to join() must be a Frame")
        if not frame.key:
            raise ValueError("The join frame is not keyed")          # py_join.cc:60-62
        self.frame = frame


def _reducer(op):
    def fn(arg=None):
        if arg is None:
            if op != "count":
                raise TypeError("%s() requires a column expression" % op)
            return Reducer("count0", None)
        if isinstance(arg, str):
            arg = ColRef(arg)
        if not isinstance(arg, (ColRef, AllCols)):
            raise NotImplementedError("%s() of a computed expression is outside the accelerated path" % op)
        return Reducer(op, arg)
    fn.__name__ = op
    return fn


sum = _reducer("sum")        # noqa: A001  (same names as datatable's)
mean = _reducer("mean")
min = _reducer("min")        # noqa: A001
max = _reducer("max")        # noqa: A001
count = _reducer("count")
first = _reducer("first")    # src/core/expr/head_reduce_unary.cc:116-190
last = _reducer("last")
sd = _reducer("sd")          # src/core/expr/head_reduce_unary.cc:194-243
median = _reducer("median")  # :424-510
nunique = _reducer("nunique")  # :377-417


def _colarg(arg, what):
    if isinstance(arg, str):
        arg = ColRef(arg)
    if not isinstance(arg, (ColRef, AllCols)):
        raise NotImplementedError("%s of a computed expression is outside the accelerated path" % what)
    return arg


def cov(a, b):
    return Reducer2("cov", _colarg(a, "cov()"), _colarg(b, "cov()"))


def corr(a, b):
    return Reducer2("corr", _colarg(a, "corr()"), _colarg(b, "corr()"))


def _cumulative(op):
    def fn(arg=None, reverse=False):
        if arg is None:
            raise TypeError("Function `datatable.%s()` requires exactly 1 positional argument, but none were given" % op)
        return Cumulative(op, _colarg(arg, op + "()"), reverse)
    fn.__name__ = op
    return fn


cumsum = _cumulative("cumsum")
cumprod = _cumulative("cumprod")
cummin = _cumulative("cummin")
cummax = _cumulative("cummax")


def fillna(cols=None, value=None, reverse=False):
    """fillna(f.col, reverse=False): NAs take the previous (reverse: the next) valid value of their group
    (FExpr_FillNA, src/core/expr/fexpr_fillna.cc:85-117).  Filling with a `value` is an ifelse over the column, not a
    group-wise operator: outside the accelerated path"""
    if value is not None:
        raise NotImplementedError("fillna(value=...) is outside the accelerated path")
    if cols is None:
        raise TypeError("Function `datatable.fillna()` requires exactly 1 positional argument, but none were given")
    return Cumulative("fillna", _colarg(cols, "fillna()"), reverse)


def cumcount(reverse=False):
    return Cumulative("cumcount", None, reverse)


def ngroup(reverse=False):
    return Cumulative("ngroup", None, reverse)


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
    """duplicate-name mangling of result frames: v, v -> v, v.0 and v1, v1 -> v1, v2 (src/core/frame/names.cc:455-510); unnamed
    columns ("") become C<k>, counting on from the largest C<num> already present (names.cc:572-607)"""
    if "" in names:
        nxt = 0
        for nm in names:
            if len(nm) > 1 and nm[0] == "C" and nm[1:].isdigit():
                nxt = builtins.max(nxt, int(nm[1:]) + 1)
        filled = []
        for nm in names:
            if nm == "":
                nm = "C%d" % nxt
                nxt += 1
            filled.append(nm)
        names = filled
    seen, out, stems = set(), [], {}
    for nm in names:
        if nm in seen:
            # _deduplicate (names.cc:455-510): a name ending in digits continues counting from that number
            # ("v1" -> "v2"), any other name gets ".<k>" appended ("v" -> "v.0"); counts already handed out
            # for a stem are skipped
            j = len(nm)
            while j > 0 and nm[j - 1].isdigit():
                j -= 1
            stem = nm[:j]
            if j < len(nm):
                cnt = int(nm[j:]) + 1
            else:
                cnt = 0
                if not nm.endswith("."):
                    stem += "."
            used = stems.setdefault(stem, set())
            while True:
                while cnt in used:
                    cnt += 1
                nm = "%s%d" % (stem, cnt)
                used.add(cnt)
                if nm not in seen:
                    break
                cnt += 1
        seen.add(nm)
        out.append(nm)
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

    def to_device(self):
        """Keep this Frame's columns resident in HBM: they are uploaded once, and the fused
        groupby-aggregate of later `DT[:, reducers, by(...)]` calls reads them there instead of copying
        them over PCIe on every call (results are small and still come back as host columns).
        An extension over the reference, whose frames live in host memory; returns self."""
        self.materialize()
        ctx = self._context()
        self._dev = [ctx.upload(self._cols[i], self._stypes[i]) for i in range(self.ncols)]
        return self

    def _operand(self, i):
        """column i as the fused aggregation takes it: the resident device copy if there is one"""
        dev = getattr(self, "_dev", None)
        if dev is not None and self._ri is None and len(dev) == self.ncols:
            return dev[i]
        return self._materialized(i)

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

    # -- keys and joins (src/core/frame/key.cc:74-133, frame/join.cc:386-446) ---------------
    @property
    def key(self):
        return tuple(self._names[:getattr(self, "_nkeys", 0)])

    @key.setter
    def key(self, cols):
        if cols is None or (isinstance(cols, (list, tuple)) and not cols):
            self._nkeys = 0
            return
        if isinstance(cols, str):
            cols = [cols]
        if not isinstance(cols, (list, tuple)):
            raise TypeError("Key should be a column name, or a list/tuple of column names")
        for i, c in enumerate(cols):
            if not isinstance(c, str):
                raise TypeError("Key should be a list/tuple of column names, instead element %d was a %s" % (i, type(c)))
        kidx = [self._index(c) for c in cols]
        for a in range(len(kidx)):
            if kidx[a] in kidx[a + 1:]:
                raise ValueError("Column %s is specified multiple times within the key" % self._names[kidx[a]])
        ctx = self._context()
        order = kidx + [c for c in range(self.ncols) if c not in kidx]
        if self.nrows:
            # group(key columns): the key is valid when every group is a single row (key.cc:92-96); the
            # columns then ride through the sort into key order (apply_rowindex + materialize, :113-126)
            mats = [self._materialized(c) for c in order]
            sts = [self._stypes[c] for c in order]
            r = ctx.groupby_rows(mats[:len(kidx)], mats, key_stypes=sts[:len(kidx)], col_stypes=sts, want_rowindex=False)
            try:
                if r.ngroups < self.nrows:
                    raise ValueError("Cannot set a key: the values are not unique")
                newcols = [r.col(c) for c in range(len(order))]
            finally:
                r.free()
        else:
            newcols = [self._cols[c] for c in order]
        self._cols, self._stypes, self._names = newcols, [self._stypes[c] for c in order], [self._names[c] for c in order]
        self._ri = None
        self._nkeys = len(kidx)
        self._dev = None            # the resident copies hold the OLD column order and row order: drop them

    def _joined(self, J):
        """X[:, :, join(J)]: X's columns, then J's non-key columns read through the join index"""
        ctx = self._context()
        nk = len(J.key)
        xidx = []
        for nm in J.key:
            if nm not in self._names:
                raise ValueError("Key column `%s` does not exist in the left Frame" % nm)     # join.cc:394-397
            xidx.append(self._names.index(nm))
        xk = [self._materialized(c) for c in xidx]
        jk = [J._cols[c] for c in range(nk)]
        if self.nrows:
            idx = ctx.join_index(xk, jk, xstypes=[self._stypes[c] for c in xidx], jstypes=J._stypes[:nk])
        else:
            idx = np.zeros(0, np.int32)
        cols = [self._materialized(c) for c in range(self.ncols)]
        sts, names = list(self._stypes), list(self._names)
        for c in range(nk, J.ncols):
            cols.append(ctx.gather(J._cols[c], idx, stype=J._stypes[c]) if J.nrows else
                        np.full(len(idx), np.nan if ST2NP[J._stypes[c]].kind == "f" else _NA_INT[ST2NP[J._stypes[c]].itemsize],
                                ST2NP[J._stypes[c]]))
            sts.append(J._stypes[c]); names.append(J._names[c])
        return Frame._from_columns(cols, sts, names)

    # -- DT[i, j, by] -------------------------------------------------------------------
    def __getitem__(self, item):
        if not isinstance(item, tuple):
            raise NotImplementedError("single-selector DT[x] is outside the accelerated path")
        i = item[0]
        j = item[1] if len(item) > 1 else slice(None)
        byx, srt, jn = None, None, None
        for r in item[2:]:
            if isinstance(r, join):
                if jn is not None:
                    raise NotImplementedError("multiple joins are outside the accelerated path")
                jn = r
            elif isinstance(r, sort):
                if srt is not None:
                    raise TypeError("Multiple sort()'s are not allowed")
                srt = r
            elif isinstance(r, (by, str, ColRef, list, tuple)):
                if byx is not None:
                    raise TypeError("Multiple by()'s are not allowed")
                byx = r if isinstance(r, by) else by(r)      # DT[:, j, "A"] == DT[:, j, by("A")]
            else:
                raise NotImplementedError("modifier %r is outside the accelerated path" % (r,))
        all_rows = i is None or i is Ellipsis or (isinstance(i, slice) and i == slice(None))
        if jn is not None:
            if byx or srt or not all_rows:
                raise NotImplementedError("join() combined with i / by() / sort() is outside the accelerated path")
            if not (j is None or j is Ellipsis or (isinstance(j, slice) and j == slice(None))):
                # a name in j refers to the left frame only (the joined columns are `g.` columns in the
                # reference, outside this mirror): an unknown one is the reference's KeyError
                items = list(j.values()) if isinstance(j, dict) else (list(j) if isinstance(j, (list, tuple)) else [j])
                for r in items:
                    if isinstance(r, (str, ColRef, int, np.integer)):
                        self._index(r)
            return self._joined(jn.frame)._select(j)
        if isinstance(i, np.ndarray) and i.dtype == np.bool_:
            i = Frame([i])
        if isinstance(i, Frame):
            # a boolean column as row selector (init_from_boolean_column, rowindex_array.cc:130-170)
            if byx or srt or jn is not None:
                raise NotImplementedError("a boolean row selector combined with by() / sort() / join() is outside the accelerated path")
            if i.ncols != 1 or i._stypes[0] != L.BOOL:
                raise TypeError("Filter expression must be boolean, instead it was of type %s" % (i._stypes[:1],))
            if i.nrows != self.nrows:
                raise ValueError("i selector has %d rows, but applied to a Frame with %d rows" % (i.nrows, self.nrows))
            ctx = self._context()
            ri = ctx.bool_to_rowindex(i._materialized(0))
            fr = Frame(self)
            fr._ri = ri if self._ri is None else ctx.gather(self._ri, ri)
            return fr._select(j)
        if isinstance(i, Filter):
            if byx or srt:
                # the reference cannot do this either: src/core/expr/fexpr_func.cc:61-73
                raise NotImplementedError("FExpr_Func::evaluate_iby() not implemented yet")
            return self._filter(i)._select(j)
        if isinstance(i, (int, np.integer)) and not isinstance(i, (bool, np.bool_)) and byx is not None:
            return self._group_nth(int(i), j, byx, srt)
        if not all_rows:
            raise NotImplementedError("row selector %r is outside the accelerated path" % (i,))
        if byx is not None:
            return self._groupby(j, byx, srt)
        if srt is not None:
            return self._sorted(srt)._select(j)
        return self._select(j)

    def _select(self, j):
        if j is None or j is Ellipsis or (isinstance(j, slice) and j == slice(None)):
            return self
        items = list(j.values()) if isinstance(j, dict) else (list(j) if isinstance(j, (list, tuple)) else [j])
        if any(isinstance(r, (Reducer, Reducer2, Cumulative)) for r in items):
            return self._groupby(j, None, None)
        idx = [self._index(r) for r in items]
        names = list(j.keys()) if isinstance(j, dict) else [self._names[k] for k in idx]
        fr = Frame._from_columns([self._cols[k] for k in idx], [self._stypes[k] for k in idx], names)
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
        desc = [bool(rev) ^ bool(c.desc) for c, rev in zip(s.cols, s.reverse)]       # sort(-f.k) == reverse
        r = ctx.groupby(keys, stypes=[self._stypes[k] for k in idx], desc=desc, na_last=s.na_last,
                        na_remove=s.na_remove)
        ri = r.rowindex()
        r.free()
        fr = Frame(self)
        fr._ri = ri if self._ri is None else ctx.gather(self._ri, ri)
        return fr

    def sort(self, *cols):
        """Frame.sort(cols): ascending, NA first (src/core/sort.cc:539-558)"""
        return self._sorted(sort(*cols))

    def _group_nth(self, i, j, byx, srt):
        """DT[i, j, by(...)] with an integer i: the i-th row of every group that has one, counted from the group's
        end for negative i (/root/reference/tests/test-groups.py:486-495).  group() -> offsets and RowIndex on the
        device, the picked rows are a gather; the by-columns lead the result like in every by() query."""
        ctx = self._context()
        kidx = [self._index(c) for c in byx.cols]
        kdesc = [bool(c.desc) for c in byx.cols]
        keys = [self._materialized(k) for k in kidx]
        kst = [self._stypes[k] for k in kidx]
        sel_all = j is None or j is Ellipsis or (isinstance(j, slice) and j == slice(None))
        items = [] if sel_all else (list(j) if isinstance(j, (list, tuple)) else [j])
        if any(not isinstance(x, (ColRef, str, int, np.integer)) or isinstance(x, AllCols) for x in items):
            raise NotImplementedError("an integer row selector with by() takes plain columns in j")
        jidx = [c for c in range(self.ncols) if c not in kidx] if sel_all else [self._index(x) for x in items]
        na_last = srt is not None and srt.na_last
        if self.nrows == 0:
            sel = np.zeros(0, np.int32)
        else:
            if srt is not None:
                gri = self._by_sort_order(ctx, keys, kst, kdesc, srt)
                g = ctx.groupby(keys, stypes=kst, desc=kdesc, na_last=na_last, want_rowindex=False)
                goff = g.offsets()
            else:
                g = ctx.groupby(keys, stypes=kst, desc=kdesc, na_last=na_last)
                gri, goff = g.rowindex(), g.offsets()
            g.free()
            sizes = np.diff(goff)
            pos = (goff[:-1][sizes > i] + i) if i >= 0 else (goff[1:][sizes >= -i] + i)
            sel = ctx.gather(gri, pos.astype(np.int32)) if len(pos) else np.zeros(0, np.int32)
        fr = Frame(self)
        fr._ri = sel if self._ri is None else (ctx.gather(self._ri, sel) if len(sel) else sel)
        return fr._select([self._names[c] for c in kidx + jidx])

    def _by_sort_order(self, ctx, keys, kst, kdesc, srt):
        """RowIndex that orders the rows by the by-columns, then (inside groups) by the sort() columns"""
        skeys, sst, sdesc = list(keys), list(kst), list(kdesc)
        for c, rev in zip(srt.cols, srt.reverse):
            ci = self._index(c)
            skeys.append(self._materialized(ci)); sst.append(self._stypes[ci]); sdesc.append(bool(rev) ^ bool(c.desc))
        res = ctx.groupby(skeys, stypes=sst, desc=sdesc, na_last=srt.na_last)
        ri = res.rowindex()
        res.free()
        return ri

    def _groupby(self, j, byx, srt):
        """DT[:, j, by(...)[, sort(...)]]  (EvalContext::evaluate, src/core/expr/eval_context.cc:144-172,
        249-288; evaluate_select :497-508).  The result has one row per group when every j item is a
        reducer or a by-column (GtoONE), else one row per input row in grouped order with the reducers
        broadcast (GtoALL, workframe.cc:384-390)."""
        ctx = self._context()
        if self.nrows > 2**31 - 1:
            raise ValueError("nrows > 2**31-1: RowIndex and group offsets are int32")
        bycols = byx.cols if byx is not None else []
        kidx = [self._index(c) for c in bycols]
        kdesc = [bool(c.desc) for c in bycols]
        keys = [self._materialized(k) for k in kidx]
        kst = [self._stypes[k] for k in kidx]
        nonby = [c for c in range(self.ncols) if c not in kidx]
        sel_all = j is None or j is Ellipsis or (isinstance(j, slice) and j == slice(None))
        # j -> flat list of (name override, item); f[:] expands to the non-by columns
        raw = [] if sel_all else (list(j.items()) if isinstance(j, dict) else
                                  [(None, x) for x in (j if isinstance(j, (list, tuple)) else [j])])

        def expand(arg):
            return [ColRef(self._names[c]) for c in nonby] if isinstance(arg, AllCols) else [arg]

        items = []
        for nm, x in raw:
            if isinstance(x, AllCols):
                items += [(None, c) for c in expand(x)]
            elif isinstance(x, Reducer) and isinstance(x.arg, AllCols):
                items += [(None, Reducer(x.op, c)) for c in expand(x.arg)]
            elif isinstance(x, Cumulative) and isinstance(x.arg, AllCols):
                items += [(None, Cumulative(x.op, c, x.reverse)) for c in expand(x.arg)]
            elif isinstance(x, Reducer2):
                la, lb = expand(x.a), expand(x.b)
                if not (len(la) == len(lb) or len(la) == 1 or len(lb) == 1):
                    raise ValueError("Cannot apply reducer function %s: argument 1 has %d columns, while argument 2 "
                                     "has %d columns" % (x.op, len(la), len(lb)))    # head_reduce_binary.cc:248-252
                m = builtins.max(len(la), len(lb))
                items += [(nm, Reducer2(x.op, la[i if len(la) > 1 else 0], lb[i if len(lb) > 1 else 0])) for i in range(m)]
            elif isinstance(x, (Reducer, Cumulative, ColRef, str, int, np.integer)):
                items.append((nm, x if isinstance(x, FExpr) else ColRef(x)))
            else:
                raise NotImplementedError("j item %r is outside the accelerated path" % (x,))
        if sel_all:
            items = [(None, ColRef(self._names[c])) for c in nonby]
        reducers = [x for _, x in items if isinstance(x, (Reducer, Reducer2))]
        cums = [x for _, x in items if isinstance(x, Cumulative)]
        plain = [self._index(x) for _, x in items if isinstance(x, ColRef)]
        bynames = [self._names[k] for k in kidx]

        def item_name(nm, x):
            if nm is not None:
                return nm
            if isinstance(x, Reducer2):
                return ""                                   # auto-named C<k>
            if isinstance(x, Reducer):
                return "count" if x.op == "count0" else self._names[self._index(x.arg)]
            if isinstance(x, Cumulative):
                return "" if x.arg is None else self._names[self._index(x.arg)]
            return self._names[self._index(x)]

        def out_stype(x):
            if isinstance(x, Reducer2):
                return ctx._lib.dthip_reduce2_out_stype(self._stypes[self._index(x.a)], self._stypes[self._index(x.b)])
            if isinstance(x, Reducer):
                return L.INT64 if x.op == "count0" else ctx._lib.dthip_reduce_out_stype(L_OPS[x.op], self._stypes[self._index(x.arg)])
            if isinstance(x, Cumulative):
                return ctx._lib.dthip_cumulate_out_stype(L_CUMOPS[x.op], L.INT64 if x.arg is None else self._stypes[self._index(x.arg)])
            return self._stypes[self._index(x)]

        names = bynames + [item_name(nm, x) for nm, x in items]
        group_level = not sel_all and all(isinstance(x, (Reducer, Reducer2)) or (isinstance(x, ColRef) and self._index(x) in kidx)
                                          for _, x in items)

        if self.nrows == 0 and (reducers or cums):
            # a reducer over an empty frame without by() still yields one row (reduce_unary.h):
            # sum / count / nunique 0, everything else NA; with by(), and for row-level items, no rows
            one = not kidx and group_level
            cols, sts = [], []
            for k in range(len(kidx)):
                cols.append(np.zeros(0, ST2NP[kst[k]])); sts.append(kst[k])
            for _, x in items:
                st = out_stype(x)
                if isinstance(x, Reducer) and x.op == "median" and not (byx is not None and self._index(x.arg) in kidx):
                    st = self._stypes[self._index(x.arg)]      # the reference leaves median's type alone on 0 rows
                                                               # (not for a by-column: compute_gmedian)
                dtp = ST2NP[st]
                if not one:
                    cols.append(np.zeros(0, dtp))
                elif isinstance(x, Reducer) and x.op in ("sum", "count", "count0", "nunique"):
                    cols.append(np.zeros(1, dtp))
                else:
                    cols.append(np.full(1, np.nan if dtp.kind == "f" else _NA_INT[dtp.itemsize], dtp))
                sts.append(st)
            return Frame._from_columns(cols, sts, names)

        if not kidx:
            # DT[:, sum(f.v)] without by(): one group over all rows
            keys, kst, kdesc = [np.zeros(self.nrows, np.int8)], [L.INT8], [False]

        # first() / last() see the rows of a group in the order by() + sort() put them in
        ordered = srt is not None and any(isinstance(x, Reducer) and x.op in ("first", "last") for x in reducers)
        fusable = all(isinstance(x, Reducer) and x.op in _FUSED_OPS for x in reducers) and not ordered
        # sort(..., na_position="last") next to by() also moves the NA groups of the by-columns last
        na_last = srt is not None and srt.na_last
        if group_level and fusable and (reducers or not items):
            # fused groupby-aggregate: one row per group
            vidx, aggs = [], []
            for x in reducers:
                if x.op == "count0":
                    aggs.append(("count0", None)); continue
                ci = self._index(x.arg)
                if ci not in vidx:
                    vidx.append(ci)
                aggs.append((x.op, vidx.index(ci)))
            resident = bool(kidx) and getattr(self, "_dev", None) is not None and self._ri is None
            vals = [self._operand(c) if resident else self._materialized(c) for c in vidx]
            fkeys = keys
            if resident:
                from .engine import DevCol
                fkeys = [DevCol(self._dev[c].ptr, self._dev[c].stype, kdesc[j], keepalive=self._dev[c].keepalive)
                         for j, c in enumerate(kidx)]
            res = ctx.groupby_agg(fkeys, vals, aggs, nrows=self.nrows, key_stypes=kst,
                                  value_stypes=[self._stypes[c] for c in vidx], desc=kdesc, na_last=na_last)
            kcols = [res.key(k) for k in range(len(kidx))]
            cols, sts, a = list(kcols), list(kst[:len(kidx)]), 0
            for _, x in items:
                if isinstance(x, Reducer):
                    cols.append(res.agg(a)); sts.append(res.agg_stype(a)); a += 1
                else:
                    k = kidx.index(self._index(x))
                    cols.append(kcols[k]); sts.append(kst[k])
            res.free()
            return Frame._from_columns(cols, sts, names)

        def grouped(ref):
            """is `ref` one of the by-columns? (Grouping::GtoONE inputs take the reference's g-variants)"""
            return byx is not None and self._index(ref) in kidx

        def reduce_item(x, order, goff):
            """one value per group through the S-red seam (dthip_reduce / dthip_reduce2)"""
            if isinstance(x, Reducer2):
                ia, ib = self._index(x.a), self._index(x.b)
                if grouped(x.a) or grouped(x.b):
                    # cov / corr with a by-column: make_na_result (head_reduce_binary.cc:47-51,238-245)
                    dtp = ST2NP[out_stype(x)]
                    return np.full(len(goff) - 1, np.nan, dtp)
                return ctx.reduce2(x.op, self._materialized(ia), self._materialized(ib), order, goff,
                                   stypes=(self._stypes[ia], self._stypes[ib]))
            if x.op == "count0":
                return ctx.reduce("count0", None, None, goff)
            ci = self._index(x.arg)
            if x.op == "sd" and grouped(x.arg):
                # sd of a by-column: 0 for every group of more than one row -- the NA group included -- else NA
                # (SdGrouped_ColumnImpl, head_reduce_unary.cc:246-283)
                cnt = ctx.reduce("count0", None, None, goff)
                dtp = ST2NP[out_stype(x)]
                return np.where(cnt > 1, dtp.type(0), dtp.type(np.nan)).astype(dtp)
            return ctx.reduce(x.op, self._materialized(ci), order, goff, stype=self._stypes[ci])

        if group_level:
            # one row per group, with reducers the fused aggregation does not carry (sd, median, nunique,
            # cov, corr), or only by-columns: group once, reduce per item; the by-columns take their value
            # at the first row of each group (eval_context.cc:473-485)
            if ordered:
                gri = self._by_sort_order(ctx, keys, kst, kdesc, srt)
                g = ctx.groupby(keys, stypes=kst, desc=kdesc, na_last=na_last, want_rowindex=False)
                goff = g.offsets()
            else:
                g = ctx.groupby(keys, stypes=kst, desc=kdesc, na_last=na_last)
                gri, goff = g.rowindex(), g.offsets()
            g.free()
            first = ctx.gather(gri, goff[:-1])
            kcols = [ctx.gather(keys[k], first, stype=kst[k]) for k in range(len(kidx))]
            cols, sts = list(kcols), list(kst[:len(kidx)])
            for _, x in items:
                if isinstance(x, (Reducer, Reducer2)):
                    cols.append(reduce_item(x, gri, goff)); sts.append(out_stype(x))
                else:
                    k = kidx.index(self._index(x))
                    cols.append(kcols[k]); sts.append(kst[k])
            return Frame._from_columns(cols, sts, names)

        # the ordering: by-columns, then (inside groups) the sort() columns
        if srt is not None:
            ri, goff = self._by_sort_order(ctx, keys, kst, kdesc, srt), None
        else:
            res = ctx.groupby(keys, stypes=kst, desc=kdesc)
            ri, goff = res.rowindex(), res.offsets()
            res.free()
        full_ri = ri if self._ri is None else ctx.gather(self._ri, ri)

        if not reducers and not cums:
            # rows in grouped order: by-columns first, then the selected columns (a view, like the reference)
            order = kidx + plain
            fr = Frame._from_columns([self._cols[c] for c in order], [self._stypes[c] for c in order], names)
            fr._ri = full_ri
            return fr

        # reducers / cumulative operators next to plain columns: one row per input row in grouped order.
        # Reducers are evaluated per group (S-red seam) and broadcast back to the rows of their group;
        # cumulative operators run along the grouped order.  A sort() inside the groups permutes rows
        # within their group only (the by-columns are the leading sort keys), so the by-only offsets
        # delimit the same groups in `ri`.
        if goff is None:
            g = ctx.groupby(keys, stypes=kst, desc=kdesc, na_last=na_last, want_rowindex=False)
            goff = g.offsets()
            g.free()
        bcast = ctx.ungroup(goff) if reducers else None
        cols = [ctx.gather(keys[k], ri, stype=kst[k]) for k in range(len(kidx))]
        sts = list(kst[:len(kidx)])
        for _, x in items:
            st = out_stype(x)
            if isinstance(x, (Reducer, Reducer2)):
                cols.append(ctx.gather(reduce_item(x, ri, goff), bcast, stype=st))
            elif isinstance(x, Cumulative):
                if x.arg is None:
                    cols.append(ctx.cumulate(x.op, None, None, goff, reverse=x.reverse))
                else:
                    ci = self._index(x.arg)
                    cols.append(ctx.cumulate(x.op, self._materialized(ci), ri, goff, reverse=x.reverse, stype=self._stypes[ci]))
            else:
                ci = self._index(x)
                cols.append(ctx.gather(self._materialized(ci), ri, stype=self._stypes[ci]))
            sts.append(st)
        return Frame._from_columns(cols, sts, names)


_FUSED_OPS = ("sum", "mean", "min", "max", "count", "count0", "first", "last")   # what dthip_groupby_agg carries


# ---- set functions (src/core/set_funcs.cc) ------------------------------------------------------

_ST_RANK = [L.BOOL, L.INT8, L.INT16, L.INT32, L.INT64, L.FLOAT32, L.FLOAT64]


def _set_columns(args, fname):
    """the single columns of the argument frames (columns_from_args, set_funcs.cc:64-101)"""
    cols, name = [], None

    def walk(a, level):
        nonlocal name
        if isinstance(a, Frame):
            if a.ncols == 0:
                return
            if a.ncols > 1:
                raise ValueError("Only single-column Frames are allowed, but received a Frame with %d columns" % a.ncols)
            cols.append((a._materialized(0), a._stypes[0]))
            if name is None:
                name = a._names[0]
        elif isinstance(a, (list, tuple)) and level < 2:
            for x in a:
                walk(x, level + 1)
        else:
            raise TypeError("%s() expects a list or sequence of Frames, but got an argument of type %s" % (fname, type(a)))

    for a in args:
        walk(a, 0)
    return cols, name


def _setop(op, cols, name):
    if not cols:
        return Frame()
    # rbind of the sources: one stype, the widest (Column::rbind up-casts)
    st = _ST_RANK[builtins.max(_ST_RANK.index(s) for _, s in cols)]
    srcs = [a if s == st else Frame._cast(a, s, st) for a, s in cols]
    if len(srcs) <= 1:
        op = "union"                                        # set_funcs.cc:281-283,338-341,437-440
    ctx = default_context()
    idx = ctx.setop(op, srcs, stype=st)
    stacked = np.concatenate(srcs)                          # the rbind of the sources
    vals = ctx.gather(stacked, idx, stype=st) if len(idx) else stacked[:0]
    return Frame._from_columns([vals], [st], [name if name is not None else ""])


def unique(frame):
    """dt.unique(frame): the distinct values of ALL columns of the frame, as one sorted column (set_funcs.cc:180-193)"""
    if not isinstance(frame, Frame):
        raise ValueError("Function `unique()` expects a Frame as a parameter")
    cols = [(frame._materialized(c), frame._stypes[c]) for c in range(frame.ncols)]
    return _setop("union", cols, frame._names[0] if frame.ncols == 1 else None)


def union(*frames):
    return _setop("union", *_set_columns(frames, "union"))


def intersect(*frames):
    return _setop("intersect", *_set_columns(frames, "intersect"))


def setdiff(*frames):
    return _setop("setdiff", *_set_columns(frames, "setdiff"))


def symdiff(*frames):
    return _setop("symdiff", *_set_columns(frames, "symdiff"))


def _unused():   # keep flake-style tools quiet about the shadowed builtins being intentional
    return builtins.sum, builtins.min, builtins.max
