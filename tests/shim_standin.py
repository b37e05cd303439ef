"""TEST INFRASTRUCTURE -- a NumPy / oracle stand-in for the few libdthip entry points the reference-side binding
(integration/datatable_hip_shim.py) calls on its fused-aggregate, S-red (other reducers, cumulative operators), row-filter, rows-in-grouped-order and sort routes, so
that the binding's HOST logic (query matching, pointer and stype plumbing, residency cache, lazy DeviceFrame results,
result assembly, names) runs end to end against the real reference on a machine without a GPU.

Nothing of the product imports this file; it only answers calls the tests route to it.  "Device" memory is host memory
(as in tests/test_integration_shim.py::_FakeLib), groups and reducers come from oracle/ (the pinned C restatement of the
reference's algorithm), the predicate rule is rowindex.hip::pred_at restated in NumPy.  The GPU suite
(tests/test_shim_e2e.py) runs the same statements through the real library."""
import ctypes as C

import numpy as np

from datatable_amd import _lib as L
from oracle import oracle as o

NP = {L.BOOL: np.int8, L.INT8: np.int8, L.INT16: np.int16, L.INT32: np.int32, L.INT64: np.int64,
      L.FLOAT32: np.float32, L.FLOAT64: np.float64}
OPNAME = {L.SUM: "sum", L.MEAN: "mean", L.MIN: "min", L.MAX: "max", L.COUNT: "count"}


def _addr(x):
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    return int(getattr(x, "value", 0) or 0)


def _view(ptr, n, st):
    """a COPY of n elements of stype st at address ptr"""
    dtype = np.dtype(NP[st])
    ptr = _addr(ptr)
    if n == 0 or ptr == 0:
        return np.empty(0, dtype)
    return np.frombuffer((C.c_char * (n * dtype.itemsize)).from_address(ptr), dtype=dtype, count=n).copy()


def _store(dst, a):
    dst = _addr(dst)
    a = np.ascontiguousarray(a)
    if a.nbytes:
        C.memmove(dst, a.ctypes.data, a.nbytes)


def _out_stype(op, in_st, res):
    if op in (L.MIN, L.MAX):
        return in_st
    return {np.dtype(np.int64): L.INT64, np.dtype(np.float32): L.FLOAT32, np.dtype(np.float64): L.FLOAT64,
            np.dtype(np.int32): L.INT32}[res.dtype]


class _Res:
    def __init__(self):
        self.keys, self.aggs, self.agg_stypes, self.cols = [], [], [], []
        self.rowindex = self.offsets = None
        self.nrows = self.ngroups = 0


class StandinLib:
    def __init__(self):
        self.live, self.results, self.next = {}, {}, 1
        self.uploads = self.downloads = 0
        self.calls = []

    # ---- memory (host memory plays HBM) ---------------------------------------------------------------------------
    def dthip_malloc(self, h, nbytes, pp):
        buf = C.create_string_buffer(int(nbytes))
        p = C.addressof(buf)
        self.live[p] = buf
        pp._obj.value = p
        return 0

    def dthip_free(self, h, p):
        self.live.pop(_addr(p), None)
        return 0

    def dthip_memcpy_h2d(self, h, dst, src, n):
        self.uploads += 1
        C.memmove(_addr(dst), _addr(src), n)
        return 0

    def dthip_memcpy_d2h(self, h, dst, src, n):
        self.downloads += 1
        C.memmove(_addr(dst), _addr(src), n)
        return 0

    def dthip_set_option(self, h, name, v):
        return 0

    def dthip_from_arrow(self, h, values, validity, nrows, st, mem, dst):
        # Arrow layout -> sentinel column (oracle.arrow_to_sentinel restates arrow_fw.cc:63-72); counted as one upload
        self.uploads += 1
        self.calls.append("from_arrow")
        n = int(nrows)
        nb = (n + 7) // 8
        bm = np.frombuffer((C.c_char * nb).from_address(_addr(validity)), np.uint8).copy() if _addr(validity) else None
        if st == L.BOOL:
            vals = np.frombuffer((C.c_char * nb).from_address(_addr(values)), np.uint8).copy()
        else:
            vals = _view(values, n, st)
        _store(dst, o.arrow_to_sentinel(vals, bm, n, st))
        return 0

    # ---- queries ----------------------------------------------------------------------------------------------------
    def _group(self, karr, nkeys, n, na_pos):
        keys = [_view(karr[i].data, n, karr[i].stype) for i in range(nkeys)]
        ri, off = o.group(keys, stypes=[karr[i].stype for i in range(nkeys)],
                          desc=[bool(karr[i].flags & L.FLAG_DESCENDING) for i in range(nkeys)], na_last=na_pos == L.NA_LAST)
        return keys, ri, off

    def _new(self, r, out):
        hid = self.next
        self.next += 1
        self.results[hid] = r
        out._obj.value = hid
        return 0

    def dthip_groupby_agg(self, h, karr, nkeys, varr, nvalues, aarr, naggs, nrows, na_pos, mem, out):
        self.calls.append(("groupby_agg", mem))
        if na_pos == L.NA_REMOVE:
            return L.ENOTIMPL
        keys, ri, off = self._group(karr, nkeys, nrows, na_pos)
        vals = [_view(varr[i].data, nrows, varr[i].stype) for i in range(nvalues)]
        r = _Res()
        r.nrows, r.ngroups = nrows, len(off) - 1
        r.keys = [np.ascontiguousarray(k[ri[off[:-1]]]) for k in keys]
        for a in range(naggs):
            op, c = aarr[a].op, aarr[a].col
            if op == L.COUNT0:
                res, st = np.diff(off).astype(np.int64), L.INT64
            else:
                res = o.reduce(OPNAME[op], vals[c], ri, off, stype=varr[c].stype)
                st = _out_stype(op, varr[c].stype, res)
            r.aggs.append(np.ascontiguousarray(res)); r.agg_stypes.append(st)
        r.offsets = np.ascontiguousarray(off, np.int32)
        return self._new(r, out)

    def dthip_groupby_rows(self, h, karr, nkeys, carr, ncols, n, na_pos, mem, want_ri, out):
        self.calls.append(("groupby_rows", mem))
        if na_pos == L.NA_REMOVE:
            return L.ENOTIMPL
        _, ri, off = self._group(karr, nkeys, n, na_pos)
        r = _Res()
        r.nrows, r.ngroups = n, len(off) - 1
        r.cols = [np.ascontiguousarray(_view(carr[i].data, n, carr[i].stype)[ri]) for i in range(ncols)]
        r.offsets = np.ascontiguousarray(off, np.int32)
        if want_ri:
            r.rowindex = np.ascontiguousarray(ri, np.int32)
        return self._new(r, out)

    @staticmethod
    def _mask(p, code, cf, cint, n):
        x = _view(p.data, n, p.stype)
        isf = p.stype in (L.FLOAT32, L.FLOAT64)
        na = np.isnan(x) if isf else x == np.iinfo(x.dtype).min
        c = NP[p.stype](cf) if isf else int(cint)
        with np.errstate(invalid="ignore"):
            return {L.GT: lambda: ~na & (x > c), L.GE: lambda: ~na & (x >= c), L.LT: lambda: ~na & (x < c), L.LE: lambda: ~na & (x <= c),
                    L.EQ: lambda: ~na & (x == c), L.NE: lambda: na | (x != c), L.NOTNA: lambda: ~na, L.ISNA: lambda: na}[code]()

    def dthip_filter_groupby_rows(self, h, pcol, code, cf, cint, karr, nkeys, carr, ncols, n, na_pos, mem, want_ri, out):
        # V = DT[pred, :]; V[:, cols, by(keys)] in one call: filter -> gather -> group (the oracle's), composed RowIndex
        self.calls.append(("filter_groupby_rows", mem))
        fri = np.flatnonzero(self._mask(pcol._obj, code, cf, cint, n)).astype(np.int32)
        r = _Res()
        if len(fri) == 0:
            r.nrows, r.ngroups = 0, 0
            r.cols = [np.empty(0, NP[carr[i].stype]) for i in range(ncols)]
            r.offsets = np.zeros(1, np.int32); r.rowindex = np.empty(0, np.int32)
            return self._new(r, out)
        keys = [_view(karr[i].data, n, karr[i].stype)[fri] for i in range(nkeys)]
        p, off = o.group(keys, stypes=[karr[i].stype for i in range(nkeys)],
                         desc=[bool(karr[i].flags & L.FLAG_DESCENDING) for i in range(nkeys)], na_last=na_pos == L.NA_LAST)
        comp = fri[p]
        r.nrows, r.ngroups = len(comp), len(off) - 1
        r.cols = [np.ascontiguousarray(_view(carr[i].data, n, carr[i].stype)[comp]) for i in range(ncols)]
        r.offsets = np.ascontiguousarray(off, np.int32)
        if want_ri:
            r.rowindex = np.ascontiguousarray(comp, np.int32)
        return self._new(r, out)

    def dthip_filter_take(self, h, pcol, code, cf, cint, carr, ncols, n, mem, out_ri, optr, k):
        self.calls.append(("filter_take", mem))
        m = self._mask(pcol._obj, code, cf, cint, n)
        for i in range(ncols):
            _store(optr[i], _view(carr[i].data, n, carr[i].stype)[m])
        if _addr(out_ri):
            _store(out_ri, np.flatnonzero(m).astype(np.int32))
        k._obj.value = int(m.sum())
        return 0

    # ---- S-red: dthip_groupby once, then one call per j item ---------------------------------------------------------
    def dthip_groupby(self, h, karr, nkeys, n, na_pos, mem, want_ri, out):
        self.calls.append(("groupby", mem))
        keys, ri, off = self._group(karr, nkeys, n, L.NA_FIRST if na_pos == L.NA_REMOVE else na_pos)
        r = _Res()
        if na_pos == L.NA_REMOVE:
            # include/dthip.h: ordered NA first, then as many rows as the LAST key column has NAs are cut off the front
            last = keys[-1]
            cut = int((np.isnan(last) if last.dtype.kind == "f" else last == np.iinfo(last.dtype).min).sum())
            ri = ri[cut:]
            off = np.unique(np.clip(off.astype(np.int64) - cut, 0, None)).astype(np.int32)
        r.nrows, r.ngroups = len(ri), len(off) - 1
        r.rowindex, r.offsets = np.ascontiguousarray(ri, np.int32), np.ascontiguousarray(off, np.int32)
        return self._new(r, out)

    def dthip_gather(self, h, col, ri, nout, mem, dst):
        c = col._obj
        idx = _view(ri, nout, L.INT32)
        n = int(idx.max()) + 1 if nout else 0
        _store(dst, _view(c.data, n, c.stype)[idx])
        return 0

    def dthip_result_group_keys(self, ctx, h, col, mem, dst):
        r, c = self._r(h), col._obj
        first = r.rowindex[r.offsets[:-1]]
        n = int(first.max()) + 1 if len(first) else 0
        _store(dst, _view(c.data, n, c.stype)[first])
        return 0

    # the stype rules are host code of the REAL library (no GPU needed to ask it)
    def dthip_reduce_out_stype(self, op, st): return L.load().dthip_reduce_out_stype(op, st)
    def dthip_reduce2_out_stype(self, a, b): return L.load().dthip_reduce2_out_stype(a, b)
    def dthip_cumulate_out_stype(self, op, st): return L.load().dthip_cumulate_out_stype(op, st)

    def dthip_reduce(self, h, op, col, ri, off, ng, n, mem, dst):
        self.calls.append(("reduce", mem))
        offs = _view(off, ng + 1, L.INT32)
        if op == L.COUNT0:
            _store(dst, np.diff(offs).astype(np.int64)); return 0
        c = col._obj
        idx = _view(ri, n, L.INT32)
        v = _view(c.data, (int(idx.max()) + 1) if n else 0, c.stype)
        if op in (L.FIRST, L.LAST):
            res = v[idx[offs[:-1]]] if op == L.FIRST else v[idx[offs[1:] - 1]]
        elif op in OPNAME:
            res = o.reduce(OPNAME[op], v, idx, offs, stype=c.stype)
        else:
            res = o.reducex({L.SD: "sd", L.MEDIAN: "median", L.NUNIQUE: "nunique"}[op], v, idx, offs, stype=c.stype)
        _store(dst, res)
        return 0

    def dthip_reduce2(self, h, op, ca, cb, ri, off, ng, n, mem, dst):
        self.calls.append(("reduce2", mem))
        offs, idx = _view(off, ng + 1, L.INT32), _view(ri, n, L.INT32)
        a, b = ca._obj, cb._obj
        m = (int(idx.max()) + 1) if n else 0
        _store(dst, o.reduce2({L.COV: "cov", L.CORR: "corr"}[op], _view(a.data, m, a.stype), _view(b.data, m, b.stype), idx, offs,
                              stypes=(a.stype, b.stype)))
        return 0

    def dthip_cumulate(self, h, op, col, ri, off, ng, n, reverse, mem, dst):
        self.calls.append(("cumulate", mem))
        offs = _view(off, ng + 1, L.INT32)
        name = {L.CUMSUM: "cumsum", L.CUMPROD: "cumprod", L.CUMMIN: "cummin", L.CUMMAX: "cummax", L.CUMCOUNT: "cumcount",
                L.NGROUP: "ngroup", L.FILLNA: "fillna"}[op]
        if col is None:
            _store(dst, o.cumulate(name, None, None, offs, reverse=bool(reverse))); return 0
        c = col._obj
        idx = _view(ri, n, L.INT32)
        v = _view(c.data, (int(idx.max()) + 1) if n else 0, c.stype)
        _store(dst, o.cumulate(name, v, idx, offs, reverse=bool(reverse), stype=c.stype))
        return 0

    # ---- results ----------------------------------------------------------------------------------------------------
    def _r(self, h):
        return self.results[_addr(h)]

    def dthip_result_ngroups(self, h): return self._r(h).ngroups
    def dthip_result_nrows(self, h): return self._r(h).nrows
    def dthip_result_key(self, h, i): return self._r(h).keys[i].ctypes.data
    def dthip_result_agg(self, h, a): return self._r(h).aggs[a].ctypes.data
    def dthip_result_agg_stype(self, h, a): return self._r(h).agg_stypes[a]
    def dthip_result_col(self, h, i): return self._r(h).cols[i].ctypes.data
    def dthip_result_offsets(self, h): return self._r(h).offsets.ctypes.data
    def dthip_result_rowindex(self, h): return self._r(h).rowindex.ctypes.data

    def dthip_result_copy_key(self, ctx, h, i, dst, mem):
        _store(dst, self._r(h).keys[i]); return 0

    def dthip_result_copy_agg(self, ctx, h, a, dst, mem):
        _store(dst, self._r(h).aggs[a]); return 0

    def dthip_result_copy_col(self, ctx, h, i, dst, mem):
        _store(dst, self._r(h).cols[i]); return 0

    def dthip_result_copy_offsets(self, ctx, h, dst, mem):
        _store(dst, self._r(h).offsets); return 0

    def dthip_result_copy_rowindex(self, ctx, h, dst, mem):
        _store(dst, self._r(h).rowindex); return 0

    def dthip_result_free(self, ctx, h):
        self.results.pop(_addr(h), None)
        return 0


class StandinCtx:
    """what the binding needs of datatable_amd.engine.Context"""

    def __init__(self):
        self._lib, self._h = StandinLib(), 1
        self.options = {}

    def set_option(self, name, value):
        self.options[name] = value

    def sync(self):
        pass

    def trim(self):
        pass
