"""Thin object layer over the C ABI (include/dthip.h): Context, Result.

Host API: numpy arrays in datatable's storage convention (sentinel NAs:
INT*_MIN, NaN; bool8 as int8 with -128 = NA).  Device API: `DevCol(ptr, stype)`
wrappers around raw HBM pointers (e.g. torch `tensor.data_ptr()`), nothing copied.
"""
import ctypes as C

import numpy as np

from . import _lib as L

NP2ST = {np.dtype(np.bool_): L.BOOL, np.dtype(np.int8): L.INT8, np.dtype(np.int16): L.INT16,
         np.dtype(np.int32): L.INT32, np.dtype(np.int64): L.INT64,
         np.dtype(np.float32): L.FLOAT32, np.dtype(np.float64): L.FLOAT64}
ST2NP = {L.BOOL: np.dtype(np.int8), L.INT8: np.dtype(np.int8), L.INT16: np.dtype(np.int16),
         L.INT32: np.dtype(np.int32), L.INT64: np.dtype(np.int64),
         L.FLOAT32: np.dtype(np.float32), L.FLOAT64: np.dtype(np.float64)}
OPS = {"sum": L.SUM, "mean": L.MEAN, "min": L.MIN, "max": L.MAX, "count": L.COUNT, "count0": L.COUNT0,
       "first": L.FIRST, "last": L.LAST, "sd": L.SD, "median": L.MEDIAN, "nunique": L.NUNIQUE,
       "prod": L.PROD, "countna": L.COUNTNA}
OPS2 = {"cov": L.COV, "corr": L.CORR}
SETOPS = {"union": L.UNION, "intersect": L.INTERSECT, "setdiff": L.SETDIFF, "symdiff": L.SYMDIFF}
CUMOPS = {"cumsum": L.CUMSUM, "cumprod": L.CUMPROD, "cummin": L.CUMMIN, "cummax": L.CUMMAX,
          "cumcount": L.CUMCOUNT, "ngroup": L.NGROUP, "fillna": L.FILLNA}
CMP = {">": L.GT, ">=": L.GE, "<": L.LT, "<=": L.LE, "==": L.EQ, "!=": L.NE}


class DevCol:
    """A typed column living in HBM: raw device pointer + stype (+ sort flags)."""
    __slots__ = ("ptr", "stype", "desc", "keepalive")

    def __init__(self, ptr, stype, desc=False, keepalive=None):
        self.ptr = int(ptr)
        self.stype = int(stype)
        self.desc = bool(desc)
        self.keepalive = keepalive


class _DevBuf:
    """owner of one dthip_malloc allocation; freed when the last DevCol referring to it goes away"""

    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr

    def __del__(self):
        try:
            if self.ctx._h is not None and self.ptr:
                self.ctx._lib.dthip_free(self.ctx._h, C.c_void_p(self.ptr))
        except Exception:
            pass


def _host_col(a, stype=None, desc=False):
    a = np.ascontiguousarray(a)
    st = stype if stype is not None else NP2ST[a.dtype]
    if a.dtype == np.bool_:
        a = a.view(np.int8)
    if st in ST2NP and a.dtype.itemsize != ST2NP[st].itemsize:
        raise TypeError("array dtype %s does not match stype %d" % (a.dtype, st))
    return a, L.Col(a.ctypes.data, st, L.FLAG_DESCENDING if desc else 0)


def _cols(cols, stypes=None, desc=None):
    """-> (ctypes array of Col, mem space, keepalive list)"""
    n = len(cols)
    arr = (L.Col * max(n, 1))()
    keep = []
    mem = None
    for i, c in enumerate(cols):
        if isinstance(c, DevCol):
            m = L.DEVICE
            arr[i] = L.Col(c.ptr, c.stype, L.FLAG_DESCENDING if c.desc else 0)
        else:
            m = L.HOST
            a, cc = _host_col(c, stypes[i] if stypes else None, bool(desc[i]) if desc else False)
            keep.append(a)
            arr[i] = cc
        if mem is None:
            mem = m
        elif mem != m:
            raise ValueError("cannot mix host arrays and device columns in one call")
    return arr, (L.HOST if mem is None else mem), keep


def _cmp_args(cmp, scalar, stype):
    """(dthip_cmp code, float scalar, int scalar) of `column <cmp> scalar`.  An integer column compared with a
    non-integral number is compared as numbers, like the reference does (the column is up-cast): == never holds,
    != always does (NA included, as for any !=), the orderings round the scalar towards the side that keeps the
    meaning.  INT64_MIN as the integer scalar of == / != is a value no valid element has."""
    import math
    if scalar is None:
        # f.x == None / f.x != None: the NA rows / the valid rows
        if cmp not in ("==", "!="):
            raise TypeError("an ordering comparison with None")
        return (L.ISNA if cmp == "==" else L.NOTNA), 0.0, 0
    code = CMP[cmp]
    if stype in (L.FLOAT32, L.FLOAT64):
        return code, float(scalar), 0
    f = float(scalar)
    if math.isfinite(f) and f != math.floor(f):
        if cmp in (">", ">="):
            return CMP[">="], f, int(math.ceil(f))
        if cmp in ("<", "<="):
            return CMP["<="], f, int(math.floor(f))
        return code, f, -2**63
    if not math.isfinite(f):
        raise NotImplementedError("comparison of an integer column with %r" % (scalar,))
    iv = int(scalar)
    if iv > 2**63 - 1 or iv < -2**63 + 1:
        # outside int64 (a ctypes c_int64 argument would wrap silently): the comparison is decided by the sign alone.
        # "every valid row": NOTNA; "no row": x > INT64_MAX, or == INT64_MIN (the NA sentinel, which no valid element
        # equals and NA never compares equal); != holds for every row, NA included, like any !=
        above = iv > 0
        if cmp == "!=":
            return CMP["!="], f, -2**63
        if cmp == "==":
            return CMP["=="], f, -2**63
        if (cmp in ("<", "<=")) == above:
            return L.NOTNA, 0.0, 0
        return CMP[">"], f, 2**63 - 1
    return code, f, iv


class Result:
    """Device-resident result of a groupby (dthip_result)."""

    def __init__(self, ctx, handle, key_stypes, naggs, col_stypes=()):
        self._ctx = ctx
        self._h = handle
        self.key_stypes = list(key_stypes)
        self.naggs = naggs
        self.col_stypes = list(col_stypes)
        lib = ctx._lib
        self.ngroups = lib.dthip_result_ngroups(handle)
        self.nrows = lib.dthip_result_nrows(handle)

    # raw device pointers (valid until free())
    @property
    def rowindex_ptr(self):
        return self._ctx._lib.dthip_result_rowindex(self._h)

    @property
    def offsets_ptr(self):
        return self._ctx._lib.dthip_result_offsets(self._h)

    def key_ptr(self, k):
        return self._ctx._lib.dthip_result_key(self._h, k)

    def agg_ptr(self, a):
        return self._ctx._lib.dthip_result_agg(self._h, a)

    def agg_stype(self, a):
        return self._ctx._lib.dthip_result_agg_stype(self._h, a)

    # host copies
    def rowindex(self):
        out = np.empty(self.nrows, np.int32)
        L.check(self._ctx._lib.dthip_result_copy_rowindex(self._ctx._h, self._h, out.ctypes.data, L.HOST))
        return out

    def offsets(self):
        out = np.empty(self.ngroups + 1, np.int32)
        L.check(self._ctx._lib.dthip_result_copy_offsets(self._ctx._h, self._h, out.ctypes.data, L.HOST))
        return out

    def key(self, k):
        out = np.empty(self.ngroups, ST2NP[self.key_stypes[k]])
        L.check(self._ctx._lib.dthip_result_copy_key(self._ctx._h, self._h, k, out.ctypes.data, L.HOST))
        return out

    def agg(self, a):
        out = np.empty(self.ngroups, ST2NP[self.agg_stype(a)])
        L.check(self._ctx._lib.dthip_result_copy_agg(self._ctx._h, self._h, a, out.ctypes.data, L.HOST))
        return out

    def col(self, c):
        """column c of a groupby_rows result, in grouped order"""
        out = np.empty(self.nrows, ST2NP[self.col_stypes[c]])
        L.check(self._ctx._lib.dthip_result_copy_col(self._ctx._h, self._h, c, out.ctypes.data, L.HOST))
        return out

    def col_ptr(self, c):
        return self._ctx._lib.dthip_result_col(self._h, c)

    def col_into(self, c, ptr):
        L.check(self._ctx._lib.dthip_result_copy_col(self._ctx._h, self._h, c, C.c_void_p(ptr), L.DEVICE))

    # device-to-device copies into caller-owned HBM (e.g. a torch tensor's data_ptr())
    def rowindex_into(self, ptr):
        L.check(self._ctx._lib.dthip_result_copy_rowindex(self._ctx._h, self._h, C.c_void_p(ptr), L.DEVICE))

    def offsets_into(self, ptr):
        L.check(self._ctx._lib.dthip_result_copy_offsets(self._ctx._h, self._h, C.c_void_p(ptr), L.DEVICE))

    def key_into(self, k, ptr):
        L.check(self._ctx._lib.dthip_result_copy_key(self._ctx._h, self._h, k, C.c_void_p(ptr), L.DEVICE))

    def agg_into(self, a, ptr):
        L.check(self._ctx._lib.dthip_result_copy_agg(self._ctx._h, self._h, a, C.c_void_p(ptr), L.DEVICE))

    def group_keys(self, key, stype=None):
        """by-column of a plain groupby result: key[rowindex[offsets[g]]]"""
        arr, mem, keep = _cols([key], [stype] if stype is not None else None)
        st = arr[0].stype
        if mem == L.HOST:
            out = np.empty(self.ngroups, ST2NP[st])
            L.check(self._ctx._lib.dthip_result_group_keys(self._ctx._h, self._h, arr, mem, out.ctypes.data))
            return out
        raise ValueError("device group_keys: call dthip_result_group_keys with your own output buffer")

    def free(self):
        if self._h is not None:
            self._ctx._lib.dthip_result_free(self._ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            if self._ctx._h is not None:
                self.free()
        except Exception:
            pass


# ---- multi-GPU: the exchange runs inside libdthip.so (csrc/comm.hip) ------------------------------------------
def comm_unique_id():
    """128 bytes created on rank 0 (ncclGetUniqueId) to be handed to every rank"""
    lib = L.load()
    buf = C.create_string_buffer(L.COMM_ID_BYTES)
    L.check(lib.dthip_comm_unique_id(buf))
    return buf.raw


def _agg_array(aggs):
    aarr = (L.Agg * max(len(aggs), 1))()
    for i, (op, col) in enumerate(aggs):
        aarr[i] = L.Agg(OPS[op] if isinstance(op, str) else int(op), -1 if col is None else int(col))
    return aarr


class _ShardMixin:
    def comm_init(self, rank, world, comm_id):
        """collective: this context becomes rank `rank` of a `world`-rank RCCL communicator (one rank per GPU)"""
        L.check(self._lib.dthip_comm_init(self._h, int(rank), int(world), comm_id))

    def comm_destroy(self):
        L.check(self._lib.dthip_comm_destroy(self._h))

    def comm_last_stats(self):
        """what the last sharded call moved: dict(bytes_to_peers, bytes_to_self, rows_sent, rows_received, allgathers)"""
        out = (C.c_int64 * 5)()
        L.check(self._lib.dthip_comm_last_stats(self._h, out, 5))
        return dict(zip(("bytes_to_peers", "bytes_to_self", "rows_sent", "rows_received", "allgathers"), [int(x) for x in out]))

    @property
    def comm_rank(self):
        return self._lib.dthip_comm_rank(self._h)

    @property
    def comm_world(self):
        return self._lib.dthip_comm_world(self._h)

    def sharded_groupby_agg(self, keys, values, aggs, nrows=None, key_stypes=None, value_stypes=None, na_last=False):
        """collective DT[:, aggs, by(keys)] over row shards: this rank's rows in, this rank's key range out"""
        karr, kmem, kkeep = _cols(keys, key_stypes)
        varr, vmem, vkeep = _cols(values, value_stypes)
        if values and kmem != vmem:
            raise ValueError("keys and values must live in the same memory space")
        if nrows is None:
            nrows = len(kkeep[0])
        h = C.c_void_p()
        L.check(self._lib.dthip_sharded_groupby_agg(self._h, karr, len(keys), varr, len(values), _agg_array(aggs), len(aggs), nrows,
                                                    L.NA_LAST if na_last else L.NA_FIRST, kmem, C.byref(h)))
        return Result(self, h, [karr[i].stype for i in range(len(keys))], len(aggs))

    def sharded_groupby_rows(self, keys, cols, row_offset, nrows=None, key_stypes=None, col_stypes=None, na_last=False):
        """collective DT[:, cols, by(keys)] over row shards; result columns: cols..., then the global row ids (int64)"""
        karr, kmem, kkeep = _cols(keys, key_stypes)
        carr, cmem, ckeep = _cols(cols, col_stypes)
        if nrows is None:
            nrows = len(kkeep[0])
        h = C.c_void_p()
        L.check(self._lib.dthip_sharded_groupby_rows(self._h, karr, len(keys), carr, len(cols), nrows, int(row_offset),
                                                     L.NA_LAST if na_last else L.NA_FIRST, kmem, C.byref(h)))
        return Result(self, h, [karr[i].stype for i in range(len(keys))], 0, [carr[i].stype for i in range(len(cols))] + [L.INT64])


class Context(_ShardMixin):
    """One device + one HIP stream + cached workspace (dthip_ctx)."""

    def __init__(self, device=0, stream=None):
        self._lib = L.load()
        h = C.c_void_p()
        rc = self._lib.dthip_init(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        L.check(rc)
        self._h = h
        self.device = device

    def close(self):
        if self._h is not None:
            self._lib.dthip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_stream(self, handle):
        """launch on the caller's HIP stream (0 / None = the device's default stream) from now on"""
        L.check(self._lib.dthip_use_stream(self._h, C.c_void_p(handle) if handle else None))

    def sync(self):
        L.check(self._lib.dthip_sync(self._h))

    def trim(self):
        L.check(self._lib.dthip_trim(self._h))

    def set_option(self, name, value):
        """tuning knobs of dthip_set_option(): 'agg_path' (0 auto / 1 sort / 2 bucketed), 'bucket_variant'"""
        L.check(self._lib.dthip_set_option(self._h, name.encode(), int(value)))

    # ---- device memory ---------------------------------------------------
    def upload(self, values, stype=None):
        """host array -> HBM (dthip_malloc + one host->device copy).  Returns a DevCol that owns the buffer."""
        a, col = _host_col(values, stype)
        p = C.c_void_p()
        L.check(self._lib.dthip_malloc(self._h, max(a.nbytes, 1), C.byref(p)))
        buf = _DevBuf(self, p.value)
        if a.nbytes:
            L.check(self._lib.dthip_memcpy_h2d(self._h, p, a.ctypes.data, a.nbytes))
        return DevCol(p.value, col.stype, keepalive=buf)

    def from_arrow(self, values, validity, nrows, stype, device=False):
        """Arrow-layout column (values buffer + validity bitmap, LSB first; validity None = no nulls; stype BOOL: values are
        Arrow's bit-packed booleans) -> a DevCol in the sentinel layout, converted on the device (dthip_from_arrow).
        values / validity: numpy arrays (host) or, with device=True, raw HBM addresses."""
        st = int(stype)
        nbytes = max(int(nrows), 0) * (ST2NP[st].itemsize if st in ST2NP else 8)      # (an unknown stype is the library's to refuse)
        p = C.c_void_p()
        L.check(self._lib.dthip_malloc(self._h, max(nbytes, 1), C.byref(p)))
        buf = _DevBuf(self, p.value)
        if device:
            vp, bp, keep = int(values), (int(validity) if validity else None), None
        else:
            va = np.ascontiguousarray(values)
            ba = np.ascontiguousarray(validity, np.uint8) if validity is not None else None
            vp, bp, keep = va.ctypes.data, (ba.ctypes.data if ba is not None else None), (va, ba)
        L.check(self._lib.dthip_from_arrow(self._h, C.c_void_p(vp), C.c_void_p(bp) if bp else None, int(nrows), st,
                                           L.DEVICE if device else L.HOST, p))
        del keep
        return DevCol(p.value, st, keepalive=buf)

    def download(self, col, nrows):
        """DevCol -> numpy (one device->host copy)"""
        out = np.empty(nrows, ST2NP[col.stype])
        if out.nbytes:
            L.check(self._lib.dthip_memcpy_d2h(self._h, out.ctypes.data, C.c_void_p(col.ptr), out.nbytes))
        return out

    def host_register(self, a):
        """page-lock a numpy array's buffer so DTHIP_HOST calls DMA from it directly (dthip_host_register)"""
        L.check(self._lib.dthip_host_register(self._h, a.ctypes.data, a.nbytes))

    def host_unregister(self, a):
        L.check(self._lib.dthip_host_unregister(self._h, a.ctypes.data))

    # ---- timing -----------------------------------------------------------
    def timer_start(self):
        L.check(self._lib.dthip_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float(0)
        L.check(self._lib.dthip_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def last_call_stats(self):
        """what the last query call did besides its result (dthip_last_call_stats): sweeps repeated after a wrong key-range
        guess / a wrong NA-free guess, routes given up after they had started, and the path that produced the result"""
        out = (C.c_int64 * 5)()
        L.check(self._lib.dthip_last_call_stats(self._h, out, 5))
        path = {0: None, 1: "sort", 2: "bucketed", 3: "hash", 4: "fused_filter", 6: "presorted"}.get(int(out[3]), int(out[3]))
        return {"retries_key_range": int(out[0]), "retries_na_guess": int(out[1]), "routes_abandoned": int(out[2]), "path": path,
                "outlier_rows_listed": int(out[4])}

    def profile(self, on=True):
        L.check(self._lib.dthip_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        L.check(self._lib.dthip_profile_reset(self._h))

    def profile_get(self, name):
        ms, n = C.c_double(0), C.c_int64(0)
        L.check(self._lib.dthip_profile_get(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_names(self):
        buf = C.create_string_buffer(8192)
        L.check(self._lib.dthip_profile_names(self._h, buf, len(buf)))
        return [s for s in buf.value.decode().split("\n") if s]

    # ---- S-grp ------------------------------------------------------------
    def groupby(self, keys, nrows=None, stypes=None, desc=None, na_last=False, want_rowindex=True, na_remove=False):
        arr, mem, keep = _cols(keys, stypes, desc)
        if nrows is None:
            nrows = len(keep[0])
        h = C.c_void_p()
        napos = L.NA_REMOVE if na_remove else (L.NA_LAST if na_last else L.NA_FIRST)
        L.check(self._lib.dthip_groupby(self._h, arr, len(keys), nrows, napos, mem,
                                        1 if want_rowindex else 0, C.byref(h)))
        return Result(self, h, [arr[i].stype for i in range(len(keys))], 0)

    def groupby_rows(self, keys, cols, nrows=None, key_stypes=None, col_stypes=None, desc=None, na_last=False,
                     want_rowindex=True):
        """rows in grouped order: group() + the materialisation of `cols` through its RowIndex, fused"""
        karr, kmem, kkeep = _cols(keys, key_stypes, desc)
        carr, cmem, ckeep = _cols(cols, col_stypes)
        if cols and kmem != cmem:
            raise ValueError("keys and columns must live in the same memory space")
        if nrows is None:
            nrows = len(kkeep[0])
        h = C.c_void_p()
        L.check(self._lib.dthip_groupby_rows(self._h, karr, len(keys), carr, len(cols), nrows,
                                             L.NA_LAST if na_last else L.NA_FIRST, kmem, 1 if want_rowindex else 0,
                                             C.byref(h)))
        return Result(self, h, [karr[i].stype for i in range(len(keys))], 0, [carr[i].stype for i in range(len(cols))])

    def filter_groupby_rows(self, pred, cmp, scalar, keys, cols, nrows=None, pred_stype=None, key_stypes=None, col_stypes=None,
                            desc=None, na_last=False, want_rowindex=True):
        """V = DT[pred <cmp> scalar, :]; V[:, cols, by(keys)] in one call (dthip_filter_groupby_rows): the passing rows in
        grouped order; result.rowindex() = the composed RowIndex (original row numbers in grouped order)"""
        parr, pmem, pkeep = _cols([pred], [pred_stype] if pred_stype is not None else None)
        karr, kmem, kkeep = _cols(keys, key_stypes, desc)
        carr, cmem, ckeep = _cols(cols, col_stypes)
        if kmem != pmem or (cols and cmem != pmem):
            raise ValueError("predicate, keys and columns must live in the same memory space")
        if nrows is None:
            nrows = len(pkeep[0])
        code, cf, ci = _cmp_args(cmp, scalar, parr[0].stype)
        h = C.c_void_p()
        L.check(self._lib.dthip_filter_groupby_rows(self._h, parr, code, cf, ci, karr, len(keys), carr, len(cols), nrows,
                                                    L.NA_LAST if na_last else L.NA_FIRST, pmem, 1 if want_rowindex else 0, C.byref(h)))
        return Result(self, h, [karr[i].stype for i in range(len(keys))], 0, [carr[i].stype for i in range(len(cols))])

    def groupby_agg(self, keys, values, aggs, nrows=None, key_stypes=None, value_stypes=None, desc=None,
                    na_last=False):
        """aggs: list of (op, value_index); op a name ('sum','mean','min','max','count','count0') or code."""
        karr, kmem, kkeep = _cols(keys, key_stypes, desc)
        varr, vmem, vkeep = _cols(values, value_stypes)
        if values and kmem != vmem:
            raise ValueError("keys and values must live in the same memory space")
        if nrows is None:
            nrows = len(kkeep[0])
        aarr = (L.Agg * max(len(aggs), 1))()
        for i, (op, col) in enumerate(aggs):
            aarr[i] = L.Agg(OPS[op] if isinstance(op, str) else int(op), -1 if col is None else int(col))
        h = C.c_void_p()
        L.check(self._lib.dthip_groupby_agg(self._h, karr, len(keys), varr, len(values), aarr, len(aggs), nrows,
                                            L.NA_LAST if na_last else L.NA_FIRST, kmem, C.byref(h)))
        return Result(self, h, [karr[i].stype for i in range(len(keys))], len(aggs))

    # ---- S-red ------------------------------------------------------------
    def reduce(self, op, values, rowindex, offsets, stype=None):
        opc = OPS[op] if isinstance(op, str) else int(op)
        offsets = np.ascontiguousarray(offsets, np.int32)
        ng = len(offsets) - 1
        nrows = int(offsets[-1]) if ng >= 0 and len(offsets) else 0
        ri = None
        if rowindex is not None:
            ri = np.ascontiguousarray(rowindex, np.int32)
        if opc == L.COUNT0:
            out = np.empty(ng, np.int64)
            L.check(self._lib.dthip_reduce(self._h, opc, None, None, offsets.ctypes.data, ng, nrows, L.HOST,
                                           out.ctypes.data))
            return out
        a, col = _host_col(values, stype)
        ost = self._lib.dthip_reduce_out_stype(opc, col.stype)
        out = np.empty(ng, ST2NP[ost])
        L.check(self._lib.dthip_reduce(self._h, opc, C.byref(col), ri.ctypes.data if ri is not None else None,
                                       offsets.ctypes.data, ng, nrows, L.HOST, out.ctypes.data))
        return out

    def reduce2(self, op, va, vb, rowindex, offsets, stypes=(None, None)):
        """cov / corr of two columns per group (head_reduce_binary.cc:113-198); host arrays"""
        opc = OPS2[op] if isinstance(op, str) else int(op)
        offsets = np.ascontiguousarray(offsets, np.int32)
        ng = len(offsets) - 1
        nrows = int(offsets[-1]) if ng >= 0 and len(offsets) else 0
        ri = np.ascontiguousarray(rowindex, np.int32) if rowindex is not None else None
        a, ca = _host_col(va, stypes[0])
        b, cb = _host_col(vb, stypes[1])
        out = np.empty(ng, ST2NP[self._lib.dthip_reduce2_out_stype(ca.stype, cb.stype)])
        L.check(self._lib.dthip_reduce2(self._h, opc, C.byref(ca), C.byref(cb), ri.ctypes.data if ri is not None else None,
                                        offsets.ctypes.data, ng, nrows, L.HOST, out.ctypes.data))
        return out

    def cumulate(self, op, values, rowindex, offsets, reverse=False, stype=None):
        """cumsum / cumprod / cummin / cummax / fillna inside groups, cumcount / ngroup; output in grouped row order"""
        opc = CUMOPS[op] if isinstance(op, str) else int(op)
        offsets = np.ascontiguousarray(offsets, np.int32)
        ng = len(offsets) - 1
        nrows = int(offsets[-1]) if ng >= 0 and len(offsets) else 0
        ri = np.ascontiguousarray(rowindex, np.int32) if rowindex is not None else None
        if opc in (L.CUMCOUNT, L.NGROUP):
            out = np.empty(nrows, np.int64)
            if nrows:
                L.check(self._lib.dthip_cumulate(self._h, opc, None, None, offsets.ctypes.data, ng, nrows,
                                                 1 if reverse else 0, L.HOST, out.ctypes.data))
            return out
        a, col = _host_col(values, stype)
        out = np.empty(nrows, ST2NP[self._lib.dthip_cumulate_out_stype(opc, col.stype)])
        if nrows:
            L.check(self._lib.dthip_cumulate(self._h, opc, C.byref(col), ri.ctypes.data if ri is not None else None,
                                             offsets.ctypes.data, ng, nrows, 1 if reverse else 0, L.HOST, out.ctypes.data))
        return out

    # ---- set functions / natural join (the other callers of group()) -------
    def setop(self, op, columns, stype=None):
        """columns: list of host arrays of ONE stype (the sources).  Returns the row ids into their
        concatenation of the result elements of union / intersect / setdiff / symdiff, ascending by value."""
        opc = SETOPS[op] if isinstance(op, str) else int(op)
        stacked = np.concatenate([np.ascontiguousarray(c) for c in columns]) if columns else np.zeros(0, np.int32)
        a, col = _host_col(stacked, stype)
        cum = np.cumsum([len(x) for x in columns]).astype(np.int64)
        out = np.empty(len(a), np.int32)
        k = C.c_int64(0)
        L.check(self._lib.dthip_setop(self._h, opc, C.byref(col), cum.ctypes.data, len(columns), len(a), L.HOST,
                                      out.ctypes.data, C.byref(k)))
        return out[:k.value].copy()

    def join_index(self, xkeys, jkeys, xstypes=None, jstypes=None):
        """per row of X the row of the keyed frame J (key columns sorted ascending, unique) it joins to, or INT32_MIN"""
        xarr, xmem, xkeep = _cols(xkeys, xstypes)
        jarr, jmem, jkeep = _cols(jkeys, jstypes)
        if xmem != L.HOST or jmem != L.HOST:
            raise ValueError("join_index takes host arrays; device columns: join_index_dev")
        out = np.empty(len(xkeep[0]), np.int32)
        L.check(self._lib.dthip_join_index(self._h, xarr, jarr, len(xkeys), len(xkeep[0]), len(jkeep[0]), L.HOST,
                                           out.ctypes.data))
        return out

    def join_index_dev(self, xcols, jcols, xrows, jrows, out_ptr):
        xarr = (L.Col * len(xcols))(*[L.Col(c.ptr, c.stype, 0) for c in xcols])
        jarr = (L.Col * len(jcols))(*[L.Col(c.ptr, c.stype, 0) for c in jcols])
        L.check(self._lib.dthip_join_index(self._h, xarr, jarr, len(xcols), xrows, jrows, L.DEVICE, C.c_void_p(out_ptr)))

    def setop_dev(self, op, col, cumsizes, nrows, out_ptr):
        opc = SETOPS[op] if isinstance(op, str) else int(op)
        c = L.Col(col.ptr, col.stype, 0)
        cum = (C.c_int64 * len(cumsizes))(*[int(x) for x in cumsizes])
        k = C.c_int64(0)
        L.check(self._lib.dthip_setop(self._h, opc, C.byref(c), cum, len(cumsizes), nrows, L.DEVICE, C.c_void_p(out_ptr), C.byref(k)))
        return k.value

    def range_bucket(self, values, bounds, stype=None):
        """int8 destination of every row in the range partition given by ascending `bounds` (host arrays)"""
        a, col = _host_col(values, stype)
        out = np.empty(len(a), np.int8)
        b = (C.c_int64 * max(len(bounds), 1))(*[min(max(int(x), -2**63), 2**63 - 1) for x in bounds])
        L.check(self._lib.dthip_range_bucket(self._h, C.byref(col), len(a), b, len(bounds), L.HOST, out.ctypes.data))
        return out

    def ungroup(self, offsets):
        """group index of every grouped position (Groupby::ungroup_rowindex)"""
        offsets = np.ascontiguousarray(offsets, np.int32)
        ng = len(offsets) - 1
        n = int(offsets[-1]) if ng >= 0 else 0
        out = np.empty(n, np.int32)
        if n:
            L.check(self._lib.dthip_ungroup(self._h, offsets.ctypes.data, ng, n, L.HOST, out.ctypes.data))
        return out

    # ---- RowIndex ---------------------------------------------------------
    def bool_to_rowindex(self, mask):
        m = np.ascontiguousarray(mask)
        if m.dtype == np.bool_:
            m = m.view(np.int8)
        out = np.empty(len(m), np.int32)
        k = C.c_int64(0)
        L.check(self._lib.dthip_bool_to_rowindex(self._h, m.ctypes.data, len(m), L.HOST, out.ctypes.data, C.byref(k)))
        return out[:k.value].copy()

    def filter_cmp(self, values, cmp, scalar, stype=None):
        a, col = _host_col(values, stype)
        out = np.empty(len(a), np.int32)
        k = C.c_int64(0)
        code, cf, ci = _cmp_args(cmp, scalar, col.stype)
        L.check(self._lib.dthip_filter_cmp(self._h, C.byref(col), len(a), code, cf, ci, L.HOST, out.ctypes.data, C.byref(k)))
        return out[:k.value].copy()

    def filter_take(self, values, cmp, scalar, cols, stype=None, col_stypes=None, want_rowindex=True):
        """rows with values <cmp> scalar: (RowIndex or None, [cols[k] restricted to those rows]) in one sweep"""
        a, pcol = _host_col(values, stype)
        carr, cmem, ckeep = _cols(cols, col_stypes)
        n = len(a)
        outs = [np.empty(n, k.dtype) for k in ckeep]
        optr = (C.c_void_p * max(len(outs), 1))(*[o.ctypes.data for o in outs])
        ri = np.empty(n, np.int32) if want_rowindex else None
        k = C.c_int64(0)
        code, cf, ci = _cmp_args(cmp, scalar, pcol.stype)
        L.check(self._lib.dthip_filter_take(self._h, C.byref(pcol), code, cf, ci, carr, len(cols), n, L.HOST,
                                            ri.ctypes.data if ri is not None else None, optr, C.byref(k)))
        return (ri[:k.value].copy() if ri is not None else None), [o[:k.value].copy() for o in outs]

    def filter_take_dev(self, col, cmp, scalar, cols, nrows, out_ri_ptr, out_ptrs):
        """device-resident filter_take: DevCol predicate column and DevCol columns, raw output pointers; returns the count"""
        c = L.Col(col.ptr, col.stype, 0)
        carr = (L.Col * max(len(cols), 1))(*[L.Col(x.ptr, x.stype, 0) for x in cols])
        optr = (C.c_void_p * max(len(out_ptrs), 1))(*[int(x) for x in out_ptrs])
        k = C.c_int64(0)
        code, cf, ci = _cmp_args(cmp, scalar, col.stype)
        L.check(self._lib.dthip_filter_take(self._h, C.byref(c), code, cf, ci, carr, len(cols), nrows, L.DEVICE,
                                            C.c_void_p(out_ri_ptr) if out_ri_ptr else None, optr, C.byref(k)))
        return k.value

    # device-resident variants (raw HBM pointers in, raw HBM pointers out)
    def filter_cmp_dev(self, col, nrows, cmp, scalar, out_ptr):
        """rows of DevCol `col` with col <cmp> scalar -> ascending int32 RowIndex at out_ptr (room for nrows); returns the count"""
        c = L.Col(col.ptr, col.stype, 0)
        k = C.c_int64(0)
        code, cf, ci = _cmp_args(cmp, scalar, col.stype)
        L.check(self._lib.dthip_filter_cmp(self._h, C.byref(c), nrows, code, cf, ci, L.DEVICE, C.c_void_p(out_ptr), C.byref(k)))
        return k.value

    def range_bucket_dev(self, col, nrows, bounds, out_ptr):
        """int8 destination (number of boundaries <= key) of every row of DevCol `col` at out_ptr"""
        c = L.Col(col.ptr, col.stype, 0)
        b = (C.c_int64 * max(len(bounds), 1))(*[min(max(int(x), -2**63), 2**63 - 1) for x in bounds])
        L.check(self._lib.dthip_range_bucket(self._h, C.byref(c), nrows, b, len(bounds), L.DEVICE, C.c_void_p(out_ptr)))

    # device-resident forms of the S-red seam: every pointer is a device address, nothing is copied
    def reduce_dev(self, op, col, rowindex_ptr, offsets_ptr, ngroups, nrows, out_ptr):
        opc = OPS[op] if isinstance(op, str) else int(op)
        c = L.Col(col.ptr, col.stype, 0) if col is not None else None
        L.check(self._lib.dthip_reduce(self._h, opc, C.byref(c) if c is not None else None,
                                       C.c_void_p(rowindex_ptr) if rowindex_ptr else None, C.c_void_p(offsets_ptr),
                                       ngroups, nrows, L.DEVICE, C.c_void_p(out_ptr)))

    def reduce2_dev(self, op, cola, colb, rowindex_ptr, offsets_ptr, ngroups, nrows, out_ptr):
        opc = OPS2[op] if isinstance(op, str) else int(op)
        ca, cb = L.Col(cola.ptr, cola.stype, 0), L.Col(colb.ptr, colb.stype, 0)
        L.check(self._lib.dthip_reduce2(self._h, opc, C.byref(ca), C.byref(cb),
                                        C.c_void_p(rowindex_ptr) if rowindex_ptr else None, C.c_void_p(offsets_ptr),
                                        ngroups, nrows, L.DEVICE, C.c_void_p(out_ptr)))

    def cumulate_dev(self, op, col, rowindex_ptr, offsets_ptr, ngroups, nrows, out_ptr, reverse=False):
        opc = CUMOPS[op] if isinstance(op, str) else int(op)
        c = L.Col(col.ptr, col.stype, 0) if col is not None else None
        L.check(self._lib.dthip_cumulate(self._h, opc, C.byref(c) if c is not None else None,
                                         C.c_void_p(rowindex_ptr) if rowindex_ptr else None, C.c_void_p(offsets_ptr),
                                         ngroups, nrows, 1 if reverse else 0, L.DEVICE, C.c_void_p(out_ptr)))

    def gather_dev(self, col, rowindex_ptr, nout, out_ptr):
        c = L.Col(col.ptr, col.stype, 0)
        L.check(self._lib.dthip_gather(self._h, C.byref(c), C.c_void_p(rowindex_ptr), nout, L.DEVICE, C.c_void_p(out_ptr)))

    def gather(self, values, rowindex, stype=None):
        a, col = _host_col(values, stype)
        ri = np.ascontiguousarray(rowindex, np.int32)
        out = np.empty(len(ri), a.dtype)
        L.check(self._lib.dthip_gather(self._h, C.byref(col), ri.ctypes.data, len(ri), L.HOST, out.ctypes.data))
        return out


import threading
_default_ctx = threading.local()


class LocalComm:
    """`world` logical shards driven from ONE process (dthip_comm_init_local): every rank has its own Context (all on
    `device` unless `devices` says otherwise); the exchange is device-to-device copies.  Same phases as the RCCL path."""

    def __init__(self, world, device=0, devices=None):
        self.ctxs = [Context(devices[r] if devices else device) for r in range(world)]
        self.world = world
        self._lib = self.ctxs[0]._lib
        self._harr = (C.c_void_p * world)(*[c._h for c in self.ctxs])
        L.check(self._lib.dthip_comm_init_local(self._harr, world))

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []

    def _percol(self, lists, stypes=None):
        """[[cols of rank 0], [cols of rank 1], ...] -> (Col** array, mem, keepalive)"""
        arrs, keep, mem = [], [], L.HOST
        for cols in lists:
            a, mem, k = _cols(cols, stypes)
            arrs.append(a); keep.append(k)
        pp = (C.POINTER(L.Col) * len(lists))(*[C.cast(a, C.POINTER(L.Col)) for a in arrs])
        return pp, mem, (arrs, keep)

    def groupby_agg(self, keys, values, aggs, key_stypes=None, value_stypes=None, na_last=False):
        """keys[r] / values[r]: rank r's columns (numpy arrays).  Returns one Result per rank (rank order = key order)."""
        kpp, kmem, kk = self._percol(keys, key_stypes)
        vpp, vmem, vk = self._percol(values, value_stypes)
        nrows = (C.c_int64 * self.world)(*[len(k[0]) for k in keys])
        outs = (C.c_void_p * self.world)()
        L.check(self._lib.dthip_sharded_groupby_agg_local(self._harr, self.world, kpp, len(keys[0]), vpp, len(values[0]) if values else 0,
                                                          _agg_array(aggs), len(aggs), nrows, L.NA_LAST if na_last else L.NA_FIRST,
                                                          kmem, outs))
        kst = [kk[0][0][i].stype for i in range(len(keys[0]))]
        return [Result(self.ctxs[r], C.c_void_p(outs[r]), kst, len(aggs)) for r in range(self.world)]

    def groupby_rows(self, keys, cols, row_offsets, key_stypes=None, col_stypes=None, na_last=False):
        kpp, kmem, kk = self._percol(keys, key_stypes)
        cpp, cmem, ck = self._percol(cols, col_stypes)
        nrows = (C.c_int64 * self.world)(*[len(k[0]) for k in keys])
        offs = (C.c_int64 * self.world)(*[int(x) for x in row_offsets])
        outs = (C.c_void_p * self.world)()
        L.check(self._lib.dthip_sharded_groupby_rows_local(self._harr, self.world, kpp, len(keys[0]), cpp, len(cols[0]) if cols else 0,
                                                           nrows, offs, L.NA_LAST if na_last else L.NA_FIRST, kmem, outs))
        kst = [kk[0][0][i].stype for i in range(len(keys[0]))]
        cst = [ck[0][0][i].stype for i in range(len(cols[0]))] + [L.INT64]
        return [Result(self.ctxs[r], C.c_void_p(outs[r]), kst, 0, cst) for r in range(self.world)]


def default_context():
    """The calling THREAD's context on device LOCAL_RANK (or 0).  A Context (dthip_ctx: stream, allocator cache, pinned
    read-back buffer) is thread-compatible, not thread-safe, and ctypes releases the GIL during calls -- so every
    Python thread gets its own."""
    ctx = getattr(_default_ctx, "ctx", None)
    if ctx is None:
        import os
        ctx = _default_ctx.ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return ctx
