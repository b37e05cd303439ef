"""ctypes binding of oracle/libdtoracle.so (see dt_oracle.c for the reference
file:line each function restates).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdtoracle.so")

# reference SType codes (src/core/stype.h:41-62)
BOOL, INT8, INT16, INT32, INT64, FLOAT32, FLOAT64 = 1, 2, 3, 4, 5, 6, 7
SUM, MEAN, MIN, MAX, COUNT, COUNT0 = 0, 1, 2, 3, 4, 5
PROD, COUNTNA = 11, 12
OPS = {"sum": SUM, "mean": MEAN, "min": MIN, "max": MAX, "count": COUNT, "count0": COUNT0, "prod": PROD, "countna": COUNTNA}
_NP2ST = {np.dtype(np.bool_): BOOL, np.dtype(np.int8): INT8, np.dtype(np.int16): INT16,
          np.dtype(np.int32): INT32, np.dtype(np.int64): INT64,
          np.dtype(np.float32): FLOAT32, np.dtype(np.float64): FLOAT64}
_ST2NP = {BOOL: np.int8, INT8: np.int8, INT16: np.int16, INT32: np.int32, INT64: np.int64,
          FLOAT32: np.float32, FLOAT64: np.float64}


class _Col(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stype", C.c_int32), ("flags", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libdtoracle.so"])


def _lib():
    srcs = [os.path.join(_HERE, n) for n in ("dt_oracle.c", "dt_oracle_groupwise.c", "dt_oracle_sets.c")]
    if not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(s) for s in srcs):
        build()
    lib = C.CDLL(_SO)
    lib.dto_group.restype = C.c_int
    lib.dto_bool_to_rowindex.restype = C.c_int64
    lib.dto_filter_cmp.restype = C.c_int64
    lib.dto_filter_cmp.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_int64, C.c_void_p]
    return lib


_L = None


def lib():
    global _L
    if _L is None:
        _L = _lib()
    return _L


def set_threads(n):
    """threads for the sort / group scan / reducers (default 1); results do not depend on it"""
    lib().dto_set_threads(int(n))


def get_threads():
    return lib().dto_get_threads()


def stype_of(a, stype=None):
    if stype is not None:
        return stype
    return _NP2ST[a.dtype]


def _col(a, stype=None, desc=False):
    a = np.ascontiguousarray(a)
    st = stype_of(a, stype)
    if a.dtype == np.bool_:
        a = a.view(np.int8)
    return a, _Col(a.ctypes.data, st, 1 if desc else 0)


def group(keys, stypes=None, desc=None, na_last=False):
    """keys: list of numpy arrays (first = most significant).
    Returns (rowindex int32[n], offsets int32[ng+1])."""
    n = len(keys[0])
    keep, cols = [], (_Col * len(keys))()
    for i, k in enumerate(keys):
        a, c = _col(k, stypes[i] if stypes else None, bool(desc[i]) if desc else False)
        keep.append(a)
        cols[i] = c
    ri = np.empty(n, np.int32)
    off = np.empty(n + 1, np.int32)
    ng = C.c_int64(0)
    rc = lib().dto_group(cols, C.c_int(len(keys)), C.c_int64(n), C.c_int(1 if na_last else 0),
                         C.c_void_p(ri.ctypes.data), C.c_void_p(off.ctypes.data), C.byref(ng))
    if rc != 0:
        raise ValueError("dto_group failed")
    return ri, off[:ng.value + 1].copy()


def reduce(op, values, ri, offsets, stype=None):
    """op: name or code. values: numpy array or None (count0). Returns numpy array[ng]
    in the reducer's output stype with NA stored as the stype sentinel."""
    opc = OPS[op] if isinstance(op, str) else op
    ng = len(offsets) - 1
    offsets = np.ascontiguousarray(offsets, np.int32)
    rip = None
    if ri is not None:
        ri = np.ascontiguousarray(ri, np.int32)
        rip = C.c_void_p(ri.ctypes.data)
    if opc == COUNT0:
        out = np.empty(ng, np.int64)
        lib().dto_reduce(C.c_int(opc), None, rip, C.c_void_p(offsets.ctypes.data), C.c_int64(ng),
                         C.c_void_p(out.ctypes.data))
        return out
    a, c = _col(values, stype)
    ost = lib().dto_reduce_out_stype(C.c_int(opc), C.c_int(c.stype))
    out = np.empty(ng, _ST2NP[ost])
    lib().dto_reduce(C.c_int(opc), C.byref(c), rip, C.c_void_p(offsets.ctypes.data), C.c_int64(ng),
                     C.c_void_p(out.ctypes.data))
    return out


def bool_to_rowindex(mask):
    m = np.ascontiguousarray(mask)
    if m.dtype == np.bool_:
        m = m.view(np.int8)
    out = np.empty(len(m), np.int32)
    k = lib().dto_bool_to_rowindex(C.c_void_p(m.ctypes.data), C.c_int64(len(m)), C.c_void_p(out.ctypes.data))
    return out[:k].copy()


CMP = {">": 0, ">=": 1, "<": 2, "<=": 3, "==": 4, "!=": 5}


def filter_cmp(values, cmp, scalar, stype=None):
    a, c = _col(values, stype)
    out = np.empty(len(a), np.int32)
    isf = c.stype in (FLOAT32, FLOAT64)
    k = lib().dto_filter_cmp(C.addressof(c), len(a), CMP[cmp], float(scalar), 0 if isf else int(scalar),
                             out.ctypes.data)
    return out[:k].copy()


def gather(values, ri, stype=None):
    a, c = _col(values, stype)
    ri = np.ascontiguousarray(ri, np.int32)
    out = np.empty(len(ri), a.dtype)
    lib().dto_gather(C.byref(c), C.c_void_p(ri.ctypes.data), C.c_int64(len(ri)), C.c_void_p(out.ctypes.data))
    return out


# ---- group-wise operators sharing the Groupby (dt_oracle_groupwise.c) --------------------------
SD, MEDIAN, NUNIQUE = 8, 9, 10
COV, CORR = 0, 1
CUMSUM, CUMPROD, CUMMIN, CUMMAX, CUMCOUNT, NGROUP, FILLNA = 0, 1, 2, 3, 4, 5, 6
OPSX = {"sd": SD, "median": MEDIAN, "nunique": NUNIQUE}
OPS2 = {"cov": COV, "corr": CORR}
CUMOPS = {"cumsum": CUMSUM, "cumprod": CUMPROD, "cummin": CUMMIN, "cummax": CUMMAX, "cumcount": CUMCOUNT,
          "ngroup": NGROUP, "fillna": FILLNA}


def _ri_off(ri, offsets):
    offsets = np.ascontiguousarray(offsets, np.int32)
    rip = None
    if ri is not None:
        ri = np.ascontiguousarray(ri, np.int32)
        rip = C.c_void_p(ri.ctypes.data)
    return ri, rip, offsets


def reducex(op, values, ri, offsets, stype=None):
    """sd / median / nunique per group (NA = the output stype's sentinel)."""
    opc = OPSX[op] if isinstance(op, str) else op
    ri, rip, offsets = _ri_off(ri, offsets)
    ng = len(offsets) - 1
    a, c = _col(values, stype)
    ost = lib().dto_reducex_out_stype(C.c_int(opc), C.c_int(c.stype))
    out = np.empty(ng, _ST2NP[ost])
    lib().dto_reducex(C.c_int(opc), C.byref(c), rip, C.c_void_p(offsets.ctypes.data), C.c_int64(ng),
                      C.c_void_p(out.ctypes.data))
    return out


def reduce2(op, va, vb, ri, offsets, stypes=(None, None)):
    """cov / corr per group."""
    opc = OPS2[op] if isinstance(op, str) else op
    ri, rip, offsets = _ri_off(ri, offsets)
    ng = len(offsets) - 1
    a, ca = _col(va, stypes[0])
    b, cb = _col(vb, stypes[1])
    ost = lib().dto_reduce2_out_stype(C.c_int(ca.stype), C.c_int(cb.stype))
    out = np.empty(ng, _ST2NP[ost])
    lib().dto_reduce2(C.c_int(opc), C.byref(ca), C.byref(cb), rip, C.c_void_p(offsets.ctypes.data), C.c_int64(ng),
                      C.c_void_p(out.ctypes.data))
    return out


def cumulate(op, values, ri, offsets, reverse=False, stype=None):
    """cumsum / cumprod / cummin / cummax / fillna / cumcount / ngroup inside groups; output in grouped order."""
    opc = CUMOPS[op] if isinstance(op, str) else op
    ri, rip, offsets = _ri_off(ri, offsets)
    ng = len(offsets) - 1
    n = int(offsets[-1]) if ng >= 0 and len(offsets) else 0
    if opc in (CUMCOUNT, NGROUP):
        out = np.empty(n, np.int64)
        lib().dto_cumulate(C.c_int(opc), None, rip, C.c_void_p(offsets.ctypes.data), C.c_int64(ng),
                           C.c_int(1 if reverse else 0), C.c_void_p(out.ctypes.data))
        return out
    a, c = _col(values, stype)
    ost = lib().dto_cumulate_out_stype(C.c_int(opc), C.c_int(c.stype))
    out = np.empty(n, _ST2NP[ost])
    lib().dto_cumulate(C.c_int(opc), C.byref(c), rip, C.c_void_p(offsets.ctypes.data), C.c_int64(ng),
                       C.c_int(1 if reverse else 0), C.c_void_p(out.ctypes.data))
    return out


# ---- set functions and natural join (dt_oracle_sets.c) --------------------------------------------
SETOPS = {"union": 0, "intersect": 1, "setdiff": 2, "symdiff": 3}


def setop(op, columns, stype=None):
    """columns: list of numpy arrays of ONE stype (the sources).  Returns the row indices into their
    concatenation of the result elements, ascending by value (NA first)."""
    opc = SETOPS[op] if isinstance(op, str) else op
    stacked = np.concatenate([np.ascontiguousarray(c) for c in columns]) if columns else np.zeros(0, np.int32)
    a, c = _col(stacked, stype)
    cum = np.cumsum([len(x) for x in columns]).astype(np.int64)
    out = np.empty(len(a), np.int32)
    lib().dto_setop.restype = C.c_int64
    k = lib().dto_setop(C.c_int(opc), C.byref(c), C.c_void_p(cum.ctypes.data), C.c_int(len(columns)), C.c_int64(len(a)),
                        C.c_void_p(out.ctypes.data))
    return out[:k].copy()


def join_index(xkeys, jkeys, xstypes=None, jstypes=None):
    """per row of X the row of the keyed frame J (key columns sorted ascending, unique) with equal key
    values, or INT32_MIN"""
    nk = len(xkeys)
    xs, js = (_Col * nk)(), (_Col * nk)()
    keep = []
    for i in range(nk):
        a, c = _col(xkeys[i], xstypes[i] if xstypes else None)
        b, d = _col(jkeys[i], jstypes[i] if jstypes else None)
        keep += [a, b]
        xs[i], js[i] = c, d
    out = np.empty(len(keep[0]), np.int32)
    rc = lib().dto_join_index(xs, js, C.c_int(nk), C.c_int64(len(keep[0])), C.c_int64(len(keep[1])), C.c_void_p(out.ctypes.data))
    if rc != 0:
        raise ValueError("dto_join_index failed")
    return out


def arrow_to_sentinel(values, validity, nrows, stype):
    """The column an Arrow-layout column materialises to in the reference: element i is valid when there is no validity
    bitmap or bit `validity[i / 8] & (1 << (i & 7))` is set (ArrowFw_ColumnImpl::_get, src/core/column/arrow_fw.cc:63-72);
    booleans are bit-packed the same way (ArrowBool_ColumnImpl::get_element, column/arrow_bool.cc); an invalid element
    reads as the stype's NA sentinel once materialised (ColumnImpl::_materialize_fw, column_impl.cc:78-101 writes
    GETNA<T>(), stype.h:186-197).  numpy restatement; TEST INFRASTRUCTURE ONLY."""
    i = np.arange(nrows)
    valid = np.ones(nrows, bool) if validity is None else ((np.asarray(validity, np.uint8)[i >> 3] >> (i & 7)) & 1).astype(bool)
    if stype == BOOL:
        data = ((np.asarray(values, np.uint8)[i >> 3] >> (i & 7)) & 1).astype(np.int8)
    else:
        data = np.asarray(values)[:nrows].astype(_ST2NP[stype], copy=True)
    out = data.copy()
    if stype in (FLOAT32, FLOAT64):
        out[~valid] = np.nan
    else:
        out[~valid] = np.iinfo(_ST2NP[stype]).min
    return out
