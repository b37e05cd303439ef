"""ctypes binding of libdthip.so -- the C ABI declared in include/dthip.h.

The library is the product: there is no CPU fallback.  Importing this module
when libdthip.so has not been built raises ImportError, and creating a
Context without a HIP device raises RuntimeError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DTHIP_LIB") or os.path.join(_HERE, "libdthip.so")   # DTHIP_LIB: an alternative build, for A/B measurements

# stype codes == the reference's SType values (src/core/stype.h:41-62)
BOOL, INT8, INT16, INT32, INT64, FLOAT32, FLOAT64 = 1, 2, 3, 4, 5, 6, 7
SUM, MEAN, MIN, MAX, COUNT, COUNT0, FIRST, LAST = 0, 1, 2, 3, 4, 5, 6, 7
SD, MEDIAN, NUNIQUE, PROD, COUNTNA = 8, 9, 10, 11, 12   # dthip_reduce only
COV, CORR = 0, 1                                    # enum dthip_op2
CUMSUM, CUMPROD, CUMMIN, CUMMAX, CUMCOUNT, NGROUP, FILLNA = 0, 1, 2, 3, 4, 5, 6   # enum dthip_cumop
UNION, INTERSECT, SETDIFF, SYMDIFF = 0, 1, 2, 3     # enum dthip_setfn
HOST, DEVICE = 0, 1
ABI_VERSION = 7                                     # DTHIP_ABI_VERSION of include/dthip.h
NA_FIRST, NA_LAST, NA_REMOVE = 0, 1, 2
FLAG_DESCENDING = 1
GT, GE, LT, LE, EQ, NE, NOTNA, ISNA = 0, 1, 2, 3, 4, 5, 6, 7

EINVAL, ENOTIMPL, ENOMEM, EDEVICE = -1, -2, -3, -4


class Col(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stype", C.c_int32), ("flags", C.c_int32)]


class Agg(C.Structure):
    _fields_ = [("op", C.c_int32), ("col", C.c_int32)]


# name -> (restype, argtypes); must list every symbol include/dthip.h declares
SIGNATURES = {
    "dthip_abi_version": (C.c_int, []),
    "dthip_build_id": (C.c_char_p, []),
    "dthip_last_error": (C.c_char_p, []),
    "dthip_device_count": (C.c_int, []),
    "dthip_init": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dthip_use_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dthip_destroy": (C.c_int, [C.c_void_p]),
    "dthip_sync": (C.c_int, [C.c_void_p]),
    "dthip_trim": (C.c_int, [C.c_void_p]),
    "dthip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "dthip_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dthip_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dthip_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dthip_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dthip_host_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dthip_host_unregister": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dthip_timer_start": (C.c_int, [C.c_void_p]),
    "dthip_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "dthip_last_call_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "dthip_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "dthip_profile_reset": (C.c_int, [C.c_void_p]),
    "dthip_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dthip_profile_names": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "dthip_groupby": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                C.POINTER(C.c_void_p)]),
    "dthip_groupby_agg": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int, C.POINTER(Col), C.c_int,
                                    C.POINTER(Agg), C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dthip_groupby_rows": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int, C.POINTER(Col), C.c_int, C.c_int64, C.c_int,
                                     C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dthip_comm_last_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "dthip_result_col": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dthip_result_copy_col": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "dthip_result_ngroups": (C.c_int64, [C.c_void_p]),
    "dthip_result_nrows": (C.c_int64, [C.c_void_p]),
    "dthip_result_rowindex": (C.c_void_p, [C.c_void_p]),
    "dthip_result_offsets": (C.c_void_p, [C.c_void_p]),
    "dthip_result_key": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dthip_result_agg": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dthip_result_agg_stype": (C.c_int, [C.c_void_p, C.c_int]),
    "dthip_result_copy_rowindex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "dthip_result_copy_offsets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "dthip_result_copy_key": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "dthip_result_copy_agg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "dthip_result_group_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Col), C.c_int, C.c_void_p]),
    "dthip_result_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dthip_reduce_out_stype": (C.c_int, [C.c_int, C.c_int]),
    "dthip_reduce": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Col), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                               C.c_int, C.c_void_p]),
    "dthip_reduce2_out_stype": (C.c_int, [C.c_int, C.c_int]),
    "dthip_reduce2": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Col), C.POINTER(Col), C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_int64, C.c_int, C.c_void_p]),
    "dthip_cumulate_out_stype": (C.c_int, [C.c_int, C.c_int]),
    "dthip_cumulate": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Col), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                 C.c_int, C.c_int, C.c_void_p]),
    "dthip_filter_take": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int, C.c_double, C.c_int64, C.POINTER(Col), C.c_int, C.c_int64,
                                    C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "dthip_setop": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Col), C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                              C.POINTER(C.c_int64)]),
    "dthip_join_index": (C.c_int, [C.c_void_p, C.POINTER(Col), C.POINTER(Col), C.c_int, C.c_int64, C.c_int64, C.c_int,
                                   C.c_void_p]),
    "dthip_range_bucket": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dthip_ungroup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "dthip_bool_to_rowindex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                         C.POINTER(C.c_int64)]),
    "dthip_filter_cmp": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int64, C.c_int, C.c_double, C.c_int64, C.c_int,
                                   C.c_void_p, C.POINTER(C.c_int64)]),
    "dthip_gather": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dthip_filter_groupby_rows": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int, C.c_double, C.c_int64, C.POINTER(Col), C.c_int,
                                            C.POINTER(Col), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dthip_from_arrow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    # multi-GPU (comm.hip)
    "dthip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dthip_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dthip_comm_init_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dthip_comm_destroy": (C.c_int, [C.c_void_p]),
    "dthip_comm_rank": (C.c_int, [C.c_void_p]),
    "dthip_comm_world": (C.c_int, [C.c_void_p]),
    "dthip_sharded_groupby_agg": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int, C.POINTER(Col), C.c_int, C.POINTER(Agg), C.c_int,
                                            C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dthip_sharded_groupby_agg_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.POINTER(Col)), C.c_int,
                                                  C.POINTER(C.POINTER(Col)), C.c_int, C.POINTER(Agg), C.c_int,
                                                  C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dthip_sharded_groupby_rows": (C.c_int, [C.c_void_p, C.POINTER(Col), C.c_int, C.POINTER(Col), C.c_int, C.c_int64, C.c_int64,
                                             C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dthip_sharded_groupby_rows_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.POINTER(Col)), C.c_int,
                                                   C.POINTER(C.POINTER(Col)), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                   C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
}
COMM_ID_BYTES = 128
FLAG_NONA = 2

_lib = None


def load():
    """Load libdthip.so and bind every declared symbol.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libdthip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C datatable_amd/csrc`.  datatable_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.dthip_abi_version() != ABI_VERSION:
        raise ImportError("libdthip.so ABI version %d != %d" % (lib.dthip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class DthipError(RuntimeError):
    pass


_EXC = {EINVAL: ValueError, ENOTIMPL: NotImplementedError, ENOMEM: MemoryError, EDEVICE: DthipError}


def check(rc):
    """Map a DTHIP_E* code to the Python exception the reference would raise
    (api.cc:34-38 sets a Python exception and returns NULL/-1)."""
    if rc == 0:
        return
    msg = load().dthip_last_error().decode("utf-8", "replace")
    raise _EXC.get(rc, DthipError)(msg or ("dthip error %d" % rc))
