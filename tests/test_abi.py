"""CPU: the C-ABI library loads, exports every symbol include/dthip.h declares, and
refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dthip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dthip_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("dthip_groupby", "dthip_groupby_agg", "dthip_reduce", "dthip_bool_to_rowindex",
                 "dthip_filter_cmp", "dthip_gather", "dthip_init", "dthip_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from datatable_amd import _lib
    lib = _lib.load()
    for s in declared_symbols():
        assert hasattr(lib, s), "libdthip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "%s has no ctypes signature in datatable_amd/_lib.py" % s
    assert set(_lib.SIGNATURES) == set(declared_symbols())
    assert lib.dthip_abi_version() == 2


def test_reduce_out_stype_rules():
    # fexpr_sumprod.cc:47-66, fexpr_mean.cc:45-74, fexpr_minmax.cc:47-68, fexpr_count.cc
    from datatable_amd import _lib as L
    lib = L.load()
    for st in (L.BOOL, L.INT8, L.INT16, L.INT32, L.INT64):
        assert lib.dthip_reduce_out_stype(L.SUM, st) == L.INT64
        assert lib.dthip_reduce_out_stype(L.MEAN, st) == L.FLOAT64
        assert lib.dthip_reduce_out_stype(L.MIN, st) == st
        assert lib.dthip_reduce_out_stype(L.COUNT, st) == L.INT64
    assert lib.dthip_reduce_out_stype(L.SUM, L.FLOAT32) == L.FLOAT32
    assert lib.dthip_reduce_out_stype(L.MEAN, L.FLOAT32) == L.FLOAT32
    assert lib.dthip_reduce_out_stype(L.SUM, L.FLOAT64) == L.FLOAT64
    assert lib.dthip_reduce_out_stype(L.COUNT0, L.FLOAT64) == L.INT64


def test_no_cpu_fallback_without_gpu():
    from datatable_amd import _lib as L
    lib = L.load()
    if lib.dthip_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.dthip_init(0, None, C.byref(h))
    assert rc == L.EDEVICE
    assert b"no CPU fallback" in lib.dthip_last_error()
    from datatable_amd.engine import Context
    with pytest.raises(RuntimeError):
        Context(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "datatable_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower() or f in (), "%s mentions the oracle" % f
