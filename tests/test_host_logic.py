"""CPU: host-side logic of the Frame mirror that needs no GPU -- result-name mangling / auto-naming
(src/core/frame/names.cc:455-607) and the translation of `column <cmp> scalar` filters."""
import json
import os

import pytest

from conftest import ROOT
from datatable_amd import _lib as L
from datatable_amd.engine import CMP, _cmp_args
from datatable_amd.frame import _mangle


@pytest.mark.parametrize("names,want", [
    (["k", "v", "v", "v"], ["k", "v", "v.0", "v.1"]),                  # no trailing digits: ".<k>"
    (["k0", "v1", "v1", "count"], ["k0", "v1", "v2", "count"]),        # trailing digits: count on from them
    (["v1", "v0", "v1", "v0"], ["v1", "v0", "v2", "v3"]),
    (["v3", "v3", "v3"], ["v3", "v4", "v5"]),
    (["v1", "v1", "v2"], ["v1", "v2", "v3"]),                          # a later original collides with a mangled name
    (["a.", "a."], ["a.", "a.0"]),
    (["k", "", ""], ["k", "C0", "C1"]),                                # unnamed -> C<k>
    (["C1", "", "v"], ["C1", "C2", "v"]),                              # ... counting on from the largest C<num>
    (["k", "", "C0", ""], ["k", "C1", "C0", "C2"]),
])
def test_mangle(names, want):
    assert _mangle(list(names)) == want


def test_mangle_matches_every_reference_result():
    """the names of every golden result frame are a fixed point of the mangling (the reference produced them
    with the same rule, so re-applying it must not change them)"""
    for fn in ("frame_queries.json", "frame_fuzz.json", "frame_fuzz2.json", "frame_fuzz4.json"):
        for q in json.load(open(os.path.join(ROOT, "tests", "golden", fn)))["queries"]:
            assert _mangle(list(q["names"])) == q["names"], (fn, q["query"])


def test_cmp_args():
    i32, f64 = L.INT32, L.FLOAT64
    assert _cmp_args(">", 2, i32) == (CMP[">"], 2.0, 2)
    assert _cmp_args(">", 1.5, f64) == (CMP[">"], 1.5, 0)
    # integer column vs non-integral scalar: compared as numbers
    assert _cmp_args(">", 1.5, i32) == (CMP[">="], 1.5, 2)
    assert _cmp_args(">=", -1.5, i32) == (CMP[">="], -1.5, -1)
    assert _cmp_args("<", 1.5, i32) == (CMP["<="], 1.5, 1)
    assert _cmp_args("<=", -1.5, i32) == (CMP["<="], -1.5, -2)
    assert _cmp_args("==", 1.5, i32) == (CMP["=="], 1.5, -2**63)        # no valid element equals the NA sentinel
    assert _cmp_args("!=", 1.5, i32) == (CMP["!="], 1.5, -2**63)
    assert _cmp_args("==", True, L.BOOL) == (CMP["=="], 1.0, 1)


def test_filter_scalars_outside_int64_do_not_wrap():
    """`DT[f.x < 1e19, :]` on an integer column: a scalar beyond int64 decides the comparison by its sign
    (ctypes would wrap it silently): every valid row / no row; != keeps every row, == none"""
    from datatable_amd.engine import _cmp_args, CMP
    from datatable_amd import _lib as L
    big, small = 10**19, -10**19
    assert _cmp_args("<", big, L.INT64)[0] == L.NOTNA and _cmp_args("<=", big, L.INT32)[0] == L.NOTNA
    assert _cmp_args(">", small, L.INT64)[0] == L.NOTNA and _cmp_args(">=", small, L.INT64)[0] == L.NOTNA
    for cmp, s in ((">", big), (">=", big), ("<", small), ("<=", small)):
        code, _, ci = _cmp_args(cmp, s, L.INT64)
        assert (code, ci) == (CMP[">"], 2**63 - 1)          # x > INT64_MAX: never
    assert _cmp_args("==", big, L.INT64)[::2] == (CMP["=="], -2**63)
    assert _cmp_args("!=", small, L.INT64)[::2] == (CMP["!="], -2**63)
    assert _cmp_args("<", 2**63 - 1, L.INT64) == (CMP["<"], float(2**63 - 1), 2**63 - 1)     # in range: unchanged
