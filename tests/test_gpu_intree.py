"""The IN-TREE bindings at HEAD, inside the default `pytest -m gpu` run.

integration/_dt_hip is the reference itself with integration/patches/*.patch applied (integration/build_dt_hip.sh, run by
__graft_entry__.build()): its `group()` (src/core/sort.cc:1411-1495, sort.h:56-58) calls dthip_groupby and its
sum / mean / min / max / count reducer columns (expr/fexpr_reduce_unary.cc:32-69) call dthip_reduce on the device, the
grouped view peeled into stored column + RowIndex (column/view.cc:140-196).  Here:

  * the reference's OWN test files for grouping and reducers (tests/test-groups.py, tests/test-reduce.py) run on that build
    with DTHIP_LIB = the library of this tree, and the build's exit report must show that group() calls and reducer columns
    were really served by libdthip.so, views peeled, and the RowIndex S-grp left on the device reused;
  * the same statements on the patched build (GPU) and on the unmodified build (oracle/_ref, CPU), a process each: names,
    stypes, keys, counts, min / max bit-exact, float sums / means within 1e-6 (float32 sums bit-exact: f32_sum = 1).
"""
import ctypes
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_close, assert_same

pytestmark = pytest.mark.gpu

PKG = os.path.join(ROOT, "integration", "_dt_hip")
REF = os.path.join(ROOT, "oracle", "_ref")
LIB = os.path.join(ROOT, "datatable_amd", "libdthip.so")


def _need_builds():
    assert glob.glob(os.path.join(PKG, "datatable", "lib", "_datatable*.so")), \
        "integration/_dt_hip is not built: run integration/build_dt_hip.sh (or __graft_entry__.build()) where /root/reference exists"
    assert os.path.exists(LIB), "datatable_amd/libdthip.so is not built"


def _env():
    env = dict(os.environ)
    env.update(DTHIP_LIB=LIB, DTHIP_SGRP_REPORT="1", PYTHONPATH=PKG, TMPDIR="/tmp")
    env.pop("DTHIP_NO_SRED", None)
    return env


def _report(stderr):
    g = re.search(r"\[dthip S-grp\] group\(\) calls served by libdthip\.so: (\d+)", stderr)
    r = re.search(r"\[dthip S-red\] reducer columns computed by libdthip\.so: (\d+) \(views peeled: (\d+), RowIndex already on "
                  r"the device: (\d+), materialised on the CPU first: (\d+)\)", stderr)
    assert g and r, "no seam report on stderr:\n" + stderr[-2000:]
    return int(g.group(1)), tuple(int(x) for x in r.groups())


def test_patched_build_maps_into_this_process():
    """the patched extension module itself is loadable here and is the reference's C API (src/datatable/include/datatable.h:32)"""
    _need_builds()
    so = glob.glob(os.path.join(PKG, "datatable", "lib", "_datatable*.so"))[0]
    h = ctypes.CDLL(so, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    h.DtABIVersion.restype = ctypes.c_size_t
    assert h.DtABIVersion() == 2
    prov = open(os.path.join(PKG, "PROVENANCE.txt")).read()
    import hashlib
    patches = b"".join(open(p, "rb").read() for p in sorted(glob.glob(os.path.join(ROOT, "integration", "patches", "*.patch"))))
    assert hashlib.sha256(patches).hexdigest()[:16] in prov, "integration/_dt_hip was built from other patches than the tree holds"


def test_reference_own_group_and_reduce_tests_on_the_patched_build():
    _need_builds()
    cmd = [sys.executable, "-m", "pytest", "tests/test-groups.py", "tests/test-reduce.py", "-q", "-p", "no:cacheprovider",
           "-o", "python_files=test*.py", "-o", "addopts="]
    p = subprocess.run(cmd, cwd=os.path.join(PKG, "ref"), env=_env(), capture_output=True, text=True, timeout=1200)
    tail = p.stdout[-1500:] + "\n" + p.stderr[-1500:]
    assert p.returncode == 0, tail
    m = re.search(r"(\d+) passed", p.stdout)
    assert m and int(m.group(1)) >= 200, tail
    assert " failed" not in p.stdout, tail
    groups, (reds, peeled, kept, mat) = _report(p.stderr)
    assert groups > 0 and reds > 0, (groups, reds)
    assert peeled > 0 and kept > 0, (peeled, kept)


@pytest.mark.parametrize("rows", [2_000_000])
def test_patched_build_on_gpu_equals_unmodified_build_on_cpu(tmp_path, rows):
    _need_builds()
    assert glob.glob(os.path.join(REF, "datatable", "lib", "_datatable*.so")), "oracle/_ref is not built (oracle/build_ref.sh)"
    worker = os.path.join(ROOT, "tests", "intree_worker.py")
    out_gpu, out_cpu = str(tmp_path / "gpu.npz"), str(tmp_path / "cpu.npz")
    p = subprocess.run([sys.executable, worker, PKG, out_gpu, str(rows)], env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    groups, (reds, peeled, kept, mat) = _report(p.stderr)
    # 5 statements: group() served for every by(); 15 + 2 + 4 + 4 + 2 reducer columns over fixed-width columns
    assert groups >= 5 and reds >= 27, (groups, reds)
    # by_k and by_a_k read their columns through the RowIndex S-grp just produced (peeled, and already on the device); the
    # filtered frame's columns are views of views (materialised first), sort() + by() hands over views of its own RowIndex
    assert peeled >= 15 + 2 and kept >= 15 + 2, (peeled, kept, mat)
    env = dict(os.environ); env.pop("DTHIP_LIB", None); env["PYTHONPATH"] = REF
    q = subprocess.run([sys.executable, worker, REF, out_cpu, str(rows)], env=env, capture_output=True, text=True, timeout=900)
    assert q.returncode == 0, q.stderr[-3000:]
    G, C = np.load(out_gpu), np.load(out_cpu)
    assert sorted(G.files) == sorted(C.files)
    checked = 0
    for name in sorted(C.files):
        g, c = G[name], C[name]
        if name.endswith("/names") or name.endswith("/stypes") or name.endswith(".na"):
            assert np.array_equal(g, c), name
            continue
        q_name, col = name.split("/")
        cname = str(C[q_name + "/names"][int(col)])
        stype = str(C[q_name + "/stypes"][int(col)])
        # float64 sums / means are re-associated on the GPU (<= 1e-6, BASELINE.json); everything else is bit-exact,
        # float32 sums included (the binding sets f32_sum = 1: column/sumprod.h:48-55's accumulation order)
        if name == "by_k/2":
            assert stype == "float32" and cname == "w"
            assert_same(g, c, what="sum(float32) " + name)
        elif stype in ("float64", "float32") and not np.array_equal(g, c, equal_nan=True):
            assert_close(g, c, rel=1e-6, what=name + " " + cname)
        else:
            assert_same(g, c, what=name + " " + cname)
        checked += 1
    assert checked >= 27 + 4
