"""The C ABI from plain C (examples/c_client.c): compiled with gcc against include/dthip.h and
datatable_amd/libdthip.so only -- no Python, no PyTorch, no HIP headers on the client side -- and run on
the GPU: fused groupby-aggregate (host and device-resident data), filter -> gather -> groupby_rows, every
result compared with scalar loops inside the program (exit status 0 = all equal)."""
import os
import subprocess

import pytest

from conftest import ROOT


def _compile(tmp_path):
    exe = str(tmp_path / "c_client")
    libdir = os.path.join(ROOT, "datatable_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_client.c"), "-o", exe, "-L" + libdir, "-ldthip",
                           "-Wl,-rpath," + libdir, "-lm"])
    return exe


def test_c_client_builds_against_the_header(tmp_path):
    """CPU: the header is valid C (not only C++) and the library exports what the client links against"""
    assert os.path.exists(_compile(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("nrows", [1000, 3_000_000])
def test_c_client_runs(tmp_path, nrows):
    p = subprocess.run([_compile(tmp_path), str(nrows)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count(" 0 mismatches") == 3, p.stdout
