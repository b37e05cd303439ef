"""The C ABI from plain C (examples/c_client.c): compiled with gcc against include/dthip.h and
datatable_amd/libdthip.so only -- no Python, no PyTorch, no HIP headers on the client side -- and run on
the GPU: fused groupby-aggregate (host and device-resident data), filter -> gather -> groupby_rows, every
result compared with scalar loops inside the program (exit status 0 = all equal)."""
import os
import subprocess

import pytest

from conftest import ROOT


def _compile(tmp_path, name="c_client"):
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "datatable_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-o", exe, "-L" + libdir, "-ldthip",
                           "-Wl,-rpath," + libdir, "-lm"])
    return exe


def test_c_client_builds_against_the_header(tmp_path):
    """CPU: the header is valid C (not only C++) and the library exports what the client links against"""
    assert os.path.exists(_compile(tmp_path))
    assert os.path.exists(_compile(tmp_path, "c_client_sharded"))


@pytest.mark.gpu
@pytest.mark.parametrize("nrows", [1000, 3_000_000])
def test_c_client_runs(tmp_path, nrows):
    p = subprocess.run([_compile(tmp_path), str(nrows)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count(" 0 mismatches") == 3, p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_c_client_sharded_logical_shards(tmp_path, world):
    """examples/c_client_sharded.c: `world` logical shards in one C process (dthip_comm_init_local), all on this box's GPU"""
    p = subprocess.run([_compile(tmp_path, "c_client_sharded"), "local", str(world), "2000000"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 mismatches" in p.stdout, p.stdout


@pytest.mark.gpu
def test_c_client_sharded_rccl(tmp_path):
    """the RCCL mode from plain C.  One process per GPU: as many ranks as this box has GPUs (1 on the test boxes -- a
    1-rank communicator still runs ncclCommInitRank, ncclAllGather and the grouped ncclSend / ncclRecv)."""
    exe = _compile(tmp_path, "c_client_sharded")
    ndev = 1
    try:
        import ctypes
        ndev = max(1, ctypes.CDLL(os.path.join(ROOT, "datatable_amd", "libdthip.so")).dthip_device_count())
    except OSError:
        pass
    world = min(ndev, 8)
    idfile = str(tmp_path / "comm.id")
    procs = [subprocess.Popen([exe, "rccl", str(world), str(r), idfile, "2000000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and " 0 mismatches" in o, o
