"""Host logic of the MSD levels of the sort path (datatable_amd/csrc/msd_plan.hpp), compiled with g++ and driven on the
CPU: the split of the key bits into two scatter digits and the final digit, the overflow forecast from the digit
histograms, and the ragged level-2 tiles -- they cover every row exactly once, never span two level-1 buckets, all but the
first tile of a bucket start 16-byte aligned, and the histogram groups stay inside one bucket."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = 8192


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("mp") / "libmsdplan.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Werror",
                           os.path.join(ROOT, "tests", "cpp", "msd_plan_harness.cpp"), "-o", so])
    return C.CDLL(so)


def split(lib, n, bits, bucket_rows=2048, rbmax=9, tile=TILE):
    s1, s2, rb = C.c_int(0), C.c_int(0), C.c_int(0)
    ok = lib.mp_split(C.c_longlong(n), bits, tile, bucket_rows, rbmax, C.byref(s1), C.byref(s2), C.byref(rb))
    return bool(ok), s1.value, s2.value, rb.value


def test_split_of_the_baseline_shapes(lib):
    assert split(lib, 499_988_561, 27) == (True, 9, 9, 9)            # C5: 5e8 passing rows, keys in [0, 1e8)
    assert split(lib, 1_000_000_000, 24) == (True, 9, 9, 6)          # dthip_groupby on C3's keys
    assert split(lib, 1_000_000_000, 28)[0] is False                 # 10 bits would be left for the final level
    assert split(lib, 1_000_000_000, 28, rbmax=10) == (True, 9, 9, 10)
    assert split(lib, 2_000_000_000, 27)[0] is False                 # 7629 rows per bucket on average: no room for skew
    assert split(lib, 100_000_000, 16)[0] is False                   # as many scatter bits as key bits: nothing left to order
    assert split(lib, 100_000_000, 17) == (True, 8, 8, 1)
    assert split(lib, 1, 20)[0] is False and split(lib, 0, 20)[0] is False


@pytest.mark.parametrize("n", [10, 1000, 300_000, 67_108_864, 1_200_000_000])
@pytest.mark.parametrize("bucket_rows", [1, 64, 700, 2048, 4096])
def test_split_invariants(lib, n, bucket_rows):
    for bits in range(2, 33):
        ok, s1, s2, rb = split(lib, n, bits, bucket_rows)
        if not ok:
            continue
        assert 1 <= s2 <= s1 <= 9 and s1 - s2 <= 1 and 1 <= rb <= 9 and s1 + s2 + rb == bits
        assert (n >> (s1 + s2)) <= max(bucket_rows, TILE * 9 // 16)
        assert (n >> (s1 + s2)) <= TILE * 9 // 16


def test_overflow_forecast(lib):
    def f(h1, h2, n):
        a, b = np.asarray(h1, np.uint32), np.asarray(h2, np.uint32)
        return bool(lib.mp_overflow(a.ctypes.data_as(C.c_void_p), len(a), b.ctypes.data_as(C.c_void_p), len(b), C.c_longlong(n), TILE))
    n = 1_500_000
    flat = [n // 64] * 64
    assert not f(flat, flat, n)                                      # uniform digits: 366 rows per cell
    few = [n // 2, n // 2] + [0] * 62
    assert f(few, [n // 3] * 3 + [0] * 61, n)                        # 6 cells for 1.5e6 rows: certain
    hot = [n // 2] + [n // 126] * 63
    assert f(hot, hot, n)                                            # one hot key: likely
    assert f([0] * 64, flat, n)                                      # (no rows at all in a digit: treated as overflow)
    assert not f(flat, [n // 32] * 32, 67_108_864 // 32)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("tpg", [1, 3, 60])
def test_level2_tiles_cover_every_row_once(lib, seed, tpg):
    rng = np.random.default_rng(seed)
    nb = [1, 2, 16, 512][seed % 4]
    kind = seed % 3
    if kind == 0:
        sizes = rng.integers(0, 40_000, nb)
    elif kind == 1:
        sizes = np.where(rng.random(nb) < 0.5, 0, rng.integers(1, 9000, nb))       # empty buckets in between
    else:
        sizes = np.full(nb, TILE)                                                   # exactly one tile each
        sizes[0] = 5
    sizes = sizes.astype(np.uint32)
    total = int(sizes.sum())
    cap = total // TILE + 2 * nb + 2
    tdesc = np.zeros(4 * cap, np.uint32); gdesc = np.zeros(2 * cap, np.uint32); gfirst = np.zeros(nb + 1, np.uint32)
    ng = C.c_int(0)
    nt = lib.mp_tiles(sizes.ctypes.data_as(C.c_void_p), nb, TILE, tpg, tdesc.ctypes.data_as(C.c_void_p),
                      gdesc.ctypes.data_as(C.c_void_p), gfirst.ctypes.data_as(C.c_void_p), C.byref(ng))
    t = tdesc[:4 * nt].reshape(nt, 4).astype(np.int64)
    g = gdesc[:2 * ng.value].reshape(ng.value, 2).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    # every row exactly once, in order
    assert t[:, 1].sum() == total and np.all(t[:, 1] >= 1) and np.all(t[:, 1] <= TILE)
    assert np.array_equal(t[:, 0], np.concatenate([[0], np.cumsum(t[:, 1])[:-1]]))
    for first, rows, grp, b in t:
        assert starts[b] <= first and first + rows <= starts[b + 1]                # inside its level-1 bucket
        assert g[grp, 0] <= np.flatnonzero((t[:, 0] == first))[0] < g[grp, 0] + g[grp, 1]
    # all but the first tile of a bucket start on a 16-byte boundary of a 4-byte array
    for b in range(nb):
        tb = t[t[:, 3] == b]
        assert np.all(tb[1:, 0] % 4 == 0)
        assert len(tb) == 0 or tb[0, 0] == starts[b]
    # groups: consecutive tiles of ONE bucket, at most tpg of them; gfirst delimits the groups of every bucket
    assert g[:, 1].sum() == nt and np.all(g[:, 1] <= tpg) and np.all(g[:, 1] >= 1)
    for gi, (t0, cnt) in enumerate(g):
        assert len(set(t[t0:t0 + cnt, 3])) == 1 and np.all(t[t0:t0 + cnt, 2] == gi)
    assert gfirst[0] == 0 and gfirst[nb] == ng.value and np.all(np.diff(gfirst.astype(np.int64)) >= 0)
    for b in range(nb):
        for gi in range(gfirst[b], gfirst[b + 1]):
            assert t[g[gi, 0], 3] == b
