"""Host logic of the multi-GPU key-range partition (datatable_amd/csrc/split_plan.hpp: global key range, histogram bin
width, splitters at the world-quantiles of the summed 4096-bin histogram), compiled with g++ and driven on the CPU:
boundaries ascend, NA images fall on rank 0 (or the last rank when NAs sort last) by themselves, no rank before the
k-th boundary holds more than k / world of the valid keys, every rank's share stays within one histogram bin of its
fair share, extreme ranges (full 64 bits, a single value, no valid key) are handled."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("sp") / "libsplitplan.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Werror",
                           os.path.join(ROOT, "tests", "cpp", "split_plan_harness.cpp"), "-o", so])
    L = C.CDLL(so)
    L.sp_bounds.restype = C.c_int
    return L


def plan(lib, imgs, world, na_img=0, uneven=True, seed=0):
    imgs = np.ascontiguousarray(imgs, np.uint64)
    n = len(imgs)
    rng = np.random.default_rng(seed)
    cuts = np.array([0] + sorted(rng.integers(0, n + 1, world - 1).tolist()) + [n] if uneven else
                    [r * n // world for r in range(world + 1)], np.int64)
    bounds = np.zeros(max(world - 1, 1), np.uint64)
    shift, gmin = C.c_int(0), C.c_ulonglong(0)
    rc = lib.sp_bounds(imgs.ctypes.data_as(C.c_void_p), cuts.ctypes.data_as(C.c_void_p), world, C.c_ulonglong(na_img),
                       bounds.ctypes.data_as(C.c_void_p), C.byref(shift), C.byref(gmin))
    assert rc == 0
    return bounds[:world - 1], shift.value, gmin.value


def dest(bounds, imgs):
    return np.searchsorted(bounds, imgs, side="right")          # number of boundaries <= image


def image_i64(k):
    return (k.astype(np.int64).view(np.uint64)) ^ np.uint64(1 << 63)


@pytest.mark.parametrize("world", [1, 2, 3, 8, 127])
@pytest.mark.parametrize("kind", ["uniform", "skew", "wide", "clustered"])
def test_splitters_balance(lib, world, kind):
    rng = np.random.default_rng(world * 7 + len(kind))
    n = 200_000
    if kind == "uniform":
        k = rng.integers(-10**6, 10**6, n)
    elif kind == "skew":
        k = (rng.random(n) ** 8 * 1e12).astype(np.int64)
    elif kind == "wide":
        k = rng.integers(-2**62, 2**62, n)
    else:
        k = np.concatenate([rng.integers(0, 1000, n // 2), rng.integers(10**15, 10**15 + 1000, n - n // 2)])
    img = image_i64(k)
    na = rng.random(n) < 0.02
    img[na] = 0                                                  # NA image (NA first)
    bounds, shift, gmin = plan(lib, img, world, na_img=0, seed=world)
    assert len(bounds) == world - 1
    assert np.all(bounds[1:] >= bounds[:-1]), "boundaries must ascend"
    d = dest(bounds, img)
    assert np.all(d[na] == 0), "NA keys (smallest image) belong to rank 0"
    valid = img[~na]
    total = len(valid)
    # heaviest histogram bin: the granularity the splitters cannot go below
    bins = ((valid - np.uint64(gmin)) >> np.uint64(shift)).astype(np.int64)
    heavy = np.bincount(bins, minlength=4096).max()
    dv = dest(bounds, valid)
    cum = np.cumsum(np.bincount(dv, minlength=world))
    for kk in range(1, world):
        assert cum[kk - 1] <= total * kk // world, "ranks < %d hold more than their share" % kk
        assert cum[kk - 1] >= total * kk // world - heavy, "ranks < %d are short by more than one bin" % kk
    # key-range partition: destinations are monotone in the key
    order = np.argsort(valid, kind="stable")
    assert np.all(np.diff(dv[order]) >= 0)


def test_na_last_and_degenerate_ranges(lib):
    rng = np.random.default_rng(1)
    k = rng.integers(0, 1000, 10_000)
    img = image_i64(k)
    img[:100] = np.uint64(2**64 - 1)                             # NAs sort last: image ~0
    bounds, _, _ = plan(lib, img, 4, na_img=2**64 - 1)
    assert np.all(dest(bounds, img[:100]) == 3), "NA keys go to the last rank when NAs sort last"
    # one single key value: everything lands on one rank, boundaries stay ordered
    img = image_i64(np.full(5000, 42))
    bounds, shift, gmin = plan(lib, img, 8)
    assert shift == 0 and gmin == int(img[0]) and np.all(bounds[1:] >= bounds[:-1])
    assert len(np.unique(dest(bounds, img))) == 1
    # no valid key at all: every boundary is ~0, NA rows (image 0) stay on rank 0
    img = np.zeros(1000, np.uint64)
    bounds, _, _ = plan(lib, img, 4)
    assert np.all(bounds == np.uint64(2**64 - 1)) and np.all(dest(bounds, img) == 0)
    # the full 64-bit range: shift 52, no overflow of the boundary arithmetic
    img = np.array([1, 2**63, 2**64 - 2] * 1000, np.uint64)
    bounds, shift, _ = plan(lib, img, 3)
    assert shift == 52 and np.all(bounds[1:] >= bounds[:-1])
    assert sorted(np.bincount(dest(bounds, img), minlength=3).tolist()) == [1000, 1000, 1000]


# ---- the aggregate path: splitters from every rank's 1024 local quantiles (sample_bounds) --------------------------
def sample_plan(lib, slices):
    """slices: one ASCENDING uint64 image array per rank"""
    world = len(slices)
    imgs = np.ascontiguousarray(np.concatenate(slices) if slices else np.zeros(0, np.uint64), np.uint64)
    cuts = np.array([0] + list(np.cumsum([len(s) for s in slices])), np.int64)
    bounds = np.zeros(max(world - 1, 1), np.uint64)
    assert lib.sp_sample_bounds(imgs.ctypes.data_as(C.c_void_p), cuts.ctypes.data_as(C.c_void_p), world,
                                bounds.ctypes.data_as(C.c_void_p)) == 0
    return bounds[:world - 1]


@pytest.mark.parametrize("world", [2, 3, 8, 64])
@pytest.mark.parametrize("kind", ["uniform", "skew", "wide", "uneven", "disjoint"])
def test_sample_splitters_balance(lib, world, kind):
    rng = np.random.default_rng(world * 13 + len(kind))
    n = 400_000
    if kind == "skew":
        k = (rng.random(n) ** 8 * 1e12).astype(np.int64)
    elif kind == "wide":
        k = rng.integers(-2**62, 2**62, n)
    else:
        k = rng.integers(0, 10**7, n)
    if kind == "uneven":
        cuts = [0] + sorted(rng.integers(0, n + 1, world - 1).tolist()) + [n]
    else:
        cuts = [r * n // world for r in range(world + 1)]
    if kind == "disjoint":       # every rank holds its own key range (already partitioned input)
        k = np.sort(k)
    slices = [np.unique(image_i64(k[cuts[r]:cuts[r + 1]])) for r in range(world)]     # partial groups: distinct, ascending
    b = sample_plan(lib, slices)
    assert np.all(b[1:] >= b[:-1])
    total = sum(len(s) for s in slices)
    got = np.zeros(world, np.int64)
    for s in slices:
        got += np.bincount(dest(b, s), minlength=world)
    assert got.sum() == total
    # every prefix of ranks holds its fair share to within one sample step per source (+ rounding)
    slack = sum(len(s) / 1024 + 1 for s in slices) + 2
    for kk in range(1, world):
        assert abs(int(got[:kk].sum()) - total * kk / world) <= slack, (kk, got, slack)


# ---- the rows path (round 5): splitters from every rank's stratified random sample of 4096 row images ---------------------
def row_sample_plan(lib, slices):
    """slices: one uint64 image array per rank, in ROW order (unsorted)"""
    world = len(slices)
    imgs = np.ascontiguousarray(np.concatenate(slices) if slices else np.zeros(0, np.uint64), np.uint64)
    cuts = np.array([0] + list(np.cumsum([len(s) for s in slices])), np.int64)
    bounds = np.zeros(max(world - 1, 1), np.uint64)
    assert lib.sp_row_sample_bounds(imgs.ctypes.data_as(C.c_void_p), cuts.ctypes.data_as(C.c_void_p), world,
                                    bounds.ctypes.data_as(C.c_void_p)) == 4096
    return bounds[:world - 1]


@pytest.mark.parametrize("world", [2, 3, 8, 64])
@pytest.mark.parametrize("kind", ["uniform", "skew", "wide", "uneven", "sorted", "periodic", "clustered"])
def test_row_sample_splitters_balance(lib, world, kind):
    """shares within a few standard errors of a random sample (sqrt(p (1 - p) / samples) of the total), whatever the row
    order: random, sorted (every rank holds its own key range), periodic with a period that divides the strata, clustered"""
    rng = np.random.default_rng(world * 17 + len(kind))
    n = 600_000
    if kind == "skew":
        k = (rng.random(n) ** 8 * 1e12).astype(np.int64)
    elif kind == "wide":
        k = rng.integers(-2**62, 2**62, n)
    elif kind == "periodic":
        k = (np.arange(n) % 4096) * 1000 + rng.integers(0, 10, n)
    elif kind == "clustered":
        k = (np.arange(n) // 5000) * 10_000 + rng.integers(0, 100, n)
    else:
        k = rng.integers(0, 10**7, n)
    if kind == "sorted":
        k = np.sort(k)
    if kind == "uneven":
        cuts = [0] + sorted(rng.integers(0, n + 1, world - 1).tolist()) + [n]
    else:
        cuts = [r * n // world for r in range(world + 1)]
    slices = [image_i64(k[cuts[r]:cuts[r + 1]]) for r in range(world)]
    b = row_sample_plan(lib, slices)
    assert np.all(b[1:] >= b[:-1])
    allimg = np.concatenate(slices)
    got = np.bincount(dest(b, allimg), minlength=world)
    assert got.sum() == n
    # destinations are monotone in the key
    order = np.argsort(allimg, kind="stable")
    assert np.all(np.diff(dest(b, allimg)[order]) >= 0)
    lib.sp_rows_recv_bound.restype = C.c_longlong
    bound = lib.sp_rows_recv_bound(C.c_longlong(n), world)
    nsamp = 4096 * sum(1 for s in slices if len(s))
    heavy = int(np.unique(allimg, return_counts=True)[1].max())           # one key is never cut: the granularity of any splitter
    for kk in range(1, world):
        p = kk / world
        se = n * (p * (1 - p) / nsamp) ** 0.5
        assert abs(int(got[:kk].sum()) - n * p) <= 6 * se + 4 * n / 4096 / world + 64 + heavy, (kind, kk, got[:kk].sum(), n * p, se)
    assert got.max() <= bound + heavy, "a share above the receive bound on well-behaved keys"


def test_row_sample_positions_and_heavy_keys(lib):
    lib.sp_row_sample_pos.restype = C.c_ulonglong
    for n in (1, 2, 100, 4095, 4096, 4097, 10**6, 2**31 - 1):
        pos = np.array([lib.sp_row_sample_pos(C.c_uint(i), C.c_ulonglong(n)) for i in range(4096)], np.int64)
        assert pos.min() >= 0 and pos.max() < n and np.all(np.diff(pos) >= 0)
        if n >= 4096:
            assert np.all(np.diff(pos) >= 0) and len(np.unique(pos)) == 4096          # one row per stratum
    # one key holding 70 % of the rows is never cut: its owner's share exceeds the bound, which the count matrix shows to
    # every rank (comm.hip then allocates exactly and runs the status round)
    rng = np.random.default_rng(3)
    n = 200_000
    k = np.where(rng.random(n) < 0.7, 5_000_000, rng.integers(0, 10**7, n))
    slices = [image_i64(k[r * n // 4:(r + 1) * n // 4]) for r in range(4)]
    b = row_sample_plan(lib, slices)
    got = np.bincount(dest(b, np.concatenate(slices)), minlength=4)
    lib.sp_rows_recv_bound.restype = C.c_longlong
    assert got.max() >= int((k == 5_000_000).sum()) and got.max() > lib.sp_rows_recv_bound(C.c_longlong(n), 4)
    assert got.min() > 0            # (round 6) ... and no rank is left without rows
    # empty ranks and a rank with fewer rows than samples
    b = row_sample_plan(lib, [np.zeros(0, np.uint64), image_i64(rng.integers(0, 1000, 50)), image_i64(rng.integers(0, 1000, 100_000))])
    assert np.all(b[1:] >= b[:-1])


def test_sample_splitters_edge_cases(lib):
    # no partial groups anywhere; one rank only; a single distinct key; the NA image (0) next to valid ones
    assert list(sample_plan(lib, [np.zeros(0, np.uint64)] * 4)) == [2**64 - 1] * 3
    b = sample_plan(lib, [np.zeros(0, np.uint64), np.arange(1, 5001, dtype=np.uint64), np.zeros(0, np.uint64)])
    assert np.all(b[1:] >= b[:-1])
    assert np.bincount(dest(b, np.arange(1, 5001, dtype=np.uint64)), minlength=3).max() <= 5000 // 3 + 8
    one = [np.array([77], np.uint64)] * 5
    b = sample_plan(lib, one)
    assert len(set(dest(b, np.array([77], np.uint64)).tolist())) == 1          # one key -> one owner, never cut
    na_first = [np.concatenate([[0], np.arange(10, 2000)]).astype(np.uint64)] * 2
    b = sample_plan(lib, na_first)
    assert dest(b, np.array([0], np.uint64))[0] == 0                           # NA image with the first rank


def test_status_agreement(lib):
    def ff(rcs, sigs):
        rank, ok = C.c_int(0), C.c_int(0)
        rc = lib.sp_first_failure((C.c_int * len(rcs))(*rcs), (C.c_uint * len(sigs))(*sigs), len(rcs), C.byref(rank), C.byref(ok))
        return rc, rank.value, bool(ok.value)
    assert ff([0, 0, 0], [5, 5, 5]) == (0, -1, True)
    assert ff([0, -3, -1], [5, 5, 5]) == (-3, 1, True)            # first failing rank and ITS code, seen by every rank
    assert ff([0, 0, 0, 0], [5, 5, 6, 5]) == (0, -1, False)       # ranks called with different queries
    assert ff([-2], [1]) == (-2, 0, True)


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("where", ["na_first", "middle", "two_heavy"])
def test_row_sample_splitters_with_heavy_keys(lib, world, where):
    """round 6: a key (the NA group of an NA-first frame, image 0) heavier than a fair share neither starves the
    ranks before it nor comes on top of a fair share: its owner holds the run (+ the few light keys before it at most), the
    other ranks split the REST -- the largest share is the heavy key's, no rank stays empty"""
    rng = np.random.default_rng(world * 31 + len(where))
    n = 400_000
    k = rng.integers(1, 10**7, n)
    img = image_i64(k)
    if where == "na_first":
        heavy_img, share = np.uint64(0), 0.30
        img[rng.random(n) < share] = heavy_img
    elif where == "middle":
        heavy_img = image_i64(np.array([5_000_000]))[0]
        img[rng.random(n) < 0.35] = heavy_img
    else:
        img[rng.random(n) < 0.25] = np.uint64(0)
        heavy_img = image_i64(np.array([7_000_000]))[0]
        img[(rng.random(n) < 0.25) & (img != 0)] = heavy_img
    slices = [img[r * n // world:(r + 1) * n // world] for r in range(world)]
    b = row_sample_plan(lib, slices)
    assert np.all(b[1:] >= b[:-1])
    d = dest(b, img)
    got = np.bincount(d, minlength=world)
    assert got.sum() == n
    vals, cnts = np.unique(img, return_counts=True)
    heavies = vals[cnts > n / world]
    owners = {int(d[img == h][0]) for h in heavies}
    assert all(len(set(d[img == h].tolist())) == 1 for h in heavies)               # never cut
    rest = n - int(cnts[cnts > n / world].sum())
    fair = rest / (world - len(heavies))
    # the largest share is a heavy key's own (+ a sample step), or a light rank's: light keys keep their ORDER around the
    # heavy ones, so a light rank may hold up to the whole stretch between two heavy keys -- never a heavy key on top of it
    assert got.max() <= max(int(cnts.max()) + 4 * n / 4096, 1.6 * fair + 4 * n / 4096), (where, got, fair)
    assert got.min() > 0, (where, got)
    for h, c in zip(heavies, cnts[cnts > n / world]):
        o = int(d[img == h][0])
        assert got[o] <= c + 0.5 * fair + 4 * n / 4096, (where, o, got, c)
