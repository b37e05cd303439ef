"""CPU, world_size 2, gloo: the exchange logic of datatable_amd.dist (range partition of
partials, all-to-all-v, merge) with the CPU oracle standing in for the per-rank HIP
groupby.  The concatenation of the ranks' outputs must equal a single-process run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

OPS = [("sum", 0), ("mean", 0), ("min", 0), ("max", 0), ("count", 0), ("count0", None), ("sum", 1), ("mean", 1)]


class OracleBackend:
    """local groupby_agg computed by oracle/ (test infrastructure) on CPU tensors"""

    def groupby_agg(self, keys, values, aggs):
        from oracle import oracle as o
        kn = [k.numpy() for k in keys]
        vn = [v.numpy() for v in values]
        ri, off = o.group(kn)
        gk = [torch.from_numpy(k[ri[off[:-1]]].copy()) for k in kn]
        out = []
        for op, col in aggs:
            if op == "count0":
                out.append(torch.from_numpy(o.reduce("count0", None, None, off)))
            else:
                out.append(torch.from_numpy(o.reduce(op, vn[col], ri, off)))
        return gk, out


    def groupby_rows(self, keys, cols):
        from oracle import oracle as o
        kn = [k.numpy() for k in keys]
        if len(kn[0]) == 0:
            return torch.zeros(1, dtype=torch.int32), [c.clone() for c in cols]
        ri, off = o.group(kn)
        return torch.from_numpy(off.copy()), [torch.from_numpy(c.numpy()[ri].copy()) for c in cols]

    def range_bucket(self, key, bounds):
        k = key.numpy()
        valid = k != np.iinfo(k.dtype).min
        d = np.zeros(len(k), np.int8)
        for b in bounds:
            d += (valid & (k.astype(np.int64) >= b)).astype(np.int8)
        return torch.from_numpy(d)


def make_data(seed, n, nkeys):
    rng = np.random.default_rng(seed)
    keys = [rng.integers(-40, 40, n).astype(np.int64 if i == 0 else np.int32) for i in range(nkeys)]
    keys[0][rng.random(n) < 0.05] = np.iinfo(np.int64).min
    v0 = rng.standard_normal(n)
    v0[rng.random(n) < 0.1] = np.nan
    v1 = rng.integers(-1000, 1000, n).astype(np.int32)
    v1[rng.random(n) < 0.1] = np.iinfo(np.int32).min
    return keys, [v0, v1]


def worker(rank, world, port, nkeys, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datatable_amd.dist import sharded_groupby_agg
    keys, vals = make_data(77, 20_000, nkeys)
    n = len(keys[0])
    lo, hi = rank * n // world, (rank + 1) * n // world
    if rank == 1 and nkeys == 1:
        hi = lo            # one rank with an empty shard
    k = [torch.from_numpy(a[lo:hi].copy()) for a in keys]
    v = [torch.from_numpy(a[lo:hi].copy()) for a in vals]
    gk, out = sharded_groupby_agg(OracleBackend(), k, v, OPS)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), *[t.numpy() for t in gk + out])
    dist.barrier()
    dist.destroy_process_group()


def rows_worker(rank, world, port, nkeys, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datatable_amd.dist import sharded_groupby_rows
    keys, vals = make_data(78, 20_000, nkeys)
    n = len(keys[0])
    lo, hi = rank * n // world, (rank + 1) * n // world
    k = [torch.from_numpy(a[lo:hi].copy()) for a in keys]
    v = [torch.from_numpy(a[lo:hi].copy()) for a in vals]
    off, rowid, cols = sharded_groupby_rows(OracleBackend(), k, v, row_offset=lo)
    np.savez(os.path.join(outdir, "rows%d.npz" % rank), off.numpy(), rowid.numpy(), *[t.numpy() for t in cols])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nkeys", [1, 2])
def test_sharded_groupby_rows_world2(tmp_path, nkeys):
    """rows in grouped order across 2 ranks == one process: global RowIndex, group offsets, columns"""
    from oracle import oracle as o
    world = 2
    mp.spawn(rows_worker, args=(world, free_port(), nkeys, str(tmp_path)), nprocs=world, join=True)
    keys, vals = make_data(78, 20_000, nkeys)
    ri, off = o.group(keys)
    parts = [np.load(os.path.join(str(tmp_path), "rows%d.npz" % r)) for r in range(world)]
    rowid = np.concatenate([p["arr_1"] for p in parts])
    assert np.array_equal(rowid, ri.astype(np.int64)), "global RowIndex differs"
    offs, base = [np.zeros(1, np.int64)], 0
    for p in parts:
        o_ = p["arr_0"].astype(np.int64)
        offs.append(o_[1:] + base)
        base += int(o_[-1])
    assert np.array_equal(np.concatenate(offs), off.astype(np.int64)), "group offsets differ"
    for c, v in enumerate(vals):
        got = np.concatenate([p["arr_%d" % (2 + c)] for p in parts])
        assert np.array_equal(got, v[ri], equal_nan=(v.dtype.kind == "f")), "column %d differs" % c
    assert all(len(p["arr_1"]) > 0 for p in parts)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("nkeys", [1, 2])
def test_sharded_groupby_world2(tmp_path, nkeys):
    from oracle import oracle as o
    world = 2
    mp.spawn(worker, args=(world, free_port(), nkeys, str(tmp_path)), nprocs=world, join=True)
    keys, vals = make_data(77, 20_000, nkeys)
    if nkeys == 1:     # rank 1 had an empty shard
        keys = [a[:10_000] for a in keys]
        vals = [a[:10_000] for a in vals]
    ri, off = o.group(keys)
    exp = [k[ri[off[:-1]]] for k in keys]
    for op, col in OPS:
        exp.append(o.reduce("count0", None, None, off) if op == "count0" else o.reduce(op, vals[col], ri, off))
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    for i, e in enumerate(exp):
        got = np.concatenate([p["arr_%d" % i] for p in parts])
        assert got.dtype == e.dtype, (i, got.dtype, e.dtype)
        if e.dtype.kind == "f":
            assert np.array_equal(np.isnan(got), np.isnan(e))
            np.testing.assert_allclose(got, e, rtol=1e-9, atol=1e-9, equal_nan=True)
        else:
            assert np.array_equal(got, e), "column %d differs" % i
    # both ranks own a non-trivial key range
    assert all(len(p["arr_0"]) > 0 for p in parts)


def test_range_boundaries():
    from datatable_amd.dist import range_boundaries
    assert range_boundaries(0, 99, 4) == [25, 50, 75]
    b = range_boundaries(-2**63 + 1, 2**63 - 1, 8)
    assert len(b) == 7 and all(b[i] < b[i + 1] for i in range(6))
