"""One rank of tests/test_gpu_rccl_2proc.py: `python rccl_rank_worker.py <rank> <world> <dir>` -- its own process, its own
HIP context on GPU 0 (or, with DTHIP_WORKER_DEVICE_PER_RANK=1 on a multi-GPU box, on GPU <rank>: RCCL's own transport), the
library's one-process-per-rank entry points (dthip_comm_init + dthip_sharded_groupby_*)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases(world):
    """name -> (keys, values / columns, query); every rank builds the same frames and takes its row block"""
    rng = np.random.default_rng(2024)
    out = {}
    n = 200_000
    k = rng.integers(-5000, 5000, n).astype(np.int64)
    k[rng.random(n) < 0.02] = np.iinfo(np.int64).min
    v = rng.standard_normal(n); v[rng.random(n) < 0.05] = np.nan
    w = rng.integers(-10**6, 10**6, n).astype(np.int64)
    ops = [("sum", 0), ("mean", 0), ("min", 0), ("max", 0), ("count", 0), ("sum", 1), ("mean", 1), ("min", 1), ("count0", None)]
    out["agg_int64"] = ("agg", [k], [v, w], ops, dict(uneven=True))
    out["agg_int64_na_last"] = ("agg", [k], [v, w], ops, dict(uneven=True, na_last=True))
    kf = (rng.integers(-300, 300, n) / 8).astype(np.float64); kf[rng.random(n) < 0.01] = np.nan
    out["agg_float_key_f32_values"] = ("agg", [kf], [v.astype(np.float32), w], ops, dict(uneven=False))
    a = rng.integers(0, 400, n).astype(np.int32); b = rng.integers(0, 30, n).astype(np.int32)
    out["agg_two_keys"] = ("agg", [a, b], [v], [("count0", None), ("sum", 0), ("max", 0)], dict(uneven=True))
    out["agg_skew_empty_rank"] = ("agg", [(rng.random(n) ** 8 * 1e9).astype(np.int64)], [v], [("sum", 0), ("count0", None)], dict(empty_last=True))
    out["rows_int64"] = ("rows", [k], [k, v, w], None, dict(uneven=True))
    out["rows_two_keys_na_last"] = ("rows", [a, b], [v, a], None, dict(uneven=True, na_last=True))
    return out


def cuts_for(n, world, opt):
    if opt.get("empty_last"):
        c = [r * n // (world - 1) for r in range(world)] + [n] if world > 1 else [0, n]
        return c[:world] + [n]
    if opt.get("uneven"):
        return [0] + sorted(np.random.default_rng(n).integers(0, n + 1, world - 1).tolist()) + [n]
    return [r * n // world for r in range(world + 1)]


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    from datatable_amd.engine import Context, comm_unique_id
    idf = os.path.join(d, "id.bin")
    if rank == 0:
        open(idf + ".tmp", "wb").write(comm_unique_id())
        os.rename(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 60:
                raise SystemExit("no communicator id after 60 s")
            time.sleep(0.01)
    cid = open(idf, "rb").read()
    ctx = Context(rank if os.environ.get("DTHIP_WORKER_DEVICE_PER_RANK") else 0)
    ctx.comm_init(rank, world, cid)
    assert ctx.comm_rank == rank and ctx.comm_world == world
    res = {}
    for name, (kind, keys, cols, ops, opt) in cases(world).items():
        c = cuts_for(len(keys[0]), world, opt)
        lo, hi = c[rank], c[rank + 1]
        if kind == "agg":
            r = ctx.sharded_groupby_agg([x[lo:hi] for x in keys], [x[lo:hi] for x in cols], ops, na_last=opt.get("na_last", False))
            for i in range(len(keys)):
                res["%s/k%d" % (name, i)] = r.key(i)
            for a_ in range(len(ops)):
                res["%s/a%d" % (name, a_)] = r.agg(a_)
        else:
            r = ctx.sharded_groupby_rows([x[lo:hi] for x in keys], [x[lo:hi] for x in cols], lo, na_last=opt.get("na_last", False))
            for i in range(len(cols) + 1):
                res["%s/c%d" % (name, i)] = r.col(i)
            res["%s/off" % name] = r.offsets()
        r.free()
    # failures must reach EVERY rank: (1) rank 1 alone fails locally (a bad row count: the same query, so only its status
    # word tells the others), (2) rank 1 alone runs another query, (3) the communicator still works afterwards
    k = np.arange(1000, dtype=np.int64) % 7
    v = np.ones(1000)
    verdicts = []
    for aggs, nrows in (([("sum", 0)], -1 if rank == 1 else None), ([("sum", 0), ("count0", None)] if rank == 1 else [("sum", 0)], None)):
        try:
            r = ctx.sharded_groupby_agg([k], [v], aggs, nrows=nrows)
            r.free()
            verdicts.append("completed")
        except Exception as e:
            verdicts.append("%s: %s" % (type(e).__name__, e))
    r = ctx.sharded_groupby_agg([k], [v], [("sum", 0)])
    res["after_errors/k0"], res["after_errors/a0"] = r.key(0), r.agg(0)
    r.free()
    np.savez(os.path.join(d, "rank%d.npz" % rank), **res)
    open(os.path.join(d, "verdicts%d.txt" % rank), "w").write("\n".join(verdicts))
    ctx.close()


if __name__ == "__main__":
    main()
