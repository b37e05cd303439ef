#!/usr/bin/env python
"""bench.py -- groupby-sum throughput of the HIP path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2] ("C3"): 1e9 rows, one int64 key with 1e7 groups,
sum(float64) -- `DT[:, sum(f.v), by(f.k)]`.  It fits one GPU (16 GB of input), so it is
the N=1 workload too; for N>1 the SAME 1e9 rows are row-sharded over the ranks (strong
scaling, as the north star states "1e9 rows at 1/2/4/8"): local fused groupby-sum,
range-partitioned all-to-all of partials over RCCL, merge on the owner (datatable_amd/dist.py).

A step = one full pass of the hot path over the (HBM-resident) batch.  Default path (dense
integer key range): sampled key range, per-tile bucket histogram (which verifies the range),
one 1024-way partition of (slot key, value), LDS-table aggregation per bucket, compaction of
the non-empty slots into (group key, sum) columns.  `--agg-path 1` takes the general path:
exact key range, key transform + digit histograms, stable LSD radix passes carrying the value,
run heads -> offsets, segmented sum.  Inputs are in HBM before the timed region; outputs stay in HBM.
One JSON line on rank 0.  `roofline` is for the dominant kernel of the timed region (the one
with the largest total time: bucket_partition_kernel on the default path), timed with HIP events
around every launch; `cpu_baseline` is the CPU oracle (oracle/, a port of the reference's
algorithm) on a bounded sample, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_ROW = 16         # key 8 B + value 8 B read once (+ ng*(8+8) written)


def cpu_baseline(sample_rows, groups, seed, threads):
    """Oracle (oracle/dt_oracle.c: restatement of group() + sum reducer) on the host cores: the rate on
    `threads` OpenMP threads (chunked radix passes and per-group reducers, as the reference parallelises
    them) over `sample_rows` rows, plus the single-thread rate on a fifth of that sample."""
    import numpy as np
    from oracle import oracle as o
    o.lib()
    rng = np.random.default_rng(seed)
    k = rng.integers(0, groups, sample_rows, dtype=np.int64)
    v = rng.standard_normal(sample_rows)

    def run(kk, vv, t):
        o.set_threads(t)
        t0 = time.perf_counter()
        ri, off = o.group([kk])
        s = o.reduce("sum", vv, ri, off)
        dt = time.perf_counter() - t0
        assert len(s) == len(off) - 1
        return dt

    n1 = max(sample_rows // 5, 1)
    try:
        dt1 = run(k[:n1], v[:n1], 1)
        dtn = run(k, v, threads) if threads > 1 else None
    finally:
        o.set_threads(1)
    if dtn is None:
        value, cores, rows_used, dt = n1 / dt1, 1, n1, dt1
    else:
        value, cores, rows_used, dt = sample_rows / dtn, threads, sample_rows, dtn
    return {"value": value, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": "%d rows (%.0f%% of the workload), int64 key uniform in [0,%d), float64 N(0,1): oracle group()+sum "
                      "on %d host thread(s), %.2f s" % (rows_used, 100.0 * rows_used / 1e9, groups, cores, dt),
            "single_thread_value": n1 / dt1, "single_thread_sample_rows": n1,
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--groups", type=int, default=10_000_000)
    ap.add_argument("--cpu-sample", type=int, default=500_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: min(host cores, 64))")
    ap.add_argument("--no-check", action="store_true", help="skip the result sanity check (kernel timing experiments)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--agg-path", type=int, default=0, help="dthip option agg_path: 0 auto, 1 sort, 2 bucketed")
    ap.add_argument("--bucket-variant", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (RCCL collectives, merge) even with one rank")
    ap.add_argument("--agg-offsets", type=int, default=0,
                    help="1: the result also carries group sizes (not part of DT[:, sum(f.v), by(f.k)]'s result Frame)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from datatable_amd.torch_bridge import context_for_current_stream, devcol
    from datatable_amd.dist import HipBackend, sharded_groupby_agg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run for N>1" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_dist
    if sharded:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)
    ctx = context_for_current_stream(local_rank)
    ctx.set_option("agg_path", args.agg_path)
    ctx.set_option("bucket_variant", args.bucket_variant)
    if not sharded:
        ctx.set_option("agg_offsets", args.agg_offsets)

    n_total = args.rows
    lo, hi = rank * n_total // world, (rank + 1) * n_total // world
    n_local = hi - lo
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + 3 + 1000 * rank)
    keys = torch.randint(0, args.groups, (n_local,), dtype=torch.int64, device=dev, generator=g)
    vals = torch.randn(n_local, dtype=torch.float64, device=dev, generator=g)
    torch.cuda.synchronize()
    aggs = [("sum", 0)]
    backend = HipBackend(ctx)
    kcol, vcol = devcol(keys), devcol(vals)

    def step():
        if not sharded:
            r = ctx.groupby_agg([kcol], [vcol], aggs, nrows=n_local)
            return r
        return sharded_groupby_agg(backend, [keys], [vals], aggs)

    def release(r):
        if not sharded:
            r.free()

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        release(step())
    ctx.profile_reset()
    if not args.no_profile:
        ctx.profile(True)
    barrier()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        if last is not None:
            release(last)
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # sanity of the last result (outside the timed region): keys strictly ascending, totals agree,
    # and (one extra untimed call with count()) every row counted exactly once
    if not sharded:
        ng = last.ngroups
        sums = torch.empty(ng, dtype=torch.float64, device=dev)
        gkeys = torch.empty(ng, dtype=torch.int64, device=dev)
        last.agg_into(0, sums.data_ptr()); last.key_into(0, gkeys.data_ptr())
        release(last)
        torch.cuda.synchronize()
        if not args.no_check:
            assert bool((gkeys[1:] > gkeys[:-1]).all())
            total, ref = float(sums.sum().item()), float(vals.sum().item())
            assert abs(total - ref) <= 1e-9 * float(vals.abs().sum().item()), (total, ref)
            rc = ctx.groupby_agg([kcol], [vcol], [("count0", None), ("sum", 0)], nrows=n_local)
            cnt = torch.empty(rc.ngroups, dtype=torch.int64, device=dev)
            s2 = torch.empty(rc.ngroups, dtype=torch.float64, device=dev)
            rc.agg_into(0, cnt.data_ptr()); rc.agg_into(1, s2.data_ptr())
            rc.free()
            torch.cuda.synchronize()
            assert rc.ngroups == ng and int(cnt.sum().item()) == n_local
            assert bool(torch.allclose(s2, sums, rtol=1e-9, atol=1e-9))
    else:
        gk, out = last
        ng_t = torch.tensor([gk[0].numel()], dtype=torch.int64, device=dev)
        tot = torch.stack([out[0].sum(), vals.sum(), vals.abs().sum()])
        dist.all_reduce(ng_t); dist.all_reduce(tot)
        ng = int(ng_t.item())
        assert abs(float(tot[0]) - float(tot[1])) <= 1e-9 * float(tot[2]), tot.tolist()

    per_kernel = {}
    for nm in ctx.profile_names():
        ms, cnt = ctx.profile_get(nm)
        per_kernel[nm] = {"launches": cnt, "avg_ms": ms / max(cnt, 1), "total_ms": ms}
    # the dominant kernel = largest share of the timed region
    dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms"]) if per_kernel else None
    rp_ms, rp_n = (per_kernel[dom]["total_ms"], per_kernel[dom]["launches"]) if dom else (0.0, 0)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt
        alg_bytes_launch = ALG_BYTES_PER_ROW * n_local          # algorithmic bytes of the rows one launch processes
        roof = None
        if rp_n:
            # single GPU: one launch of the dominant kernel per step.  Sharded: the merge of the exchanged
            # partials launches it a second time on a few rows; its time is charged, its bytes are not.
            per_step = max(1, round(rp_n / args.steps))
            avg_s = rp_ms / rp_n * per_step * 1e-3
            ach = alg_bytes_launch / avg_s / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc) and not sharded:
                try:
                    traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch_at_rows", {}).get(str(n_local))
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "launches": rp_n, "launches_per_step": per_step,
                    "avg_launch_ms": rp_ms / rp_n * per_step,
                    "alg_bytes_per_launch": alg_bytes_launch,
                    "note": "achieved = 16 B/row (SURVEY 8d, C3) x rows of one launch / HIP-event time of that launch; "
                            "traffic = HBM bytes of one launch from rocprofv3 FETCH_SIZE x2 + WRITE_SIZE (profiles/)",
                    "whole_step_alg_GBs": (ALG_BYTES_PER_ROW * n_total + 16 * ng) / (dt / args.steps) / 1e9,
                    "whole_step_frac": (ALG_BYTES_PER_ROW * n_total + 16 * ng) / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world)}
        line = {
            "metric": "groupby-sum rows/sec (1e9 rows, int64 key, 1e7 groups, float64 value)",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int64 keys / f64 sums", "data": "synthetic",
            "config": {"workload": "C3: DT[:, sum(f.v), by(f.k)], %d rows, int64 key uniform in [0,%d), float64 N(0,1)"
                                   % (n_total, args.groups),
                       "rows": n_total, "groups_found": ng, "rows_per_gpu": n_local,
                       "parallelism": "row-sharded x%d, range-partitioned all-to-all of partials" % world if world > 1 else "single GPU"},
            "roofline": roof,
            "kernels": per_kernel,
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 64)
            line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, n_total), args.groups, 1234 + 3, threads)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
