#!/usr/bin/env python
"""Timing of the group-wise operators on device-resident data:
    python scripts/groupwise_bench.py [--rows 1e8] [--groups 1e5] [--reps 3]
Prints one line per operator: ms, rows/s and algorithmic GB/s (bytes read once + written once)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datatable_amd import torch_bridge as tb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=1e8)
    ap.add_argument("--groups", type=float, default=1e5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    n, ng = int(a.rows), int(a.groups)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1234)
    k = torch.randint(0, ng, (n,), device=dev, dtype=torch.int64, generator=g)
    v = torch.randn(n, device=dev, dtype=torch.float64, generator=g)
    w = torch.randn(n, device=dev, dtype=torch.float64, generator=g)
    vi = torch.randint(-1000, 1000, (n,), device=dev, dtype=torch.int64, generator=g)
    ctx = tb.context_for_current_stream(0)
    off, ri, cols = tb.groupby_rows_tensors(ctx, [k], [], want_rowindex=True)
    ngr = off.numel() - 1
    if a.profile:
        ctx.profile(True)

    def timed(name, fn, alg_bytes):
        fn()
        best = 1e30
        for _ in range(a.reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(json.dumps({"op": name, "ms": round(best, 3), "rows_per_s": round(n / best * 1e3, 1),
                          "alg_GBps": round(alg_bytes / best / 1e6, 1)}), flush=True)

    R = n * 4                      # the RowIndex read
    timed("sum (seg_reduce)", lambda: tb.group_reduce_tensor(ctx, "sum", v, ri, off), R + n * 8 + ngr * 8)
    timed("sd", lambda: tb.group_reduce_tensor(ctx, "sd", v, ri, off), R + n * 8 + ngr * 8)
    timed("cov", lambda: tb.group_reduce_tensor(ctx, "cov", v, ri, off, value2=w), R + n * 16 + ngr * 8)
    timed("corr", lambda: tb.group_reduce_tensor(ctx, "corr", v, ri, off, value2=w), R + n * 16 + ngr * 8)
    timed("median", lambda: tb.group_reduce_tensor(ctx, "median", v, ri, off), R + n * 8 + ngr * 8)
    timed("nunique(int64)", lambda: tb.group_reduce_tensor(ctx, "nunique", vi, ri, off), R + n * 8 + ngr * 8)
    vc = torch.randint(0, 50, (n,), device=dev, dtype=torch.int32, generator=g)       # a categorical column
    timed("nunique(int32, 50 levels)", lambda: tb.group_reduce_tensor(ctx, "nunique", vc, ri, off), R + n * 4 + ngr * 8)
    timed("median(int32, 50 levels)", lambda: tb.group_reduce_tensor(ctx, "median", vc, ri, off), R + n * 4 + ngr * 8)
    timed("cumsum(f64)", lambda: tb.group_cumulate_tensor(ctx, "cumsum", v, ri, off), R + n * 16)
    timed("cumsum(int64) rev", lambda: tb.group_cumulate_tensor(ctx, "cumsum", vi, ri, off, reverse=True), R + n * 16)
    timed("cummax(f64)", lambda: tb.group_cumulate_tensor(ctx, "cummax", v, ri, off), R + n * 16)
    timed("cumcount", lambda: tb.group_cumulate_tensor(ctx, "cumcount", None, None, off), n * 8)
    # set function / natural join: union of the two halves of k; join of k against its distinct values
    import numpy as np
    from datatable_amd.engine import DevCol
    from datatable_amd import _lib as L
    half = n // 2
    oidx = torch.empty(n, dtype=torch.int32, device=dev)
    res = {}
    def run_set(op):
        res["n"] = ctx.setop_dev(op, DevCol(k.data_ptr(), L.INT64), [half, n], n, oidx.data_ptr())
    timed("union(2 x n/2)", lambda: run_set("union"), n * 8 + ngr * 4)
    timed("symdiff(2 x n/2)", lambda: run_set("symdiff"), n * 8 + ngr * 4)
    off2, gk, _ = tb.groupby_agg_tensors(ctx, [k], [], [("count0", None)])
    jk = gk[0]                                   # sorted distinct keys = a keyed frame's key column
    jidx = torch.empty(n, dtype=torch.int32, device=dev)
    timed("join_index(n x %d keys)" % jk.numel(),
          lambda: ctx.join_index_dev([DevCol(k.data_ptr(), L.INT64)], [DevCol(jk.data_ptr(), L.INT64)], n, jk.numel(), jidx.data_ptr()),
          n * 12)
    torch.cuda.synchronize()
    assert bool((jk[jidx.long()[:1000000]] == k[:1000000]).all())
    if a.profile:
        rows = [(nm,) + ctx.profile_get(nm) for nm in ctx.profile_names()]
        for nm, ms, cnt in sorted(rows, key=lambda t: -t[1]):
            print("%-32s %9.3f ms  %5d launches  %8.3f ms/launch" % (nm, ms, cnt, ms / max(cnt, 1)))


if __name__ == "__main__":
    main()
