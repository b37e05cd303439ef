#!/usr/bin/env python
"""float64 keys from a pool of 1e7 values, sum(float64): the hash combiner's general route (packed-key pass + pseudo keys)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from datatable_amd.torch_bridge import context_for_current_stream, devcol
dev = torch.device("cuda", 0)
ctx = context_for_current_stream(0)
g = torch.Generator(device=dev); g.manual_seed(7)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
pool = torch.randn(10_000_000, dtype=torch.float64, device=dev, generator=g)
k = pool[torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)]
del pool
v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
ctx.profile(True)
for hm in (0, 3, 0, 3):
    ctx.set_option("hash_mode", hm)
    best = 1e9
    for rep in range(3):
        if rep == 1: ctx.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0)], nrows=n); ng = r.ngroups
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        r.free()
    names = sorted(ctx.profile_names(), key=lambda nm: -ctx.profile_get(nm)[0])[:6]
    print("hash_mode=%d groups=%d  %.2f ms   %s" % (hm, ng, best * 1e3,
          " ".join("%s=%.2fx%d" % (nm.replace("_kernel", ""), ctx.profile_get(nm)[0] / max(ctx.profile_get(nm)[1], 1), ctx.profile_get(nm)[1] // 2) for nm in names)))
