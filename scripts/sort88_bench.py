#!/usr/bin/env python
"""Rows in grouped order with TWO 8-byte columns riding (the <8, 8> variant of radix_pass_kernel: 96 spilled VGPRs when both
are prefetched): DTHIP_RP_PREFETCH=2 against 1.  5e8 rows, int64 key in [0, 1e8) -- the shape of config 5's second
statement through the binding (V[:, :, by(f.k)] with V = {k, x})."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from datatable_amd.torch_bridge import context_for_current_stream, devcol
dev = torch.device("cuda", 0)
ctx = context_for_current_stream(0)
g = torch.Generator(device=dev); g.manual_seed(5)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
k = torch.randint(0, 100_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
a = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
b = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
kc, ac, bc = devcol(k), devcol(a), devcol(b)
ctx.profile(True)
for want_ri in (False, True):
    best = 1e9
    for rep in range(3):
        if rep == 1: ctx.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.groupby_rows([kc], [ac, bc], nrows=n, want_rowindex=want_ri)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        r.free()
    names = sorted(ctx.profile_names(), key=lambda nm: -ctx.profile_get(nm)[0])[:6]
    print("PREFETCH=%s rowindex=%s  %.2f ms   %s" % (os.environ.get("DTHIP_RP_PREFETCH", "2"), want_ri, best * 1e3,
          " ".join("%s=%.2fx%d" % (nm.replace("_kernel", ""), ctx.profile_get(nm)[0] / max(ctx.profile_get(nm)[1], 1), ctx.profile_get(nm)[1] // 2) for nm in names)))
