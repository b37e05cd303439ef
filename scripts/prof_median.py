import sys, os, torch
sys.path.insert(0, "/root/repo")
from datatable_amd import torch_bridge as tb
n, ng = 100_000_000, 100_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1234)
k = torch.randint(0, ng, (n,), device=dev, dtype=torch.int64, generator=g)
v = torch.randn(n, device=dev, dtype=torch.float64, generator=g)
ctx = tb.context_for_current_stream(0)
off, ri, _ = tb.groupby_rows_tensors(ctx, [k], [], want_rowindex=True)
tb.group_reduce_tensor(ctx, "median", v, ri, off)
ctx.profile(True)
tb.group_reduce_tensor(ctx, "median", v, ri, off)
torch.cuda.synchronize()
rows = [(nm,) + ctx.profile_get(nm) for nm in ctx.profile_names()]
for nm, ms, cnt in sorted(rows, key=lambda t: -t[1])[:25]:
    print("%-32s %9.3f ms  %5d launches" % (nm, ms, cnt))
