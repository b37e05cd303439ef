import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datatable_amd.torch_bridge import context_for_current_stream, devcol
dev = torch.device("cuda", 0)
ctx = context_for_current_stream(0)
n = 1_000_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
for name in ("random", "sorted", "constant"):
    if name == "random":
        k = torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
    elif name == "sorted":
        k = torch.sort(torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)).values
    else:
        k = torch.full((n,), 12345, dtype=torch.int64, device=dev)
    def run():
        r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0)], nrows=n); ng = r.ngroups; r.free(); return ng
    run(); torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
        run()
        e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(name, "wall %.2f ms  events %.2f ms" % ((t1 - t0) * 1e3, e0.elapsed_time(e1)), flush=True)
    ctx.profile_reset(); ctx.profile(True); run(); torch.cuda.synchronize(); ctx.profile(False)
    tot = sum(ctx.profile_get(nm)[0] for nm in ctx.profile_names())
    print(name, "sum of kernels %.2f ms" % tot, flush=True)
    del k
