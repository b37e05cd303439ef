import sys, time, torch
sys.path.insert(0, "/root/repo")
from datatable_amd.torch_bridge import context_for_current_stream, devcol
dev = torch.device("cuda:0"); g = torch.Generator(device=dev); g.manual_seed(5)
ctx = context_for_current_stream(0)
n = 1_000_000_000
pool = torch.randint(-2**62, 2**62, (10_000_000,), dtype=torch.int64, device=dev, generator=g)
k = pool[torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)]; del pool
v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
w = torch.randint(-1000, 1000, (n,), dtype=torch.int64, device=dev, generator=g)
for aggs, vals in (([("sum", 0)], [v]), ([("sum", 0), ("sum", 1)], [v, w]), ([("sum", 0), ("mean", 1), ("min", 1), ("count0", None)], [v, w])):
    for hm in (0, 1):
        ctx.set_option("hash_mode", hm)
        def run():
            r = ctx.groupby_agg([devcol(k)], [devcol(x) for x in vals], aggs, nrows=n); ng = r.ngroups; r.free(); return ng
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ng = run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
        print("aggs=%s hash_mode=%d: %.1f ms, %d groups" % ("+".join(a for a, _ in aggs), hm, t * 1e3, ng), flush=True)
ctx.set_option("hash_mode", 0)
