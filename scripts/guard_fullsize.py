#!/usr/bin/env python
"""Full-size runs of the BASELINE configs in GUARD-PAGE mode (dthip option "guard", include/dthip.h): every buffer the
library allocates -- and, here, every INPUT column too (copied into dthip_malloc'ed buffers) -- is its own mapping with
its end (guard 1) or start (guard 2) flush against unmapped pages, every launch is synchronised.  A kernel that reads
or writes outside a buffer dies with a GPU memory access fault and the library's SIGABRT handler names it; a clean run
prints one OK line per config and "[dthip guard] context closed: ... no fault".

    DTHIP_GUARD=1 python scripts/guard_fullsize.py --configs C3,C4,C5,C3_hard --scale 1.0
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C3,C4,C5,C3_hard,C2,C1")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--guard", type=int, default=int(os.environ.get("DTHIP_GUARD", "1")))
    args = ap.parse_args()
    os.environ["DTHIP_GUARD"] = str(args.guard)
    import torch
    from datatable_amd import _lib as L
    from datatable_amd.engine import DevCol
    from datatable_amd.torch_bridge import context_for_current_stream, T2ST
    hip = C.CDLL("libamdhip64.so")
    dev = torch.device("cuda", 0)
    ctx = context_for_current_stream(0)
    ctx.set_option("guard", args.guard)
    g = torch.Generator(device=dev)

    def guarded(t):
        """torch tensor -> a guarded library buffer holding the same bytes"""
        nb = t.numel() * t.element_size()
        p = C.c_void_p()
        L.check(ctx._lib.dthip_malloc(ctx._h, max(nb, 1), C.byref(p)))
        torch.cuda.synchronize()
        assert hip.hipMemcpy(p, C.c_void_p(t.data_ptr()), C.c_size_t(nb), 3) == 0
        return DevCol(p.value, T2ST[t.dtype])

    def free(*cols):
        for c in cols:
            ctx._lib.dthip_free(ctx._h, C.c_void_p(c.ptr))

    for c in [x for x in args.configs.split(",") if x]:
        t0 = time.perf_counter()
        if c == "C1":
            n = int(1e6 * args.scale); g.manual_seed(1235)
            k = guarded(torch.randint(0, 100, (n,), dtype=torch.int32, device=dev, generator=g))
            v = guarded(torch.randn(n, dtype=torch.float64, device=dev, generator=g))
            r = ctx.groupby_agg([k], [v], [("sum", 0), ("count0", None)], nrows=n)
            ng, cnt = r.ngroups, int(r.agg(1).sum()); r.free(); free(k, v)
            assert cnt == n
        elif c == "C2":
            n = int(1e8 * args.scale); g.manual_seed(1236)
            k = guarded(torch.randint(0, 100_000, (n,), dtype=torch.int64, device=dev, generator=g))
            vs = [guarded(torch.randn(n, dtype=torch.float64, device=dev, generator=g)) for _ in range(4)]
            aggs = [(op, i) for op in ("sum", "mean", "min", "max") for i in range(4)] + [("count0", None)]
            r = ctx.groupby_agg([k], vs, aggs, nrows=n)
            ng, cnt = r.ngroups, int(r.agg(16).sum()); r.free(); free(k, *vs)
            assert cnt == n
        elif c in ("C3", "C3_hard"):
            n = int(1e9 * args.scale)
            if c == "C3":
                g.manual_seed(1237)
                kt = torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
            else:
                g.manual_seed(1240)
                pool = torch.randint(-2**62, 2**62, (10_000_000,), dtype=torch.int64, device=dev, generator=g)
                kt = pool[torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)]
                del pool
            k = guarded(kt); del kt
            v = guarded(torch.randn(n, dtype=torch.float64, device=dev, generator=g))
            torch.cuda.empty_cache()
            r = ctx.groupby_agg([k], [v], [("sum", 0)], nrows=n)          # the benchmarked query
            ng = r.ngroups; r.free()
            r = ctx.groupby_agg([k], [v], [("sum", 0), ("count0", None)], nrows=n)
            cnt = int(r.agg(1).sum()); assert r.ngroups == ng; r.free(); free(k, v)
            assert cnt == n
        elif c == "C4":
            n = int(1e9 * args.scale); g.manual_seed(1238)
            a = guarded(torch.randint(0, 3163, (n,), dtype=torch.int32, device=dev, generator=g))
            b = guarded(torch.randint(0, 3163, (n,), dtype=torch.int32, device=dev, generator=g))
            v = guarded(torch.randn(n, dtype=torch.float64, device=dev, generator=g))
            torch.cuda.empty_cache()
            r = ctx.groupby_agg([a, b], [v], [("count0", None), ("sum", 0)], nrows=n)
            ng, cnt = r.ngroups, int(r.agg(0).sum()); r.free(); free(a, b, v)
            assert cnt == n
        elif c == "C5":
            n = int(1e9 * args.scale); g.manual_seed(1239)
            k = guarded(torch.randint(0, 100_000_000, (n,), dtype=torch.int64, device=dev, generator=g))
            xt = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            npass_exp = int((xt > 0).sum().item())
            x = guarded(xt); del xt
            torch.cuda.empty_cache()
            # outputs sized EXACTLY for the passing rows: a write past them faults
            ri = guarded(torch.empty(npass_exp, dtype=torch.int32, device=dev))
            kb = guarded(torch.empty(npass_exp, dtype=torch.int64, device=dev))
            xb = guarded(torch.empty(npass_exp, dtype=torch.float64, device=dev))
            torch.cuda.empty_cache()
            npass = ctx.filter_take_dev(x, ">", 0.0, [k, x], n, ri.ptr, [kb.ptr, xb.ptr])
            assert npass == npass_exp
            r = ctx.groupby_rows([kb], [kb, xb, ri], nrows=npass, want_rowindex=False)
            ng = r.ngroups; cnt = int(r.offsets()[-1]); r.free(); free(k, x, ri, kb, xb)
            assert cnt == npass
        elif c == "C5f":
            # round 5: the same query as ONE call (dthip_filter_groupby_rows: tile-local levels, records, windows); the sort
            # route can be forced through DTHIP_TL_LEVEL2 / DTHIP_FILTER_ROWS_FUSED like everywhere else
            n = int(1e9 * args.scale); g.manual_seed(1239)
            k = guarded(torch.randint(0, 100_000_000, (n,), dtype=torch.int64, device=dev, generator=g))
            xt = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            npass_exp = int((xt > 0).sum().item())
            x = guarded(xt); del xt
            torch.cuda.empty_cache()
            r = ctx.filter_groupby_rows(x, ">", 0.0, [k], [k, x], nrows=n, want_rowindex=True)
            ng = r.ngroups; cnt = int(r.offsets()[-1]); r.free(); free(k, x)
            assert cnt == npass_exp
        else:
            raise SystemExit("unknown config " + c)
        ctx.sync()
        print("guard=%d %s rows=%d groups=%d OK (%.1f s)" % (args.guard, c, n, ng, time.perf_counter() - t0), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
