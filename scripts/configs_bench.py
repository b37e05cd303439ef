#!/usr/bin/env python
"""Single-GPU timing of BASELINE.json's five configurations through the C ABI with
HBM-resident inputs (the parity of the same shapes is in tests/).  Prints one line per config."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="multiply every row count")
    ap.add_argument("--configs", default="1,2,3,4,5")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--once", action="store_true", help="evaluate every query exactly once (rocprofv3 PMC passes: counters per query)")
    ap.add_argument("--agg-path", type=int, default=0)
    ap.add_argument("--bucket-variant", type=int, default=0)
    ap.add_argument("--c5-two-calls", action="store_true", help="config 5 as dthip_filter_take + dthip_groupby_rows (round 4's form)")
    ap.add_argument("--c5-unfused", action="store_true", help="config 5 with filter_cmp + two gathers instead of filter_take")
    args = ap.parse_args()
    import torch
    from datatable_amd import _lib as L
    from datatable_amd.torch_bridge import context_for_current_stream, devcol
    dev = torch.device("cuda", 0)
    ctx = context_for_current_stream(0)
    ctx.set_option("agg_path", args.agg_path)
    ctx.set_option("bucket_variant", args.bucket_variant)
    g = torch.Generator(device=dev)

    def timed(fn):
        if args.once:
            t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
            return time.perf_counter() - t0, r
        fn()                                   # warm-up (allocator, first-touch)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(args.reps):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best, r

    out = []
    for c in [int(x) for x in args.configs.split(",")]:
        g.manual_seed(1234 + c)
        if c == 1:
            n = int(1e6 * args.scale)
            k = torch.randint(0, 100, (n,), dtype=torch.int32, device=dev, generator=g)
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            def run():
                r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0)], nrows=n); ng = r.ngroups; r.free(); return ng
            alg = n * 12
        elif c == 2:
            n = int(1e8 * args.scale)
            k = torch.randint(0, 100_000, (n,), dtype=torch.int64, device=dev, generator=g)
            vs = [torch.randn(n, dtype=torch.float64, device=dev, generator=g) for _ in range(4)]
            aggs = [(op, i) for op in ("sum", "mean", "min", "max") for i in range(4)]
            def run():
                r = ctx.groupby_agg([devcol(k)], [devcol(x) for x in vs], aggs, nrows=n); ng = r.ngroups; r.free(); return ng
            alg = n * 40
        elif c == 3:
            n = int(1e9 * args.scale)
            k = torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            def run():
                r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0)], nrows=n); ng = r.ngroups; r.free(); return ng
            alg = n * 16
        elif c == 4:
            n = int(1e9 * args.scale)
            a = torch.randint(0, 3163, (n,), dtype=torch.int32, device=dev, generator=g)
            b = torch.randint(0, 3163, (n,), dtype=torch.int32, device=dev, generator=g)
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            def run():
                r = ctx.groupby_agg([devcol(a), devcol(b)], [devcol(v)], [("count0", None), ("sum", 0)], nrows=n)
                ng = r.ngroups; r.free(); return ng
            alg = n * 16
        elif c in (7, 8, 9, 11):
            # robustness variants of C3: 7 = heavy skew (key = 1e7 * u^6: a few keys hold most rows),
            # 8 = keys already sorted, 9 = one single key, 11 = bench.py's C3_hotkey (7 % of the rows in ONE key)
            n = int(1e9 * args.scale)
            if c == 7:
                k = (torch.rand(n, dtype=torch.float64, device=dev, generator=g) ** 6 * 1e7).to(torch.int64)
            elif c == 8:
                k = (torch.arange(n, dtype=torch.int64, device=dev) // 100)
            elif c == 11:
                k = torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
                k = torch.where(torch.rand(n, device=dev, generator=g) < 0.07, torch.tensor(12_345, dtype=torch.int64, device=dev), k)
            else:
                k = torch.full((n,), 12345, dtype=torch.int64, device=dev)
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            def run(check=False):
                r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0), ("count0", None)], nrows=n)
                ng = r.ngroups
                if check:        # outside the timed calls: sums of 1e9 elements cost as much as the query
                    tot = torch.empty(ng, dtype=torch.float64, device=dev); cnt = torch.empty(ng, dtype=torch.int64, device=dev)
                    r.agg_into(0, tot.data_ptr()); r.agg_into(1, cnt.data_ptr())
                    torch.cuda.synchronize()
                    assert int(cnt.sum().item()) == n, (int(cnt.sum().item()), n)
                    d = abs(float(tot.sum().item()) - float(v.sum().item()))
                    assert d <= 1e-9 * float(v.abs().sum().item()), (d, float(tot.sum().item()), float(v.sum().item()), ng)
                r.free()
                return ng
            run(check=True)
            alg = n * 16
        elif c == 10:
            # C3 as the LITERAL seam: dthip_groupby (RowIndex + offsets) then dthip_reduce(SUM) gathering through the RowIndex
            n = int(1e9 * args.scale)
            k = torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            sums = torch.empty(min(n, 10_000_000) + 16, dtype=torch.float64, device=dev)
            def run():
                r = ctx.groupby([devcol(k)], nrows=n, want_rowindex=True)
                ng = r.ngroups
                ctx.reduce_dev("sum", devcol(v), r.rowindex_ptr, r.offsets_ptr, ng, n, sums.data_ptr())
                r.free(); return ng
            alg = n * 20
        elif c == 6:
            # hard-keys variant of C3: full-range int64 keys drawn from a pool of 1e7 values
            n = int(1e9 * args.scale)
            pool = torch.randint(-2**62, 2**62, (10_000_000,), dtype=torch.int64, device=dev, generator=g)
            k = pool[torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)]
            del pool
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            def run():
                r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0)], nrows=n); ng = r.ngroups; r.free(); return ng
            alg = n * 16
        else:
            n = int(1e9 * args.scale)
            k = torch.randint(0, 100_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
            x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            ri = torch.empty(n, dtype=torch.int32, device=dev)
            kbuf = torch.empty(n, dtype=torch.int64, device=dev)
            xbuf = torch.empty(n, dtype=torch.float64, device=dev)
            def run():
                # V = DT[f.x > 0, :]; V[:, :, by(f.k)]: round 5 = ONE call (dthip_filter_groupby_rows: the filter fused into the
                # first sort level, tile-local levels); --c5-two-calls = round 4's filter_take + groupby_rows with the filter's
                # RowIndex riding through the sort; --c5-unfused = round 1's filter_cmp + two gathers
                if not (args.c5_unfused or args.c5_two_calls):
                    r = ctx.filter_groupby_rows(devcol(x), ">", 0.0, [devcol(k)], [devcol(k), devcol(x)], nrows=n, want_rowindex=True)
                    ng = r.ngroups
                    r.free()
                    return ng
                if args.c5_unfused:
                    npass = ctx.filter_cmp_dev(devcol(x), n, ">", 0.0, ri.data_ptr())
                    kv = torch.empty(npass, dtype=torch.int64, device=dev)
                    xv = torch.empty(npass, dtype=torch.float64, device=dev)
                    ctx.gather_dev(devcol(k), ri.data_ptr(), npass, kv.data_ptr())
                    ctx.gather_dev(devcol(x), ri.data_ptr(), npass, xv.data_ptr())
                else:
                    # the view's columns are materialised by the filter sweep itself (dthip_filter_take)
                    npass = ctx.filter_take_dev(devcol(x), ">", 0.0, [devcol(k), devcol(x)], n, ri.data_ptr(),
                                                [kbuf.data_ptr(), xbuf.data_ptr()])
                    kv, xv = kbuf[:npass], xbuf[:npass]
                r = ctx.groupby_rows([devcol(kv)], [devcol(kv), devcol(xv), devcol(ri[:npass])], nrows=npass, want_rowindex=False)
                ng = r.ngroups
                r.free()
                return ng
            alg = int(n * 30.4)
        t, ng = timed(run)
        if args.profile:
            ctx.profile_reset(); ctx.profile(True); run(); torch.cuda.synchronize(); ctx.profile(False)
            prof = sorted(((nm,) + ctx.profile_get(nm) for nm in ctx.profile_names()), key=lambda r: -r[1])
            print("   " + " ".join("%s=%.2fx%d" % (nm.replace("_kernel", ""), ms / max(c_, 1), c_) for nm, ms, c_ in prof[:12]), flush=True)
        line = {"config": c, "rows": n, "groups": ng, "ms": t * 1e3, "rows_per_s": n / t,
                "alg_GBps": alg / t / 1e9, "frac_of_8TBps": alg / t / 8e12}
        out.append(line)
        print(json.dumps(line), flush=True)
        del run
        torch.cuda.empty_cache()
        ctx.trim()
    ctx.close()


if __name__ == "__main__":
    main()
