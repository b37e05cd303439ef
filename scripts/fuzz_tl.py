#!/usr/bin/env python
"""Randomised cross-check of the tile-local layout of the bucketed aggregation (the default from 2^22 rows on when the
buckets are evenly filled) against the exact-position layout (option bucket_variant = 2) and the sort path
(agg_path = 1) on device-resident data: random row counts, key ranges / offsets / NA shares, one or two keys, value
stypes and reducer sets.  Integer results must be identical, float sums / means agree to 1e-9 relative.
    python scripts/fuzz_tl.py [cases] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datatable_amd.torch_bridge import context_for_current_stream, groupby_agg_tensors  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(seed)
cpu = torch.Generator(); cpu.manual_seed(seed)
ctx = context_for_current_stream(0)


def ri(lo, hi):
    return int(torch.randint(lo, hi, (1,), generator=cpu).item())


def run(variant, path, keys, vals, aggs):
    ctx.set_option("bucket_variant", variant); ctx.set_option("agg_path", path)
    try:
        return groupby_agg_tensors(ctx, keys, vals, aggs)
    finally:
        ctx.set_option("bucket_variant", 0); ctx.set_option("agg_path", 0)


nbad = 0
for case in range(cases):
    n = ri(4_200_000, 30_000_000)
    nkeys = 1 if ri(0, 3) else 2
    kd = torch.int64 if ri(0, 2) else torch.int32
    keys = []
    bits_left = ri(18, 25)
    for i in range(nkeys):
        b = bits_left if i == nkeys - 1 else ri(4, bits_left - 4)
        bits_left -= b
        span = max(2, int(2 ** b * (0.55 + 0.45 * torch.rand(1, generator=cpu).item())))
        off = ri(-10**6, 10**6)
        k = torch.randint(0, span, (n,), device=dev, dtype=kd, generator=g) + off
        if ri(0, 2):
            na = torch.rand(n, device=dev, generator=g) < 0.01
            k[na] = torch.iinfo(kd).min
        keys.append(k)
    vkind = ri(0, 4)
    if vkind == 0:
        v = torch.randn(n, device=dev, dtype=torch.float64, generator=g)
        v[torch.rand(n, device=dev, generator=g) < 0.02] = float("nan")
    elif vkind == 1:
        v = torch.randint(-10**9, 10**9, (n,), device=dev, dtype=torch.int64, generator=g)
    elif vkind == 2:
        v = torch.randn(n, device=dev, dtype=torch.float32, generator=g)
    else:
        v = torch.randint(-1000, 1000, (n,), device=dev, dtype=torch.int32, generator=g)
    aggsets = [[("sum", 0)], [("sum", 0), ("count0", None)], [("sum", 0), ("mean", 0), ("min", 0), ("max", 0), ("count", 0)],
               [("count0", None)], [("min", 0), ("max", 0)]]
    aggs = aggsets[ri(0, len(aggsets))]
    vals = [v] if any(c is not None for _, c in aggs) else []
    a = run(0, 0, keys, vals, aggs)
    b = run(2, 0, keys, vals, aggs)
    ok = True
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        if x.dtype.is_floating_point:
            same = bool(torch.allclose(x.double(), y.double(), rtol=1e-9 if x.dtype == torch.float64 else 1e-5, atol=1e-9, equal_nan=True))
        else:
            same = bool(torch.equal(x, y))
        ok &= same and x.shape == y.shape
    ok &= bool(torch.equal(a[0], b[0]))
    if case % 8 == 0:         # the sort path as a third opinion (slower)
        c = run(0, 1, keys, vals, aggs)
        ok &= bool(torch.equal(a[0], c[0])) and all(bool(torch.equal(x, y)) for x, y in zip(a[1], c[1]))
    names = [nm for nm in ctx.profile_names()] if False else []
    print("case %2d n=%9d keys=%d %s val=%s aggs=%s groups=%d %s" % (case, n, nkeys, str(kd)[6:], str(v.dtype)[6:],
          "+".join(op for op, _ in aggs), a[1][0].numel(), "ok" if ok else "MISMATCH"), flush=True)
    nbad += 0 if ok else 1
    del keys, v, a, b
    torch.cuda.empty_cache()
print("FUZZ %s: %d cases, %d mismatches" % ("PASSED" if nbad == 0 else "FAILED", cases, nbad))
sys.exit(1 if nbad else 0)
