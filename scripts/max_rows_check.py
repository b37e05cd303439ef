#!/usr/bin/env python
"""Maximum-size check: nrows = 2**31 - 1 (the int32 RowIndex limit; one more row must be refused).
Runs the fused aggregation (bucketed and sort path), dthip_groupby, and a cumulative operator on
device-resident data and checks size-independent properties.  Needs ~120 GB of HBM; not part of the
pytest suite (run: python scripts/max_rows_check.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datatable_amd import torch_bridge as tb  # noqa: E402
from datatable_amd import _lib as L  # noqa: E402

n = 2**31 - 1
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
ng = 10_000_000
k = torch.randint(0, ng, (n,), device=dev, dtype=torch.int64, generator=g)
v = torch.randint(-1000, 1000, (n,), device=dev, dtype=torch.int64, generator=g)
ctx = tb.context_for_current_stream(0)
total = int(v.sum().item())
for path in (2, 1):
    ctx.set_option("agg_path", path)
    off, gk, out = tb.groupby_agg_tensors(ctx, [k], [v], [("sum", 0), ("count0", None), ("min", 0), ("max", 0)])
    assert gk[0].numel() == ng and bool((gk[0][1:] > gk[0][:-1]).all()), "keys not strictly ascending"
    assert int(out[1].sum().item()) == n, "counts do not sum to n"
    assert int(out[0].sum().item()) == total, "sum of group sums != sum of values"
    assert bool((out[2] <= out[3]).all())
    assert int(off[-1].item()) == n and int(off[0].item()) == 0
    print("agg_path", path, "ok: ngroups", gk[0].numel(), flush=True)
    del off, gk, out
ctx.set_option("agg_path", 0)
# RowIndex + cumulative operator at full size
off, ri, _ = tb.groupby_rows_tensors(ctx, [k], [], want_rowindex=True)
assert ri.numel() == n and int(off[-1].item()) == n
kk = k[ri.long()[:50_000_000]]
assert bool((kk[1:] >= kk[:-1]).all()), "grouped order not sorted"
del kk
cs = tb.group_cumulate_tensor(ctx, "cumsum", v, ri, off)
last = cs[(off[1:] - 1).long()]
assert int(last.sum().item()) == total, "last cumsum of every group must be the group sum"
print("rowindex + cumsum ok", flush=True)
del cs, last
# sd per group, ngroup and the natural-join index at the same size
sd = tb.group_reduce_tensor(ctx, "sd", v, ri, off)
# ~215 uniform integers of [-1000, 1000) per group: sd = 577 +- 28 per group, 1e7 groups
assert sd.numel() == ng and abs(float(sd.mean().item()) - 577.0) < 3.0 and bool((sd > 350).all()) and bool((sd < 800).all())
ngr = tb.group_cumulate_tensor(ctx, "ngroup", None, None, off)
assert int(ngr[-1].item()) == ng - 1 and int(ngr[0].item()) == 0 and bool((ngr[1:] >= ngr[:-1]).all())
del sd, ngr, ri, off
from datatable_amd.engine import DevCol
jk = torch.arange(0, ng, dtype=torch.int64, device=dev)                 # keyed frame: every key once, ascending
jidx = torch.empty(n, dtype=torch.int32, device=dev)
ctx.join_index_dev([DevCol(k.data_ptr(), L.INT64)], [DevCol(jk.data_ptr(), L.INT64)], n, ng, jidx.data_ptr())
torch.cuda.synchronize()
assert bool((jidx[:100_000_000].long() == k[:100_000_000]).all()) and bool((jidx[-100_000_000:].long() == k[-100_000_000:]).all())
print("sd + ngroup + join_index ok", flush=True)
del jk, jidx
# one more row is refused (int32 RowIndex), not silently wrapped like the reference (sort.cc:505-506)
import ctypes as C
h = C.c_void_p()
col = (L.Col * 1)(L.Col(k.data_ptr(), L.INT64, 0))
rc = ctx._lib.dthip_groupby(ctx._h, col, 1, n + 1, L.NA_FIRST, L.DEVICE, 1, C.byref(h))
assert rc != 0, "nrows = 2**31 must be rejected"
print("nrows 2**31 rejected:", ctx._lib.dthip_last_error().decode()[:80])
print("MAX ROWS CHECK PASSED")
