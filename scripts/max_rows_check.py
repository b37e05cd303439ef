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
del cs, last, ri, off
# one more row is refused (int32 RowIndex), not silently wrapped like the reference (sort.cc:505-506)
import ctypes as C
h = C.c_void_p()
col = (L.Col * 1)(L.Col(k.data_ptr(), L.INT64, 0))
rc = ctx._lib.dthip_groupby(ctx._h, col, 1, n + 1, L.NA_FIRST, L.DEVICE, 1, C.byref(h))
assert rc != 0, "nrows = 2**31 must be rejected"
print("nrows 2**31 rejected:", ctx._lib.dthip_last_error().decode()[:80])
print("MAX ROWS CHECK PASSED")
