#!/usr/bin/env python
"""PCIe-inclusive rate of the host-pointer mode (DTHIP_HOST): numpy buffers in, numpy columns out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from datatable_amd.engine import Context
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(3)
k = rng.integers(0, n // 100, n, dtype=np.int64)
v = rng.standard_normal(n)
ctx = Context(0)
for rep in range(3):
    t0 = time.perf_counter()
    r = ctx.groupby_agg([k], [v], [("sum", 0)])
    keys, sums = r.key(0), r.agg(0)
    r.free()
    dt = time.perf_counter() - t0
    print("host mode: %d rows -> %d groups in %.1f ms = %.3g rows/s (%.1f GB/s of input incl. PCIe staging)" % (n, len(keys), dt * 1e3, n / dt, 16 * n / dt / 1e9))
ctx.close()
