import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import test_gpu_parity as T
from datatable_amd.engine import Context
ctx = Context(0)
bad = 0
for seed in range(40, 400):
    try:
        T.test_fuzz_fused_agg_all_paths.__wrapped__(ctx, seed) if hasattr(T.test_fuzz_fused_agg_all_paths, "__wrapped__") else T.test_fuzz_fused_agg_all_paths(ctx, seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED:", repr(e)[:300])
print("done, failures:", bad)
