#!/usr/bin/env python
"""More seeds of tests/test_gpu_parity.py::test_fuzz_fused_agg_all_paths (every aggregation route -- bucketed, tile-local,
hash combiner, sort path -- against the oracle on random frames) than the default suite runs.
    python scripts/fuzz_more.py [seconds] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402
from datatable_amd.engine import Context  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = Context(0)
fn = getattr(T.test_fuzz_fused_agg_all_paths, "__wrapped__", T.test_fuzz_fused_agg_all_paths)
t0 = time.time()
n = bad = 0
while time.time() - t0 < budget:
    try:
        fn(ctx, seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED:", repr(e)[:300], flush=True)
    seed += 1; n += 1
print("fuzz_more: %d seeds in %.0f s, %d failures" % (n, time.time() - t0, bad), flush=True)
sys.exit(1 if bad else 0)
