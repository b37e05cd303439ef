#!/usr/bin/env python
"""Randomised check of dthip_filter_groupby_rows on its three routes (fused with a tile-local second level, fused with a
scattering second level, two calls) against the oracle, bit for bit (tests/test_gpu_filter_rows.py::run): random row
counts, key ranges / offsets / stypes, hot keys (buckets far beyond a tile), NA keys, NA predicate values, selectivity,
comparison, riding columns, descending / NA-last, final-bucket sizes, with and without the composed RowIndex.
    python scripts/fuzz_fused_rows.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_filter_rows as T  # noqa: E402
from datatable_amd.engine import Context  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = Context(0)
ctx.set_option("sort_path", 2); ctx.set_option("msd_min_rows", 1)
t0 = time.time()
case = bad = 0
routes = {}
while time.time() - t0 < budget:
    seed = seed0 * 100000 + case; case += 1
    rng = np.random.default_rng(seed)
    n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 300_000), rng.integers(300_000, 6_000_000)], p=[0.15, 0.35, 0.5]))
    brows = int(rng.choice([64, 256, 1024, 2048]))
    aim = rng.random() < float(os.environ.get('FUZZ_AIM', '0.7'))    # aim at the fused route: the levels need  key bits - scatter bits <= 9
    bits = int(rng.integers(1, 33))
    if aim:
        bits = int(rng.integers(1, max(2, int(np.log2(max(n // (2 * brows), 1))) + 10)))
    hi = int(min(2**bits, 2**32 - 2))
    kt = np.int64 if rng.random() < 0.6 else np.int32
    if kt == np.int32:
        hi = min(hi, 2**31 - 2)
    off = int(rng.integers(-2**20, 2**20)) if rng.random() < 0.5 else 0
    if kt == np.int32:
        off = max(-2**31 + 1, min(off, 2**31 - 1 - hi))
    k = (rng.integers(0, hi, n) + off).astype(kt)
    shape = rng.random()
    if shape < 0.25 and n > 10:                       # one hot key: a final bucket far beyond a tile
        k[rng.random(n) < rng.uniform(0.05, 0.7)] = k[0]
    elif shape < 0.4 and n > 10:                      # few distinct keys inside a wide range
        pool = (rng.integers(0, hi, int(rng.integers(1, 50))) + off).astype(kt)
        k = pool[rng.integers(0, len(pool), n)]
    elif shape < 0.5:
        k = np.sort(k)                                # sorted keys: whole tiles fall into one bucket
    if rng.random() < 0.5 and n > 4:
        k[rng.random(n) < rng.choice([0.0001, 0.01, 0.3])] = np.iinfo(kt).min
    if rng.random() < 0.7:
        x = rng.standard_normal(n)
        if rng.random() < 0.5:
            x[rng.random(n) < rng.choice([0.001, 0.2])] = np.nan
        scalar = float(rng.choice([0.0, -1.5, 1.0, 2.5, -4.0, 4.0]))
    else:
        x = rng.integers(-5, 6, n).astype(np.int64)
        if rng.random() < 0.5:
            x[rng.random(n) < 0.1] = np.iinfo(np.int64).min
        scalar = int(rng.integers(-6, 7))
    cmp = str(rng.choice([">", ">=", "<", "<=", "==", "!="]))
    if rng.random() < 0.08:
        scalar = None; cmp = str(rng.choice(["==", "!="]))
    y = rng.standard_normal(n)
    w = rng.integers(-10**6, 10**6, n).astype(np.int32)
    sets = [[k, x], [x], [], [k], [w], [w, x], [x, y], [x, k, x, k], [y, w]]
    cols = sets[int(rng.integers(0, len(sets)))]
    want_ri = bool(rng.random() < 0.7)
    if aim:                                          # ... and at most two riding columns of widths 8 / 8+4 / 8+8 / 4, the RowIndex being one of them
        cols, want_ri = [([k, x], True), ([x], True), ([], True), ([k], True), ([w], False), ([w, x], False), ([x, y], False), ([x, k, x, k], True)][int(rng.integers(0, 8))]
    kw = {}
    if rng.random() < 0.2:
        kw["na_last"] = True
    if rng.random() < 0.2:
        kw["desc"] = [True]
    ctx.set_option("msd_bucket_rows", brows)
    try:
        T.run(ctx, x, cmp, scalar, [k], cols, expect_fused=None, want_rowindex=want_ri, expect_tl2=None, stats=routes, **kw)
    except AssertionError as e:
        bad += 1
        print("seed %d n=%d bits=%d kt=%s cmp=%s scalar=%r ncols=%d kw=%r ri=%r FAILED: %s" % (seed, n, bits, kt.__name__, cmp, scalar, len(cols), kw, want_ri, str(e)[:300]), flush=True)
    except Exception as e:
        bad += 1
        print("seed %d n=%d ERROR: %r" % (seed, n, e), flush=True)
print("fuzz_fused_rows: %d cases in %.0f s, %d failures; cases that ran to the end on: %r" % (case, time.time() - t0, bad, routes), flush=True)
ctx.close()
sys.exit(1 if bad else 0)
