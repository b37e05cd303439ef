#!/usr/bin/env python
"""Randomised check of the sharded paths of csrc/comm.hip (logical shards on one GPU: the same phases as the RCCL path,
round 5's rows exchange with two all-gathers -- stratified key sample, receive bound, overflow -> status round) against
the oracle: dthip_sharded_groupby_rows (global row ids == the oracle's RowIndex, a column in grouped order, offsets) and
dthip_sharded_groupby_agg on the same rows.  Random world sizes, row counts, uneven / empty shards, key shapes (uniform,
skewed, wide, float, SORTED -- every shard one key range --, ONE HOT KEY -- a receive bound any sample underestimates).
    python scripts/fuzz_sharded_rows.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_sharded as T  # noqa: E402
from conftest import assert_same  # noqa: E402
from datatable_amd.engine import LocalComm  # noqa: E402
from oracle import oracle as o  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
comms = {}
t0 = time.time()
case = bad = 0
while time.time() - t0 < budget:
    seed = seed0 * 100000 + case; case += 1
    rng = np.random.default_rng(seed)
    world = int(rng.choice([2, 3, 4, 5, 7, 8, 16]))
    if world not in comms:
        comms[world] = LocalComm(world, device=0)
    n = int(rng.choice([rng.integers(0, 200), rng.integers(200, 50_000), rng.integers(50_000, 1_500_000)], p=[0.15, 0.35, 0.5]))
    kind = str(rng.choice(["uniform", "skew", "wide", "float", "int32"]))
    k, v, w = T.make(n, seed, kind)
    shape = rng.random()
    if shape < 0.2 and n > 10:
        k = np.sort(k)                                           # every shard holds one key range: all its rows go to one or two owners
    elif shape < 0.45 and n > 10:
        k[rng.random(n) < rng.uniform(0.3, 0.95)] = k[n // 2]    # one hot key
    elif shape < 0.55 and n > 10:
        k[:] = k[0]                                              # a constant column
    na_last = bool(rng.random() < 0.25)
    uneven = bool(rng.random() < 0.6)
    try:
        ri, off = o.group([k], na_last=na_last)
        ksh, cuts = T.shard([k], world, uneven=uneven)
        csh, _ = T.shard([k, v, w], world, uneven=uneven)
        res = comms[world].groupby_rows(ksh, csh, cuts[:-1], na_last=na_last)
        assert_same(T.concat(res, lambda r: r.col(3)).astype(np.int32), ri, "global row ids == the oracle's RowIndex")
        got_v = T.concat(res, lambda r: r.col(1)); exp_v = v[ri]
        assert_same(got_v.view(np.uint64), exp_v.view(np.uint64), "value column in grouped order")
        assert_same(T.concat(res, lambda r: r.col(2)), w[ri], "int64 column in grouped order")
        goff = [0]
        for r in res:
            oo = r.offsets().astype(np.int64)
            goff += (oo[1:] + goff[-1]).tolist() if r.ngroups else []
        assert_same(np.array(goff, np.int32), off, "offsets")
        for r in res:
            r.free()
        T.check_agg_oracle(comms[world], [k], [v, w], T.OPS, uneven=uneven, na_last=na_last)
    except AssertionError as e:
        bad += 1
        print("seed %d world=%d n=%d kind=%s shape=%.2f na_last=%r uneven=%r FAILED: %s" % (seed, world, n, kind, shape, na_last, uneven, str(e)[:300]), flush=True)
    except Exception as e:
        bad += 1
        print("seed %d world=%d n=%d kind=%s ERROR: %r" % (seed, world, n, kind, e), flush=True)
print("fuzz_sharded_rows: %d cases in %.0f s, %d failures (world sizes used: %s)" % (case, time.time() - t0, bad, sorted(comms)), flush=True)
sys.exit(1 if bad else 0)
