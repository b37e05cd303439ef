#!/usr/bin/env python
"""bench.py -- groupby-sum throughput of the HIP path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 5 --warmup 2          (starts its 8 ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2] ("C3"): 1e9 rows, one int64 key with 1e7 groups,
sum(float64) -- `DT[:, sum(f.v), by(f.k)]`.  It fits one GPU (16 GB of input), so it is
the N=1 workload too; for N>1 the SAME 1e9 rows are row-sharded over the ranks (strong
scaling, as the north star states "1e9 rows at 1/2/4/8"): local fused groupby-sum,
range-partitioned all-to-all of partials over RCCL, merge on the owner.

A step = one full pass of the hot path over the (HBM-resident) batch.  Inputs are in HBM before the
timed region; outputs stay in HBM.  One JSON line on rank 0:

  roofline      the dominant kernel of the timed region (largest total time), timed with HIP events
                around every launch on the stream the library launches on
  cpu_baseline  kind "reference": the UNMODIFIED reference (oracle/_ref, built by oracle/build_ref.sh)
                evaluating the same query with its own thread pool on the FIRST `--cpu-sample` rows of the
                very tensors the GPU leg ran on (copied to the host), fresh Frame per run; the rate of the
                CPU restatement (oracle/, OpenMP) is kept as `port`
  parity        outside the timed region: (1) GPU vs the reference on that sample -- group keys bit-exact,
                sums <= 1e-6; (2) GPU vs the OpenMP port on ALL rows of the workload -- keys and group
                sizes bit-exact, sums <= 1e-6.  A mismatch aborts the bench.
  configs       the other BASELINE.json configs (C1, C2, C4, C5) and the hard-keys variant of C3 at their
                largest single-GPU sizes: ms, rows/s, algorithmic GB/s (SURVEY 8(d) byte formulas), fraction
                of 8 TB/s and the dominant kernel of each
  host_mode     PCIe-inclusive rate of the host-pointer mode (never `value`)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_ROW = 16         # key 8 B + value 8 B read once (+ ng*(8+8) written)
RTOL, ATOL = 1e-6, 1e-9        # float64 sums: BASELINE.json's tolerance (+ an absolute floor for sums near 0)


def pmc_records():
    """(profiles/pmc_traffic.json, None) -- or ({}, why) when the counter passes were recorded on ANOTHER build of the library:
    the file carries dthip_build_id() of the library that was profiled (scripts/prof.sh), a hash of the library's sources"""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception as e:
        return {}, "no profiles/pmc_traffic.json (%s)" % type(e).__name__
    from datatable_amd import _lib
    bid = _lib.load().dthip_build_id().decode()
    if doc.get("_build_id") != bid:
        return {}, "stale: recorded on library build %s, this library is build %s -- re-run scripts/prof.sh" % (doc.get("_build_id"), bid)
    return doc, None


def to_host(t, chunk=1 << 27):
    """device tensor -> numpy, through a pinned bounce buffer (the inputs of the CPU legs are the GPU leg's)"""
    import numpy as np
    import torch
    out = np.empty(t.numel(), dtype={torch.int64: np.int64, torch.float64: np.float64, torch.int32: np.int32}[t.dtype])
    ot = torch.from_numpy(out)
    pin = torch.empty(min(chunk, t.numel()), dtype=t.dtype).pin_memory()
    for s in range(0, t.numel(), chunk):
        e = min(s + chunk, t.numel())
        pin[:e - s].copy_(t[s:e]); torch.cuda.synchronize()
        ot[s:e].copy_(pin[:e - s])
    return out


def sums_close(got, exp):
    import numpy as np
    d = np.abs(got - exp)
    ok = bool(np.all(d <= ATOL + RTOL * np.abs(exp)))
    big = np.abs(exp) > 1e-3
    return ok, float(d.max()) if len(d) else 0.0, float((d[big] / np.abs(exp[big])).max()) if big.any() else 0.0


def gpu_groupby_sum(ctx, keys, vals, with_counts=False):
    """(group keys, sums[, counts]) of DT[:, sum(f.v)[, count()], by(f.k)] on device tensors, as numpy"""
    import torch
    from datatable_amd.torch_bridge import devcol
    aggs = [("sum", 0)] + ([("count0", None)] if with_counts else [])
    r = ctx.groupby_agg([devcol(keys)], [devcol(vals)], aggs, nrows=keys.numel())
    ng = r.ngroups
    gk = torch.empty(ng, dtype=torch.int64, device=keys.device)
    gs = torch.empty(ng, dtype=torch.float64, device=keys.device)
    r.key_into(0, gk.data_ptr()); r.agg_into(0, gs.data_ptr())
    out = [gk, gs]
    if with_counts:
        gc = torch.empty(ng, dtype=torch.int64, device=keys.device)
        r.agg_into(1, gc.data_ptr()); out.append(gc)
    torch.cuda.synchronize()
    r.free()
    return [t.cpu().numpy() for t in out]


def reference_leg(ctx, keys, vals, sample_rows, thread_list):
    """cpu_baseline (kind "reference") + parity vs the reference on the first `sample_rows` rows."""
    import numpy as np
    from oracle import ref
    dt = ref.load()
    if dt is None:
        return None, None
    S = min(sample_rows, keys.numel())
    hk, hv = to_host(keys[:S]), to_host(vals[:S])
    ncpu = os.cpu_count() or 1
    by_threads, res = {}, None
    for t in thread_list:
        nt = ncpu if t == 0 else min(t, ncpu)
        if str(nt) in by_threads:
            continue
        res, sec = ref.groupby_agg({"k": hk, "v": hv}, ["k"], [("sum", "v")], nthreads=nt, reps=1)
        by_threads[str(nt)] = {"seconds": sec, "rows_per_s": S / sec}
    best = max(by_threads, key=lambda k: by_threads[k]["rows_per_s"])
    rk = res[:, 0].to_numpy().ravel()
    rs = res[:, 1].to_numpy().ravel()
    gk, gs = gpu_groupby_sum(ctx, keys[:S], vals[:S])
    keys_ok = bool(rk.dtype == np.int64 and np.array_equal(gk, rk))
    s_ok, max_abs, max_rel = sums_close(gs, rs) if keys_ok else (False, None, None)
    parity = {"against": "reference (oracle/_ref, unmodified h2oai/datatable %s)" % dt.__version__,
              "rows": S, "groups": int(len(rk)), "keys_bit_exact": keys_ok, "sums_within_tol": s_ok,
              "sum_max_abs_err": max_abs, "sum_max_rel_err": max_rel, "rtol": RTOL, "atol": ATOL}
    base = {"value": by_threads[best]["rows_per_s"], "unit": "rows/s", "cores": int(best), "kind": "reference",
            "sample": "first %d rows (%.0f%%) of the workload's own key/value tensors copied to the host: "
                      "DT[:, sum(f.v), by(f.k)] on datatable %s (oracle/_ref), dt.options.nthreads=%s, fresh Frame, "
                      "%.1f s" % (S, 100.0 * S / keys.numel(), dt.__version__, best, by_threads[best]["seconds"]),
            "by_threads": by_threads, "cpu_model": ref.cpu_model(), "host_cores_available": ncpu,
            "groups_in_sample": int(len(rk))}
    return base, parity


def port_leg(ctx, keys, vals, last_gpu, threads):
    """full-size parity: the OpenMP restatement (oracle/dt_oracle.c) on ALL rows vs the GPU result; its rate is
    the `port` CPU number."""
    import numpy as np
    from oracle import oracle as o
    o.lib()
    hk, hv = to_host(keys), to_host(vals)
    n = len(hk)
    o.set_threads(threads)
    try:
        t0 = time.perf_counter()
        ri, off = o.group([hk])
        es = o.reduce("sum", hv, ri, off)
        sec = time.perf_counter() - t0
        ek = hk[ri[off[:-1]]]
        ec = np.diff(off).astype(np.int64)
    finally:
        o.set_threads(1)
    del ri, hk, hv
    gk, gs, gc = last_gpu
    keys_ok = bool(np.array_equal(gk, ek))
    cnt_ok = bool(keys_ok and np.array_equal(gc, ec))
    s_ok, max_abs, max_rel = sums_close(gs, es) if keys_ok else (False, None, None)
    parity = {"against": "oracle/dt_oracle.c (OpenMP restatement, pinned to reference goldens)", "rows": n,
              "groups": int(len(ek)), "keys_bit_exact": keys_ok, "group_sizes_bit_exact": cnt_ok,
              "sums_within_tol": s_ok, "sum_max_abs_err": max_abs, "sum_max_rel_err": max_rel, "rtol": RTOL, "atol": ATOL}
    port = {"value": n / sec, "unit": "rows/s", "cores": threads, "kind": "port",
            "sample": "all %d rows: oracle group()+sum on %d OpenMP threads, %.2f s" % (n, threads, sec)}
    return port, parity


# ---- the other BASELINE configs: timed, then VERIFIED at full size, then the reference timed on a sample -----------
class Budget:
    """wall-clock budget of the optional legs (verification, CPU samples): a leg that would start after the budget is
    spent is skipped and says so in the line -- the one JSON line must come out whatever the host's speed"""

    def __init__(self, seconds):
        self.t0, self.seconds = time.perf_counter(), seconds

    def left(self):
        return self.seconds - (time.perf_counter() - self.t0)

    def ok(self, need=0.0):
        return self.left() > need


def _cmp_exact(got, exp):
    import numpy as np
    return bool(got.dtype == exp.dtype and got.shape == exp.shape and np.array_equal(got, exp, equal_nan=(exp.dtype.kind == "f")))


def verify_agg(ctx, name, keys_t, vals_t, aggs, threads, keep=None, host=None):
    """GPU result of DT[:, aggs, by(keys)] on ALL rows of the config vs the OpenMP oracle (oracle/dt_oracle.c): group
    keys, counts, min / max bit-exact, float64 sums / means within RTOL (+ATOL); raises on mismatch.
    keep (dict): receives the single-GPU result arrays ("keys", "aggs") for the sharded-vs-single-GPU comparison"""
    import numpy as np
    from oracle import oracle as o
    from datatable_amd.torch_bridge import devcol
    t0 = time.perf_counter()
    r = ctx.groupby_agg([devcol(k) for k in keys_t], [devcol(v) for v in vals_t], aggs, nrows=keys_t[0].numel())
    gk = [r.key(i) for i in range(len(keys_t))]
    ga = [r.agg(a) for a in range(len(aggs))]
    r.free()
    if keep is not None:
        keep["keys"], keep["aggs"] = gk, ga
    hk, hv = host if host is not None else ([to_host(k) for k in keys_t], [to_host(v) for v in vals_t])
    o.lib(); o.set_threads(threads)
    try:
        ri, off = o.group(hk)
        first = ri[off[:-1]]
        keys_ok = all(_cmp_exact(gk[i], hk[i][first]) for i in range(len(hk)))
        out = {"against": "oracle/dt_oracle.c, %d OpenMP threads" % threads, "rows": int(len(ri)), "groups": int(len(off) - 1),
               "keys_bit_exact": keys_ok}
        worst_abs, worst_rel, ok_all = 0.0, 0.0, keys_ok
        for a, (op, c) in enumerate(aggs):
            if not keys_ok:
                break
            exp = np.diff(off).astype(np.int64) if c is None else o.reduce(op, hv[c], ri, off)
            if exp.dtype.kind == "f" and op in ("sum", "mean"):
                ok, ma, mr = sums_close(ga[a], exp)
                worst_abs, worst_rel = max(worst_abs, ma), max(worst_rel, mr)
            else:
                ok = _cmp_exact(ga[a], exp)
            out["%s(%s)" % (op, "" if c is None else "v%d" % c)] = bool(ok)
            ok_all = ok_all and ok
    finally:
        o.set_threads(1)
    out.update({"ok": bool(ok_all), "sum_max_abs_err": worst_abs, "sum_max_rel_err": worst_rel, "rtol": RTOL, "atol": ATOL,
                "seconds": time.perf_counter() - t0})
    return out, (hk, hv)


def verify_c5(ctx, k_t, x_t, threads, keep=None):
    """config 5 on ALL rows: the filter's RowIndex, the rows in grouped order (key, x, the composed RowIndex riding
    through the sort) and the offsets, all bit-exact against the oracle's filter -> gather -> group.
    keep (dict): receives the single-GPU result arrays (offsets, key / x / composed-RowIndex columns)"""
    import numpy as np
    import torch
    from oracle import oracle as o
    from datatable_amd.torch_bridge import devcol
    t0 = time.perf_counter()
    n = k_t.numel()
    dev = k_t.device
    # (1) the TIMED form: one call, dthip_filter_groupby_rows (fused route: filter + key transform + first sort level in one
    # sweep, csrc/tlsort.hip)
    rf = ctx.filter_groupby_rows(devcol(x_t), ">", 0.0, [devcol(k_t)], [devcol(k_t), devcol(x_t)], nrows=n, want_rowindex=True)
    npf = rf.nrows
    f_off = torch.empty(rf.ngroups + 1, dtype=torch.int32, device=dev); rf.offsets_into(f_off.data_ptr())
    f_k = torch.empty(npf, dtype=torch.int64, device=dev); f_x = torch.empty(npf, dtype=torch.float64, device=dev)
    f_ri = torch.empty(npf, dtype=torch.int32, device=dev)
    rf.col_into(0, f_k.data_ptr()); rf.col_into(1, f_x.data_ptr()); rf.rowindex_into(f_ri.data_ptr())
    torch.cuda.synchronize()
    rf.free()
    # (2) the two statements as two calls (round 4's form): filter -> RowIndex + the view's columns, then the rows in grouped
    # order with the filter's RowIndex riding along
    ri_t = torch.empty(n, dtype=torch.int32, device=dev)
    kb = torch.empty(n, dtype=torch.int64, device=dev)
    xb = torch.empty(n, dtype=torch.float64, device=dev)
    npass = ctx.filter_take_dev(devcol(x_t), ">", 0.0, [devcol(k_t), devcol(x_t)], n, ri_t.data_ptr(), [kb.data_ptr(), xb.data_ptr()])
    r = ctx.groupby_rows([devcol(kb[:npass])], [devcol(kb[:npass]), devcol(xb[:npass]), devcol(ri_t[:npass])], nrows=npass, want_rowindex=False)
    t_off = torch.empty(r.ngroups + 1, dtype=torch.int32, device=dev); r.offsets_into(t_off.data_ptr())
    t_k = torch.empty(npass, dtype=torch.int64, device=dev); t_x = torch.empty(npass, dtype=torch.float64, device=dev)
    t_ri = torch.empty(npass, dtype=torch.int32, device=dev)
    r.col_into(0, t_k.data_ptr()); r.col_into(1, t_x.data_ptr()); r.col_into(2, t_ri.data_ptr())
    torch.cuda.synchronize()
    r.free()
    one_call_equals_two_calls = bool(npf == npass and torch.equal(f_off, t_off) and torch.equal(f_ri, t_ri) and torch.equal(f_k, t_k)
                                     and torch.equal(f_x.view(torch.int64), t_x.view(torch.int64)))
    # ... and the filter's own RowIndex is the composed RowIndex sorted ascending
    filter_is_sorted_composed = bool(torch.equal(torch.sort(f_ri).values, ri_t[:npass])) if npf == npass else False
    del t_off, t_k, t_x, t_ri
    g_off, g_k, g_x, g_ri = to_host(f_off), to_host(f_k), to_host(f_x), to_host(f_ri)
    del f_off, f_k, f_x, f_ri
    if keep is not None:
        keep.update(offsets=g_off, k=g_k, x=g_x, ri=g_ri)
    g_filter = to_host(ri_t[:npass])
    del ri_t, kb, xb
    torch.cuda.empty_cache()
    hk, hx = to_host(k_t), to_host(x_t)
    o.lib(); o.set_threads(threads)
    try:
        fri = o.filter_cmp(hx, ">", 0.0)
        filter_ok = _cmp_exact(g_filter, fri)
        kv = hk[fri]
        p, off = o.group([kv])
        comp = fri[p]                                  # RowIndex composition ab*bc (rowindex_array.cc:258-269)
        res = {"against": "oracle/dt_oracle.c, %d OpenMP threads" % threads, "rows": int(n), "rows_passing": int(len(fri)),
               "groups": int(len(off) - 1), "filter_rowindex_bit_exact": filter_ok,
               "one_call_equals_two_calls_bit_exact": one_call_equals_two_calls,
               "sorted_composed_rowindex_is_filter_rowindex_bit_exact": filter_is_sorted_composed,
               "offsets_bit_exact": _cmp_exact(g_off, off), "composed_rowindex_bit_exact": _cmp_exact(g_ri, comp),
               "key_column_bit_exact": _cmp_exact(g_k, hk[comp]), "x_column_bit_exact": _cmp_exact(g_x, hx[comp])}
    finally:
        o.set_threads(1)
    res["ok"] = all(v for k, v in res.items() if k.endswith("bit_exact"))
    res["seconds"] = time.perf_counter() - t0
    return res, (hk, hx)


def verify_sgrp(ctx, k_t, v_t, threads):
    """the LITERAL seam on ALL rows: dthip_groupby's RowIndex and offsets bit-exact against the oracle's group()
    (sort.cc:1411-1495), dthip_reduce(SUM) THROUGH that RowIndex within RTOL of the oracle's reducer (column/sumprod.h:34-59)"""
    import numpy as np
    import torch
    from oracle import oracle as o
    from datatable_amd.torch_bridge import devcol
    t0 = time.perf_counter()
    n = k_t.numel()
    r = ctx.groupby([devcol(k_t)], nrows=n, want_rowindex=True)
    ng = r.ngroups
    sums = torch.empty(ng, dtype=torch.float64, device=k_t.device)
    ctx.reduce_dev("sum", devcol(v_t), r.rowindex_ptr, r.offsets_ptr, ng, n, sums.data_ptr())
    g_ri = torch.empty(n, dtype=torch.int32, device=k_t.device); r.rowindex_into(g_ri.data_ptr())
    g_off = torch.empty(ng + 1, dtype=torch.int32, device=k_t.device); r.offsets_into(g_off.data_ptr())
    torch.cuda.synchronize()
    r.free()
    h_ri, h_off, h_s = to_host(g_ri), to_host(g_off), sums.cpu().numpy()
    del g_ri, g_off, sums
    hk, hv = to_host(k_t), to_host(v_t)
    o.lib(); o.set_threads(threads)
    try:
        ri, off = o.group([hk])
        es = o.reduce("sum", hv, ri, off)
    finally:
        o.set_threads(1)
    ri_ok, off_ok = _cmp_exact(h_ri, ri), _cmp_exact(h_off, off)
    s_ok, ma, mr = sums_close(h_s, es) if (ri_ok and off_ok) else (False, None, None)
    res = {"against": "oracle/dt_oracle.c, %d OpenMP threads" % threads, "rows": int(n), "groups": int(len(off) - 1),
           "rowindex_bit_exact": ri_ok, "offsets_bit_exact": off_ok, "sums_within_tol": s_ok, "sum_max_abs_err": ma,
           "sum_max_rel_err": mr, "rtol": RTOL, "atol": ATOL, "ok": bool(ri_ok and off_ok and s_ok), "seconds": time.perf_counter() - t0}
    return res, None


def adversarial_legs(ctx, dev, n, groups, steps, verify, threads, budget, c3_ms):
    """Inputs the DEFAULT path's guesses do not like: the key range and NA-freeness are guessed from
    samples and verified on every row, so one outlier key / one NA costs a second sweep; sorted and hot keys serialise LDS
    atomics.  Every variant of C3 is timed, its retries counted (dthip_last_call_stats) and its result verified on ALL rows."""
    import numpy as np
    import torch
    from datatable_amd.torch_bridge import devcol
    (k,), (v,) = gen_c3(dev, 0, 1, n, groups)
    out = {}

    def timed(name, keys, vals, aggs, desc, host=None):
        kc, vc = devcol(keys), devcol(vals)
        def run():
            r = ctx.groupby_agg([kc], [vc], aggs, nrows=n); ng = r.ngroups; r.free(); return ng
        run(); torch.cuda.synchronize()
        st = ctx.last_call_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            ng = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        ctx.profile_reset(); ctx.profile(True); run(); torch.cuda.synchronize(); ctx.profile(False)
        prof = {nm: ctx.profile_get(nm) for nm in ctx.profile_names()}
        out[name] = {"workload": desc, "rows": n, "groups": int(ng), "ms": ms, "rows_s": n / (ms * 1e-3), "x_C3": (ms / c3_ms) if c3_ms else None,
                     "retries": {"key_range": st["retries_key_range"], "na_guess": st["retries_na_guess"]}, "path": st["path"],
                     "outlier_rows_listed": st["outlier_rows_listed"],
                     "kernel_ms": {q: round(w[0], 4) for q, w in sorted(prof.items(), key=lambda kv: -kv[1][0])[:6]}}
        if verify and budget.ok(40):
            par, _ = verify_agg(ctx, name, [keys], [vals], aggs + ([("count0", None)] if ("count0", None) not in aggs else []), threads, host=host)
            out[name]["parity"] = par
            out[name]["ok"] = par["ok"]
            assert par["ok"], (name, par)
        elif verify:
            out[name]["parity"] = {"skipped": "time budget (%.0f s) spent" % budget.seconds}
        torch.cuda.empty_cache(); ctx.trim()

    hk = hv = None
    if verify:
        hk, hv = to_host(k), to_host(v)
    # (1) ONE key outside the range the 2^17-piece sample sees (widened by 1/64): the partition's per-row check raises the
    # flag, the query runs again with the exact range (DTHIP_RETRY_EXACT).  A row the sample does not visit is looked for.
    outlier_val = 3 * groups
    for row in (n // 3 + 7, n // 3 + 100_003, n // 5 + 11):
        old = int(k[row].item()); k[row] = outlier_val
        r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0)], nrows=n); r.free()
        st0 = ctx.last_call_stats()
        hit = st0["retries_key_range"] > 0 or st0["outlier_rows_listed"] > 0
        if hit:
            break
        k[row] = old
    if hk is not None:
        hk_old = int(hk[row]); hk[row] = outlier_val
    timed("C3_outlier", k, v, [("sum", 0)], "C3 with ONE key = %d at row %d (outside the sampled range [0,%d): round 6 lists the row and splices its group in; rounds 2-5 ran a second sweep with the exact range)" % (outlier_val, row, groups),
          host=([hk], [hv]) if hk is not None else None)
    out["C3_outlier"]["sample_missed_the_outlier"] = bool(hit)
    k[row] = old
    if hk is not None:
        hk[row] = hk_old
    # (2) ONE NA in a value column whose 65536-row sample shows none, under a reducer that needs valid counts (mean): the
    # aggregation runs again with counters (DTHIP_RETRY_NA); the same query without the NA is timed beside it
    aggs2 = [("sum", 0), ("mean", 0)]
    timed("C3_mean_clean", k, v, aggs2, "DT[:, [sum(f.v), mean(f.v)], by(f.k)] on C3's rows, no NA", host=([hk], [hv]) if hk is not None else None)
    rowv = n // 2 + 12_345
    oldv = float(v[rowv].item()); v[rowv] = float("nan")
    if hv is not None:
        hv[rowv] = np.nan
    timed("C3_na_planted", k, v, aggs2, "the same with ONE NaN planted at row %d (value column guessed NA-free -> the NA row is skipped and counted apart: no second aggregation since round 6)" % rowv,
          host=([hk], [hv]) if hk is not None else None)
    out["C3_na_planted"]["x_clean"] = out["C3_na_planted"]["ms"] / out["C3_mean_clean"]["ms"]
    v[rowv] = oldv
    if hv is not None:
        hv[rowv] = oldv
    # (3) sorted keys: whole waves address one bucket / one slot (cluster variants)
    ks = torch.sort(k).values
    timed("C3_sorted", ks, v, [("sum", 0)], "C3 with the key column sorted ascending", host=([to_host(ks)], [hv]) if hk is not None else None)
    del ks
    torch.cuda.empty_cache()
    # (4) one hot key: 7 % of the rows
    g = torch.Generator(device=dev); g.manual_seed(4242)
    kh = torch.where(torch.rand(n, device=dev, generator=g) < 0.07, torch.tensor(12_345, dtype=torch.int64, device=dev), k)
    timed("C3_hotkey", kh, v, [("sum", 0)], "C3 with 7 % of the rows in ONE key", host=([to_host(kh)], [hv]) if hk is not None else None)
    del kh, k, v, hk, hv
    torch.cuda.empty_cache(); ctx.trim()
    return out


def run_configs(ctx, dev, which, steps, scale, verify, cpu_sample, ref_threads, port_threads, budget):
    import numpy as np
    import torch
    from datatable_amd.torch_bridge import devcol
    from oracle import ref
    g = torch.Generator(device=dev)
    out = {}
    dt_ref = ref.load() if cpu_sample else None

    def measure(name, n, alg_bytes, run, desc):
        run(); torch.cuda.synchronize()                      # warm-up: allocator, first touch
        t0 = time.perf_counter()
        for _ in range(steps):
            ng = run()
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / steps
        ctx.profile_reset(); ctx.profile(True); run(); torch.cuda.synchronize(); ctx.profile(False)
        prof = {nm: ctx.profile_get(nm) for nm in ctx.profile_names()}
        dom = max(prof, key=lambda k: prof[k][0]) if prof else None
        out[name] = {"workload": desc, "rows": n, "groups": int(ng), "ms": sec * 1e3, "rows_s": n / sec,
                     "alg_bytes": alg_bytes, "alg_GBs": alg_bytes / sec / 1e9, "frac": alg_bytes / sec / 1e9 / HBM_PEAK_GBS,
                     "dominant_kernel": dom, "dominant_ms": prof[dom][0] if dom else None,
                     "kernel_ms": {k: round(v[0], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]}}
        # HBM bytes of one whole query from the rocprofv3 PMC passes recorded under profiles/ (scripts/prof.sh):
        # FETCH_SIZE x 2 + WRITE_SIZE over every kernel of the query; amplification = traffic / algorithmic bytes
        doc, why = pmc_records()
        rec = doc.get("_configs", {}).get(name)
        if why:
            out[name]["traffic_note"] = why
        if rec and rec.get("rows") == n:
            out[name]["traffic"] = rec["hbm_bytes_per_query"]
            out[name]["amplification"] = rec["hbm_bytes_per_query"] / alg_bytes
            out[name]["traffic_GBs"] = rec["hbm_bytes_per_query"] / sec / 1e9
        else:
            out[name]["traffic"] = None
        torch.cuda.empty_cache(); ctx.trim()

    def check(name, fn):
        """full-size verification outside the timed loop; a mismatch aborts the bench"""
        if not verify:
            return None
        if not budget.ok(30):
            out[name]["parity"] = {"skipped": "time budget (%.0f s) spent" % budget.seconds}
            return None
        par, host = fn()
        out[name]["parity"] = par
        assert par["ok"], (name, par)
        return host

    def cpu(name, rows, fn):
        """the reference's own CPU path on the first `rows` rows of this config's tensors"""
        if dt_ref is None or not budget.ok(40):
            if dt_ref is not None:
                out[name]["cpu_reference"] = {"skipped": "time budget (%.0f s) spent" % budget.seconds}
            return
        try:
            sec = fn()
            out[name]["cpu_reference"] = {"rows": rows, "seconds": sec, "cpu_rows_s": rows / sec, "threads": ref_threads,
                                          "gpu_over_cpu": out[name]["rows_s"] / (rows / sec),
                                          "what": "unmodified reference (oracle/_ref), dt.options.nthreads=%d, fresh Frame, first %d rows" % (ref_threads, rows)}
        except Exception as e:
            out[name]["cpu_reference"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    for c in which:
        if c == "C1":
            n = int(1e6 * scale); g.manual_seed(1235)
            k = torch.randint(0, 100, (n,), dtype=torch.int32, device=dev, generator=g)
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            def run():
                r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0)], nrows=n); ng = r.ngroups; r.free(); return ng
            measure(c, n, n * 12 + 100 * 12, run, "DT[:, sum(f.v), by(f.k)], int32 key in [0,100), float64")
            host = check(c, lambda: verify_agg(ctx, c, [k], [v], [("sum", 0), ("count0", None)], port_threads))
            if host:
                cpu(c, n, lambda: ref.groupby_agg({"k": host[0][0], "v": host[1][0]}, ["k"], [("sum", "v")], nthreads=ref_threads)[1])
            del k, v, host
        elif c == "C2":
            n = int(1e8 * scale); g.manual_seed(1236)
            k = torch.randint(0, 100_000, (n,), dtype=torch.int64, device=dev, generator=g)
            vs = [torch.randn(n, dtype=torch.float64, device=dev, generator=g) for _ in range(4)]
            aggs = [(op, i) for op in ("sum", "mean", "min", "max") for i in range(4)]
            def run():
                r = ctx.groupby_agg([devcol(k)], [devcol(x) for x in vs], aggs, nrows=n); ng = r.ngroups; r.free(); return ng
            measure(c, n, n * 40 + 100_000 * 136, run, "DT[:, [sum,mean,min,max](f[1:]), by(f.k)], int64 key in [0,1e5), 4 x float64")
            host = check(c, lambda: verify_agg(ctx, c, [k], vs, aggs + [("count0", None)], port_threads))
            if host:
                S = min(n, cpu_sample)
                cols = {"k": host[0][0][:S]}
                cols.update({"v%d" % i: host[1][i][:S] for i in range(4)})
                cpu(c, S, lambda: ref.groupby_agg(cols, ["k"], [(op, "v%d" % i) for op, i in aggs], nthreads=ref_threads)[1])
            del k, vs, host
        elif c in ("C3_hard",):
            n = int(1e9 * scale); g.manual_seed(1240)
            pool = torch.randint(-2**62, 2**62, (10_000_000,), dtype=torch.int64, device=dev, generator=g)
            k = pool[torch.randint(0, 10_000_000, (n,), dtype=torch.int64, device=dev, generator=g)]
            del pool
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            def run():
                r = ctx.groupby_agg([devcol(k)], [devcol(v)], [("sum", 0)], nrows=n); ng = r.ngroups; r.free(); return ng
            measure(c, n, n * 16 + 10_000_000 * 16, run, "C3 with 63-bit keys drawn from a pool of 1e7 values (SURVEY 8(d) hard keys)")
            host = check(c, lambda: verify_agg(ctx, c, [k], [v], [("sum", 0), ("count0", None)], port_threads))
            if host:
                S = min(n, cpu_sample)
                cpu(c, S, lambda: ref.groupby_agg({"k": host[0][0][:S], "v": host[1][0][:S]}, ["k"], [("sum", "v")], nthreads=ref_threads)[1])
            del k, v, host
        elif c == "C3_sgrp":
            # the LITERAL seam north_star names (VERDICT r05 missing 2): group() -> RowIndex + Groupby (S-grp, sort.h:56-58,
            # sort.cc:1411-1495), then the reducer gathering value[rowindex[i]] per group (S-red, column/sumprod.h:34-59) --
            # two calls, the RowIndex materialised in HBM, on the C3 tensors.  Algorithmic bytes: key 8 + value 8 read,
            # RowIndex 4 written per row ("+ n x 4 B when the RowIndex is a requested output"), offsets + sums per group
            n = int(1e9 * scale)
            (k,), (v,) = gen_c3(dev, 0, 1, n, 10_000_000)
            sums = torch.empty(min(n, 10_000_000) + 16, dtype=torch.float64, device=dev)
            def run():
                r = ctx.groupby([devcol(k)], nrows=n, want_rowindex=True)
                ng = r.ngroups
                ctx.reduce_dev("sum", devcol(v), r.rowindex_ptr, r.offsets_ptr, ng, n, sums.data_ptr())
                r.free(); return ng
            measure(c, n, n * 20 + 10_000_000 * 12, run, "C3 as TWO calls: dthip_groupby(want_rowindex=1) -> RowIndex + offsets in HBM, then "
                                                       "dthip_reduce(SUM, v, rowindex, offsets): the S-grp -> S-red seam of the in-tree binding")
            check(c, lambda: verify_sgrp(ctx, k, v, port_threads))
            del k, v, sums
        elif c == "C4":
            n = int(1e9 * scale); g.manual_seed(1238)
            a = torch.randint(0, 3163, (n,), dtype=torch.int32, device=dev, generator=g)
            b = torch.randint(0, 3163, (n,), dtype=torch.int32, device=dev, generator=g)
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            def run():
                r = ctx.groupby_agg([devcol(a), devcol(b)], [devcol(v)], [("count0", None), ("sum", 0)], nrows=n)
                ng = r.ngroups; r.free(); return ng
            measure(c, n, n * 16 + 10_004_569 * 24, run, "DT[:, [count(), sum(f.v)], by(f.a, f.b)], 2 x int32 keys in [0,3163), float64")
            host = check(c, lambda: verify_agg(ctx, c, [a, b], [v], [("count0", None), ("sum", 0)], port_threads))
            if host:
                S = min(n, cpu_sample)
                cpu(c, S, lambda: ref.groupby_agg({"a": host[0][0][:S], "b": host[0][1][:S], "v": host[1][0][:S]}, ["a", "b"],
                                                  [("count", None), ("sum", "v")], nthreads=ref_threads)[1])
            del a, b, v, host
        elif c == "C5":
            n = int(1e9 * scale); g.manual_seed(1239)
            k = torch.randint(0, 100_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
            x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            ri = torch.empty(n, dtype=torch.int32, device=dev)
            kbuf = torch.empty(n, dtype=torch.int64, device=dev)
            xbuf = torch.empty(n, dtype=torch.float64, device=dev)
            def run_two_calls():
                # V = DT[f.x > 0, :]; V[:, :, by(f.k)] (two-step form) as TWO calls (round 4): filter -> RowIndex +
                # the view's columns in one sweep, then the rows in grouped order (key, x and the RowIndex ride through the sort)
                npass = ctx.filter_take_dev(devcol(x), ">", 0.0, [devcol(k), devcol(x)], n, ri.data_ptr(),
                                            [kbuf.data_ptr(), xbuf.data_ptr()])
                kv, xv = kbuf[:npass], xbuf[:npass]
                r = ctx.groupby_rows([devcol(kv)], [devcol(kv), devcol(xv), devcol(ri[:npass])], nrows=npass, want_rowindex=False)
                ng = r.ngroups; r.free(); return ng
            def run():
                # the same two statements as ONE call (round 5, dthip_filter_groupby_rows): same outputs -- offsets, key and x
                # columns in grouped order, the composed RowIndex -- the filter fused into the first sort level
                r = ctx.filter_groupby_rows(devcol(x), ">", 0.0, [devcol(k)], [devcol(k), devcol(x)], nrows=n, want_rowindex=True)
                ng = r.ngroups; r.free(); return ng
            measure("C5_two_calls", n, int(n * 30.4), run_two_calls, "config 5 as dthip_filter_take + dthip_groupby_rows (round 4's form)")
            two = out.pop("C5_two_calls")
            del ri, kbuf, xbuf
            torch.cuda.empty_cache(); ctx.trim()
            measure(c, n, int(n * 30.4), run, "V = DT[f.x > 0, :]; V[:, :, by(f.k)], int64 key in [0,1e8), float64 x, ~50% pass; ONE call "
                                               "(dthip_filter_groupby_rows): offsets, key and x columns in grouped order, composed RowIndex")
            out[c]["two_calls"] = {"ms": two["ms"], "kernel_ms": two["kernel_ms"], "what": two["workload"]}
            torch.cuda.empty_cache(); ctx.trim()
            host = check(c, lambda: verify_c5(ctx, k, x, port_threads))
            if host:
                S = min(n, cpu_sample)
                cpu(c, S, lambda: ref.filter_group_rows({"k": host[0][:S], "x": host[1][:S]}, "x", "k", nthreads=ref_threads)[1])
            del k, x, host
        torch.cuda.empty_cache(); ctx.trim()
    return out


def dist_one_rank_leg(ctx, keys, vals, steps, local_ms):
    """the sharded code path (RCCL all-gathers, all-to-all-v to self, merge) with ONE rank: its fixed cost over the
    plain local call, per kernel -- what every rank of an N-GPU run pays on top of its 1/N of the rows"""
    import torch
    from datatable_amd.torch_bridge import devcol
    from datatable_amd.engine import comm_unique_id
    ctx.comm_init(0, 1, comm_unique_id())
    try:
        kc, vc = devcol(keys), devcol(vals)
        n = keys.numel()
        for _ in range(2):
            ctx.sharded_groupby_agg([kc], [vc], [("sum", 0)], nrows=n).free()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.sharded_groupby_agg([kc], [vc], [("sum", 0)], nrows=n).free()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        ctx.profile_reset(); ctx.profile(True)
        ctx.sharded_groupby_agg([kc], [vc], [("sum", 0)], nrows=n).free()
        torch.cuda.synchronize(); ctx.profile(False)
        prof = {nm: ctx.profile_get(nm) for nm in ctx.profile_names()}
        return {"ms_per_step": ms, "local_ms_per_step": local_ms, "overhead_ms": ms - local_ms, "rows": n,
                "what": "dthip_sharded_groupby_agg on a 1-rank RCCL communicator: 2 all-gathers (samples; send counts + status), "
                        "all-to-all-v to self, merge of the partial groups",
                "kernel_ms": {k: round(v[0], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]},
                "kernel_launches": int(sum(v[1] for v in prof.values()))}
    finally:
        ctx.comm_destroy()


def host_mode_leg(ctx, rows):
    """PCIe-inclusive: numpy buffers in, numpy results out (DTHIP_HOST) -- what the reference-side shim uses"""
    import numpy as np
    rng = np.random.default_rng(99)
    k = rng.integers(0, 1_000_000, rows, dtype=np.int64)
    v = rng.standard_normal(rows)
    def timed():
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            r = ctx.groupby_agg([k], [v], [("sum", 0)])
            s = r.agg(0); kk = r.key(0)
            r.free()
            sec = time.perf_counter() - t0
            best = sec if best is None or sec < best else best
        return best, len(kk)
    best, ng = timed()
    out = {"rows": rows, "groups": int(ng), "ms": best * 1e3, "rows_s": rows / best, "input_GBs": rows * 16 / best / 1e9,
           "note": "DTHIP_HOST mode: pageable numpy buffers in, numpy results out; includes PCIe both ways"}
    try:
        t0 = time.perf_counter()
        ctx.host_register(k); ctx.host_register(v)
        reg = time.perf_counter() - t0
        pb, _ = timed()
        ctx.host_unregister(k); ctx.host_unregister(v)
        out["registered"] = {"ms": pb * 1e3, "rows_s": rows / pb, "input_GBs": rows * 16 / pb / 1e9, "register_ms": reg * 1e3,
                             "note": "same call on buffers page-locked once with dthip_host_register (DMA without the runtime's bounce copy)"}
    except Exception as e:           # registration is an optimisation: report, do not fail the bench
        out["registered"] = {"error": str(e)[:200]}
    return out


XGMI_GBS_PER_GPU = 7 * 153.0          # 7 point-to-point xGMI links x ~153 GB/s out of every GPU (MI355X_MICROARCH.md)


def shim_resident_leg(steps, rows_c3, groups_c3, rows_c5, raw_c3_ms, raw_c5_ms):
    """The reference-side binding with device-resident columns (integration/datatable_hip_shim.py, options.residency =
    "lazy"): a real datatable.Frame (oracle/_ref), columns uploaded once, then the SAME `DT[...]` expressions repeated --
    C3 as DT[:, sum(f.v), by(f.k)] and config 5 in its two-step form V = DT[f.x > 0, :]; V[:, :, by(f.k)] -- timed from the
    Python statement to the end of the device work (results stay in HBM; the raw C-ABI times stand beside them)"""
    import numpy as np
    import torch
    from oracle import ref
    dtm = ref.load()
    if dtm is None:
        return {"error": "oracle/_ref (the reference build) is not on this machine"}
    from datatable import f, sum as dsum
    from integration import datatable_hip_shim as shim
    old = shim.options.residency
    shim.options.residency = "lazy"
    out = {"residency": "lazy", "steps": steps,
           "what": "datatable.Frame -> shim.Frame.to_device() once; every step = one Python `DT[...]` statement, device work "
                   "finished (dthip_sync), nothing copied to the host"}
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev)
    try:
        sctx = shim._context()
        # ---- C3
        g.manual_seed(1237)
        k = torch.randint(0, groups_c3, (rows_c3,), dtype=torch.int64, device=dev, generator=g)
        v = torch.randn(rows_c3, dtype=torch.float64, device=dev, generator=g)
        hk, hv = to_host(k), to_host(v)
        del k, v
        torch.cuda.empty_cache()
        DT = shim.Frame(k=hk, v=hv)
        t0 = time.perf_counter(); DT.to_device(); sctx.sync(); up = time.perf_counter() - t0
        q = lambda: DT[:, dsum(f.v), shim.by(f.k)]
        R = q(); sctx.sync()
        ng = R.nrows
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter(); R = q(); sctx.sync(); ts.append(time.perf_counter() - t0)
        out["C3"] = {"rows": rows_c3, "groups": int(ng), "upload_s": up, "ms_best": min(ts) * 1e3, "ms_mean": sum(ts) / len(ts) * 1e3,
                     "raw_c_abi_ms": raw_c3_ms, "over_raw": (min(ts) * 1e3 / raw_c3_ms) if raw_c3_ms else None,
                     "rows_per_s": rows_c3 / min(ts), "result": type(R).__name__}
        # first host access of the lazy result, once (PCIe-inclusive, not part of the step)
        t0 = time.perf_counter(); F = R.to_frame(); out["C3"]["download_result_ms"] = (time.perf_counter() - t0) * 1e3
        assert F.nrows == ng and F.names == ("k", "v")
        del DT, R, F, hk, hv
        sctx.trim()
        # ---- C5, two-step form
        g.manual_seed(1239)
        k = torch.randint(0, 100_000_000, (rows_c5,), dtype=torch.int64, device=dev, generator=g)
        x = torch.randn(rows_c5, dtype=torch.float64, device=dev, generator=g)
        hk, hx = to_host(k), to_host(x)
        del k, x
        torch.cuda.empty_cache()
        DT = shim.Frame(k=hk, x=hx)
        DT.to_device(); sctx.sync()
        pred = f.x > 0                    # one FExpr object: its exact threshold is probed once and remembered

        def q5():
            V = DT[pred, :]
            return V, V[:, :, shim.by(f.k)]
        V, R = q5(); sctx.sync()
        ts = []
        for _ in range(steps):
            del V, R
            t0 = time.perf_counter(); V, R = q5(); sctx.sync(); ts.append(time.perf_counter() - t0)
        out["C5"] = {"rows": rows_c5, "rows_passing": int(V.nrows), "ms_best": min(ts) * 1e3, "ms_mean": sum(ts) / len(ts) * 1e3,
                     "raw_c_abi_ms": raw_c5_ms, "over_raw": (min(ts) * 1e3 / raw_c5_ms) if raw_c5_ms else None,
                     "statements": "V = DT[f.x > 0, :]; R = V[:, :, by(f.k)]", "result": type(R).__name__,
                     "note": "the raw figure carries the filter's RowIndex through the sort as a third column; the two-step "
                             "datatable form has no such column"}
        del DT, V, R, hk, hx
        sctx.trim()
    finally:
        shim.options.residency = old
    return out


# ---- --gpus N: the sharded results are CHECKED, not only timed ---------------------------------------------------------
# Every rank draws its row block [r n / W, (r + 1) n / W) from its own seed; rank 0 can therefore rebuild the WHOLE frame
# on its GPU (16 GB at 1e9 rows), run the single-GPU path on it -- itself compared with the OpenMP oracle on all rows,
# exactly as the N = 1 line does -- and compare what the ranks returned (gathered over gloo, outside every timed region)
# with that: keys, counts, offsets and the RowIndex bit for bit, float64 sums within RTOL.  A mismatch aborts all ranks.
def shard_bounds(r, world, n_total):
    return r * n_total // world, (r + 1) * n_total // world


def gen_c3(dev, r, world, n_total, groups):
    import torch
    lo, hi = shard_bounds(r, world, n_total)
    g = torch.Generator(device=dev); g.manual_seed(1234 + 3 + 1000 * r)
    keys = torch.randint(0, groups, (hi - lo,), dtype=torch.int64, device=dev, generator=g)
    vals = torch.randn(hi - lo, dtype=torch.float64, device=dev, generator=g)
    return [keys], [vals]


def gen_c4(dev, r, world, n_total):
    import torch
    lo, hi = shard_bounds(r, world, n_total)
    g = torch.Generator(device=dev); g.manual_seed(1238 + 1000 * r)
    a = torch.randint(0, 3163, (hi - lo,), dtype=torch.int32, device=dev, generator=g)
    b = torch.randint(0, 3163, (hi - lo,), dtype=torch.int32, device=dev, generator=g)
    v = torch.randn(hi - lo, dtype=torch.float64, device=dev, generator=g)
    return [a, b], [v]


def gen_c5(dev, r, world, n_total):
    import torch
    lo, hi = shard_bounds(r, world, n_total)
    g = torch.Generator(device=dev); g.manual_seed(1239 + 1000 * r)
    k = torch.randint(0, 100_000_000, (hi - lo,), dtype=torch.int64, device=dev, generator=g)
    x = torch.randn(hi - lo, dtype=torch.float64, device=dev, generator=g)
    return [k], [x]


def tensor_checksum(ts):
    """order-dependent 64-bit checksums of device tensors (wrapping int64 arithmetic on the bit patterns): enough to tell
    that a shard rebuilt on rank 0 from the rank's seed IS the shard that rank computed on"""
    import torch
    out = []
    for t in ts:
        b = t.view(torch.int64) if t.element_size() == 8 else t.to(torch.int64)
        w = torch.arange(1, b.numel() + 1, dtype=torch.int64, device=t.device)
        out.append(int(((b * 0x9E3779B1 + 12345) * (w | 1)).sum().item()))
        del w, b
    return out


def gather_to_rank0(dist, rank, world, t, chunk=1 << 25):
    """every rank's 1-D device tensor -> ONE numpy array on rank 0, slabs in rank order (gloo point-to-point through a
    pinned bounce buffer; control plane only, never inside a timed region).  Returns (array or None, slab lengths)."""
    import numpy as np
    import torch
    t = t.contiguous().view(-1)
    mine = torch.tensor([t.numel()], dtype=torch.int64)
    ns = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(ns, mine)
    ns = [int(x.item()) for x in ns]
    pin = torch.empty(max(1, min(chunk, max(ns))), dtype=t.dtype).pin_memory()
    if rank == 0:
        out = torch.empty(sum(ns), dtype=t.dtype)
        for s in range(0, ns[0], chunk):
            e = min(s + chunk, ns[0])
            pin[:e - s].copy_(t[s:e]); torch.cuda.synchronize()
            out[s:e].copy_(pin[:e - s])
        pos = ns[0]
        for r in range(1, world):
            for s in range(0, ns[r], chunk):
                e = min(s + chunk, ns[r])
                dist.recv(out[pos + s:pos + e], src=r)
            pos += ns[r]
        return out.numpy(), ns
    for s in range(0, ns[rank], chunk):
        e = min(s + chunk, ns[rank])
        pin[:e - s].copy_(t[s:e]); torch.cuda.synchronize()
        dist.send(pin[:e - s], dst=0)
    return None, ns


def agree(dist, rank, verdict):
    """rank 0's verdict reaches every rank; a failed check takes ALL ranks down (nobody waits for a peer that left)"""
    box = [verdict if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    v = box[0]
    assert v is None or v.get("ok") or v.get("skipped"), v
    return v


def rebuild_frame(gen, dev, rank, world, own, sums_by_rank):
    """rank 0: the whole frame on ITS GPU = the ranks' shards in rank order, every shard drawn again from its seed and
    checked against the checksum the owning rank took of the tensors it really used"""
    import torch
    parts_k, parts_v, same = None, None, True
    for r in range(world):
        kk, vv = own if r == rank else gen(r)
        if r != rank:
            same = same and tensor_checksum(kk + vv) == sums_by_rank[r]
        parts_k = [[x] for x in kk] if parts_k is None else [p + [x] for p, x in zip(parts_k, kk)]
        parts_v = [[x] for x in vv] if parts_v is None else [p + [x] for p, x in zip(parts_v, vv)]
    full_k = [torch.cat(p) for p in parts_k]
    full_v = [torch.cat(p) for p in parts_v]
    return full_k, full_v, same


def verify_sharded_agg(name, ctx, dev, rank, world, dist, gen, own, aggs, threads, budget, with_oracle=True):
    """One more sharded call of the config's query (+ count(), so that group sizes are compared too), its result slabs
    gathered on rank 0 and compared with the SINGLE-GPU call over the rebuilt frame, which verify_agg() compares with the
    OpenMP oracle on all rows.  Collective; returns the verdict on every rank."""
    import numpy as np
    import torch
    from datatable_amd.torch_bridge import devcol, ST2T
    t0 = time.perf_counter()
    skip = torch.tensor([0 if budget.ok(60) else 1], dtype=torch.int64)
    dist.broadcast(skip, src=0)
    if int(skip.item()):
        return {"skipped": "time budget (%.0f s) spent" % budget.seconds}
    keys, vals = own
    n = keys[0].numel()
    vaggs = list(aggs) + ([] if ("count0", None) in aggs else [("count0", None)])
    r = ctx.sharded_groupby_agg([devcol(k) for k in keys], [devcol(v) for v in vals], vaggs, nrows=n)
    ngl = r.ngroups
    slabs = []
    for i, k in enumerate(keys):
        t = torch.empty(ngl, dtype=k.dtype, device=dev)
        if ngl:
            r.key_into(i, t.data_ptr())
        slabs.append(t)
    for a in range(len(vaggs)):
        t = torch.empty(ngl, dtype=ST2T[r.agg_stype(a)], device=dev)
        if ngl:
            r.agg_into(a, t.data_ptr())
        slabs.append(t)
    torch.cuda.synchronize()
    r.free()
    got = [gather_to_rank0(dist, rank, world, t) for t in slabs]
    del slabs
    sums = [None] * world
    dist.all_gather_object(sums, tensor_checksum(keys + vals))
    verdict = None
    if rank == 0:
        full_k, full_v, same = rebuild_frame(gen, dev, rank, world, own, sums)
        keep = {}
        if with_oracle:
            par, _host = verify_agg(ctx, name, full_k, full_v, vaggs, threads, keep=keep)
            del _host
        else:
            par = None
            rr = ctx.groupby_agg([devcol(k) for k in full_k], [devcol(v) for v in full_v], vaggs, nrows=full_k[0].numel())
            keep["keys"] = [rr.key(i) for i in range(len(full_k))]; keep["aggs"] = [rr.agg(a) for a in range(len(vaggs))]
            rr.free()
        del full_k, full_v
        torch.cuda.empty_cache(); ctx.trim()
        nk = len(keys)
        keys_ok = all(_cmp_exact(got[i][0], keep["keys"][i]) for i in range(nk))
        verdict = {"against": "the single-GPU path (dthip_groupby_agg) over ALL rows, rebuilt on rank 0's GPU from the ranks' seeds",
                   "groups": int(len(keep["keys"][0])), "groups_per_rank": got[0][1], "inputs_rebuilt_identical": bool(same),
                   "keys_bit_exact": bool(keys_ok), "single_gpu_vs_oracle_all_rows": par}
        ok = bool(keys_ok and (par is None or par["ok"]))
        wa, wr = 0.0, 0.0
        for a, (op, c) in enumerate(vaggs):
            if not keys_ok:
                break
            g_, e_ = got[nk + a][0], keep["aggs"][a]
            if e_.dtype.kind == "f" and op in ("sum", "mean"):
                o_, ma, mr = sums_close(g_, e_)
                wa, wr = max(wa, ma), max(wr, mr)
            else:
                o_ = _cmp_exact(g_, e_)
            verdict["%s(%s)" % (op, "" if c is None else "v%d" % c)] = bool(o_)
            ok = ok and bool(o_)
        verdict.update(ok=ok, sum_max_abs_err=wa, sum_max_rel_err=wr, rtol=RTOL, atol=ATOL, seconds=time.perf_counter() - t0)
        if not same:
            # the frame rank 0 rebuilt is NOT the one the ranks computed on (the generator streams differ between these devices):
            # the comparison says nothing either way -- reported, never fatal
            verdict = {"skipped": "inputs rebuilt on rank 0 differ from the ranks' own tensors (device generator streams differ): "
                                  "sharded-vs-single-GPU comparison not possible on this node", "inputs_rebuilt_identical": False}
    return agree(dist, rank, verdict)


def verify_sharded_c5(ctx, dev, rank, world, dist, n_total, own, threads, budget):
    """config 5 sharded vs single GPU: every rank's slab of V[:, :, by(f.k)] -- group sizes, key and x columns and the
    composed RowIndex (first row of the source rank's shard + the filter's RowIndex that travelled with the row) -- gathered
    on rank 0 and compared bit for bit with the single-GPU filter -> rows-in-grouped-order over the rebuilt frame, which
    verify_c5() compares with the OpenMP oracle on all rows."""
    import numpy as np
    import torch
    from datatable_amd.torch_bridge import devcol
    t0 = time.perf_counter()
    skip = torch.tensor([0 if budget.ok(120) else 1], dtype=torch.int64)
    dist.broadcast(skip, src=0)
    if int(skip.item()):
        return {"skipped": "time budget (%.0f s) spent" % budget.seconds}
    (k,), (x,) = own
    n = k.numel()
    lo = shard_bounds(rank, world, n_total)[0]
    ri = torch.empty(n, dtype=torch.int32, device=dev)
    kb = torch.empty(n, dtype=torch.int64, device=dev)
    xb = torch.empty(n, dtype=torch.float64, device=dev)
    npass = ctx.filter_take_dev(devcol(x), ">", 0.0, [devcol(k), devcol(x)], n, ri.data_ptr(), [kb.data_ptr(), xb.data_ptr()])
    r = ctx.sharded_groupby_rows([devcol(kb[:npass])], [devcol(kb[:npass]), devcol(xb[:npass]), devcol(ri[:npass])],
                                 row_offset=lo, nrows=npass)
    nr, ngl = r.nrows, r.ngroups
    off = torch.empty(ngl + 1, dtype=torch.int32, device=dev)
    ck = torch.empty(nr, dtype=torch.int64, device=dev); cx = torch.empty(nr, dtype=torch.float64, device=dev)
    cri = torch.empty(nr, dtype=torch.int32, device=dev); cid = torch.empty(nr, dtype=torch.int64, device=dev)
    r.offsets_into(off.data_ptr())
    if nr:
        r.col_into(0, ck.data_ptr()); r.col_into(1, cx.data_ptr()); r.col_into(2, cri.data_ptr()); r.col_into(3, cid.data_ptr())
    torch.cuda.synchronize()
    r.free()
    del ri, kb, xb
    # global row id = (first row of the SOURCE rank's shard) + (position among that rank's passing rows): the source rank
    # is the shard the id falls into, and the row's place in the unfiltered frame is that shard's start + the filter's RowIndex
    los = torch.tensor([shard_bounds(q, world, n_total)[0] for q in range(world)], dtype=torch.int64, device=dev)
    src = torch.searchsorted(los, cid, right=True) - 1
    comp = (los[src] + cri.to(torch.int64)).to(torch.int32)
    sizes = (off[1:] - off[:-1]).contiguous()
    del src, cri, cid, off
    got = {nm: gather_to_rank0(dist, rank, world, t) for nm, t in (("sizes", sizes), ("k", ck), ("x", cx), ("ri", comp))}
    del sizes, ck, cx, comp
    sums = [None] * world
    dist.all_gather_object(sums, tensor_checksum([k, x]))
    torch.cuda.empty_cache(); ctx.trim()
    verdict = None
    if rank == 0:
        full_k, full_v, same = rebuild_frame(lambda q: gen_c5(dev, q, world, n_total), dev, rank, world, own, sums)
        keep = {}
        par, _host = verify_c5(ctx, full_k[0], full_v[0], threads, keep=keep)
        del _host, full_k, full_v
        torch.cuda.empty_cache(); ctx.trim()
        exp_sizes = np.diff(keep["offsets"]).astype(np.int32)
        verdict = {"against": "the single-GPU path (dthip_filter_take + dthip_groupby_rows) over ALL rows, rebuilt on rank 0's GPU "
                              "from the ranks' seeds",
                   "rows_passing": int(len(keep["ri"])), "groups": int(len(exp_sizes)), "rows_per_rank": got["k"][1],
                   "inputs_rebuilt_identical": bool(same), "single_gpu_vs_oracle_all_rows": par,
                   "group_sizes_bit_exact": _cmp_exact(got["sizes"][0], exp_sizes),
                   "composed_rowindex_bit_exact": _cmp_exact(got["ri"][0], keep["ri"]),
                   "key_column_bit_exact": _cmp_exact(got["k"][0], keep["k"]),
                   "x_column_bit_exact": _cmp_exact(got["x"][0], keep["x"])}
        verdict["ok"] = bool(par["ok"] and all(v for kk, v in verdict.items() if kk.endswith("bit_exact")))
        verdict["seconds"] = time.perf_counter() - t0
        if not same:
            verdict = {"skipped": "inputs rebuilt on rank 0 differ from the ranks' own tensors (device generator streams differ): "
                                  "sharded-vs-single-GPU comparison not possible on this node", "inputs_rebuilt_identical": False}
    return agree(dist, rank, verdict)


def sharded_config_legs(ctx, dev, rank, world, n_total, steps, dist, verify=True, threads=16, budget=None):
    """--gpus N: BASELINE configs 4 and 5 through the sharded entry points, row-block shards like C3's.  Per config: wall
    time (barrier, max over ranks), rows/s, bytes every rank sent to its peers in the all-to-all-v and what that is of the
    xGMI bound, and the wall-clock phases of one profiled call."""
    import torch
    from datatable_amd.torch_bridge import devcol
    lo, hi = rank * n_total // world, (rank + 1) * n_total // world
    n = hi - lo
    g = torch.Generator(device=dev)
    out = {}

    def barrier():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()

    def timed(run, alg_bytes):
        run().free()                                   # warm-up
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run().free()
        barrier()
        t = torch.tensor([(time.perf_counter() - t0) / steps], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = float(t.item())
        ctx.profile_reset(); ctx.profile(True)
        r = run(); torch.cuda.synchronize(); ctx.profile(False)
        ngl = r.ngroups
        r.free()
        st = ctx.comm_last_stats()
        prof = {nm: ctx.profile_get(nm) for nm in ctx.profile_names()}
        phases = {nm[6:]: round(ms, 4) for nm, (ms, _) in prof.items() if nm.startswith("phase_")}
        kern = {nm: round(ms, 4) for nm, (ms, _) in sorted(prof.items(), key=lambda kv: -kv[1][0]) if not nm.startswith("phase_")}
        tot = torch.tensor([st["bytes_to_peers"], ngl], dtype=torch.float64)
        mx = tot.clone()
        dist.all_reduce(tot); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        a2a = phases.get("alltoallv") or None
        return {"ms": sec * 1e3, "rows_per_s": n_total / sec, "groups": int(tot[1].item()),
                "alg_GBs": alg_bytes / sec / 1e9, "hbm_frac": alg_bytes / sec / 1e9 / (HBM_PEAK_GBS * world),
                "bytes_to_peers_rank0": st["bytes_to_peers"], "bytes_to_peers_max": int(mx[0].item()), "bytes_to_peers_all": int(tot[0].item()),
                "xgmi_frac_of_step": mx[0].item() / sec / 1e9 / XGMI_GBS_PER_GPU,
                "xgmi_frac_of_exchange": (st["bytes_to_peers"] / (a2a * 1e-3) / 1e9 / XGMI_GBS_PER_GPU) if a2a else None,
                "xgmi_peak_GBs_per_gpu": XGMI_GBS_PER_GPU, "allgathers": st["allgathers"],
                "phases_ms_rank0": phases, "kernel_ms_rank0": dict(list(kern.items())[:8])}

    # C4: DT[:, [count(), sum(f.v)], by(f.a, f.b)]
    (a, b), (v,) = gen_c4(dev, rank, world, n_total)
    out["C4"] = timed(lambda: ctx.sharded_groupby_agg([devcol(a), devcol(b)], [devcol(v)], [("count0", None), ("sum", 0)], nrows=n),
                      n_total * 16 + 10_004_569 * 24)
    out["C4"]["query"] = "DT[:, [count(), sum(f.v)], by(f.a, f.b)], %d rows over %d ranks" % (n_total, world)
    if verify:
        out["C4"]["parity"] = verify_sharded_agg("C4", ctx, dev, rank, world, dist, lambda q: gen_c4(dev, q, world, n_total),
                                                 ([a, b], [v]), [("count0", None), ("sum", 0)], threads, budget)
    del a, b, v
    torch.cuda.empty_cache(); ctx.trim()
    # C5: V = DT[f.x > 0, :] on the shard (local: a filter needs no exchange), then V[:, :, by(f.k)] sharded
    (k,), (x,) = gen_c5(dev, rank, world, n_total)
    ri = torch.empty(n, dtype=torch.int32, device=dev)
    kb = torch.empty(n, dtype=torch.int64, device=dev)
    xb = torch.empty(n, dtype=torch.float64, device=dev)

    def c5():
        npass = ctx.filter_take_dev(devcol(x), ">", 0.0, [devcol(k), devcol(x)], n, ri.data_ptr(), [kb.data_ptr(), xb.data_ptr()])
        return ctx.sharded_groupby_rows([devcol(kb[:npass])], [devcol(kb[:npass]), devcol(xb[:npass]), devcol(ri[:npass])],
                                        row_offset=lo, nrows=npass)
    out["C5"] = timed(c5, int(n_total * 30.4))
    out["C5"]["query"] = "V = DT[f.x > 0, :] per shard; V[:, :, by(f.k)] sharded (rows + the filter's RowIndex travel), %d rows over %d ranks" % (n_total, world)
    del ri, kb, xb
    torch.cuda.empty_cache(); ctx.trim()
    if verify:
        out["C5"]["parity"] = verify_sharded_c5(ctx, dev, rank, world, dist, n_total, ([k], [x]), threads, budget)
    del k, x
    torch.cuda.empty_cache(); ctx.trim()
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the environment
    torch.distributed.run would have set), pass rank 0's stdout -- the JSON line -- through, and take the whole group
    down if one rank dies so nobody waits in a collective for a peer that is gone."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DTHIP_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    q.terminate()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--groups", type=int, default=10_000_000)
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="rows of the workload the reference is timed on")
    ap.add_argument("--ref-threads", default="16,1",
                    help="dt.options.nthreads values to time (0 = all host cores; on the 256-thread box the reference is fastest at 16 and 40x slower at 256: profiles/r02a_bench_refsweep.json)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-parity", action="store_true", help="skip the all-rows GPU vs OpenMP-port comparison")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the OpenMP port (0: min(host cores, 64))")
    ap.add_argument("--configs", default="C1,C2,C4,C5,C3_hard,C3_sgrp,C3_adv",
                    help="other BASELINE configs to time after C3 ('' = none); C3_sgrp = C3 as dthip_groupby + dthip_reduce (the literal "
                         "S-grp -> S-red seam); C3_adv = adversarial variants of C3 (outlier key, planted NA, sorted keys, hot key)")
    ap.add_argument("--config-steps", type=int, default=3)
    ap.add_argument("--config-scale", type=float, default=1.0)
    ap.add_argument("--no-verify-configs", action="store_true", help="time the other configs only (no full-size oracle check, no CPU sample)")
    ap.add_argument("--time-budget", type=float, default=600.0,
                    help="seconds after which the remaining OPTIONAL legs (full-size checks of the other configs, CPU samples, "
                         "the all-rows reference run, the 1-rank sharded leg) are skipped and reported as skipped")
    ap.add_argument("--no-ref-full", action="store_true", help="skip the reference's run over ALL rows of C3 (~1 min of CPU)")
    ap.add_argument("--no-dist-1rank", action="store_true", help="skip the 1-rank run of the sharded (RCCL) code path")
    ap.add_argument("--host-rows", type=int, default=100_000_000, help="rows of the host-pointer (PCIe-inclusive) leg, 0 = skip")
    ap.add_argument("--no-shim-resident", action="store_true", help="skip the reference-side binding leg with device-resident columns")
    ap.add_argument("--shim-scale", type=float, default=1.0, help="row-count factor of the shim_resident leg")
    ap.add_argument("--no-sharded-configs", action="store_true", help="--gpus N: skip the sharded legs of configs 4 and 5")
    ap.add_argument("--no-check", action="store_true", help="skip the result sanity check (kernel timing experiments)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--agg-path", type=int, default=0, help="dthip option agg_path: 0 auto, 1 sort, 2 bucketed")
    ap.add_argument("--bucket-variant", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (RCCL collectives, merge) even with one rank")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher contract only (CPU): ranks meet over gloo, the communicator id travels, one JSON line; no GPU work")
    ap.add_argument("--agg-offsets", type=int, default=0,
                    help="1: the result also carries group sizes (not part of DT[:, sum(f.v), by(f.k)]'s result Frame)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    budget = Budget(args.time_budget)

    # stdout carries exactly ONE line, the JSON: gloo and RCCL announce themselves on the C-level stdout (RCCL's banner
    # even after the line, when its buffer is flushed at exit), so fd 1 is pointed at stderr for the whole run and the
    # line goes to a private copy of the original stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        # the control plane of a sharded run and nothing else: rendezvous, id hand-off, barrier, max over ranks
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        box = [os.urandom(128) if rank == 0 else None]          # stands for dthip_comm_unique_id() (needs librccl + a GPU)
        dist.broadcast_object_list(box, src=0)
        seen = [None] * world
        dist.all_gather_object(seen, (rank, len(box[0])))
        dist.barrier()
        t0 = time.perf_counter()
        dist.barrier()
        tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "dry run: launcher contract only", "value": 0.0, "unit": "rows/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(tmax.item()) * 1e3,
                              "dry_run": True, "ranks_seen": sorted(r for r, _ in seen), "id_bytes": min(n for _, n in seen)}),
                  file=json_out, flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    from datatable_amd.torch_bridge import context_for_current_stream, devcol
    from datatable_amd.engine import comm_unique_id
    if os.environ.get("DTHIP_BENCH_ONE_GPU"):
        local_rank = 0      # tests only: every rank on GPU 0 (with DTHIP_RCCL_LIB = the shared-memory stand-in; RCCL refuses that)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_dist
    if sharded:
        # torch.distributed carries CONTROL only (the communicator id, the barrier and the max over ranks of the wall
        # time) over gloo on the CPU; the data path -- RCCL all-gathers and the all-to-all-v -- is inside libdthip.so
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("gloo", rank=0, world_size=1)
        else:
            dist.init_process_group("gloo")
        box = [comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        dist.barrier()
    ctx = context_for_current_stream(local_rank)
    if sharded:
        ctx.comm_init(rank, world, box[0])
    ctx.set_option("agg_path", args.agg_path)
    ctx.set_option("bucket_variant", args.bucket_variant)
    if not sharded:
        ctx.set_option("agg_offsets", args.agg_offsets)

    n_total = args.rows
    lo, hi = rank * n_total // world, (rank + 1) * n_total // world
    n_local = hi - lo
    (keys,), (vals,) = gen_c3(dev, rank, world, n_total, args.groups)
    torch.cuda.synchronize()
    aggs = [("sum", 0)]
    kcol, vcol = devcol(keys), devcol(vals)

    def step():
        if not sharded:
            return ctx.groupby_agg([kcol], [vcol], aggs, nrows=n_local)
        return ctx.sharded_groupby_agg([kcol], [vcol], aggs, nrows=n_local)

    def release(r):
        r.free()

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        release(step())
    ctx.profile_reset()
    if not args.no_profile:
        ctx.profile(True)
    barrier()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        if last is not None:
            release(last)
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    tmax = torch.tensor([dt], dtype=torch.float64)
    if sharded:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    per_kernel = {}
    for nm in ctx.profile_names():
        ms, cnt = ctx.profile_get(nm)
        per_kernel[nm] = {"launches": cnt, "avg_ms": ms / max(cnt, 1), "total_ms": ms}
    # the dominant kernel = largest share of the timed region
    dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms"]) if per_kernel else None
    rp_ms, rp_n = (per_kernel[dom]["total_ms"], per_kernel[dom]["launches"]) if dom else (0.0, 0)

    # sanity of the last result (outside the timed region): keys strictly ascending, totals agree,
    # and (one extra untimed call with count()) every row counted exactly once
    parity = {}
    if not sharded:
        ng = last.ngroups
        sums = torch.empty(ng, dtype=torch.float64, device=dev)
        gkeys = torch.empty(ng, dtype=torch.int64, device=dev)
        last.agg_into(0, sums.data_ptr()); last.key_into(0, gkeys.data_ptr())
        release(last)
        torch.cuda.synchronize()
        if not args.no_check:
            assert bool((gkeys[1:] > gkeys[:-1]).all())
            total, ref_total = float(sums.sum().item()), float(vals.sum().item())
            assert abs(total - ref_total) <= 1e-9 * float(vals.abs().sum().item()), (total, ref_total)
            rc = ctx.groupby_agg([kcol], [vcol], [("count0", None), ("sum", 0)], nrows=n_local)
            cnt = torch.empty(rc.ngroups, dtype=torch.int64, device=dev)
            s2 = torch.empty(rc.ngroups, dtype=torch.float64, device=dev)
            rc.agg_into(0, cnt.data_ptr()); rc.agg_into(1, s2.data_ptr())
            rc.free()
            torch.cuda.synchronize()
            assert rc.ngroups == ng and int(cnt.sum().item()) == n_local
            assert bool(torch.allclose(s2, sums, rtol=1e-9, atol=1e-9))
            parity["properties"] = {"rows": n_local, "keys_strictly_ascending": True, "group_sizes_sum_to_rows": True,
                                    "sum_of_group_sums_equals_sum_of_values": True}
            last_gpu = (gkeys.cpu().numpy(), sums.cpu().numpy(), cnt.cpu().numpy())
            del cnt, s2
        del sums, gkeys
    else:
        # every rank owns one key range: ranges ascend with the rank, keys ascend inside, totals agree
        ngl = last.ngroups
        sums = torch.empty(ngl, dtype=torch.float64, device=dev)
        gkeys = torch.empty(ngl, dtype=torch.int64, device=dev)
        if ngl:
            last.agg_into(0, sums.data_ptr()); last.key_into(0, gkeys.data_ptr())
        release(last)
        torch.cuda.synchronize()
        assert ngl < 2 or bool((gkeys[1:] > gkeys[:-1]).all())
        edge = torch.tensor([int(gkeys[0]) if ngl else 2**62, int(gkeys[-1]) if ngl else -2**62], dtype=torch.int64)
        edges = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(edges, edge)
        prev = -2**63
        for e in edges:
            if int(e[0]) <= int(e[1]):
                assert int(e[0]) > prev, "key ranges of the ranks overlap"
                prev = int(e[1])
        ng_t = torch.tensor([ngl], dtype=torch.int64)
        tot = torch.stack([sums.sum(), vals.sum(), vals.abs().sum()]).cpu()
        dist.all_reduce(ng_t); dist.all_reduce(tot)
        ng = int(ng_t.item())
        assert abs(float(tot[0]) - float(tot[1])) <= 1e-9 * float(tot[2]), tot.tolist()
        parity["properties"] = {"rows": n_total, "keys_strictly_ascending_within_and_across_ranks": True,
                                "sum_of_group_sums_equals_sum_of_values": True, "groups": ng}
        del sums, gkeys

    sharded_extra = None
    if sharded:
        # one profiled call of the timed query: wall-clock phases + what crossed the fabric (rank 0's view, max over ranks)
        ctx.profile_reset(); ctx.profile(True)
        release(step()); torch.cuda.synchronize(); ctx.profile(False)
        st = ctx.comm_last_stats()
        prof = {nm: ctx.profile_get(nm) for nm in ctx.profile_names()}
        mxb = torch.tensor([float(st["bytes_to_peers"])], dtype=torch.float64)
        dist.all_reduce(mxb, op=dist.ReduceOp.MAX)
        phases = {nm[6:]: round(ms, 4) for nm, (ms, _) in prof.items() if nm.startswith("phase_")}
        a2a = phases.get("alltoallv") or None
        sharded_extra = {"phases_ms_rank0": phases, "bytes_to_peers_rank0": st["bytes_to_peers"], "bytes_to_peers_max": int(mxb.item()),
                         "allgathers": st["allgathers"], "xgmi_peak_GBs_per_gpu": XGMI_GBS_PER_GPU,
                         "xgmi_frac_of_step": mxb.item() / (dt / args.steps) / 1e9 / XGMI_GBS_PER_GPU,
                         "xgmi_frac_of_exchange": (st["bytes_to_peers"] / (a2a * 1e-3) / 1e9 / XGMI_GBS_PER_GPU) if a2a else None}
    ranks_seen = None
    if sharded:
        # who took part, as the LIBRARY's communicator sees it (dthip_comm_rank / dthip_comm_world) and on which device
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            devid = "%s %s" % (pr.name, getattr(pr, "pci_bus_id", "?"))
        except Exception:
            devid = "?"
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "comm_rank": int(ctx.comm_rank), "comm_world": int(ctx.comm_world),
                                      "device": local_rank, "gpu": devid, "pid": os.getpid()})
        ranks_seen = seen
        assert sorted(x["comm_rank"] for x in seen) == list(range(world)) and all(x["comm_world"] == world for x in seen), seen
    line = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt
        alg_bytes_launch = ALG_BYTES_PER_ROW * n_local          # algorithmic bytes of the rows one launch processes
        roof = None
        if rp_n:
            # single GPU: one launch of the dominant kernel per step.  Sharded: the merge of the exchanged
            # partials launches it a second time on a few rows; its time is charged, its bytes are not.
            per_step = max(1, round(rp_n / args.steps))
            avg_s = rp_ms / rp_n * per_step * 1e-3
            ach = alg_bytes_launch / avg_s / 1e9
            traffic, traffic_note = None, None
            if not sharded:
                doc, traffic_note = pmc_records()
                traffic = (doc.get(dom) or {}).get("hbm_bytes_per_launch_at_rows", {}).get(str(n_local))
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "launches": rp_n, "launches_per_step": per_step,
                    "avg_launch_ms": rp_ms / rp_n * per_step,
                    "alg_bytes_per_launch": alg_bytes_launch,
                    "note": "achieved = 16 B/row (SURVEY 8d, C3) x rows of one launch / HIP-event time of that launch; "
                            "traffic = HBM bytes of one launch from rocprofv3 FETCH_SIZE x2 + WRITE_SIZE (profiles/)",
                    "whole_step_alg_GBs": (ALG_BYTES_PER_ROW * n_total + 16 * ng) / (dt / args.steps) / 1e9,
                    "whole_step_frac": (ALG_BYTES_PER_ROW * n_total + 16 * ng) / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world)}
        line = {
            "metric": "groupby-sum rows/sec (1e9 rows, int64 key, 1e7 groups, float64 value)",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int64 keys / f64 sums", "data": "synthetic",
            "config": {"workload": "C3: DT[:, sum(f.v), by(f.k)], %d rows, int64 key uniform in [0,%d), float64 N(0,1)"
                                   % (n_total, args.groups),
                       "rows": n_total, "groups_found": ng, "rows_per_gpu": n_local,
                       "parallelism": ("row-sharded x%d: local combiner, histogram splitters, range-partitioned all-to-all-v of partials "
                                       "(RCCL inside libdthip.so), merge on the owner" % world) if sharded else "single GPU"},
            "roofline": roof,
            "kernels": per_kernel,
            "library_build_id": __import__("datatable_amd._lib", fromlist=["load"]).load().dthip_build_id().decode(),
            "cpu_baseline": None,
        }
        if sharded and not args.no_cpu_baseline:
            # N > 1: the reference's CPU path, timed on rank 0's host cores on the first rows of rank 0's shard (= the first
            # rows of the frame) while the other ranks sleep in a gloo barrier; GPU-vs-reference parity on that sample too
            try:
                base, par = reference_leg(ctx, keys, vals, min(args.cpu_sample, n_local), [int(t) for t in args.ref_threads.split(",") if t != ""])
            except Exception as e:
                base, par = None, None
                parity["vs_reference"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            if base is not None:
                line["cpu_baseline"] = base
                parity["vs_reference"] = par
                assert par["keys_bit_exact"] and par["sums_within_tol"], par
        if world == 1 and not sharded and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 64)
            try:
                base, par = reference_leg(ctx, keys, vals, args.cpu_sample, [int(t) for t in args.ref_threads.split(",") if t != ""])
            except Exception as e:       # the reference build could not be imported / run here: report it, keep the port
                base, par = None, None
                parity["vs_reference"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            if base is not None:
                line["cpu_baseline"] = base
                parity["vs_reference"] = par
                assert par["keys_bit_exact"] and par["sums_within_tol"], par
            if not args.no_full_parity and not args.no_check:
                port, par = port_leg(ctx, keys, vals, last_gpu, threads)
                parity["vs_port_all_rows"] = par
                assert par["keys_bit_exact"] and par["group_sizes_bit_exact"] and par["sums_within_tol"], par
                if line["cpu_baseline"] is None:
                    line["cpu_baseline"] = port          # no oracle/_ref on this machine: the port is all there is
                else:
                    line["cpu_baseline"]["port"] = port
        line["parity"] = parity or None
        if ranks_seen is not None:
            line["ranks_seen"] = sorted(x["comm_rank"] for x in ranks_seen)
            line["ranks"] = ranks_seen
        if sharded_extra:
            line["exchange"] = sharded_extra
            if line.get("roofline"):
                line["roofline"]["xgmi"] = {k: sharded_extra[k] for k in ("xgmi_peak_GBs_per_gpu", "xgmi_frac_of_step", "xgmi_frac_of_exchange", "bytes_to_peers_max")}
    threads = args.cpu_threads or min(os.cpu_count() or 1, 64)
    if rank == 0 and world == 1 and not sharded and not args.no_dist_1rank and not args.no_check:
        # the fixed cost of the sharded call, driver-visible (one rank: local work + collectives + merge)
        if budget.ok(20):
            try:
                line["dist_1rank"] = dist_one_rank_leg(ctx, keys, vals, args.steps, ms_per_step)
            except Exception as e:
                line["dist_1rank"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        else:
            line["dist_1rank"] = {"skipped": "time budget"}
    ref_full_inputs = None
    if rank == 0 and world == 1 and not sharded and not args.no_cpu_baseline and not args.no_ref_full and line.get("cpu_baseline") \
            and line["cpu_baseline"].get("kind") == "reference":
        ref_full_inputs = (to_host(keys), to_host(vals))
    raw_c3_ms = dt / args.steps * 1e3
    verify_sharded = sharded and not (args.no_check or args.no_verify_configs)
    if sharded:
        dist.barrier()                  # (the other ranks waited here, in gloo, while rank 0 timed the reference)
    if verify_sharded:
        # the timed query's sharded result vs the single-GPU path over ALL rows (itself vs the OpenMP oracle): see
        # verify_sharded_agg; a mismatch aborts every rank
        par3 = verify_sharded_agg("C3", ctx, dev, rank, world, dist, lambda q: gen_c3(dev, q, world, n_total, args.groups),
                                  ([keys], [vals]), aggs, threads, budget, with_oracle=not args.no_full_parity)
        if rank == 0:
            line["parity"] = line.get("parity") or {}
            line["parity"]["sharded_vs_single_gpu"] = par3
            line["parity"]["configs"] = {"C3": par3.get("ok", par3.get("skipped"))}
    del keys, vals, kcol, vcol
    torch.cuda.empty_cache(); ctx.trim()
    if sharded and world > 1 and not args.no_sharded_configs:
        cfgs = sharded_config_legs(ctx, dev, rank, world, int(n_total * args.config_scale), max(1, min(args.steps, 5)), dist,
                                   verify=verify_sharded, threads=threads, budget=budget)
        if rank == 0:
            line["configs"] = cfgs
            if verify_sharded:
                for c in ("C4", "C5"):
                    pc = cfgs[c].get("parity") or {}
                    line["parity"]["configs"][c] = pc.get("ok", pc.get("skipped"))
    if rank == 0 and world == 1 and not sharded:
        which = [c for c in args.configs.split(",") if c]
        adv = "C3_adv" in which
        which = [c for c in which if c != "C3_adv"]
        if which:
            best_threads = int(line["cpu_baseline"]["cores"]) if line.get("cpu_baseline") and line["cpu_baseline"].get("kind") == "reference" else 16
            cfg = run_configs(ctx, dev, which, args.config_steps, args.config_scale,
                              verify=not (args.no_verify_configs or args.no_check),
                              cpu_sample=0 if args.no_cpu_baseline else args.cpu_sample,
                              ref_threads=best_threads, port_threads=threads, budget=budget)
            line["configs"] = cfg
            line["parity"] = line.get("parity") or {}
            line["parity"]["configs"] = {c: (v["parity"].get("ok", v["parity"].get("skipped")) if "parity" in v else None) for c, v in cfg.items()}
        if adv:
            try:
                acfg = adversarial_legs(ctx, dev, int(n_total * args.config_scale), args.groups, args.config_steps,
                                        not (args.no_verify_configs or args.no_check), threads, budget, raw_c3_ms * args.config_scale)
            except AssertionError:
                raise
            except Exception as e:
                acfg = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            line.setdefault("configs", {}).update(acfg)
            line["parity"] = line.get("parity") or {}
            line["parity"].setdefault("configs", {}).update(
                {c: (v["parity"].get("ok", v["parity"].get("skipped")) if isinstance(v, dict) and "parity" in v else None) for c, v in acfg.items() if c != "error"})
        if ref_full_inputs is not None:
            # the reference's CPU path on the WHOLE workload, once, at the thread count the sample leg found best
            # (sort.cc:1243-1282: past ~1e7 groups it pays a thread-team wake-up per tiny radix bucket -- the "cliff")
            if budget.ok(150):
                from oracle import ref
                nt = int(line["cpu_baseline"]["cores"])
                try:
                    _, sec = ref.groupby_agg({"k": ref_full_inputs[0], "v": ref_full_inputs[1]}, ["k"], [("sum", "v")], nthreads=nt, reps=1)
                    line["cpu_baseline"]["all_rows"] = {"rows": n_total, "seconds": sec, "rows_per_s": n_total / sec, "threads": nt,
                                                        "gpu_over_cpu": value / (n_total / sec)}
                except Exception as e:
                    line["cpu_baseline"]["all_rows"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            else:
                line["cpu_baseline"]["all_rows"] = {"skipped": "time budget"}
            ref_full_inputs = None
        if args.host_rows:
            line["host_mode"] = host_mode_leg(ctx, args.host_rows)
        if not args.no_shim_resident:
            if budget.ok(90):
                try:
                    c5 = (line.get("configs") or {}).get("C5") or {}
                    line["shim_resident"] = shim_resident_leg(max(3, min(args.steps, 10)), int(n_total * args.shim_scale), args.groups,
                                                              int(1e9 * args.config_scale * args.shim_scale), raw_c3_ms, c5.get("ms"))
                except Exception as e:
                    line["shim_resident"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            else:
                line["shim_resident"] = {"skipped": "time budget"}
        line["seconds_total"] = time.perf_counter() - budget.t0
    if rank == 0:
        print(json.dumps(line), file=json_out, flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
