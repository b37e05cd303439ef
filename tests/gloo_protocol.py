"""TEST INFRASTRUCTURE: the multi-GPU protocol of datatable_amd/csrc/comm.hip re-enacted over gloo on CPU tensors, one
process per rank, with the CPU oracle standing in for the per-rank HIP kernels.  What it shares with the product is the
host-side planning code itself -- csrc/split_plan.hpp compiled by g++ (tests/cpp/split_plan_harness.cpp): quantile
splitters (sample_bounds), histogram splitters (reduce_key_ranges / split_bounds), the {status, signature, n} header of
every all-gathered blob and first_failure -- and the phase order: all-gather A (quantile samples: exact on the aggregate
path, a stratified random sample on the rows path), B (send counts + status), [C status when a share exceeds the receive
bound], all-to-all-v, merge.  Used by tests/test_dist_gloo.py (world_size 2)."""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = 1024
BINS = 4096
_H = None


def harness():
    global _H
    if _H is None:
        so = os.path.join(tempfile.mkdtemp(prefix="splitplan"), "libsplitplan.so")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC",
                               os.path.join(ROOT, "tests", "cpp", "split_plan_harness.cpp"), "-o", so])
        _H = C.CDLL(so)
    return _H


def image(k, na_last=False):
    """order-preserving uint64 image of an integer key column (comm.hip key_image); NA -> 0 / ~0"""
    k64 = k.astype(np.int64)
    img = k64.view(np.uint64) ^ np.uint64(1 << 63)
    na = k == np.iinfo(k.dtype).min
    img[na] = np.uint64(2**64 - 1) if na_last else np.uint64(0)
    return img


def allgather_blob(rc, sig, n, payload):
    """every rank's {rc, sig, n} + payload bytes -> list of (rc, sig, n, payload) in rank order"""
    blob = struct.pack("<iIq", rc, sig, n) + payload
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    res = []
    for o in outs:
        b = o.numpy().tobytes()
        res.append(struct.unpack("<iIq", b[:16]) + (b[16:],))
    return res


def agree(blobs):
    """first_failure of split_plan.hpp on the gathered headers: (rc, failing rank, signatures equal)"""
    world = len(blobs)
    rank, ok = C.c_int(0), C.c_int(0)
    rc = harness().sp_first_failure((C.c_int * world)(*[b[0] for b in blobs]), (C.c_uint * world)(*[b[1] for b in blobs]),
                                    world, C.byref(rank), C.byref(ok))
    return rc, rank.value, bool(ok.value)


class Disagreement(Exception):
    pass


def _check(blobs):
    rc, rank, ok = agree(blobs)
    if not ok:
        raise Disagreement("ranks were called with different queries")
    if rc:
        raise Disagreement("rank %d failed with code %d" % (rank, rc))


def _a2av(t, send_counts, recv_counts):
    out = t.new_empty((int(sum(recv_counts)),) + tuple(t.shape[1:]))
    dist.all_to_all_single(out, t.contiguous(), list(map(int, recv_counts)), list(map(int, send_counts)))
    return out


def _exchange_counts(send_cnt, sig, rc=0):
    blobs = allgather_blob(rc, sig, int(sum(send_cnt)), np.asarray(send_cnt, np.int64).tobytes())
    _check(blobs)
    me = dist.get_rank()
    return [int(np.frombuffer(b[3], np.int64)[me]) for b in blobs]


def partial_plan(aggs):
    """comm.hip plan_partials: mean -> (weighted sum, count); duplicates shared"""
    partial, recipe = [], []

    def need(op, col):
        if (op, col) not in partial:
            partial.append((op, col))
        return partial.index((op, col))
    for op, col in aggs:
        if op == "mean":
            recipe.append((1, need("mean", col), need("count", col)))
        elif op == "count0":
            recipe.append((0, need("count0", None), -1))
        else:
            recipe.append((0, need(op, col), -1))
    return partial, recipe


def sharded_groupby_agg(local_agg, keys, values, aggs, sig=1, fail=False, na_last=False):
    """keys / values: this rank's numpy columns.  local_agg(keys, values, aggs) -> (group key columns, agg columns) in key
    order (the oracle).  Returns (group keys, aggregates) of this rank's key range."""
    world = dist.get_world_size()
    partial, recipe = partial_plan(aggs)
    rc = -3 if fail else 0
    gk, cols, smp, ng = [], [], np.zeros(Q, np.uint64), 0
    if rc == 0:
        gk, cols = local_agg(keys, values, partial)
        ng = len(gk[0])
        for i, (op, col) in enumerate(partial):          # mean partials travel as weighted sums
            if op == "mean":
                cnt = cols[partial.index(("count", col))]
                cols[i] = np.where(cnt > 0, cols[i].astype(np.float64) * cnt, 0.0)
        img = image(gk[0], na_last)
        if ng:
            smp = img[(np.arange(Q, dtype=np.uint64) * np.uint64(ng)) // np.uint64(Q)]
    blobs = allgather_blob(rc, sig, ng, smp.tobytes())                                  # all-gather A
    _check(blobs)
    allsmp = np.concatenate([np.frombuffer(b[3], np.uint64) for b in blobs])
    counts = np.array([b[2] for b in blobs], np.int64)
    bounds = np.zeros(max(world - 1, 1), np.uint64)
    harness().sp_bounds_from_samples(allsmp.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), world,
                                     bounds.ctypes.data_as(C.c_void_p))
    cuts = [0] + [int(np.searchsorted(img, b, side="left")) for b in bounds[:world - 1]] + [ng]    # lower_bound_kernel
    send_cnt = np.diff(cuts)
    # round 4: the receive buffers are sized BEFORE the counts are known -- the splitters' guaranteed bound, the same number
    # on every rank -- so their allocation status travels with the counts and the status-only all-gather C is gone;
    # only a share above the bound (every rank sees the whole count matrix) brings it back
    total = int(counts.sum())
    recv_bound = total // world + total // Q + 2 * world + 16
    recv_cnt = _exchange_counts(send_cnt, sig)                                          # all-gather B (counts + status)
    assert int(sum(recv_cnt)) <= recv_bound, "a rank share exceeds the bound of the splitters"
    rk = [_a2av(torch.from_numpy(np.ascontiguousarray(k)), send_cnt, recv_cnt).numpy() for k in gk]
    rp = [_a2av(torch.from_numpy(np.ascontiguousarray(c)), send_cnt, recv_cnt).numpy() for c in cols]
    # merge on the owner: sums of sums / counts, min of mins, max of maxs
    mops = [("sum" if op in ("sum", "mean", "count", "count0") else op, i) for i, (op, _) in enumerate(partial)]
    mk, mc = local_agg(rk, rp, mops, nona=[op in ("sum", "mean", "count", "count0") for op, _ in partial])
    out = []
    for (kind, a, b), (op, col) in zip(recipe, aggs):
        if kind == 0:
            out.append(mc[a])
        else:
            with np.errstate(invalid="ignore", divide="ignore"):
                out.append(np.where(mc[b] > 0, mc[a] / np.maximum(mc[b], 1), np.nan))
    return mk, out


ROW_Q = 4096


def sharded_groupby_rows(local_rows, keys, cols, row_offset, sig=2, na_last=False, stats=None):
    """rows in grouped order (round 5: TWO all-gathers): A = every rank's stratified sample of ROW_Q key images, sorted
    (positions: split_plan.hpp row_sample_pos); splitters from the weighted union (sample_bounds_q); B = send counts +
    status, receive buffers sized by rows_recv_bound beforehand; a status-only round C only when a share exceeds that bound;
    all-to-all-v; one stable local grouping.  local_rows(keys, cols) -> (offsets, cols in grouped order)."""
    world = dist.get_world_size()
    H = harness()
    H.sp_row_sample_pos.restype = C.c_ulonglong
    H.sp_rows_recv_bound.restype = C.c_longlong
    n = len(keys[0])
    img = image(keys[0], na_last)
    smp = np.zeros(ROW_Q, np.uint64)
    if n:
        pos = np.array([H.sp_row_sample_pos(C.c_uint(i), C.c_ulonglong(n)) for i in range(ROW_Q)], np.int64)
        smp = np.sort(img[pos])                                                            # row_sample_kernel + std::sort
    blobs = allgather_blob(0, sig, n, smp.tobytes())                                      # A
    _check(blobs)
    allsmp = np.concatenate([np.frombuffer(b[3], np.uint64) for b in blobs])
    ns = (C.c_longlong * world)(*[b[2] for b in blobs])
    bounds = np.zeros(max(world - 1, 1), np.uint64)
    assert H.sp_bounds_from_row_samples(allsmp.ctypes.data_as(C.c_void_p), ns, world, bounds.ctypes.data_as(C.c_void_p)) == ROW_Q
    dest = np.searchsorted(bounds[:world - 1], img, side="right")                         # image_dest_kernel
    order = np.argsort(dest, kind="stable")                                               # stable partition by destination
    send_cnt = np.bincount(dest, minlength=world)
    bound = H.sp_rows_recv_bound(C.c_longlong(sum(b[2] for b in blobs)), world)
    blobs = allgather_blob(0, sig, n, send_cnt.astype(np.int64).tobytes())                # B: counts + status
    _check(blobs)
    mat = np.stack([np.frombuffer(b[3], np.int64) for b in blobs])                       # [sender][receiver]
    recv_cnt = mat[:, dist.get_rank()].copy()
    rounds = 2
    if (mat.sum(axis=0) > bound).any():                                                   # every rank sees the same matrix
        _check(allgather_blob(0, sig, 0, b""))                                            # C: exact buffers allocated
        rounds = 3
    if stats is not None:
        stats["allgathers"] = rounds
    rowid = np.arange(row_offset, row_offset + n, dtype=np.int64)
    rk = [_a2av(torch.from_numpy(np.ascontiguousarray(k[order])), send_cnt, recv_cnt).numpy() for k in keys]
    rc = [_a2av(torch.from_numpy(np.ascontiguousarray(c[order])), send_cnt, recv_cnt).numpy() for c in list(cols) + [rowid]]
    return local_rows(rk, rc, na_last)
