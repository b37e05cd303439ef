"""LEGACY Python wrapper of the multi-GPU exchange (round 1).  The product path is INSIDE libdthip.so since round 2
(datatable_amd/csrc/comm.hip: dthip_comm_* / dthip_sharded_groupby_*, RCCL called directly, histogram splitters; bench.py
--gpus N and engine.Context.sharded_groupby_agg use that).  This module stays for its CPU-testable exchange logic
(tests/test_dist_gloo.py, gloo, world_size 2) and as documentation of the algorithm in 100 lines of Python.

Multi-GPU DT[:, aggs, by(keys)]: one process per GPU, torch.distributed (RCCL over xGMI).

The reference is single-process; this exchange step is new.  Design
(SURVEY §8e): rows are sharded by row block; every rank first runs the fused local
groupby-aggregate (a combiner, so at most one partial per local group crosses the
fabric), then partials are range-partitioned on the FIRST key column and exchanged
with ONE all-to-all-v per column; the owner merges the <= world_size partials of
each group with the same HIP groupby kernel (sum of sums, sum of counts, min of mins,
max of maxes; mean = sum(mean_i*count_i)/sum(count_i)).

Range (not hash) partitioning keeps the global group order equal to the
concatenation of the ranks' outputs in rank order -- the order the reference
returns (ascending keys, NA group first: NA is the smallest integer sentinel and
lands on rank 0).  xGMI is point-to-point, so an all-to-all uses all 7 links of a
GPU concurrently; the payload is partials (<= ngroups x 16..40 B per GPU), not rows.

The local compute is injected as a backend so the exchange logic is testable on CPU
(gloo, world_size 2) with a CPU reference backend supplied by the tests; the product backend is
`HipBackend` (libdthip through torch_bridge).  Documented deviations of the merge from
a single-GPU run: a partial int64 sum equal to INT64_MIN, or a NaN partial float sum
(inf - inf inside one shard), is treated as NA by the merge.
"""
import torch
import torch.distributed as dist

INT_KEY_DTYPES = (torch.int8, torch.int16, torch.int32, torch.int64)


class HipBackend:
    """local groupby_agg on this rank's GPU through the C ABI"""

    def __init__(self, ctx):
        self.ctx = ctx

    def groupby_agg(self, keys, values, aggs):
        from .torch_bridge import groupby_agg_tensors
        _, gk, out = groupby_agg_tensors(self.ctx, keys, values, aggs, want_offsets=False)
        return gk, out

    def groupby_rows(self, keys, cols):
        """-> (offsets int32[ng+1], [cols in grouped order]); stable, NA group first"""
        from .torch_bridge import groupby_rows_tensors
        off, _, out = groupby_rows_tensors(self.ctx, keys, cols)
        return off, out

    def range_bucket(self, key, bounds):
        from .torch_bridge import range_bucket_tensor
        return range_bucket_tensor(self.ctx, key, bounds)


def _partials_for(aggs):
    """requested aggs -> deduplicated list of local partial aggs + recipe per requested agg"""
    partial, index = [], {}

    def need(op, col):
        key = (op, col)
        if key not in index:
            index[key] = len(partial)
            partial.append(key)
        return index[key]

    recipe = []
    for op, col in aggs:
        if op == "mean":
            recipe.append(("mean", need("mean", col), need("count", col)))
        elif op in ("sum", "min", "max", "count"):
            recipe.append((op, need(op, col)))
        elif op == "count0":
            recipe.append(("count0", need("count0", None)))
        else:
            raise ValueError("unknown reducer %r" % (op,))
    return partial, recipe


def _all_to_all_v(t, send_counts, recv_counts, group):
    out = torch.empty(int(sum(recv_counts)), dtype=t.dtype, device=t.device)
    dist.all_to_all_single(out, t.contiguous(), list(recv_counts), list(send_counts), group=group)
    return out


def range_boundaries(gmin, gmax, world):
    """world-1 ascending key boundaries splitting [gmin, gmax] evenly (python ints, no overflow)"""
    width = gmax - gmin + 1
    return [gmin + (width * j) // world for j in range(1, world)]


def sharded_groupby_agg(backend, keys, values, aggs, group=None):
    """keys / values: this rank's row shard (tensors on the backend's device).
    aggs: [(op, value_index)], op in sum/mean/min/max/count/count0 (value_index None for count0).
    Returns (group_key_tensors, agg_tensors) for the key range this rank owns."""
    world = dist.get_world_size(group)
    if keys[0].dtype not in INT_KEY_DTYPES:
        raise NotImplementedError("distributed groupby partitions on an integer first key column")
    partial, recipe = _partials_for(aggs)
    gk, parts = backend.groupby_agg(keys, values, partial)
    dev = gk[0].device
    # weighted sums for means (float64)
    cols, merge_ops = [], []
    for i, (op, col) in enumerate(partial):
        p = parts[i]
        if op == "mean":
            cnt = parts[partial.index(("count", col))]
            p64 = p.to(torch.float64)
            p = torch.where(cnt > 0, p64 * cnt.to(torch.float64), torch.zeros_like(p64))
            merge_ops.append("sum")
        elif op in ("sum", "count", "count0"):
            merge_ops.append("sum")
        else:
            merge_ops.append(op)
        cols.append(p)

    # global key range of the first key (valid keys only).  The local group keys are ascending with the
    # NA group (sentinel = dtype min) first, so the range is read off three elements: one host sync.
    k0 = gk[0].to(torch.int64)
    na = torch.iinfo(gk[0].dtype).min
    big = torch.iinfo(torch.int64)
    nk = k0.numel()
    lo, neg_hi = big.max, big.max
    if nk:
        probe = k0[torch.tensor([0, min(1, nk - 1), nk - 1], device=dev)].tolist()
        first_valid = probe[0] if probe[0] != na else (probe[1] if nk > 1 else None)
        if first_valid is not None and first_valid != na:
            lo, neg_hi = int(first_valid), -int(probe[2])
    mm = torch.tensor([lo, neg_hi], dtype=torch.int64, device=dev)
    dist.all_reduce(mm, op=dist.ReduceOp.MIN, group=group)
    gmin, neg = mm.tolist()
    gmax = -neg
    if gmin > gmax:                       # no valid key anywhere: everything (NA groups) goes to rank 0
        gmin, gmax = 0, 0
    bounds = torch.tensor(range_boundaries(gmin, gmax, world), dtype=torch.int64, device=dev)
    # NA rows (sentinel) must land on rank 0: the sentinel is below every boundary
    cut = torch.searchsorted(k0.contiguous(), bounds) if world > 1 else bounds
    edges = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), cut.to(torch.int64),
                       torch.tensor([nk], dtype=torch.int64, device=dev)])
    send_counts = (edges[1:] - edges[:-1])
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = torch.stack([send_counts, recv_counts]).tolist()

    rk = [_all_to_all_v(k, sc, rc, group) for k in gk]
    rp = [_all_to_all_v(c, sc, rc, group) for c in cols]

    # merge the partials of each group on its owner
    mk, merged = backend.groupby_agg(rk, rp, [(merge_ops[i], i) for i in range(len(rp))])

    out = []
    for r in recipe:
        if r[0] == "mean":
            s, c = merged[r[1]], merged[r[2]]
            m = s / c.to(torch.float64)
            m = torch.where(c > 0, m, torch.full_like(m, float("nan")))
            src = partial[r[1]][1]
            out.append(m.to(values[src].dtype) if values[src].dtype == torch.float32 else m)
        else:
            out.append(merged[r[1]])
    return mk, out


def sharded_groupby_rows(backend, keys, cols, row_offset=0, group=None):
    """Rows in grouped order across ranks: `DT[:, cols, by(keys)]` (configuration 5's second step).

    keys / cols: this rank's row shard.  Every rank ends up owning a contiguous range of the first
    key; the concatenation of the ranks' outputs in rank order is what a single process returns
    (groups ascending, NA group first, original row order inside a group).
    Returns (offsets int32[ng_local+1], global row ids int64[n_local_out], [cols in grouped order]).

    One exchange step: rows are grouped by destination rank on the sender (a stable
    groupby_rows on the int8 destination, so every slab keeps the sender's row order), slabs travel
    with one all-to-all-v per column, and because a receiver concatenates slabs in source-rank order
    -- and shards are row blocks in rank order -- arrival order IS global row order: one stable
    local groupby_rows reproduces the reference's permutation exactly."""
    world = dist.get_world_size(group)
    if keys[0].dtype not in INT_KEY_DTYPES:
        raise NotImplementedError("distributed groupby partitions on an integer first key column")
    if world > 16:
        raise NotImplementedError("range partition into at most 16 destinations")
    dev = keys[0].device
    n = keys[0].numel()
    k0 = keys[0]
    na = torch.iinfo(k0.dtype).min
    big = torch.iinfo(torch.int64)
    # global range of the first key over its valid rows (two scalars through one all-reduce)
    if n:
        valid = k0 != na
        lo = int(torch.where(valid, k0, torch.full_like(k0, torch.iinfo(k0.dtype).max)).min().item())
        hi = int(torch.where(valid, k0, torch.full_like(k0, na)).max().item())
        if hi == na:            # no valid key on this rank
            lo, hi = big.max, -big.max
    else:
        lo, hi = big.max, -big.max
    mm = torch.tensor([lo, -hi], dtype=torch.int64, device=dev)
    dist.all_reduce(mm, op=dist.ReduceOp.MIN, group=group)
    gmin, neg = mm.tolist()
    gmax = -neg
    if gmin > gmax:
        gmin, gmax = 0, 0
    bounds = range_boundaries(gmin, gmax, world)
    rowid = torch.arange(row_offset, row_offset + n, dtype=torch.int64, device=dev)
    payload = list(keys) + list(cols) + [rowid]
    if world > 1:
        dest = backend.range_bucket(k0, bounds)
        _, slabs = backend.groupby_rows([dest], payload)             # rows grouped by destination, order kept
        send_counts = torch.bincount(dest.to(torch.int64), minlength=world)[:world]
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=group)
        sc, rc = torch.stack([send_counts, recv_counts]).tolist()
        recv = [_all_to_all_v(t, sc, rc, group) for t in slabs]
    else:
        recv = payload
    rkeys, rcols, rrow = recv[:len(keys)], recv[len(keys):-1], recv[-1]
    off, out = backend.groupby_rows(rkeys, rcols + [rrow])
    return off, out[-1], out[:-1]
