"""Plumbing between torch device tensors and the C ABI: torch is used only for
device memory, streams and torch.distributed (RCCL); all compute is libdthip."""
import torch

from . import _lib as L
from .engine import Context, DevCol

T2ST = {torch.bool: L.BOOL, torch.int8: L.INT8, torch.int16: L.INT16, torch.int32: L.INT32,
        torch.int64: L.INT64, torch.float32: L.FLOAT32, torch.float64: L.FLOAT64}
ST2T = {L.BOOL: torch.int8, L.INT8: torch.int8, L.INT16: torch.int16, L.INT32: torch.int32,
        L.INT64: torch.int64, L.FLOAT32: torch.float32, L.FLOAT64: torch.float64}


def context_for_current_stream(device_index):
    """A Context that launches on torch's current stream of that device -- also when that is the
    default stream (handle 0) -- so torch ops and libdthip kernels are ordered without extra
    synchronisation."""
    with torch.cuda.device(device_index):
        stream = torch.cuda.current_stream().cuda_stream
    ctx = Context(device_index)
    ctx.use_stream(stream)
    return ctx


def devcol(t, desc=False):
    assert t.is_cuda and t.is_contiguous()
    return DevCol(t.data_ptr(), T2ST[t.dtype], desc, keepalive=t)


def groupby_agg_tensors(ctx, keys, values, aggs, na_last=False, want_offsets=True):
    """keys/values: CUDA tensors.  Returns (offsets or None, [group key tensors], [agg tensors])."""
    n = keys[0].numel()
    # the context may launch on its own (non-blocking) stream: order it after torch's producers ...
    torch.cuda.current_stream(keys[0].device).synchronize()
    ctx.set_option("agg_offsets", 1 if want_offsets else 0)
    try:
        r = ctx.groupby_agg([devcol(k) for k in keys], [devcol(v) for v in values], aggs, nrows=n, na_last=na_last)
    finally:
        ctx.set_option("agg_offsets", 1)
    ng = r.ngroups
    dev = keys[0].device
    off = None
    if want_offsets:
        off = torch.empty(ng + 1, dtype=torch.int32, device=dev)
        r.offsets_into(off.data_ptr())
    gk = []
    for i, k in enumerate(keys):
        t = torch.empty(ng, dtype=ST2T[T2ST[k.dtype]], device=dev)
        if ng:
            r.key_into(i, t.data_ptr())
        gk.append(t)
    out = []
    for a in range(len(aggs)):
        t = torch.empty(ng, dtype=ST2T[r.agg_stype(a)], device=dev)
        if ng:
            r.agg_into(a, t.data_ptr())
        out.append(t)
    ctx.sync()          # ... and torch's consumers after the copies out of the result
    r.free()
    return off, gk, out


def groupby_rows_tensors(ctx, keys, cols, want_rowindex=False):
    """keys/cols: CUDA tensors.  Returns (offsets, rowindex or None, [cols in grouped order])."""
    n = keys[0].numel()
    dev = keys[0].device
    torch.cuda.current_stream(dev).synchronize()
    r = ctx.groupby_rows([devcol(k) for k in keys], [devcol(c) for c in cols], nrows=n, want_rowindex=want_rowindex)
    ng = r.ngroups
    off = torch.empty(ng + 1, dtype=torch.int32, device=dev)
    r.offsets_into(off.data_ptr())
    ri = None
    if want_rowindex:
        ri = torch.empty(n, dtype=torch.int32, device=dev)
        if n:
            r.rowindex_into(ri.data_ptr())
    out = []
    for c, col in enumerate(cols):
        t = torch.empty(n, dtype=col.dtype, device=dev)
        if n:
            r.col_into(c, t.data_ptr())
        out.append(t)
    ctx.sync()
    r.free()
    return off, ri, out


def range_bucket_tensor(ctx, key, bounds):
    """int8 destination of every row of the CUDA tensor `key` in the range partition given by `bounds`"""
    n = key.numel()
    out = torch.empty(n, dtype=torch.int8, device=key.device)
    torch.cuda.current_stream(key.device).synchronize()
    if n:
        ctx.range_bucket_dev(devcol(key), n, bounds, out.data_ptr())
    ctx.sync()
    return out


def group_reduce_tensor(ctx, op, value, rowindex, offsets, value2=None):
    """one value per group of a CUDA tensor: op in sum..last, sd, median, nunique; cov / corr with value2.
    rowindex (int32 CUDA tensor or None) and offsets (int32 CUDA tensor, ngroups+1) describe the grouping."""
    dev = offsets.device
    ng = offsets.numel() - 1
    n = rowindex.numel() if rowindex is not None else value.numel()
    torch.cuda.current_stream(dev).synchronize()
    if value2 is not None:
        st = ctx._lib.dthip_reduce2_out_stype(T2ST[value.dtype], T2ST[value2.dtype])
        out = torch.empty(ng, dtype=ST2T[st], device=dev)
        if ng:
            ctx.reduce2_dev(op, devcol(value), devcol(value2), rowindex.data_ptr() if rowindex is not None else 0,
                            offsets.data_ptr(), ng, n, out.data_ptr())
    else:
        from .engine import OPS
        st = ctx._lib.dthip_reduce_out_stype(OPS[op], T2ST[value.dtype])
        out = torch.empty(ng, dtype=ST2T[st], device=dev)
        if ng:
            ctx.reduce_dev(op, devcol(value), rowindex.data_ptr() if rowindex is not None else 0, offsets.data_ptr(), ng, n,
                           out.data_ptr())
    ctx.sync()
    return out


def group_cumulate_tensor(ctx, op, value, rowindex, offsets, reverse=False):
    """cumsum / cumprod / cummin / cummax of a CUDA tensor inside groups (value=None: cumcount / ngroup),
    one value per row in grouped order"""
    from .engine import CUMOPS
    dev = offsets.device
    ng = offsets.numel() - 1
    n = rowindex.numel() if rowindex is not None else (value.numel() if value is not None else int(offsets[-1].item()))
    torch.cuda.current_stream(dev).synchronize()
    st = ctx._lib.dthip_cumulate_out_stype(CUMOPS[op], T2ST[value.dtype] if value is not None else L.INT64)
    out = torch.empty(n, dtype=ST2T[st], device=dev)
    if n:
        ctx.cumulate_dev(op, devcol(value) if value is not None else None, rowindex.data_ptr() if rowindex is not None else 0,
                         offsets.data_ptr(), ng, n, out.data_ptr(), reverse=reverse)
    ctx.sync()
    return out
