"""Parity at BASELINE.json's sizes: the HIP path through the C ABI against the CPU oracle (oracle/dt_oracle.c,
pinned bit-exactly to reference-generated goldens, run here on all host cores with OpenMP) on the SAME seeded
inputs -- per-group values, not only properties.

  C2  1e8 rows, int64 key (1e5 groups), sum+mean+min+max of 4 float64 columns            full size
  C3  1e8 rows (10 % of 1e9; all 1e9 rows are compared inside bench.py), int64 key, 1e7 groups, sum
  C3h 1e8 rows, 63-bit keys from a pool of 1e7 (the 64-significant-bit path), sum + count
  C4  1e8 rows (10 %), (int32, int32) composite key ~1e7 pairs, count() + sum
  C5  1e8 rows (10 %), filter x > 0 -> RowIndex -> view -> rows in grouped order, ~1e7..5e7 groups

Bit-exact: group keys, group sizes / offsets, RowIndex permutations, min/max, counts.  float64 sums / means:
|got - exp| <= 1e-6 |exp| + 1e-12 * sum|v| of the group (BASELINE.json's tolerance).
"""
import os

import numpy as np
import pytest

from conftest import assert_close, assert_same
from oracle import oracle as o

pytestmark = pytest.mark.gpu

N = int(os.environ.get("DTHIP_FULLSIZE_ROWS", 100_000_000))


@pytest.fixture(scope="module", autouse=True)
def oracle_threads():
    o.set_threads(min(os.cpu_count() or 1, 64))
    yield
    o.set_threads(1)


def group_abs(v, ri, off):
    a = np.abs(v)[ri]
    return np.add.reduceat(a, off[:-1].astype(np.int64))


def check_fused(ctx, keys, vals, ops, with_count0=True, paths=(0,)):
    ri, off = o.group(keys)
    gkeys = [k[ri[off[:-1]]] for k in keys]
    alist = [(op, vi) for vi in range(len(vals)) for op in ops] + ([("count0", None)] if with_count0 else [])
    for path in paths:
        ctx.set_option("agg_path", path)
        try:
            r = ctx.groupby_agg(keys, vals, alist)
        finally:
            ctx.set_option("agg_path", 0)
        assert r.ngroups == len(off) - 1
        assert_same(r.offsets(), off, "offsets")
        for i in range(len(keys)):
            assert_same(r.key(i), gkeys[i], "group key %d" % i)
        for a, (op, vi) in enumerate(alist):
            if op == "count0":
                assert_same(r.agg(a), np.diff(off).astype(np.int64), "count()")
                continue
            exp = o.reduce(op, vals[vi], ri, off)
            if op in ("sum", "mean"):
                assert_close(r.agg(a), exp, scale=group_abs(vals[vi], ri, off), rel=1e-6, what="%s(v%d) path %d" % (op, vi, path))
            else:
                assert_same(r.agg(a), exp, "%s(v%d) path %d" % (op, vi, path))
        r.free()
    return ri, off


def test_c2_full_size(ctx):
    rng = np.random.default_rng(1234 + 2)
    n = N
    k = rng.integers(0, 100_000, n, dtype=np.int64)
    vals = [rng.standard_normal(n) for _ in range(4)]
    check_fused(ctx, [k], vals, ("sum", "mean", "min", "max"))


def test_c3_tenth(ctx):
    rng = np.random.default_rng(1234 + 3)
    n = N
    k = rng.integers(0, 10_000_000, n, dtype=np.int64)
    v = rng.standard_normal(n)
    ri, off = check_fused(ctx, [k], [v], ("sum",), paths=(0, 1))
    # S-grp: the RowIndex itself at this size
    r = ctx.groupby([k])
    assert_same(r.offsets(), off, "offsets")
    assert_same(r.rowindex(), ri, "rowindex")
    r.free()


def test_c3_hard_keys(ctx):
    rng = np.random.default_rng(1234 + 6)
    n = N
    pool = rng.integers(-2**62, 2**62, 10_000_000, dtype=np.int64)
    k = pool[rng.integers(0, len(pool), n)]
    v = rng.standard_normal(n)
    check_fused(ctx, [k], [v], ("sum", "count"))


def test_c3_hard_keys_two_value_columns(ctx):
    """the hash combiner with several value columns (one partition, one pass of hash tables + merge per column)"""
    rng = np.random.default_rng(1234 + 7)
    n = N // 2
    pool = rng.integers(-2**62, 2**62, 3_000_000, dtype=np.int64)
    k = pool[rng.integers(0, len(pool), n)]
    v = rng.standard_normal(n)
    w = rng.integers(-1000, 1000, n).astype(np.int64)
    check_fused(ctx, [k], [v, w], ("sum",))


def test_c4_two_keys(ctx):
    rng = np.random.default_rng(1234 + 4)
    n = N
    a = rng.integers(0, 3163, n).astype(np.int32)
    b = rng.integers(0, 3163, n).astype(np.int32)
    v = rng.standard_normal(n)
    check_fused(ctx, [a, b], [v], ("sum",))


def test_c5_filter_rowindex_group_rows(ctx):
    """V = DT[f.x > 0, :]; V[:, :, by(f.k)] -- the RowIndex of the filter, the view's columns, the grouping
    permutation and the rows in grouped order, each bit-exact against the oracle"""
    rng = np.random.default_rng(1234 + 5)
    n = N
    k = rng.integers(0, 100_000_000, n, dtype=np.int64)
    x = rng.standard_normal(n)
    ri_f = o.filter_cmp(x, ">", 0.0)
    kv_e, xv_e = k[ri_f], x[ri_f]
    got_ri, (kv, xv) = ctx.filter_take(x, ">", 0.0, [k, x])
    assert_same(got_ri, ri_f, "filter RowIndex")
    assert_same(kv, kv_e, "view key column"); assert_same(xv, xv_e, "view x column")
    ri, off = o.group([kv_e])
    r = ctx.groupby_rows([kv], [kv, xv, got_ri], want_rowindex=True)
    assert_same(r.offsets(), off, "offsets")
    assert_same(r.rowindex(), ri, "grouping RowIndex")
    assert_same(r.col(0), kv_e[ri], "key in grouped order")
    assert_same(r.col(1), xv_e[ri], "x in grouped order")
    assert_same(r.col(2), ri_f[ri], "composed RowIndex (rows of DT in grouped order)")
    r.free()
