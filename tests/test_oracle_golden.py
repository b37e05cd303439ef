"""CPU: the oracle restatement (oracle/dt_oracle.c) reproduces, bit for bit, what the
unmodified reference returned for every golden case (incl. float64 sums)."""
import numpy as np
import pytest

from conftest import assert_same, golden_names
from oracle import oracle as o


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(gold, name):
    c = gold.by_name[name]
    keys = gold.keys(name)
    ri, off = o.group(keys, stypes=c["key_stypes"])
    assert_same(ri, gold.get(name, "ri"), "rowindex")
    assert_same(off, gold.get(name, "off"), "offsets")
    for i in range(len(keys)):
        gk = keys[i][ri[off[:-1]]] if c["n"] else keys[i][:0]
        assert_same(gk, gold.get(name, "gk%d" % i), "group key %d" % i)
    vals = gold.vals(name)
    for opn, vi, ost in c["aggs"]:
        r = o.reduce(opn, vals[vi], ri, off, stype=c["val_stypes"][vi])
        assert_same(r, gold.get(name, "%s.v%d" % (opn, vi)), "%s(v%d)" % (opn, vi))


def test_oracle_bool_to_rowindex():
    m = np.array([1, 0, -128, 1, 1, 0, -128], np.int8)
    assert o.bool_to_rowindex(m).tolist() == [0, 3, 4]


def test_oracle_filter_and_gather():
    # V = DT[f.v > 1.6, :]; V[:, sum(f.v), by(f.k)] -> k=[NA,2,3], sum=[10,16,inf]
    k = np.array([3, -2**31, 1, 3, 1, -2**31, 2, 3], np.int32)
    v = np.array([1.5, 2.0, np.nan, 4.0, np.nan, 8.0, 16.0, np.inf])
    ri = o.filter_cmp(v, ">", 1.6)
    assert ri.tolist() == [1, 3, 5, 6, 7]
    kv, vv = o.gather(k, ri), o.gather(v, ri)
    gri, off = o.group([kv])
    assert kv[gri[off[:-1]]].tolist() == [-2**31, 2, 3]
    s = o.reduce("sum", vv, gri, off)
    assert s.tolist() == [10.0, 16.0, np.inf]
    assert o.gather(v, np.array([0, -2**31, 6], np.int32)).tolist()[0::2] == [1.5, 16.0]
    assert np.isnan(o.gather(v, np.array([-2**31], np.int32))[0])


def test_oracle_threads_do_not_change_results():
    """the multi-threaded oracle (bench.py's cpu_baseline) returns exactly what the single-threaded one
    does: same RowIndex, offsets and bit-identical float sums"""
    import numpy as np
    from oracle import oracle as o
    rng = np.random.default_rng(5)
    n = 400_000
    keys = [rng.integers(-500, 500, n).astype(np.int32), rng.integers(0, 70_000, n).astype(np.int64)]
    keys[1][rng.random(n) < 0.01] = np.iinfo(np.int64).min
    v = rng.standard_normal(n)
    ref = {}
    try:
        for t in (1, 3, 8):
            o.set_threads(t)
            ri, off = o.group(keys)
            red = [o.reduce(op, v, ri, off) for op in ("sum", "mean", "min", "max", "count")]
            if t == 1:
                ref = dict(ri=ri, off=off, red=red)
            else:
                assert np.array_equal(ri, ref["ri"]) and np.array_equal(off, ref["off"])
                for a, b in zip(red, ref["red"]):
                    assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f")
    finally:
        o.set_threads(1)
