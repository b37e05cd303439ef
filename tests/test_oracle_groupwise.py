"""CPU: the oracle restatement of the group-wise operators (oracle/dt_oracle_groupwise.c) reproduces,
bit for bit, what the unmodified reference returned for every case of tests/golden/groupwise_cases.npz
(sd / median / nunique, cov / corr, cumsum / cumprod / cummin / cummax forward and reverse,
cumcount / ngroup)."""
import numpy as np
import pytest

from conftest import assert_same, groupwise_golden
from oracle import oracle as o

GW = groupwise_golden()


def oracle_out(c, key, ri, off, vals):
    """evaluate the golden output `key` of case `c` with the oracle"""
    parts = key.split(".")
    op = parts[0]
    rev = parts[-1] == "rev"
    if op in ("cumcount", "ngroup"):
        return o.cumulate(op, None, ri, off, reverse=rev)
    if op in ("cov", "corr"):
        i, j = int(parts[1]), int(parts[2])
        return o.reduce2(op, vals[i], vals[j], ri, off, stypes=(c["val_stypes"][i], c["val_stypes"][j]))
    vi = int(parts[1][1:])
    if op in ("sd", "median", "nunique"):
        return o.reducex(op, vals[vi], ri, off, stype=c["val_stypes"][vi])
    return o.cumulate(op, vals[vi], ri, off, reverse=rev, stype=c["val_stypes"][vi])


@pytest.mark.parametrize("name", GW.names())
def test_oracle_groupwise_matches_reference(name):
    c = GW.by_name[name]
    keys, vals = GW.keys(name), GW.vals(name)
    ri, off = o.group(keys, stypes=c["key_stypes"])
    assert_same(ri, GW.get(name, "ri"), "rowindex")
    assert_same(off, GW.get(name, "off"), "offsets")
    assert len(c["outs"]) >= 15
    for key, ost in c["outs"].items():
        want = GW.get(name, key)
        got = oracle_out(c, key, ri, off, vals)
        assert_same(got, want, "%s/%s" % (name, key))
