"""CPU: the oracle restatement of the set functions and of the natural join (oracle/dt_oracle_sets.c)
reproduces what the unmodified reference returned for every case of tests/golden/sets_join_cases.npz."""
import numpy as np
import pytest

from conftest import assert_same, sets_join_golden
from oracle import oracle as o

GS = sets_join_golden()


@pytest.mark.parametrize("name", GS.names("set"))
def test_oracle_setops_match_reference(name):
    c = GS.by_name[name]
    srcs = [GS.get(name, "src%d" % i) for i in range(len(c["stypes"]))]
    if len(set(c["stypes"])) != 1:
        pytest.skip("mixed stypes are up-cast by the Frame layer")
    st = c["stypes"][0]
    stacked = np.concatenate(srcs)
    for op in ("union", "intersect", "setdiff", "symdiff"):
        idx = o.setop(op, srcs, stype=st)
        assert_same(stacked[idx], GS.get(name, op), "%s/%s" % (name, op))
    if "unique" in c["outs"]:
        # dt.unique(frame) = union of the frame's columns (set_funcs.cc:180-193)
        assert_same(stacked[o.setop("union", [stacked], stype=st)], GS.get(name, "unique"), name + "/unique")


@pytest.mark.parametrize("name", GS.names("join"))
def test_oracle_join_matches_reference(name):
    c = GS.by_name[name]
    nk = len(c["xstypes"])
    x = [GS.get(name, "x%d" % k) for k in range(nk)]
    j = [GS.get(name, "j%d" % k) for k in range(nk)]
    got = o.join_index(x, j, xstypes=c["xstypes"], jstypes=c["jstypes"])
    assert_same(got, GS.get(name, "index"), name)
