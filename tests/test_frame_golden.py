"""Frame-level golden answers: tests/golden/frame_queries.json holds 80 `DT[i, j, by/sort]` queries
together with what the UNMODIFIED reference returned for them (tests/golden/make_frame_golden.py,
run in the dev container).  The same query strings are evaluated here against
datatable_amd.frame on the GPU: names, stypes and every value must agree (float sums/means to 1e-6
relative, everything else exactly)."""
import json
import math
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_queries.json")))


def _dec(x):
    if x == "inf":
        return math.inf
    if x == "-inf":
        return -math.inf
    return x


@pytest.mark.parametrize("qi", range(len(GOLD["queries"])), ids=[q["query"][:70] for q in GOLD["queries"]])
def test_frame_query_matches_reference(qi):
    from datatable_amd import frame as dt
    from datatable_amd.frame import f, by, sort, sum, mean, min, max, count, first, last   # noqa: F401,A004
    from datatable_amd.frame import sd, median, nunique, cov, corr, cumsum, cumprod, cummin, cummax, cumcount, ngroup, fillna  # noqa: F401
    q = GOLD["queries"][qi]
    spec = GOLD["frames"][q["frame"]]
    DT = dt.Frame({nm: [_dec(x) for x in c["values"]] for nm, c in spec.items()},
                  stypes={nm: c["stype"] for nm, c in spec.items()})
    R = eval(q["query"])
    assert list(R.names) == q["names"]
    assert list(R.stypes) == q["stypes"]
    got = R.to_list()
    assert len(got) == len(q["columns"])
    is_float_red = any(k in q["query"] for k in ("sum(", "mean(", "sd(", "cov(", "corr(", "cumprod("))
    for ci, (g, e) in enumerate(zip(got, q["columns"])):
        e = [_dec(x) for x in e]
        assert len(g) == len(e), "column %d: %d rows, expected %d" % (ci, len(g), len(e))
        for x, y in zip(g, e):
            if isinstance(y, float) and isinstance(x, float) and is_float_red and math.isfinite(y):
                assert abs(x - y) <= 1e-6 * abs(y) + 1e-9, (q["names"][ci], x, y)
            else:
                assert x == y and type(x) is type(y), (q["names"][ci], g[:12], e[:12])


# ---- randomised queries (tests/golden/make_frame_fuzz_golden.py) -----------------------------------
FUZZ = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_fuzz.json")))
FUZZ2 = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_fuzz2.json")))     # other query templates (batch 2)
FUZZ4 = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_fuzz4.json")))     # chains of views (batch 4)
FUZZ5 = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_fuzz5.json")))     # corner cases (batch 5)
FUZZ6 = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_fuzz6.json")))     # no-by mixtures, f[:] forms (batch 6)
for _F in (FUZZ2, FUZZ4, FUZZ5, FUZZ6):
    for _k, _v in _F["frames"].items():
        FUZZ["frames"][_k] = _v
    FUZZ["queries"] += _F["queries"]


def _dec2(x):
    if x == "-0.0":
        return -0.0
    return _dec(x)


@pytest.mark.parametrize("qi", range(len(FUZZ["queries"])), ids=["%d:%s" % (i, q["query"][:60]) for i, q in enumerate(FUZZ["queries"])])
def test_fuzz_query_matches_reference(qi):
    """2500 seeded random queries over 625 random frames (all fixed-width stypes, NAs, +-inf, -0.0 keys):
    names, stypes and values as the unmodified reference returned them"""
    from datatable_amd import frame as dt
    from datatable_amd.frame import f, by, sort, sum, mean, min, max, count, first, last   # noqa: F401,A004
    from datatable_amd.frame import sd, median, nunique, cov, corr, cumsum, cumprod, cummin, cummax, cumcount, ngroup, fillna  # noqa: F401
    _run_fuzz_query(FUZZ["queries"][qi], resident=False)


@pytest.mark.parametrize("qi", range(0, 700, 2), ids=["%d" % i for i in range(0, 700, 2)])
def test_fuzz_query_on_resident_frame(qi):
    """every second query of batch 1 again, on a Frame whose columns were moved to HBM first (Frame.to_device)"""
    _run_fuzz_query(FUZZ["queries"][qi], resident=True)


def _run_fuzz_query(q, resident):
    from datatable_amd import frame as dt
    from datatable_amd.frame import f, by, sort, sum, mean, min, max, count, first, last   # noqa: F401,A004
    from datatable_amd.frame import sd, median, nunique, cov, corr, cumsum, cumprod, cummin, cummax, cumcount, ngroup, fillna  # noqa: F401
    spec = FUZZ["frames"][q["frame"]]
    DT = dt.Frame({nm: [_dec2(x) for x in c["values"]] for nm, c in spec.items()},
                  stypes={nm: c["stype"] for nm, c in spec.items()})
    assert list(DT.stypes) == [c["stype"] for c in spec.values()]
    if resident:
        DT.to_device()
    R = eval(q["query"])
    assert list(R.names) == q["names"]
    assert list(R.stypes) == q["stypes"]
    got = R.to_list()
    assert len(got) == len(q["columns"])
    for ci, (g, e) in enumerate(zip(got, q["columns"])):
        e = [_dec2(x) for x in e]
        assert len(g) == len(e), "column %d: %d rows, expected %d" % (ci, len(g), len(e))
        rel = 2e-5 if q["stypes"][ci] == 6 else 1e-6
        for ri, (x, y) in enumerate(zip(g, e)):
            if isinstance(y, float) and isinstance(x, float) and math.isfinite(y) and math.isfinite(x):
                assert abs(x - y) <= rel * abs(y) + 1e-9, (q["names"][ci], ri, x, y)
            else:
                assert x == y and type(x) is type(y), (q["names"][ci], ri, g[:12], e[:12])


# ---- set functions, unique, keys, joins over several frames (batch 3 of make_frame_fuzz_golden.py) ------
FUZZ3 = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_fuzz3.json")))["cases"]


@pytest.mark.parametrize("ci", range(len(FUZZ3)), ids=["%d:%s" % (i, c["query"][:50]) for i, c in enumerate(FUZZ3)])
def test_fuzz_sets_keys_joins_match_reference(ci):
    """300 seeded random cases: union / intersect / setdiff / symdiff over 1-4 frames of possibly different
    stypes, unique(), Frame.key, natural joins with 1-2 keys of possibly different stypes -- names, stypes,
    values, key count, or the exception type, as the unmodified reference produced them"""
    from datatable_amd import frame as dt
    from datatable_amd.frame import union, intersect, setdiff, symdiff, unique, join   # noqa: F401
    c = FUZZ3[ci]

    def keyed(F, *names):
        G = dt.Frame(F)
        G.key = list(names) if len(names) > 1 else names[0]
        return G

    ns = dict(union=union, intersect=intersect, setdiff=setdiff, symdiff=symdiff, unique=unique, join=join, keyed=keyed)
    for fname, spec in c["frames"].items():
        ns[fname] = dt.Frame({nm: [_dec2(x) for x in col["values"]] for nm, col in spec.items()},
                             stypes={nm: col["stype"] for nm, col in spec.items()})
    if "error" in c:
        with pytest.raises(Exception) as e:
            eval(c["query"], ns)
        assert type(e.value).__name__ == c["error"], (type(e.value).__name__, str(e.value), c["message"])
        return
    R = eval(c["query"], ns)
    assert list(R.names) == c["names"]
    assert list(R.stypes) == c["stypes"]
    assert len(R.key) == c["nkeys"]
    got = R.to_list()
    assert len(got) == len(c["columns"])
    for k, (g, e) in enumerate(zip(got, c["columns"])):
        e = [_dec2(x) for x in e]
        assert len(g) == len(e), (c["names"][k], len(g), len(e))
        for x, y in zip(g, e):
            assert x == y and type(x) is type(y), (c["names"][k], g[:12], e[:12])
            if isinstance(y, float) and y == 0:
                assert math.copysign(1, x) == math.copysign(1, y), (c["names"][k], "sign of zero")
