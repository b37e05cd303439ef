"""Evidence matches HEAD ("stale evidence"): the library in the tree was built from the sources in the tree,
and the committed PMC traffic record -- which bench.py puts into the driver's line -- was recorded on exactly that build.
dthip_build_id() is a hash of csrc's sources and include/dthip.h taken by the Makefile (csrc/Makefile: build_id.inc)."""
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "datatable_amd", "csrc")


def source_hash():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    srcs = re.search(r"^SRCS := (.*)$", mk, re.M).group(1).split()
    hdrs = re.search(r"^HDRS := (.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for f in srcs + hdrs:                       # the order of the Makefile's `cat`
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def test_library_was_built_from_the_sources_in_the_tree():
    from datatable_amd import _lib as L
    assert L.load().dthip_build_id().decode() == source_hash(), "datatable_amd/libdthip.so is older than csrc/: run build()"


def test_committed_counter_records_belong_to_this_build():
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert rec.get("_build_id") == source_hash(), \
        "profiles/pmc_traffic.json was recorded on another library build: re-run scripts/prof.sh + scripts/prof_keep.sh (bench.py drops the record otherwise)"


def test_committed_bench_lines_name_their_build():
    """every round-6 bench line kept under profiles/ says which library produced it, and it is this one"""
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if name.startswith("r06_bench") and name.endswith(".json"):
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            assert d.get("library_build_id") == source_hash(), name
            assert d["parity"]["configs"] and all(d["parity"]["configs"].values()), name
