#!/bin/bash
# One GPU call that records a round's evidence on the library of the tree (run through gpurun; everything lands under gpurun_out/):
#   scripts/final.sh <round tag>
# 1. scripts/prof.sh: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of C3 and of every other config -> pmc_traffic.json
# 2. scripts/prof_keep.sh ON THE BOX, so that the bench line of step 3 carries the traffic of this very build
# 3. python bench.py (the driver's default invocation) -> gpurun_out/<tag>_bench_1e9.json
# 4. the reference's own test files on the patched in-tree build (scripts/run_ref_suite.sh)
TAG=${1:?round tag}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/prof.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
bash scripts/prof_keep.sh $TAG >> gpurun_out/prof_$TAG.log 2>&1
python bench.py > gpurun_out/${TAG}_bench_1e9.json 2> gpurun_out/${TAG}_bench_1e9.err
tail -2 gpurun_out/${TAG}_bench_1e9.err
SKIP_CPU_BASELINE=${SKIP_CPU_BASELINE:-} bash scripts/run_ref_suite.sh > gpurun_out/ref_suite_$TAG.log 2>&1
tail -12 gpurun_out/ref_suite_$TAG.log
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_1e9.json"))
print("C3 ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel"], "whole", d["roofline"]["whole_step_frac"], "total s", d.get("seconds_total"))
for k, v in d.get("configs", {}).items():
    print(k, round(v["ms"], 3), v.get("amplification"), (v.get("parity") or {}).get("ok"))
print(d["parity"]["configs"])
PY
