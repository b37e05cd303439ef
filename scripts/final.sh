#!/bin/bash
# round-end evidence: default bench line (with cpu baseline), rocprofv3 stats + PMC
TAG=${1:-r01c}
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/bench_default_$TAG.log 2>&1
grep '^{' gpurun_out/bench_default_$TAG.log | tail -1 > gpurun_out/bench_default_$TAG.json
tail -4 gpurun_out/bench_default_$TAG.log | cut -c1-600
bash scripts/prof.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
tail -30 gpurun_out/prof_$TAG.log
