#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun). $1 = tag (e.g. r01a)
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $OUT/bench_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $OUT/bench_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $OUT/bench_write.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -30
ls -la $OUT/*
