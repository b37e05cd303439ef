#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun). $1 = tag (e.g. r01b); rest = bench args
TAG=${1:-r01}; shift
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-check "$@" > $OUT/bench_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-check "$@" > $OUT/bench_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-check "$@" > $OUT/bench_write.log 2>&1
cd $REPO
for d in stats pmc_fetch pmc_write; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/$d.txt 2>&1
done
tail -3 $OUT/bench_stats.log
head -25 $OUT/stats.txt
grep -i "bucket\|table_agg\|minmax" $OUT/pmc_fetch.txt $OUT/pmc_write.txt
