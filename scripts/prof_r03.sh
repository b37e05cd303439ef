#!/bin/bash
# round-3 rocprofv3 evidence at HEAD: C3 through bench.py (kernel stats + FETCH_SIZE / WRITE_SIZE passes -> pmc_traffic.json)
# and BASELINE configs C2 / C5 / hard-keys C3 through scripts/configs_bench.py (kernel stats).  Summaries land in
# gpurun_out/prof_r03*/; the ones to keep are copied into profiles/.
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_r03; rm -rf $OUT; mkdir -p $OUT
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-check --configs= --host-rows 0"
( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1 )
( cd /tmp; timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- $BENCH --steps 1 --warmup 0 > $OUT/fetch.log 2>&1 )
( cd /tmp; timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- $BENCH --steps 1 --warmup 0 > $OUT/write.log 2>&1 )
for d in stats fetch write; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/$d.txt 2>&1 && rm -rf $OUT/$d
done
python scripts/make_pmc_json.py $OUT/fetch.txt $OUT/write.txt 1000000000 $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
for C in 2 5 6; do
  ( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/c$C -o cfg -- python $REPO/scripts/configs_bench.py --configs $C --reps 3 > $OUT/c$C.log 2>&1 )
  db=$(find $OUT/c$C -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/c$C.txt 2>&1 && rm -rf $OUT/c$C
done
head -16 $OUT/stats.txt; cat $OUT/pmc_traffic.txt | head -12
