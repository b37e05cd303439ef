#!/bin/bash
# round-4 rocprofv3 evidence: C3 through bench.py (kernel stats + FETCH_SIZE / WRITE_SIZE passes -> per-kernel entries of
# pmc_traffic.json) and, NEW, the HBM traffic of one whole query of C2 / C4 / C5 / hard-keys C3 (-> "_configs" entries,
# which bench.py reports as configs.*.traffic / amplification).  Counters are collected in their own passes with
# --kernel-trace only.  Summaries land in gpurun_out/prof_r04/; the ones to keep are copied into profiles/.
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_r04; rm -rf $OUT; mkdir -p $OUT
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-check --configs= --host-rows 0 --no-shim-resident --no-dist-1rank"
( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1 )
( cd /tmp; timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- $BENCH --steps 1 --warmup 0 > $OUT/fetch.log 2>&1 )
( cd /tmp; timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- $BENCH --steps 1 --warmup 0 > $OUT/write.log 2>&1 )
for d in stats fetch write; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/$d.txt 2>&1 && rm -rf $OUT/$d
done
python scripts/make_pmc_json.py $OUT/fetch.txt $OUT/write.txt 1000000000 $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
# configs: name, configs_bench id, algorithmic bytes
for spec in "C2 2 4013600000" "C4 4 16240109656" "C5 5 30400000000" "C3_hard 6 16160000000"; do
  set -- $spec
  ( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/$1 -o cfg -- python $REPO/scripts/configs_bench.py --configs $2 --reps 3 > $OUT/$1.log 2>&1 )
  db=$(find $OUT/$1 -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/$1_stats.txt 2>&1 && rm -rf $OUT/$1
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout -k 5 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/$1_$C -o cfg -- python $REPO/scripts/configs_bench.py --configs $2 --once > $OUT/$1_$C.log 2>&1 )
    db=$(find $OUT/$1_$C -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/$1_$C.txt 2>&1 && rm -rf $OUT/$1_$C
  done
  ROWS=1000000000; [ $1 = C2 ] && ROWS=100000000
  python scripts/pmc_config_json.py $1 $OUT/$1_FETCH_SIZE.txt $OUT/$1_WRITE_SIZE.txt 1 $ROWS $3 $OUT/pmc_traffic.json >> $OUT/pmc_traffic.txt 2>&1
done
head -12 $OUT/stats.txt; cat $OUT/pmc_traffic.txt
