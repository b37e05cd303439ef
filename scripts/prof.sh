#!/bin/bash
# rocprofv3 evidence of one round, one script (replaces the per-round prof_r0N.sh / r0N_*.sh lab notes):
#   scripts/prof.sh <round tag, e.g. r05> [configs, default "C2 C4 C5 C3_hard"]
# C3 (the headline) through bench.py: kernel stats + FETCH_SIZE / WRITE_SIZE passes -> per-kernel entries of
# pmc_traffic.json (+ "_build_id" = dthip_build_id() of the library profiled: bench.py drops the record on another build);
# every other config through scripts/configs_bench.py: kernel stats + the HBM traffic of one whole query -> "_configs".
# Counters are collected in their own passes with --kernel-trace only (never with other trace domains).
# Output: gpurun_out/prof_<tag>/ ; the files to keep are copied to profiles/<tag>_*.txt and profiles/pmc_traffic.json
# by the caller (scripts/prof_keep.sh <tag>).
export TMPDIR=/tmp
TAG=${1:?round tag}; shift
CFGS=${*:-C2 C4 C5 C3_hard C3_sgrp}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-check --configs= --host-rows 0 --no-shim-resident --no-dist-1rank"
( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1 )
( cd /tmp; timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- $BENCH --steps 1 --warmup 0 > $OUT/fetch.log 2>&1 )
( cd /tmp; timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- $BENCH --steps 1 --warmup 0 > $OUT/write.log 2>&1 )
for d in stats fetch write; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/$d.txt 2>&1 && rm -rf $OUT/$d
done
python scripts/make_pmc_json.py $OUT/fetch.txt $OUT/write.txt 1000000000 $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
for name in $CFGS; do
  case $name in
    C2) id=2; alg=4013600000; rows=100000000;;
    C4) id=4; alg=16240109656; rows=1000000000;;
    C5) id=5; alg=30400000000; rows=1000000000;;
    C3_hard) id=6; alg=16160000000; rows=1000000000;;
    C3_sgrp) id=10; alg=20120000000; rows=1000000000;;
    *) echo "unknown config $name"; continue;;
  esac
  ( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/$name -o cfg -- python $REPO/scripts/configs_bench.py --configs $id --reps 3 > $OUT/$name.log 2>&1 )
  db=$(find $OUT/$name -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/${name}_stats.txt 2>&1 && rm -rf $OUT/$name
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout -k 5 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/${name}_$C -o cfg -- python $REPO/scripts/configs_bench.py --configs $id --once > $OUT/${name}_$C.log 2>&1 )
    db=$(find $OUT/${name}_$C -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/${name}_$C.txt 2>&1 && rm -rf $OUT/${name}_$C
  done
  python scripts/pmc_config_json.py $name $OUT/${name}_FETCH_SIZE.txt $OUT/${name}_WRITE_SIZE.txt 1 $rows $alg $OUT/pmc_traffic.json >> $OUT/pmc_traffic.txt 2>&1
done
# C2: LDS instruction / bank-conflict counters of the aggregation (SQ_INSTS_LDS, SQ_LDS_BANK_CONFLICT)
( cd /tmp; timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $OUT/C2_lds -o cfg -- python $REPO/scripts/configs_bench.py --configs 2 --once > $OUT/C2_lds.log 2>&1 )
db=$(find $OUT/C2_lds -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/C2_lds.txt 2>&1 && rm -rf $OUT/C2_lds
head -12 $OUT/stats.txt; cat $OUT/pmc_traffic.txt
