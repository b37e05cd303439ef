#!/bin/bash
# SQ counters of one config's kernels (two passes of 8 SQ slots): scripts/pmc_one.sh <config id> <tag> [kernel filter]
export TMPDIR=/tmp
ID=${1:?config id}; TAG=${2:?tag}; REPO=$PWD; OUT=$REPO/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 300 rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o cfg -- python $REPO/scripts/configs_bench.py --configs $ID --once > $OUT/p$i.log 2>&1 )
  db=$(find $OUT/p$i -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/p$i.txt 2>&1 && rm -rf $OUT/p$i
done
grep -h "${3:-hash_agg}" $OUT/p1.txt $OUT/p2.txt
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp; timeout -k 5 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/$C -o cfg -- python $REPO/scripts/configs_bench.py --configs $ID --once > $OUT/$C.log 2>&1 )
  db=$(find $OUT/$C -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/$C.txt 2>&1 && rm -rf $OUT/$C
  grep -h "${3:-hash_agg}" $OUT/$C.txt | grep "$C"
done
