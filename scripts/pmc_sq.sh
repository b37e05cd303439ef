#!/bin/bash
# SQ / LDS counters of the bucketed pipeline (separate rocprofv3 --pmc passes, kernel-trace only)
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-check > $OUT/p$i.log 2>&1
  db=$(find $OUT/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/scripts/rocpd_summary.py $db | grep -E "bucket_partition|table_agg|bucket_hist" | grep -v "^void" | grep -E "SQ_|GRBM|TCC" >> $OUT/summary.txt
done
cat $OUT/summary.txt
