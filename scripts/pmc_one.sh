#!/bin/bash
# usage: pmc_one.sh <kernel-substring> <python command...>   -- SQ/LDS counters of one kernel
export TMPDIR=/tmp
REPO=$PWD; K=$1; shift
OUT=$REPO/gpurun_out/pmc_one; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  ( cd $REPO && timeout 120 rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o run -- "$@" > $OUT/p$i.log 2>&1 )
  db=$(find $OUT/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/scripts/rocpd_summary.py $db | grep "$K" | grep -E "SQ_|GRBM"
done
