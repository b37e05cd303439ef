#!/bin/bash
# SQ / LDS / fabric-request counters of the sort path (C5), the hash combiner (hard-keys C3) and C2, one counter group
# per rocprofv3 pass (kernel-trace only), at HEAD.   usage: scripts/pmc_sq_r03.sh "5 6 2" [scale]
export TMPDIR=/tmp
REPO=$PWD
CFGS=${1:-"5 6 2"}; SCALE=${2:-0.25}
OUT=$REPO/gpurun_out/pmc_sq_r03; rm -rf $OUT; mkdir -p $OUT
cd /tmp
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
G3="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
for c in $CFGS; do
  sc=$SCALE; [ "$c" = "2" ] && sc=1.0
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    ( cd $REPO && timeout -k 5 120 rocprofv3 --pmc $G --kernel-trace -d $OUT/c${c}_p$i -o run -- python scripts/configs_bench.py --configs $c --scale $sc --reps 1 > $OUT/c${c}_p$i.log 2>&1 )
    db=$(find $OUT/c${c}_p$i -name "*.db" | head -1)
    if [ -n "$db" ]; then python $REPO/scripts/rocpd_summary.py $db > $OUT/c${c}_p$i.txt 2>&1; rm -rf $OUT/c${c}_p$i; else echo "c$c p$i: no db (rc)"; tail -3 $OUT/c${c}_p$i.log; fi
  done
done
ls $OUT
