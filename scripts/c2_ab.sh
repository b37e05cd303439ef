#!/bin/bash
# C2 A/B on one box, alternating runs of one library under environment switches (round 6, last session):
#   scripts/c2_ab.sh [rounds]      -> gpurun_out/c2ab/ab.txt
# DTHIP_TL_FEW=0: <= 128 buckets keep the exact-position layout (histogram pass); DTHIP_TAB_KB: LDS bytes a table may take
export TMPDIR=/tmp
OUT=gpurun_out/c2ab; mkdir -p $OUT
R=${1:-2}
run() { echo "--- $*"; env "$@" python scripts/configs_bench.py --configs 2 --reps 5 --profile 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids"; }
for i in $(seq $R); do
  run DTHIP_TL_FEW=0 DTHIP_TAB_KB=144
  run DTHIP_TL_FEW=0
  run DTHIP_TL_FEW=1
  run DTHIP_TL_FEW=1 DTHIP_TL_ITEMS=16
done 2>&1 | tee $OUT/ab.txt
