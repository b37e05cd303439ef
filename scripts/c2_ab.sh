#!/bin/bash
# C2 A/B on one box, alternating runs of one library under environment switches: scripts/c2_ab.sh [rounds]
export TMPDIR=/tmp
OUT=gpurun_out/c2ab; mkdir -p $OUT
R=${1:-2}
run() { echo "--- $*"; env "$@" python scripts/configs_bench.py --configs 2 --reps 5 --profile 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr"; }
for i in $(seq $R); do
  run DTHIP_PAY_PIPE=0
  run DTHIP_PAY_PIPE=1
  run DTHIP_PAY_PIPE=1 DTHIP_TAB_KB=78
  run DTHIP_PAY_PIPE=1 DTHIP_TAB_KB=40
  run DTHIP_PAY_PIPE=1 DTHIP_TAB_KB=78 DTHIP_PART_DIV=8
  run DTHIP_PAY_PIPE=1 DTHIP_TAB_KB=78 DTHIP_PART_DIV=32
done 2>&1 | tee $OUT/ab.txt
echo "--- variant 1 (512-thread tiles)" | tee -a $OUT/ab.txt
python scripts/configs_bench.py --configs 2 --reps 5 --profile --bucket-variant 1 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" | tee -a $OUT/ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
