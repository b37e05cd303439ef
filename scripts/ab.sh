#!/bin/bash
# A/B of two builds of the library on the same box, alternating runs: scripts/ab.sh <alt.so> <configs> [scale] [rounds]
ALT=$1; CFG=${2:-5}; SC=${3:-1.0}; R=${4:-2}
for i in $(seq $R); do
  echo "--- base"; python scripts/configs_bench.py --configs $CFG --scale $SC --reps 3 --profile 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr"
  echo "--- alt $ALT"; DTHIP_LIB=$PWD/$ALT python scripts/configs_bench.py --configs $CFG --scale $SC --reps 3 --profile 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr"
done
