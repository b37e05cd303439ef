#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_guard2; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do for M in 1 2; do
  DTHIP_GUARD=$M timeout 600 python scripts/guard_fullsize.py --configs C3_hard,C2,C1 > $OUT/hard_m${M}_$i.txt 2>&1; echo "guard $M run $i rc=$?" | tee -a $OUT/summary.txt
  grep -E "guard=|Memory access|abort while|context closed" $OUT/hard_m${M}_$i.txt | tee -a $OUT/summary.txt
done; done
