#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_guard3; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do
  DTHIP_GUARD=1 timeout 600 python scripts/guard_fullsize.py --configs C3,C4,C5,C3_hard,C2,C1 > $OUT/full_$i.txt 2>&1; echo "guard 1 full sequence run $i rc=$?" | tee -a $OUT/summary.txt
  grep -E "guard=|Memory access|abort while|context closed" $OUT/full_$i.txt | tee -a $OUT/summary.txt
done
DTHIP_GUARD=2 timeout 600 python scripts/guard_fullsize.py --configs C3,C4,C5,C3_hard > $OUT/full_m2.txt 2>&1; echo "guard 2 full sequence rc=$?" | tee -a $OUT/summary.txt
grep -E "guard=|Memory access|abort while|context closed" $OUT/full_m2.txt | tee -a $OUT/summary.txt
