#!/bin/bash
# round 4, after the small-table sequence and the NA-free guess: guard-page runs of the suites that exercise them
export TMPDIR=/tmp
OUT=gpurun_out/r04_guard4; rm -rf $OUT; mkdir -p $OUT
FILES="tests/test_gpu_parity.py tests/test_gpu_tilelocal.py tests/test_reference_suite.py"
for MODE in 1 2; do
  GUARD_TIMEOUT=600 bash scripts/guard_check.sh $MODE $FILES > $OUT/mode$MODE.txt 2>&1
done
DTHIP_GUARD=1 timeout 400 python scripts/guard_fullsize.py --configs C2,C1,C3 > $OUT/fullsize1.txt 2>&1; echo "fullsize rc=$?" >> $OUT/fullsize1.txt
for f in mode1 mode2; do echo "# $f"; cat $OUT/$f.txt; done
grep -v amdgpu.ids $OUT/fullsize1.txt | tail -6
