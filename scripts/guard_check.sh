#!/bin/bash
# Guard-page run of the GPU suites + the full-size configs (VERDICT r02 "next" 1a).  Every test FILE runs in its own
# process, so a GPU memory access fault (process abort) in one file does not hide the others.
#   usage: scripts/guard_check.sh [1|2] [file ...]     -> gpurun_out/guard<mode>/
MODE=${1:-1}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/guard$MODE; mkdir -p $OUT
FILES=${@:-tests/test_gpu_tilelocal.py tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_groupwise.py tests/test_gpu_sets_join.py tests/test_reference_suite.py tests/test_frame_golden.py tests/test_gpu_fullsize.py}
export DTHIP_GUARD=$MODE
: > $OUT/summary.txt
for f in $FILES; do
  b=$(basename $f .py)
  timeout ${GUARD_TIMEOUT:-600} python -m pytest $f -m gpu -x -q -p no:cacheprovider > $OUT/$b.log 2>&1
  rc=$?
  echo "$b rc=$rc :: $(grep -E 'passed|failed|error' $OUT/$b.log | tail -1) :: $(grep -c 'dthip guard\] context closed' $OUT/$b.log) clean context closes :: $(grep -E 'Memory access fault|dthip guard\] abort' $OUT/$b.log | head -2 | tr '\n' ' ')" >> $OUT/summary.txt
done
cat $OUT/summary.txt
