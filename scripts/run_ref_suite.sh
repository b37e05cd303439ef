#!/bin/bash
# The reference's OWN test files (grouping, sorting, reducers, cumulative operators, keys, joins, set functions: tests/
# test-groups.py, test-reduce.py, test-keys.py, test-sets.py, test-join.py, ijby/, dt/) against the patched reference build
# integration/_dt_hip (integration/build_dt_hip.sh): its group() hands every fixed-width key to libdthip.so on the GPU
# (S-grp) and its sum / mean / min / max / count reducer columns are computed by dthip_reduce (S-red).
# Baseline on the UNMODIFIED reference (oracle/_ref), same files, same interpreter: 5 failed (3 x np.NaN removed in NumPy 2,
# 2 x test_strXX_large6), 1202 passed, 11 skipped, 4 xfailed.  Run on the GPU box; summary -> gpurun_out/ref_suite/.
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/ref_suite; rm -rf $OUT; mkdir -p $OUT
cd $ROOT/integration/_dt_hip/ref
( DTHIP_LIB=$ROOT/datatable_amd/libdthip.so DTHIP_SGRP_REPORT=1 PYTHONPATH=$ROOT/integration/_dt_hip timeout 1500 \
  python -m pytest tests -q -p no:cacheprovider -o python_files="test*.py" -o addopts="" > $OUT/hip.log 2>&1; echo "rc=$?" >> $OUT/hip.log )
if [ -z "$SKIP_CPU_BASELINE" ]; then
( PYTHONPATH=$ROOT/oracle/_ref timeout 1500 python -m pytest tests -q -p no:cacheprovider -o python_files="test*.py" -o addopts="" > $OUT/cpu.log 2>&1; echo "rc=$?" >> $OUT/cpu.log )
else echo "(skipped: SKIP_CPU_BASELINE)" > $OUT/cpu.log; fi
cd $ROOT
{
  echo "== patched reference, group() on the GPU (DTHIP_LIB=datatable_amd/libdthip.so) =="
  grep -a "^FAILED\|passed\|failed\|dthip S-grp\|dthip S-red" $OUT/hip.log
  echo "== unmodified reference (oracle/_ref), CPU =="
  grep -a "^FAILED\|passed\|failed" $OUT/cpu.log
} > $OUT/summary.txt
cat $OUT/summary.txt
