#!/bin/bash
# round 4: guard-page runs (every library buffer between unmapped pages, every launch synchronised) of the suites that
# exercise the new kernels: lane-mask ranking, ragged MSD levels (forced on every sort), windowed final level, one-pass
# filter, filter take from registers; then the 1e9-row configs.
export TMPDIR=/tmp
OUT=gpurun_out/r04_guard; rm -rf $OUT; mkdir -p $OUT
FILES="tests/test_gpu_msd.py tests/test_gpu_parity.py tests/test_reference_suite.py tests/test_gpu_sharded.py tests/test_gpu_tilelocal.py"
for MODE in 1 2; do
  GUARD_TIMEOUT=900 bash scripts/guard_check.sh $MODE $FILES > $OUT/mode$MODE.txt 2>&1
  DTHIP_SORT_PATH=2 DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=64 DTHIP_FILTER_PATH=0 GUARD_TIMEOUT=900 bash scripts/guard_check.sh $MODE tests/test_gpu_parity.py tests/test_reference_suite.py tests/test_gpu_msd.py > $OUT/mode${MODE}_forced_msd.txt 2>&1
done
DTHIP_GUARD=1 timeout 600 python scripts/guard_fullsize.py --configs C3,C4,C5,C3_hard,C2,C1 > $OUT/fullsize1.txt 2>&1; echo "fullsize rc=$?" >> $OUT/fullsize1.txt
DTHIP_GUARD=1 DTHIP_SORT_PATH=2 timeout 600 python scripts/guard_fullsize.py --configs C5 > $OUT/fullsize1_msd.txt 2>&1; echo "fullsize (MSD levels) rc=$?" >> $OUT/fullsize1_msd.txt
for f in mode1 mode1_forced_msd mode2 mode2_forced_msd; do echo "# $f"; cat $OUT/$f.txt; done
grep -v amdgpu.ids $OUT/fullsize1.txt | tail -9; grep -v amdgpu.ids $OUT/fullsize1_msd.txt | tail -3
