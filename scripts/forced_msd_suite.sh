#!/bin/bash
# The GPU suites with the MSD levels FORCED onto every sort (they are automatic only from 2^26 rows): ragged tiles, windowed
# final level with 64-row buckets, on every golden / ported / full-size case.   -> gpurun_out/forced_msd.txt
export TMPDIR=/tmp
OUT=gpurun_out/forced_msd.txt; mkdir -p gpurun_out; : > $OUT
python -c "from datatable_amd import _lib as L; print('library build', L.load().dthip_build_id().decode())" >> $OUT 2>&1
export DTHIP_SORT_PATH=2 DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=${1:-64}
echo "DTHIP_SORT_PATH=2 DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=$DTHIP_MSD_BUCKET_ROWS" >> $OUT
for f in tests/test_gpu_parity.py tests/test_reference_suite.py tests/test_frame_golden.py tests/test_gpu_groupwise.py tests/test_gpu_sets_join.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py tests/test_shim_e2e.py; do
  timeout 400 python -m pytest $f -m gpu -q -p no:cacheprovider > /tmp/forced.log 2>&1; rc=$?
  echo "$(basename $f .py) rc=$rc :: $(grep -E 'passed|failed|error' /tmp/forced.log | tail -1)" >> $OUT
  [ $rc -ne 0 ] && grep -E "^(FAILED|ERROR)" /tmp/forced.log | head -8 >> $OUT
done
cat $OUT
