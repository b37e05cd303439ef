#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_heads; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_msd.py -x -q 2>&1 | grep -E "passed|failed" | tail -1
DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=64 timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_suite.py tests/test_frame_golden.py tests/test_gpu_sharded.py tests/test_gpu_groupwise.py tests/test_gpu_sets_join.py -x -q 2>&1 | grep -E "passed|failed" | tail -1
DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=64 DTHIP_MSD_WINDOWS=0 timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_suite.py tests/test_gpu_groupwise.py -x -q 2>&1 | grep -E "passed|failed" | tail -1
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed" | tail -1
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_SORT_PATH=1
run DTHIP_FUSE_HEADS=0
run DTHIP_FUSE_HEADS=1
run DTHIP_SORT_PATH=1
run DTHIP_FUSE_HEADS=1
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|msd_|radix_pass|config" | cut -c1-170
