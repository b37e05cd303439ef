#!/bin/bash
# after compiling the head-marking phase out of the product build: the MSD / sort suites, then C5 and the LSD pass time
export TMPDIR=/tmp
OUT=gpurun_out/r04_regress; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_msd.py tests/test_reference_suite.py -x -q > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"
grep -E "passed|failed" $OUT/pytest1.log | tail -2
DTHIP_SORT_PATH=2 DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=64 timeout -k 5 600 python -m pytest tests/test_gpu_msd.py tests/test_gpu_parity.py -x -q > $OUT/pytest2.log 2>&1; echo "pytest2 (forced MSD) rc=$?"
grep -E "passed|failed" $OUT/pytest2.log | tail -2
timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile > $OUT/c5.log 2>&1
DTHIP_SORT_PATH=1 timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile > $OUT/c5_lsd.log 2>&1
grep -v amdgpu.ids $OUT/c5.log $OUT/c5_lsd.log | grep -E "msd_|radix_pass|config" | cut -c1-420
