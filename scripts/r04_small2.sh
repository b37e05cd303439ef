#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_small2; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tilelocal.py tests/test_reference_suite.py tests/test_shim_e2e.py tests/test_gpu_sharded.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
for i in 1 2 3; do timeout -k 5 300 python scripts/configs_bench.py --configs 1 --reps 9 --profile >> $OUT/ab.log 2>&1; done
grep -v amdgpu.ids $OUT/ab.log | grep -E "config|bucket_plan" | cut -c1-260
