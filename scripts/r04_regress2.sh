#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_regress2; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_gpu_msd.py -x -q > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"
grep -E "passed|failed|fault" $OUT/pytest1.log | tail -2
timeout -k 5 200 python scripts/configs_bench.py --configs 5 --reps 3 --profile > $OUT/c5.log 2>&1; echo "c5 rc=$?"
grep -v amdgpu.ids $OUT/c5.log | grep -E "msd_|radix_pass|config|fault" | cut -c1-420
