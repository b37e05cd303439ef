#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_small; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_table or guessed_na_free" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"
grep -v amdgpu.ids $OUT/pytest1.log | tail -15
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tilelocal.py tests/test_gpu_sharded.py tests/test_shim_e2e.py tests/test_reference_suite.py tests/test_gpu_msd.py tests/test_gpu_groupwise.py -x -q > $OUT/pytest2.log 2>&1; echo "pytest2 rc=$?"
grep -E "passed|failed|error" $OUT/pytest2.log | tail -3
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 1 --reps 9 --profile >> $OUT/ab.log 2>&1; }
for i in 1 2 3; do
run DTHIP_SMALL_PATH=0
run DTHIP_SMALL_PATH=1
run DTHIP_SMALL_PATH=2
done
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|config" | cut -c1-160
grep -v amdgpu.ids $OUT/ab.log | grep -E "small_groups" | head -2 | cut -c1-300
