#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd6; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_msd.py -x -q > $OUT/pytest_msd.log 2>&1; echo "msd tests rc=$?" | tee -a $OUT/pytest_msd.log
DTHIP_SORT_PATH=2 DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=64 timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_suite.py -x -q > $OUT/pytest_forced.log 2>&1; echo "forced-MSD parity rc=$?"
tail -2 $OUT/pytest_forced.log
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_SORT_PATH=1
run DTHIP_SORT_PATH=2
run DTHIP_SORT_PATH=2 DTHIP_RP_PREFETCH=1
run DTHIP_SORT_PATH=2 DTHIP_MSD_WINDOWS=0
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|msd_|radix_pass|config" | cut -c1-250
