#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd5; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_msd.py -x -q > $OUT/pytest_msd.log 2>&1; echo "msd tests rc=$?" | tee -a $OUT/pytest_msd.log
tail -12 $OUT/pytest_msd.log
DTHIP_MSD_WINDOWS=0 timeout -k 5 600 python -m pytest tests/test_gpu_msd.py -x -q > $OUT/pytest_msd_nowin.log 2>&1; echo "msd tests (per-bucket final level) rc=$?"
for BR in 64 700; do
DTHIP_SORT_PATH=2 DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=$BR timeout -k 5 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_msd.py -k "not shim and not rccl and not sharded and not sets and not jay" > $OUT/pytest_forced_$BR.log 2>&1; echo "forced-MSD suite (bucket rows $BR) rc=$?"
tail -3 $OUT/pytest_forced_$BR.log
done
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_SORT_PATH=1
run DTHIP_SORT_PATH=2
run DTHIP_SORT_PATH=2 DTHIP_MSD_WINDOWS=0
run DTHIP_SORT_PATH=1
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|msd_|radix_pass|config" | cut -c1-250
