#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd11; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_msd.py -x -q > $OUT/pytest_msd.log 2>&1; echo "msd tests rc=$?"
DTHIP_SORT_PATH=2 DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=64 timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_suite.py -x -q > $OUT/pytest_forced.log 2>&1; echo "forced-MSD rc=$?"; tail -1 $OUT/pytest_forced.log
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
E=datatable_amd/libdthip_exp.so
run DTHIP_SORT_PATH=1
run DTHIP_SORT_PATH=2
run DTHIP_SORT_PATH=2 DTHIP_FUSE_UKEY=0
run DTHIP_LIB=$E DTHIP_SORT_PATH=2 DTHIP_FUSE_UKEY=0 DTHIP_MSD_R1ONLY=1
run DTHIP_SORT_PATH=2 DTHIP_MSD_WINDOWS=0
run DTHIP_SORT_PATH=1
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|msd_|radix_pass|config" | cut -c1-140
