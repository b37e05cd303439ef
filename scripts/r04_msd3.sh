#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd3; rm -rf $OUT; mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_SORT_PATH=1
run DTHIP_SORT_PATH=0
run DTHIP_SORT_PATH=0 DTHIP_MSD_RBMAX=10 DTHIP_MSD_BUCKET_ROWS=4096
run DTHIP_SORT_PATH=0 DTHIP_MSD_RBMAX=10 DTHIP_MSD_BUCKET_ROWS=4096 DTHIP_RP_PREFETCH=0
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|radix_pass|msd_|config"
