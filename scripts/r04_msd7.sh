#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd7; rm -rf $OUT; mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_SORT_PATH=1
run DTHIP_LIB=datatable_amd/libdthip_exp.so DTHIP_SORT_PATH=2
run DTHIP_LIB=datatable_amd/libdthip_exp.so DTHIP_SORT_PATH=2 DTHIP_MSD_R1ONLY=1
run DTHIP_LIB=datatable_amd/libdthip_exp.so DTHIP_SORT_PATH=2 DTHIP_MSD_R1ONLY=1 DTHIP_RP_PREFETCH=0
run DTHIP_LIB=datatable_amd/libdthip_exp.so DTHIP_SORT_PATH=2 DTHIP_RP_PREFETCH=0
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|msd_|radix_pass" | cut -c1-150
