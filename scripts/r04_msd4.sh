#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd4; rm -rf $OUT; mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_SORT_PATH=1
for W in 8192 7000 6000 5000 4000; do run DTHIP_LIB=datatable_amd/libdthip_exp.so DTHIP_SORT_PATH=2 DTHIP_MSD_FAKE_WINDOW=$W; done
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|radix_pass|msd_" | sed -e 's/compact.*//'
