#!/bin/bash
# round 4, GPU call 4: LSD pass variants on C5 (same box): streaming hints on loads / stores, 1024-thread tiles
export TMPDIR=/tmp
OUT=gpurun_out/r04_ab3; rm -rf $OUT; mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_SORT_PATH=1
run DTHIP_SORT_PATH=1 DTHIP_LIB=datatable_amd/libdthip_nt.so
run DTHIP_SORT_PATH=1 DTHIP_LIB=datatable_amd/libdthip_nts.so
run DTHIP_SORT_PATH=1 DTHIP_LIB=datatable_amd/libdthip_b1024.so
run DTHIP_SORT_PATH=1
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|radix_pass"
