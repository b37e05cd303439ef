#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd8; rm -rf $OUT; mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
E=datatable_amd/libdthip_exp.so
run DTHIP_SORT_PATH=1
run DTHIP_LIB=$E DTHIP_SORT_PATH=2 DTHIP_FUSE_UKEY=0
run DTHIP_LIB=$E DTHIP_SORT_PATH=2 DTHIP_FUSE_UKEY=0 DTHIP_MSD_WIN_NOR2=1
run DTHIP_LIB=$E DTHIP_SORT_PATH=2 DTHIP_FUSE_UKEY=0 DTHIP_MSD_FAKE_WINDOW=6080
run DTHIP_LIB=$E DTHIP_SORT_PATH=2 DTHIP_FUSE_UKEY=0 DTHIP_MSD_FAKE_WINDOW=6081
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|msd_|radix_pass" | cut -c1-150
