#!/bin/bash
# round 4, GPU call 3: per-level times of the MSD levels (C5), final level with 256- and 512-thread workgroups, LSD for reference
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd2; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_msd.py -x -q > $OUT/pytest_msd.log 2>&1; echo "msd tests rc=$?" | tee -a $OUT/pytest_msd.log
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_SORT_PATH=1
run DTHIP_SORT_PATH=0
run DTHIP_SORT_PATH=0 DTHIP_MSD_FINAL_BLOCK=512
run DTHIP_SORT_PATH=0 DTHIP_MSD_BUCKET_ROWS=4096
run DTHIP_SORT_PATH=0 DTHIP_MSD_BUCKET_ROWS=1024
run DTHIP_SORT_PATH=1
grep -v amdgpu.ids $OUT/ab.log
