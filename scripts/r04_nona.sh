#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_nona; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -k "guessed_na_free" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"
tail -15 $OUT/pytest1.log
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tilelocal.py tests/test_gpu_sharded.py tests/test_shim_e2e.py -x -q > $OUT/pytest2.log 2>&1; echo "pytest2 rc=$?"
tail -5 $OUT/pytest2.log
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 1,2,3,4 --reps 5 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_NONA_GUESS=0
run DTHIP_NONA_GUESS=1
run DTHIP_NONA_GUESS=0
run DTHIP_NONA_GUESS=1
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|table_agg|value_na|config" | cut -c1-250
