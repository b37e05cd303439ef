#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_uk; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_msd.py tests/test_gpu_sharded.py tests/test_shim_e2e.py tests/test_reference_suite.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest.log
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_FUSE_UKEY=0
run DTHIP_FUSE_UKEY=1
run DTHIP_FUSE_UKEY=1 DTHIP_SORT_PATH=2
run DTHIP_FUSE_UKEY=0
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|msd_|radix_pass|config" | cut -c1-250
