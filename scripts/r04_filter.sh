#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_filter; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests -m gpu -x -q -k "filter or compact or rowindex or bool or msd or shim or frame or reference_suite" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -4 $OUT/pytest.log
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run DTHIP_FILTER_PATH=1
run DTHIP_FILTER_PATH=0
run DTHIP_FILTER_PATH=1
run DTHIP_FILTER_PATH=0
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|compact|config"
