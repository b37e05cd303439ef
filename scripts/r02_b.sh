#!/bin/bash
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_shim_e2e.py tests/test_gpu_fullsize.py -q -m gpu ) > gpurun_out/r02b_tests.log 2>&1
tail -25 gpurun_out/r02b_tests.log
export PYTHONFAULTHANDLER=1
( time timeout 1200 python -X faulthandler bench.py --ref-threads 0,64,16,1 ) > gpurun_out/r02b_bench.log 2>&1
grep '^{' gpurun_out/r02b_bench.log | tail -1 > gpurun_out/r02b_bench.json
tail -40 gpurun_out/r02b_bench.log | cut -c1-3000
