#!/bin/bash
# round 2, first GPU call: new tests + the new bench legs (reference thread sweep)
mkdir -p gpurun_out
nproc > gpurun_out/r02a_host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r02a_host.txt; free -g | head -2 >> gpurun_out/r02a_host.txt
( time timeout 900 python -m pytest tests/test_shim_e2e.py tests/test_gpu_fullsize.py -x -q -m gpu ) > gpurun_out/r02a_tests.log 2>&1
tail -15 gpurun_out/r02a_tests.log
( time timeout 1200 python bench.py --ref-threads 0,64,16,1 ) > gpurun_out/r02a_bench.log 2>&1
grep '^{' gpurun_out/r02a_bench.log | tail -1 > gpurun_out/r02a_bench.json
tail -5 gpurun_out/r02a_bench.log | cut -c1-3000
