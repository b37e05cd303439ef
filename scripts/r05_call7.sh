#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05g; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_rccl_2proc.py tests/test_c_client.py -m gpu -q --maxfail=12 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
