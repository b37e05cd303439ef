#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_shim; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_shim_e2e.py tests/test_integration_shim.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -30 $OUT/pytest.log
