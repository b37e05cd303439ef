#!/bin/bash
# quick GPU check of the sort / filter routes after a kernel change: their tests, then config 5 timed (scripts/c5_ab.sh)
export TMPDIR=/tmp
OUT=gpurun_out/quick; rm -rf $OUT; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_filter_rows.py tests/test_gpu_msd.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=10 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
fi
bash scripts/c5_ab.sh ${1:-DTHIP_TL_LEVEL2} ${2:-1} $3 $4
