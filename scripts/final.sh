#!/bin/bash
# end-of-round evidence in one GPU call: the whole GPU suite, the default bench line, the rocprofv3 profiles
export TMPDIR=/tmp
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 2400 bash scripts/prof.sh r05 > $OUT/prof.log 2>&1; tail -40 $OUT/prof.log
