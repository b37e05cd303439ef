#!/bin/bash
# end-of-round evidence in one GPU call: the whole GPU suite, the default bench line, the rocprofv3 profiles, the reference's
# own suite on the in-tree binding
export TMPDIR=/tmp
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 2400 bash scripts/prof.sh r05 > $OUT/prof.log 2>&1; tail -45 $OUT/prof.log
cp gpurun_out/prof_r05/pmc_traffic.json profiles/pmc_traffic.json           # (so that the bench line below carries the counters of THIS build)
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
if [ -z "$SKIP_REF" ]; then SKIP_CPU_BASELINE=1 bash scripts/run_ref_suite.sh > $OUT/ref_suite.log 2>&1; tail -12 $OUT/ref_suite.log; fi
