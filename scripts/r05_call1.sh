#!/bin/bash
# round 5, first GPU call: the ADVICE r04 use-after-free shown on the `make uaf` flavour and gone at HEAD, the whole GPU
# suite (both sort routes, Arrow ingestion, the self-verifying 2-rank bench), the default bench line, a first fault hunt.
export TMPDIR=/tmp
OUT=gpurun_out/r05a; rm -rf $OUT; mkdir -p $OUT
( DTHIP_LIB=$PWD/datatable_amd/libdthip_uaf.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k mapped_words > $OUT/uaf_test.log 2>&1; echo "rc=$?" >> $OUT/uaf_test.log )
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 700 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 400 python scripts/fault_hunt.py --settings uaf,head --runs 8 --budget 240 > $OUT/fault_hunt.txt 2>&1
cat $OUT/fault_hunt.txt; tail -3 $OUT/uaf_test.log
