#!/bin/bash
# round 4, GPU call 1: GPU test suite on the new default (LDS lane-mask ranking), then C5 / hard-keys timing A/B:
# ballot ranking vs lane masks, plus the two timing-only experiment switches (unordered ranks, tile-local writes)
export TMPDIR=/tmp
OUT=gpurun_out/r04_ab1; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
for R in 0 1; do
  echo "== DTHIP_RP_RANK=$R" | tee -a $OUT/ab.log
  DTHIP_RP_RANK=$R timeout -k 5 300 python scripts/configs_bench.py --configs 5,6 --reps 3 --profile >> $OUT/ab.log 2>&1
done
echo "== exp lib, RANK=2 (unordered; wrong results, timing only)" | tee -a $OUT/ab.log
DTHIP_LIB=datatable_amd/libdthip_exp.so DTHIP_RP_RANK=2 timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1
echo "== exp lib, RANK=1 SEQ=1 (tile-local writes; wrong results, timing only)" | tee -a $OUT/ab.log
DTHIP_LIB=datatable_amd/libdthip_exp.so DTHIP_RP_RANK=1 DTHIP_RP_SEQ=1 timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1
echo "== exp lib, RANK=2 SEQ=1" | tee -a $OUT/ab.log
DTHIP_LIB=datatable_amd/libdthip_exp.so DTHIP_RP_RANK=2 DTHIP_RP_SEQ=1 timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1
cat $OUT/ab.log
