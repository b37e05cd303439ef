#!/bin/bash
# Guard-page evidence for the code of round 5 (tile-local sort levels + records, fused filter -> by, Arrow ingestion, the rows
# exchange with 2 all-gathers): the suites that drive it, in modes 1 (over-runs) and 2 (under-runs), and config 5 at full
# size as one call and as two.   -> gpurun_out/guard_r05.txt
export TMPDIR=/tmp
OUT=gpurun_out/guard_r05.txt; mkdir -p gpurun_out; : > $OUT
python -c "from datatable_amd import _lib as L; print('library build', L.load().dthip_build_id().decode())" >> $OUT 2>&1
FILES="tests/test_gpu_filter_rows.py tests/test_gpu_arrow.py tests/test_gpu_msd.py tests/test_gpu_sharded.py tests/test_gpu_parity.py"
for MODE in 1 2; do
  GUARD_TIMEOUT=400 bash scripts/guard_check.sh $MODE $FILES > /dev/null 2>&1
  echo "--- DTHIP_GUARD=$MODE, suites" >> $OUT; cat gpurun_out/guard$MODE/summary.txt >> $OUT
  echo "--- DTHIP_GUARD=$MODE, config 5 at 1e9 rows (C5f = one fused call, C5 = two calls)" >> $OUT
  DTHIP_GUARD=$MODE timeout 400 python scripts/guard_fullsize.py --configs C5f,C5 --guard $MODE 2>&1 | grep -E "OK|guard\]|fault|Error|error" | tail -8 >> $OUT
done
cat $OUT
