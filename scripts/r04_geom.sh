#!/bin/bash
# round 4: radix-pass tile geometries without register spills (A/B on C5, one box) + the filter fix
export TMPDIR=/tmp
OUT=gpurun_out/r04_geom; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_suite.py -x -q -k "filter or rows or rowindex or take or config5" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
run() { echo "== $*" | tee -a $OUT/ab.log; env "$@" timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; }
run A=base
for v in b384i20 b384i16 b256i32; do
  DTHIP_LIB=datatable_amd/libdthip_$v.so timeout 300 python -m pytest tests/test_gpu_msd.py tests/test_gpu_parity.py -x -q -k "msd or rows or golden_groupby" > $OUT/pytest_$v.log 2>&1; echo "$v pytest rc=$?"
  run DTHIP_LIB=datatable_amd/libdthip_$v.so
done
run A=base
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|radix_pass|config" | cut -c1-200
