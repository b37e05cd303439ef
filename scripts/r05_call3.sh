#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05c; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_filter_rows.py tests/test_gpu_arrow.py tests/test_gpu_rccl_2proc.py tests/test_jay.py "tests/test_gpu_parity.py::test_small_path_mapped_words_survive_read_back" -m gpu -q --maxfail=12 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -40 $OUT/pytest.log
DTHIP_TL_LEVEL2=0 timeout 300 python -m pytest tests/test_gpu_filter_rows.py -m gpu -q --maxfail=5 > $OUT/pytest_tl2off.log 2>&1; echo "rc=$?" >> $OUT/pytest_tl2off.log; tail -5 $OUT/pytest_tl2off.log
( DTHIP_LIB=$PWD/datatable_amd/libdthip_uaf.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k mapped_words > $OUT/uaf_test.log 2>&1; echo "rc=$?" >> $OUT/uaf_test.log ); tail -12 $OUT/uaf_test.log
for T in 1 0; do
DTHIP_TL_LEVEL2=$T DTHIP_MSD_DEBUG=1 timeout 600 python bench.py --steps 5 --configs C5 --no-cpu-baseline --no-dist-1rank --no-shim-resident --host-rows 0 --no-full-parity > $OUT/bench_c5_$T.json 2> $OUT/bench_c5_$T.err; echo "bench tl2=$T rc=$?"
grep -a "dthip fused" $OUT/bench_c5_$T.err | head -1
python - $OUT/bench_c5_$T.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["configs"]["C5"]
    print("C5 one call %.2f ms" % c["ms"], c["kernel_ms"], (c.get("parity") or {}).get("ok"))
    print("C5 two calls %.2f ms" % c["two_calls"]["ms"])
except Exception as e:
    print("no bench line:", e)
PY
tail -3 $OUT/bench_c5_$T.err
done
