#!/bin/bash
# round 5, second GPU call: the fused filter -> group-rows route (tlsort.hip) against the oracle, then config 5 timed both ways
export TMPDIR=/tmp
OUT=gpurun_out/r05b; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_filter_rows.py tests/test_gpu_arrow.py tests/test_gpu_rccl_2proc.py tests/test_jay.py "tests/test_gpu_parity.py::test_small_path_mapped_words_survive_read_back" -m gpu -q --maxfail=12 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
DTHIP_MSD_DEBUG=1 timeout 600 python bench.py --steps 5 --configs C5 --no-cpu-baseline --no-dist-1rank --no-shim-resident --host-rows 0 --no-full-parity > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench rc=$?"
grep -a "dthip fused\|dthip msd" $OUT/bench_c5.err | head -5
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05b/bench_c5.json"))
    c = d["configs"]["C5"]
    print("C5 one call %.2f ms" % c["ms"], c["kernel_ms"], c.get("parity"))
    print("C5 two calls %.2f ms" % c["two_calls"]["ms"], c["two_calls"]["kernel_ms"])
except Exception as e:
    print("no bench line:", e)
PY
tail -5 $OUT/bench_c5.err
