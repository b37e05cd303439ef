#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_bench1; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_rccl_2proc.py tests/test_gpu_sharded.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout -k 5 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -5 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench1/bench.json'))
print(d["ms_per_step"], d["roofline"]["frac"], d.get("seconds_total"))
print(json.dumps(d.get("shim_resident"), indent=1)[:2500])
print({k:(v.get("ms"), v.get("traffic"), v.get("parity",{}).get("ok")) for k,v in d.get("configs",{}).items()})
PY
