#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_full; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 5 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_full/bench.json'))
print(d["ms_per_step"], d["roofline"]["frac"], d.get("seconds_total"))
print({k:(round(v.get("ms"),3), v.get("amplification"), v.get("parity",{}).get("ok")) for k,v in d.get("configs",{}).items()})
print({k:(v.get("ms_best"), v.get("over_raw")) for k,v in d["shim_resident"].items() if isinstance(v, dict)})
PY
