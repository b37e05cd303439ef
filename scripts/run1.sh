#!/bin/bash
# first GPU run of the bucketed aggregation: parity tests, then bench at 1e9 on both paths
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r1.log 2>&1
tail -15 gpurun_out/pytest_r1.log
summ() { python -c '
import json,sys
for l in sys.stdin:
    if not l.startswith("{"): print(l.strip()[:300]); continue
    d=json.loads(l); k=d["kernels"]
    print("ms/step %.2f | " % d["ms_per_step"] + " ".join("%s=%.2fx%d" % (n.replace("_kernel",""), v["avg_ms"], v["launches"]) for n,v in sorted(k.items(), key=lambda kv:-kv[1]["total_ms"])[:10]))
'; }
for P in 2 1; do
  echo "== agg_path $P" | tee -a gpurun_out/bench_r1.log
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --agg-path $P 2>&1 | grep '^{\|rror\|Traceback' | tee -a gpurun_out/bench_r1.json | summ | tee -a gpurun_out/bench_r1.log
done
for V in 1 2 3; do
  echo "== agg_path 2 variant $V" | tee -a gpurun_out/bench_r1.log
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --agg-path 2 --bucket-variant $V 2>&1 | grep '^{\|rror\|Traceback' | tee -a gpurun_out/bench_r1.json | summ | tee -a gpurun_out/bench_r1.log
done
