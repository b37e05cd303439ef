#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/run4.log; : > $L
summ() { python -c '
import json,sys
for l in sys.stdin:
    if not l.startswith("{"): print(l.strip()[:300]); continue
    d=json.loads(l); k=d["kernels"]
    print("ms/step %.2f | " % d["ms_per_step"] + " ".join("%s=%.2fx%d" % (n.replace("_kernel",""), v["avg_ms"], v["launches"]) for n,v in sorted(k.items(), key=lambda kv:-kv[1]["total_ms"])[:4]))
'; }
run() { echo "== $*" | tee -a $L; env $1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check $2 2>&1 | grep '^{\|rror\|Traceback' | summ | tee -a $L; }
run "DTHIP_BK_DEBUG=2" "--agg-path 0"
run "DTHIP_BK_DEBUG=4" "--agg-path 0"
run "DTHIP_BK_DEBUG=10" "--agg-path 0"
run "DTHIP_BK_DEBUG=14" "--agg-path 0"
