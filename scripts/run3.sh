#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/run3.log; : > $L
summ() { python -c '
import json,sys
for l in sys.stdin:
    if not l.startswith("{"): print(l.strip()[:300]); continue
    d=json.loads(l); k=d["kernels"]
    print("ms/step %.2f | " % d["ms_per_step"] + " ".join("%s=%.2fx%d" % (n.replace("_kernel",""), v["avg_ms"], v["launches"]) for n,v in sorted(k.items(), key=lambda kv:-kv[1]["total_ms"])[:6]))
'; }
run() { echo "== $*" | tee -a $L; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | grep '^{\|rror\|Traceback' | summ | tee -a $L; }
for a in "$@"; do run $a; done
if [ -n "$PYTEST" ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee -a $L; fi
