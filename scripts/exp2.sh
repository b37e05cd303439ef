#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/exp2.log
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA 2>&1 | grep '^{\|Error\|error' | python -c '
import json,sys
for l in sys.stdin:
    if not l.startswith("{"): print(l.strip()[:200]); continue
    d=json.loads(l); k=d["kernels"]
    print("ms/step %.2f | " % d["ms_per_step"] + " ".join("%s=%.2f" % (n.replace("_kernel",""), v["avg_ms"]) for n,v in sorted(k.items(), key=lambda kv:-kv[1]["total_ms"])[:8]))
' >> $OUT 2>&1
}
EXTRA="" run DTHIP_RP_VARIANT=0
EXTRA="" run DTHIP_RP_VARIANT=0 DTHIP_RP_DEBUG=2
EXTRA="" run DTHIP_RP_VARIANT=4 DTHIP_RP_DEBUG=2
EXTRA="" run DTHIP_RP_VARIANT=1 DTHIP_RP_DEBUG=2
cat $OUT
