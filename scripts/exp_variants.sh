#!/bin/bash
# kernel-variant timing experiments for the radix pass (run through gpurun)
mkdir -p gpurun_out
OUT=gpurun_out/exp_variants.log
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA 2>&1 | grep '^{' | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); k=d["kernels"]
    print("ms/step %.2f | " % d["ms_per_step"] + " ".join("%s=%.2f" % (n.replace("_kernel",""), v["avg_ms"]) for n,v in sorted(k.items(), key=lambda kv:-kv[1]["total_ms"])[:8]))
' >> $OUT 2>&1
}
for V in 0 1 2 3 4 5; do EXTRA="" run DTHIP_RP_VARIANT=$V; done
for V in 0 1 3; do EXTRA="--no-check" run DTHIP_RP_VARIANT=$V DTHIP_RP_DEBUG=1; done
cat $OUT
