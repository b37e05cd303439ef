#!/bin/bash
# A/B of one environment switch on config 5 inside ONE gpurun call (boxes differ by ~5 %): scripts/c5_ab.sh VAR value [value ...]
#   scripts/c5_ab.sh DTHIP_TL_LEVEL2 1 0        (tile-local / scattering second level of the fused route)
#   scripts/c5_ab.sh DTHIP_TL_BLOCK 512 1024    (first-level tile size)
# Every run is bench.py's config-5 leg: the one call and the two calls timed, both verified on all rows (profiles/r05_c5_ab.txt).
export TMPDIR=/tmp
VAR=${1:?environment variable}; shift
OUT=gpurun_out/c5_ab; mkdir -p $OUT
for V in "$@"; do
  env $VAR=$V DTHIP_MSD_DEBUG=1 timeout 600 python bench.py --steps 5 --configs C5 --no-cpu-baseline --no-dist-1rank --no-shim-resident --host-rows 0 --no-full-parity \
      > $OUT/bench_${VAR}_$V.json 2> $OUT/bench_${VAR}_$V.err; echo "$VAR=$V rc=$?"
  python - $OUT/bench_${VAR}_$V.json <<'PY'
import json, sys
try:
    c = json.load(open(sys.argv[1]))["configs"]["C5"]
    print("  one call %.2f ms  two calls %.2f ms  parity %s" % (c["ms"], c["two_calls"]["ms"], (c.get("parity") or {}).get("ok")), c["kernel_ms"])
except Exception as e:
    print("  no bench line:", e)
PY
done
