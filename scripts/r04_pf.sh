#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_pf; rm -rf $OUT; mkdir -p $OUT
for pf in 2 1 2 1; do echo "== DTHIP_RP_PREFETCH=$pf" >> $OUT/ab.log; DTHIP_RP_PREFETCH=$pf timeout -k 5 100 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1; done
grep -v amdgpu.ids $OUT/ab.log | grep -E "^==|msd_|\"ms\"" | cut -c1-330
