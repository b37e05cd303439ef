#!/bin/bash
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/tcc_counters.txt
for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_WRITE_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/$tag -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $OUT/$tag.log 2>&1
  db=$(find $OUT/$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/scripts/rocpd_summary.py $db | grep -A30 "counter" | grep -i "bucket\|table_agg\|counter" 
done
