#!/bin/bash
# rocprofv3 kernel stats for the group-wise / set / join operators (scripts/groupwise_bench.py); $1 = tag
TAG=${1:-r01e}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_gw_$TAG
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o gw -- python $REPO/scripts/groupwise_bench.py --rows 1e8 --groups 1e5 --reps 3 > $OUT/gw.log 2>&1
cd $REPO
db=$(find $OUT/stats -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/stats.txt 2>&1
cat $OUT/gw.log | tail -16
head -40 $OUT/stats.txt
