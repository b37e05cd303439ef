#!/bin/bash
# round-2 rocprofv3 evidence (run on the GPU box through gpurun): C3 through bench.py (kernel stats + HBM PMC passes)
# and BASELINE configs C2 / C5 / hard-keys C3 through scripts/configs_bench.py.  Summaries land in gpurun_out/prof_r02*/;
# the ones to keep are copied into profiles/ by hand.
TAG=${1:-r02}
export TMPDIR=/tmp
REPO=$PWD
bash scripts/prof.sh $TAG --configs '' --host-rows 0 > gpurun_out/prof_$TAG.log 2>&1
python scripts/make_pmc_json.py gpurun_out/prof_$TAG/pmc_fetch.txt gpurun_out/prof_$TAG/pmc_write.txt 1000000000 gpurun_out/prof_$TAG/pmc_traffic.json > gpurun_out/prof_$TAG/pmc_traffic.txt 2>&1
for C in 2 5 6; do
  OUT=$REPO/gpurun_out/prof_${TAG}_c$C; mkdir -p $OUT
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o cfg -- python $REPO/scripts/configs_bench.py --configs $C --reps 3 > $OUT/stats.log 2>&1 )
  ( cd /tmp; timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o cfg -- python $REPO/scripts/configs_bench.py --configs $C --reps 1 > $OUT/fetch.log 2>&1 )
  ( cd /tmp; timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o cfg -- python $REPO/scripts/configs_bench.py --configs $C --reps 1 > $OUT/write.log 2>&1 )
  for d in stats pmc_fetch pmc_write; do
    db=$(find $OUT/$d -name "*.db" | head -1)
    [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/$d.txt 2>&1
  done
  tail -2 $OUT/stats.log; head -14 $OUT/stats.txt
done
tail -20 gpurun_out/prof_$TAG.log
