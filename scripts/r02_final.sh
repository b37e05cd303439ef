#!/bin/bash
# round-2 evidence at HEAD: full GPU suite, rocprofv3 (stats + HBM PMC) of C3, default bench line
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r02f_fullsuite.log 2>&1
grep -E "passed|failed" gpurun_out/r02f_fullsuite.log | tail -2
bash scripts/prof.sh r02f --configs '' --host-rows 0 > gpurun_out/prof_r02f.log 2>&1
python scripts/make_pmc_json.py gpurun_out/prof_r02f/pmc_fetch.txt gpurun_out/prof_r02f/pmc_write.txt 1000000000 gpurun_out/prof_r02f/pmc_traffic.json > gpurun_out/prof_r02f/pmc_traffic.txt 2>&1
head -12 gpurun_out/prof_r02f/stats.txt; cat gpurun_out/prof_r02f/pmc_traffic.txt | grep -E "partition|seg|transpose|totals"
cp gpurun_out/prof_r02f/pmc_traffic.json profiles/pmc_traffic.json
( time timeout 900 python bench.py ) > gpurun_out/r02f_bench.log 2>&1
grep '^{' gpurun_out/r02f_bench.log | tail -1 > gpurun_out/r02f_bench.json
tail -3 gpurun_out/r02f_bench.log | cut -c1-1500
# C5 kernel stats at HEAD (the sort path changed: 9-bit passes) and the driver's smoke entry point
OUT=$PWD/gpurun_out/prof_r02f_c5; mkdir -p $OUT
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o cfg -- python $OLDPWD/scripts/configs_bench.py --configs 5 --reps 3 > $OUT/stats.log 2>&1 )
db=$(find $OUT/stats -name "*.db" | head -1); [ -n "$db" ] && python scripts/rocpd_summary.py $db > $OUT/stats.txt 2>&1
head -8 $OUT/stats.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
