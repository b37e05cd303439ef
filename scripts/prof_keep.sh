#!/bin/bash
# copy the summaries of gpurun_out/prof_<tag>/ (scripts/prof.sh) that are to be judged into profiles/
TAG=${1:?round tag}; S=gpurun_out/prof_$TAG
cp $S/stats.txt profiles/${TAG}_kernel_stats.txt
cp $S/fetch.txt profiles/${TAG}_pmc_fetch.txt; cp $S/write.txt profiles/${TAG}_pmc_write.txt
cp $S/pmc_traffic.txt profiles/${TAG}_pmc_traffic.txt; cp $S/pmc_traffic.json profiles/pmc_traffic.json
for c in C2 C4 C5 C3_hard C3_sgrp; do
  l=$(echo $c | tr A-Z a-z)
  [ -f $S/${c}_stats.txt ] && cp $S/${c}_stats.txt profiles/${TAG}_${l}_kernel_stats.txt
  [ -f $S/${c}_FETCH_SIZE.txt ] && { echo "== FETCH_SIZE"; cat $S/${c}_FETCH_SIZE.txt; echo "== WRITE_SIZE"; cat $S/${c}_WRITE_SIZE.txt; } > profiles/${TAG}_${l}_pmc.txt
done
[ -f $S/C2_lds.txt ] && cp $S/C2_lds.txt profiles/${TAG}_c2_lds_pmc.txt
ls -la profiles/${TAG}_*
