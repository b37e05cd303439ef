#!/bin/bash
# round 4, GPU call 2: MSD levels -- the new tests, the whole GPU suite with the levels forced on every sort
# (tiny inputs included), then C5 timing: LSD passes vs MSD levels
export TMPDIR=/tmp
OUT=gpurun_out/r04_msd1; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_msd.py -x -q > $OUT/pytest_msd.log 2>&1; echo "msd tests rc=$?" | tee -a $OUT/pytest_msd.log
tail -15 $OUT/pytest_msd.log
DTHIP_SORT_PATH=2 DTHIP_MSD_MIN_ROWS=1 DTHIP_MSD_BUCKET_ROWS=64 timeout -k 5 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_msd.py > $OUT/pytest_forced.log 2>&1; echo "forced-MSD suite rc=$?" | tee -a $OUT/pytest_forced.log
tail -5 $OUT/pytest_forced.log
for SP in 1 0; do
  echo "== DTHIP_SORT_PATH=$SP" | tee -a $OUT/ab.log
  DTHIP_SORT_PATH=$SP timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile >> $OUT/ab.log 2>&1
done
grep -v amdgpu.ids $OUT/ab.log
