#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r04_last; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_table or guessed_na_free or speculative" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"
grep -v amdgpu.ids $OUT/pytest1.log | tail -4
( rocm-smi --showmemorypartition --showcomputepartition 2>&1; rocm-smi --showclocks --showpower --showperflevel 2>&1; rocm-smi --showmeminfo vram 2>&1 ) > $OUT/smi.txt
grep -v "^$\|====" $OUT/smi.txt | head -40
timeout -k 5 300 python scripts/configs_bench.py --configs 5 --reps 3 --profile > $OUT/c5.log 2>&1
grep -v amdgpu.ids $OUT/c5.log | grep -E "msd_|config" | cut -c1-400
