#!/bin/bash
# `python bench.py --gpus 2` -- the command the driver issues for N > 1 -- on a ONE-GPU box: both ranks on device 0, the nine RCCL
# entry points served by tests/cpp/fake_rccl.cpp (shared memory; RCCL itself refuses two ranks on one device).  Everything above
# that call boundary is the product: self-started ranks, gloo control plane, sharded C3 / C4 / C5, and (round 5) the line's
# self-verification: parity.configs, cpu_baseline, ranks_seen.  The JSON line -> gpurun_out/bench_2ranks.json
export TMPDIR=/tmp
ROWS=${1:-200000000}; NGRP=${2:-2000000}; SCALE=${3:-0.2}
D=$(mktemp -d /tmp/fakerccl.XXXX)
/opt/rocm/bin/hipcc -O2 -shared -fPIC tests/cpp/fake_rccl.cpp -o $D/libfakerccl.so -lpthread || exit 1
mkdir -p gpurun_out
env -u RANK -u WORLD_SIZE -u LOCAL_RANK -u MASTER_ADDR -u MASTER_PORT DTHIP_RCCL_LIB=$D/libfakerccl.so DTHIP_BENCH_ONE_GPU=1 FAKE_RCCL_DIR=$D FAKE_RCCL_OUTBOX_MB=1024 \
  timeout 900 python bench.py --gpus 2 --rows $ROWS --groups $NGRP --steps 3 --warmup 1 --config-scale $SCALE > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
echo "rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_2ranks.json"))
    print("n_gpus", d["n_gpus"], "value %.3g rows/s" % d["value"], "ms/step %.2f" % d["ms_per_step"])
    print("parity.configs", d["parity"]["configs"], "ranks_seen", d.get("ranks_seen"), "cpu_baseline", (d.get("cpu_baseline") or {}).get("kind"), (d.get("cpu_baseline") or {}).get("value"))
    print("allgathers C3/C4/C5", d["exchange"]["allgathers"], d["configs"]["C4"]["allgathers"], d["configs"]["C5"]["allgathers"])
    print("C5 parity", {k: v for k, v in d["configs"]["C5"]["parity"].items() if k != "single_gpu_vs_oracle_all_rows"})
except Exception as e:
    print("no line:", e)
PY
tail -3 gpurun_out/bench_2ranks.err
