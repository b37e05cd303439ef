#!/bin/bash
# AddressSanitizer flavour of libdthip (make -C datatable_amd/csrc asan) on the GPU: host + device instrumentation,
# xnack on.  usage: scripts/asan_check.sh   -> gpurun_out/asan.log
REPO=$PWD
RT=$(/opt/rocm/bin/hipcc --print-file-name=libclang_rt.asan-x86_64.so)
cat > /tmp/asan_probe.py <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from datatable_amd.engine import Context
ctx = Context(0)
rng = np.random.default_rng(0)
for n in (1000, 300_000, 20_000_000):
    k = rng.integers(0, max(n // 100, 7), n).astype(np.int64); v = rng.standard_normal(n)
    r = ctx.groupby_agg([k], [v], [("sum", 0), ("count0", None)]); assert int(r.agg(1).sum()) == n; r.free()
    g = ctx.groupby([k]); assert len(g.rowindex()) == n; g.free()
    r = ctx.groupby_rows([k], [k, v]); assert len(r.col(1)) == n; r.free()
    print("asan probe n=%d ok" % n, flush=True)
ctx.close()
PY
( cd $REPO && HSA_XNACK=1 LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 DTHIP_LIB=$REPO/datatable_amd/libdthip_asan.so timeout -k 5 ${ASAN_TIMEOUT:-240} python /tmp/asan_probe.py ) > $REPO/gpurun_out/asan.log 2>&1
echo "asan rc=$?" >> $REPO/gpurun_out/asan.log
tail -30 $REPO/gpurun_out/asan.log
