#!/bin/bash
# A/B of bucket_variant values on C3 (kernel times only): scripts/ab.sh "0 2" [extra bench args]
mkdir -p gpurun_out
for rep in 1 2; do
for v in $1; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --configs '' --host-rows 0 --bucket-variant $v ${@:2} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernels']
print('variant $v: %.3f ms/step  ' % d['ms_per_step'] + ' '.join('%s=%.3f' % (n.replace('_kernel','').replace('bucket_',''), k[n]['avg_ms']) for n in sorted(k, key=lambda n:-k[n]['total_ms']) if k[n]['total_ms']/d['steps'] > 0.03))
"
done; done
