#!/bin/bash
# Register / scratch / occupancy table of every kernel of one .hip file, at one or two commits -- the check that found
# round 4's silent regression (a phase that never ran made radix_pass_kernel spill 60 instead of 24 VGPRs).
#   scripts/kernel_resources.sh radix.hip                 # working tree
#   scripts/kernel_resources.sh radix.hip bb2b7d6         # that commit against the working tree: only the differences
set -e
F=${1:?file under datatable_amd/csrc}; REV=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/kres.XXXXXX)
stage() {   # $1 = dir, $2 = rev or empty
  mkdir -p $1/datatable_amd/csrc $1/include
  if [ -z "$2" ]; then cp $ROOT/datatable_amd/csrc/*.h* $1/datatable_amd/csrc/; cp $ROOT/include/dthip.h $1/include/
  else for f in $(git -C $ROOT ls-tree --name-only $2 datatable_amd/csrc/ | grep -E '\.(hip|hpp)$') include/dthip.h; do git -C $ROOT show $2:$f > $1/$f; done; fi
  ( cd $1/datatable_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics \
      -Rpass-analysis=kernel-resource-usage -c $F -o /dev/null 2> res.txt || { tail -5 res.txt; exit 1; } )
}
stage $W/new ""
[ -n "$REV" ] && stage $W/old $REV
python3 - $W/new/datatable_amd/csrc/res.txt ${REV:+$W/old/datatable_amd/csrc/res.txt} <<'PY'
import re, subprocess, sys
def parse(f):
    d, cur = {}, None
    for ln in open(f):
        m = re.search(r'Function Name: (\S+)', ln)
        if m: cur = m.group(1); d[cur] = {}; continue
        m = re.search(r'remark:\s+(.+?): (\d+) \[-R', ln)
        if m and cur: d[cur][re.sub(r' \[.*\]', '', m.group(1)).strip()] = int(m.group(2))
    return d
def name(k):
    try: return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', k], capture_output=True, text=True).stdout.strip()[:110]
    except Exception: return k[:110]
COLS = ('VGPRs', 'VGPRs Spill', 'ScratchSize', 'Occupancy', 'LDS Size')
new = parse(sys.argv[1])
if len(sys.argv) == 2:
    for k in sorted(new): print("%-110s %s" % (name(k), " ".join("%s=%s" % (c.replace(' ', ''), new[k].get(c)) for c in COLS)))
else:
    old = parse(sys.argv[2]); nd = 0
    for k in sorted(set(old) | set(new)):
        a, b = old.get(k), new.get(k)
        if a is None or b is None: print("%-8s %s" % ("only new" if a is None else "only old", name(k))); continue
        if any(a.get(c) != b.get(c) for c in COLS):
            nd += 1; print("DIFF     %s  %s" % (name(k), {c: (a.get(c), b.get(c)) for c in COLS if a.get(c) != b.get(c)}))
    print("%d kernels, %d differ" % (len(new), nd))
PY
rm -rf $W
