#!/bin/bash
# integration/build_dt_hip.sh -- the IN-TREE bindings, built: the reference (h2oai/datatable, /root/reference) with
# integration/patches/*.patch applied --
#   s_grp_sort_cc.patch              S-grp: `group()` (src/core/sort.cc:1411) hands fixed-width key columns to dthip_groupby
#   s_red_fexpr_reduce_unary_cc.patch  S-red: FExpr_ReduceUnary::evaluate_n (expr/fexpr_reduce_unary.cc:32) computes the
#                                    sum / mean / min / max / count columns with dthip_reduce ON THE DEVICE: the grouped
#                                    view is peeled into stored column + RowIndex (no CPU gather), and the RowIndex /
#                                    offsets S-grp produced a moment ago are still in HBM (dthip_seam.h)
#   s_red_view_peel.patch            the accessor that peeling needs: Column::view_rowindex32() (column.h, column_impl.h,
#                                    column/view.h: ArrayView_ColumnImpl<int32_t> hands out its RowIndex)
# -- when DTHIP_LIB points at libdthip.so (dlopen: no link dependency; everything the library does not take falls through
# to the reference's own code).  Every `by()`, `sort()`, `Frame.sort`, `unique`, set
# function, `Frame.key = ...` and join of the patched build then runs its grouping on the MI355X -- and the reference's OWN
# test-suite becomes a parity suite of the HIP path (scripts/run_ref_suite.sh, summary under profiles/).
#
# Like oracle/build_ref.sh: the reference tree is copied to a scratch directory OUTSIDE the repo (the build of
# oracle/_ref is reused when it is there: one translation unit is recompiled), built with the reference's own driver, and
# only OUTPUTS land under integration/_dt_hip/ (git-ignored, travels to the GPU box):
#     integration/_dt_hip/datatable/            the package as sourceless .pyc + the stripped .so
#     integration/_dt_hip/ref/tests/            the reference's own test files the suite run needs (the GPU box has no
#                                               /root/reference); never committed
#   usage: integration/build_dt_hip.sh [--force]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
SRC="${DT_REF_SRC:-/root/reference}"
BASE="${DT_REF_WORK:-/tmp/dt_ref_build}"
WORK="${DT_HIP_WORK:-/tmp/dt_hip_build}"
OUT="$HERE/_dt_hip"
FORCE=0; [ "${1:-}" = "--force" ] && FORCE=1
if [ ! -d "$SRC/src/core" ]; then
  echo "build_dt_hip: $SRC not present (GPU box?) -- using the prebuilt integration/_dt_hip as is" >&2; exit 0
fi
PSUM="$(cat "$HERE"/patches/*.patch | sha256sum | cut -c1-16)"
if ls "$OUT"/datatable/lib/_datatable*.so >/dev/null 2>&1 && [ $FORCE -eq 0 ] && [ -f "$OUT/PROVENANCE.txt" ] \
   && grep -q "patches/\*.patch ($PSUM)" "$OUT/PROVENANCE.txt"; then
  echo "build_dt_hip: integration/_dt_hip already built (use --force to rebuild)"; exit 0
fi
rm -rf "$WORK"; mkdir -p "$WORK"
if [ -d "$BASE/build" ]; then cp -r "$BASE"/. "$WORK"/; else cp -r "$SRC"/. "$WORK"/; fi
chmod -R u+w "$WORK"
# (the unpatched files, whatever the base held; files a patch ADDS are removed)
for f in $(grep -h '^+++ b/' "$HERE"/patches/*.patch | sed 's#^+++ b/##; s#\t.*##'); do
  if [ -f "$SRC/$f" ]; then cp "$SRC/$f" "$WORK/$f"; else rm -f "$WORK/$f"; fi
done
for pf in "$HERE"/patches/*.patch; do ( cd "$WORK" && patch -p1 --no-backup-if-mismatch < "$pf" ); done
( cd "$WORK" && python ci/ext.py build > "$WORK/build_hip.log" 2>&1 ) || { tail -30 "$WORK/build_hip.log"; exit 1; }
SO=$(ls "$WORK"/src/datatable/lib/_datatable*.so | head -1)
rm -rf "$OUT"; mkdir -p "$OUT"
(cd "$WORK/src" && find datatable -name __pycache__ -prune -o -type f -name '*.py' -print0 | xargs -0 -I{} cp --parents {} "$OUT"/)
python -m compileall -b -q "$OUT/datatable" > /dev/null
find "$OUT/datatable" -name '*.py' -delete
find "$OUT/datatable" -name __pycache__ -prune -exec rm -rf {} + 2>/dev/null || true
cp "$SO" "$OUT/datatable/lib/"; strip --strip-unneeded "$OUT/datatable/lib/$(basename "$SO")"
# the reference's own tests that exercise group(): grouping, sorting, reducers, cumulative operators, keys, joins, sets
mkdir -p "$OUT/ref/tests"
cp "$SRC/tests/__init__.py" "$SRC/tests/conftest.py" "$SRC/tests/test-groups.py" "$SRC/tests/test-reduce.py" "$SRC/tests/test-keys.py" \
   "$SRC/tests/test-sets.py" "$SRC/tests/test-join.py" "$OUT/ref/tests/"
cp -r "$SRC/tests/ijby" "$SRC/tests/dt" "$OUT/ref/tests/"
{
  echo "reference: $SRC + integration/patches/*.patch ($PSUM)"
  echo "built: $(date -u +%Y-%m-%dT%H:%M:%SZ) by integration/build_dt_hip.sh"
} > "$OUT/PROVENANCE.txt"
PYTHONPATH="$OUT" python -c "import datatable as dt; print('build_dt_hip: integration/_dt_hip ok, datatable', dt.__version__)"
du -sh "$OUT" | sed 's/^/build_dt_hip: /'
