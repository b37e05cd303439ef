// ops.hip -- result accessors, the S-red family (dthip_reduce / reduce2 / cumulate), set functions, join index, RowIndex
// construction and gather (split out of api.hip in round 6)
#include <algorithm>
#include "host.hpp"

using namespace dthip;

extern "C" {

int64_t dthip_result_ngroups(const dthip_result* r) { return r ? r->ngroups : -1; }
int64_t dthip_result_nrows(const dthip_result* r) { return r ? r->nrows : -1; }
const int32_t* dthip_result_rowindex(const dthip_result* r) { return r ? r->rowindex : nullptr; }
const int32_t* dthip_result_offsets(const dthip_result* r) { return r ? r->offsets : nullptr; }
const void* dthip_result_key(const dthip_result* r, int k) { return (r && k >= 0 && k < r->nkeys) ? r->key[k] : nullptr; }
const void* dthip_result_agg(const dthip_result* r, int a) { return (r && a >= 0 && a < r->naggs) ? r->agg[a] : nullptr; }
const void* dthip_result_col(const dthip_result* r, int c) { return (r && c >= 0 && c < (int)r->col.size()) ? r->col[c] : nullptr; }
int dthip_result_copy_col(dthip_ctx* ctx, const dthip_result* r, int c, void* dst, int mem) {
  if (!ctx || !r || c < 0 || c >= (int)r->col.size()) { set_error("bad column index"); return DTHIP_EINVAL; }
  if (r->nrows == 0) return DTHIP_OK;
  return copy_out(ctx, dst, r->col[c], (size_t)r->nrows * stype_size(r->col_stype[c]), mem);
}
int dthip_result_agg_stype(const dthip_result* r, int a) { return (r && a >= 0 && a < r->naggs) ? r->agg_stype[a] : 0; }

int dthip_result_copy_rowindex(dthip_ctx* ctx, const dthip_result* r, int32_t* dst, int mem) {
  if (!ctx || !r) { set_error("null argument"); return DTHIP_EINVAL; }
  if (r->nrows == 0) return DTHIP_OK;
  if (!r->rowindex) { set_error("result holds no RowIndex (want_rowindex=0 or fused aggregation)"); return DTHIP_EINVAL; }
  return copy_out(ctx, dst, r->rowindex, sizeof(int32_t) * (size_t)r->nrows, mem);
}
int dthip_result_copy_offsets(dthip_ctx* ctx, const dthip_result* r, int32_t* dst, int mem) {
  if (!ctx || !r) { set_error("null argument"); return DTHIP_EINVAL; }
  if (!r->offsets) { set_error("result holds no group offsets (option agg_offsets=0 and no count() requested)"); return DTHIP_EINVAL; }
  return copy_out(ctx, dst, r->offsets, sizeof(int32_t) * (size_t)(r->ngroups + 1), mem);
}
int dthip_result_copy_key(dthip_ctx* ctx, const dthip_result* r, int k, void* dst, int mem) {
  if (!ctx || !r || k < 0 || k >= r->nkeys) { set_error("bad key index"); return DTHIP_EINVAL; }
  if (r->ngroups == 0) return DTHIP_OK;
  if (!r->key[k]) { set_error("result holds no group-key columns (use dthip_result_group_keys)"); return DTHIP_EINVAL; }
  return copy_out(ctx, dst, r->key[k], (size_t)r->ngroups * stype_size(r->key_stype[k]), mem);
}
int dthip_result_copy_agg(dthip_ctx* ctx, const dthip_result* r, int a, void* dst, int mem) {
  if (!ctx || !r || a < 0 || a >= r->naggs) { set_error("bad agg index"); return DTHIP_EINVAL; }
  if (r->ngroups == 0) return DTHIP_OK;
  return copy_out(ctx, dst, r->agg[a], (size_t)r->ngroups * stype_size(r->agg_stype[a]), mem);
}

int dthip_result_group_keys(dthip_ctx* ctx, const dthip_result* r, const dthip_col* key, int mem, void* dst) {
  if (!ctx || !r || !key) { set_error("null argument"); return DTHIP_EINVAL; }
  if (r->ngroups == 0) return DTHIP_OK;
  if (!dst) { set_error("null argument"); return DTHIP_EINVAL; }
  if (!r->rowindex) { set_error("result holds no RowIndex"); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  const int sz = stype_size(key->stype);
  if (!sz) { set_error("unsupported stype %d", key->stype); return DTHIP_ENOTIMPL; }
  Scratch sc(ctx);
  const void* kd = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, key->data, (size_t)r->nrows * sz, mem, &kd));
  int32_t* firstrow = nullptr;
  DTHIP_TRY(sc.get<int32_t>((size_t)r->ngroups, &firstrow));
  DTHIP_TRY(launch_gather(ctx, r->rowindex, DTHIP_INT32, r->offsets, r->ngroups, firstrow));
  if (mem == DTHIP_DEVICE) return launch_gather(ctx, kd, key->stype, firstrow, r->ngroups, dst);
  unsigned char* tmp = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)r->ngroups * sz, &tmp));
  DTHIP_TRY(launch_gather(ctx, kd, key->stype, firstrow, r->ngroups, tmp));
  return copy_out(ctx, dst, tmp, (size_t)r->ngroups * sz, mem);
}

int dthip_result_free(dthip_ctx* ctx, dthip_result* r) {
  if (!ctx) return DTHIP_EINVAL;
  if (r) result_destroy(ctx, r);
  return DTHIP_OK;
}

int dthip_reduce_out_stype(int op, int st) {
  switch (op) {
    case DTHIP_SUM: case DTHIP_PROD: return st == DTHIP_FLOAT32 ? DTHIP_FLOAT32 : st == DTHIP_FLOAT64 ? DTHIP_FLOAT64 : DTHIP_INT64;
    case DTHIP_MEAN: return st == DTHIP_FLOAT32 ? DTHIP_FLOAT32 : DTHIP_FLOAT64;
    case DTHIP_MIN: case DTHIP_MAX: case DTHIP_FIRST: case DTHIP_LAST: return st;
    case DTHIP_SD: case DTHIP_MEDIAN: return st == DTHIP_FLOAT32 ? DTHIP_FLOAT32 : DTHIP_FLOAT64;   // head_reduce_unary.cc:221-229,484-491
    default: return DTHIP_INT64;
  }
}

namespace dthip {

// value column of a reducer on the device: it may be longer than nrows when read through a
// RowIndex (the caller guarantees the indices fit); host staging copies max(index)+1 rows
static int stage_value_col(dthip_ctx* ctx, Scratch& sc, const dthip_col* value, const int32_t* rowindex, int64_t nrows,
                           int mem, const void** d_val) {
  const int sz = stype_size(value->stype);
  if (!sz) { set_error("unsupported stype %d", value->stype); return DTHIP_ENOTIMPL; }
  *d_val = value->data;
  if (mem == DTHIP_HOST) {
    int64_t vrows = nrows;
    if (rowindex) { vrows = 0; for (int64_t i = 0; i < nrows; i++) if (rowindex[i] >= vrows) vrows = (int64_t)rowindex[i] + 1; }
    DTHIP_TRY(stage_in(ctx, sc, value->data, (size_t)vrows * sz, mem, d_val));
  }
  return DTHIP_OK;
}

// head bitmap (1 bit per grouped position) + per-tile "heads before this tile" from the offsets
static int heads_from_offsets(dthip_ctx* ctx, Scratch& sc, const int32_t* d_off, int64_t ngroups, int64_t nrows,
                              unsigned long long** bitmap, uint32_t** tile_counts) {
  DTHIP_TRY(sc.get<unsigned long long>((size_t)((nrows + 63) / 64) + 1, bitmap));
  const uint32_t nt = (uint32_t)((nrows + SEG_TILE - 1) / SEG_TILE);
  DTHIP_TRY(sc.get<uint32_t>((size_t)nt + 4 + nt / 8192, tile_counts));
  return launch_bitmap_from_offsets(ctx, d_off, ngroups, nrows, *bitmap, *tile_counts, *tile_counts + nt);
}

// column as float64 in grouped order (NA -> NaN); a float64 column already in order is used as is
static int grouped_f64(dthip_ctx* ctx, Scratch& sc, const void* d_val, int stype, const int32_t* d_ri, int64_t nrows,
                       const double** out) {
  if (stype == DTHIP_FLOAT64 && !d_ri) { *out = static_cast<const double*>(d_val); return DTHIP_OK; }
  double* t = nullptr;
  DTHIP_TRY(sc.get<double>((size_t)nrows, &t));
  DTHIP_TRY(launch_gather_f64(ctx, d_val, stype, d_ri, nrows, t));
  *out = t;
  return DTHIP_OK;
}

// median / nunique: the distinct (group, value) pairs in (group, value) order with their row counts =
// the fused groupby-aggregate on keys (group id, value) with count().  It takes the sort-free bucketed /
// hash paths when the composite key is dense or has few distinct values, the sort path otherwise.  The
// reference sorts every group separately (Column::sort_grouped, head_reduce_unary.cc:442-444) or fills
// a std::set per group (:379-385).
static int median_nunique(dthip_ctx* ctx, Scratch& sc, int op, const void* d_val, int stype, const int32_t* d_ri,
                          const int32_t* d_off, int64_t ngroups, int64_t nrows, void* d_out) {
  int32_t* gid = nullptr;
  DTHIP_TRY(sc.get<int32_t>((size_t)nrows, &gid));
  DTHIP_TRY(launch_ungroup(ctx, d_off, ngroups, nrows, gid));
  const void* vg = d_val;
  if (d_ri) {
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>((size_t)nrows * stype_size(stype), &t));
    DTHIP_TRY(launch_gather(ctx, d_val, stype, d_ri, nrows, t));
    vg = t;
  }
  dthip_col keys[2] = {{gid, DTHIP_INT32, 0}, {vg, stype, 0}};
  if (stype_is_float(stype) && !ctx->pairs_always) {
    // mostly-distinct values: order the rows by (group, value) and read them through that order
    dthip_result* r1 = nullptr;
    DTHIP_TRY(dthip_groupby(ctx, keys, 2, nrows, DTHIP_NA_FIRST, DTHIP_DEVICE, 1, &r1));
    int rc1;
    if (op == DTHIP_MEDIAN) rc1 = launch_median_sorted(ctx, vg, stype, r1->rowindex, d_off, ngroups, d_out);
    else rc1 = launch_nunique_sorted(ctx, vg, stype, gid, r1->rowindex, r1->offsets, r1->ngroups, ngroups, static_cast<int64_t*>(d_out));
    result_destroy(ctx, r1);
    return rc1;
  }
  const dthip_agg cnt{DTHIP_COUNT0, -1};
  dthip_result* r2 = nullptr;
  const int saved_off = ctx->agg_offsets;
  ctx->agg_offsets = 1;                                         // the pairs' row offsets are needed (median)
  int rc = dthip_groupby_agg(ctx, keys, 2, nullptr, 0, &cnt, 1, nrows, DTHIP_NA_FIRST, DTHIP_DEVICE, &r2);
  ctx->agg_offsets = saved_off;
  if (rc != DTHIP_OK) return rc;
  if (op == DTHIP_MEDIAN) rc = launch_median(ctx, r2->key[1], stype, r2->offsets, r2->ngroups, d_off, ngroups, d_out);
  else rc = launch_nunique(ctx, r2->key[1], stype, static_cast<const int32_t*>(r2->key[0]), r2->ngroups, ngroups, static_cast<int64_t*>(d_out));
  result_destroy(ctx, r2);
  return rc;
}

}  // namespace dthip

int dthip_reduce(dthip_ctx* ctx, int op, const dthip_col* value, const int32_t* rowindex, const int32_t* offsets,
                 int64_t ngroups, int64_t nrows, int mem, void* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (ngroups < 0 || ngroups > nrows) { set_error("ngroups=%lld inconsistent with nrows=%lld", (long long)ngroups, (long long)nrows); return DTHIP_EINVAL; }
  if (ngroups == 0) return DTHIP_OK;
  if (!offsets || !out) { set_error("null argument"); return DTHIP_EINVAL; }
  if (op < DTHIP_SUM || op > DTHIP_COUNTNA) { set_error("bad reducer op %d", op); return DTHIP_EINVAL; }
  if (op != DTHIP_COUNT0 && (!value || !value->data)) { set_error("reducer needs a value column"); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  const void* d_off = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, offsets, sizeof(int32_t) * (size_t)(ngroups + 1), mem, &d_off));
  const int32_t* off32 = static_cast<const int32_t*>(d_off);
  const int ost = dthip_reduce_out_stype(op, op == DTHIP_COUNT0 ? DTHIP_INT64 : value->stype);
  const size_t obytes = (size_t)ngroups * stype_size(ost);
  void* d_out = out;
  if (mem == DTHIP_HOST) {
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>(obytes, &t));
    d_out = t;
  }
  if (op == DTHIP_COUNT0) {
    DTHIP_TRY(launch_count0(ctx, off32, ngroups, static_cast<int64_t*>(d_out)));
  } else {
    const void* d_ri = nullptr;
    DTHIP_TRY(stage_in(ctx, sc, rowindex, sizeof(int32_t) * (size_t)nrows, mem, &d_ri));
    const int32_t* ri32 = static_cast<const int32_t*>(d_ri);
    const void* d_val = nullptr;
    DTHIP_TRY(stage_value_col(ctx, sc, value, rowindex, nrows, mem, &d_val));
    if (op == DTHIP_FIRST || op == DTHIP_LAST) {
      DTHIP_TRY(launch_firstlast(ctx, d_val, value->stype, ri32, off32, ngroups, op == DTHIP_LAST, d_out));
    } else if (op == DTHIP_MEDIAN || op == DTHIP_NUNIQUE) {
      DTHIP_TRY(median_nunique(ctx, sc, op, d_val, value->stype, ri32, off32, ngroups, nrows, d_out));
    } else if (op == DTHIP_PROD && stype_is_float(value->stype)) {
      DTHIP_TRY(launch_prod_float_seq(ctx, d_val, value->stype, ri32, off32, ngroups, d_out));
    } else {
      unsigned long long* bitmap = nullptr;
      uint32_t* tile_counts = nullptr;
      DTHIP_TRY(heads_from_offsets(ctx, sc, off32, ngroups, nrows, &bitmap, &tile_counts));
      if (op == DTHIP_SD) {
        const double* xg = nullptr;
        DTHIP_TRY(grouped_f64(ctx, sc, d_val, value->stype, ri32, nrows, &xg));
        DTHIP_TRY(launch_moments(ctx, xg, nullptr, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, 0, d_out,
                                 ost == DTHIP_FLOAT32, off32, ngroups));
      } else if (op == DTHIP_PROD) {
        DTHIP_TRY(launch_reduce_prod_int(ctx, d_val, value->stype, ri32, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, d_out));
      } else if (op == DTHIP_COUNTNA) {
        ReduceOuts ro;
        DTHIP_TRY(reduce_outs_for(DTHIP_COUNT, d_out, &ro));
        DTHIP_TRY(launch_reduce(ctx, d_val, value->stype, ri32, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, ro));
        DTHIP_TRY(launch_countna_from_count(ctx, off32, ngroups, static_cast<int64_t*>(d_out)));
      } else {
        ReduceOuts ro;
        DTHIP_TRY(reduce_outs_for(op, d_out, &ro));
        if (op == DTHIP_SUM && value->stype == DTHIP_FLOAT32 && ctx->f32_sum_ref)
          DTHIP_TRY(launch_sum_f32_seq(ctx, d_val, ri32, off32, ngroups, d_out));
        else
          DTHIP_TRY(launch_reduce(ctx, d_val, value->stype, ri32, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, ro));
      }
    }
  }
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, obytes, mem));
  return DTHIP_OK;
}

int dthip_reduce2_out_stype(int stype_a, int stype_b) {
  return (stype_a == DTHIP_FLOAT32 && stype_b == DTHIP_FLOAT32) ? DTHIP_FLOAT32 : DTHIP_FLOAT64;
}

int dthip_reduce2(dthip_ctx* ctx, int op, const dthip_col* a, const dthip_col* b, const int32_t* rowindex,
                  const int32_t* offsets, int64_t ngroups, int64_t nrows, int mem, void* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (ngroups < 0 || ngroups > nrows) { set_error("ngroups=%lld inconsistent with nrows=%lld", (long long)ngroups, (long long)nrows); return DTHIP_EINVAL; }
  if (ngroups == 0) return DTHIP_OK;
  if (!offsets || !out || !a || !b || !a->data || !b->data) { set_error("null argument"); return DTHIP_EINVAL; }
  if (op != DTHIP_COV && op != DTHIP_CORR) { set_error("bad binary reducer op %d", op); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  const void *d_off = nullptr, *d_ri = nullptr, *d_a = nullptr, *d_b = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, offsets, sizeof(int32_t) * (size_t)(ngroups + 1), mem, &d_off));
  DTHIP_TRY(stage_in(ctx, sc, rowindex, sizeof(int32_t) * (size_t)nrows, mem, &d_ri));
  DTHIP_TRY(stage_value_col(ctx, sc, a, rowindex, nrows, mem, &d_a));
  DTHIP_TRY(stage_value_col(ctx, sc, b, rowindex, nrows, mem, &d_b));
  const int ost = dthip_reduce2_out_stype(a->stype, b->stype);
  const size_t obytes = (size_t)ngroups * stype_size(ost);
  void* d_out = out;
  if (mem == DTHIP_HOST) {
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>(obytes, &t));
    d_out = t;
  }
  unsigned long long* bitmap = nullptr;
  uint32_t* tile_counts = nullptr;
  DTHIP_TRY(heads_from_offsets(ctx, sc, static_cast<const int32_t*>(d_off), ngroups, nrows, &bitmap, &tile_counts));
  const double *xg = nullptr, *yg = nullptr;
  DTHIP_TRY(grouped_f64(ctx, sc, d_a, a->stype, static_cast<const int32_t*>(d_ri), nrows, &xg));
  if (d_b == d_a && b->stype == a->stype) yg = xg;
  else DTHIP_TRY(grouped_f64(ctx, sc, d_b, b->stype, static_cast<const int32_t*>(d_ri), nrows, &yg));
  DTHIP_TRY(launch_moments(ctx, xg, yg, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, op == DTHIP_COV ? 1 : 2,
                           d_out, ost == DTHIP_FLOAT32, static_cast<const int32_t*>(d_off), ngroups));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, obytes, mem));
  return DTHIP_OK;
}

int dthip_cumulate_out_stype(int op, int st) {
  if (op == DTHIP_CUMCOUNT || op == DTHIP_NGROUP) return DTHIP_INT64;
  if (op == DTHIP_CUMSUM || op == DTHIP_CUMPROD) return st == DTHIP_FLOAT32 ? DTHIP_FLOAT32 : st == DTHIP_FLOAT64 ? DTHIP_FLOAT64 : DTHIP_INT64;
  return st;
}

int dthip_cumulate(dthip_ctx* ctx, int op, const dthip_col* value, const int32_t* rowindex, const int32_t* offsets,
                   int64_t ngroups, int64_t nrows, int reverse, int mem, void* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (ngroups < 0 || ngroups > nrows) { set_error("ngroups=%lld inconsistent with nrows=%lld", (long long)ngroups, (long long)nrows); return DTHIP_EINVAL; }
  if (nrows == 0) return DTHIP_OK;
  if (!offsets || !out || ngroups == 0) { set_error("null argument"); return DTHIP_EINVAL; }
  if (op < DTHIP_CUMSUM || op > DTHIP_FILLNA) { set_error("bad cumulative op %d", op); return DTHIP_EINVAL; }
  const bool counting = op == DTHIP_CUMCOUNT || op == DTHIP_NGROUP;
  if (!counting && (!value || !value->data)) { set_error("cumulative op needs a value column"); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  const void* d_off = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, offsets, sizeof(int32_t) * (size_t)(ngroups + 1), mem, &d_off));
  const int32_t* off32 = static_cast<const int32_t*>(d_off);
  const int ost = dthip_cumulate_out_stype(op, counting ? DTHIP_INT64 : value->stype);
  const size_t obytes = (size_t)nrows * stype_size(ost);
  if (!obytes) { set_error("unsupported stype"); return DTHIP_ENOTIMPL; }
  void* d_out = out;
  if (mem == DTHIP_HOST) {
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>(obytes, &t));
    d_out = t;
  }
  if (counting) {
    DTHIP_TRY(launch_cumcount(ctx, off32, ngroups, nrows, op == DTHIP_NGROUP, reverse, static_cast<int64_t*>(d_out)));
  } else {
    const void *d_ri = nullptr, *d_val = nullptr;
    DTHIP_TRY(stage_in(ctx, sc, rowindex, sizeof(int32_t) * (size_t)nrows, mem, &d_ri));
    DTHIP_TRY(stage_value_col(ctx, sc, value, rowindex, nrows, mem, &d_val));
    unsigned long long* bitmap = nullptr;
    uint32_t* tile_counts = nullptr;
    DTHIP_TRY(heads_from_offsets(ctx, sc, off32, ngroups, nrows, &bitmap, &tile_counts));
    DTHIP_TRY(launch_cumulate(ctx, d_val, value->stype, static_cast<const int32_t*>(d_ri), reinterpret_cast<const uint8_t*>(bitmap),
                              nrows, op, reverse, d_out, ost));
  }
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, obytes, mem));
  return DTHIP_OK;
}

int dthip_setop(dthip_ctx* ctx, int op, const dthip_col* stacked, const int64_t* cumsizes, int nsources, int64_t nrows,
                int mem, int32_t* out_indices, int64_t* nout) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (!nout) { set_error("null argument"); return DTHIP_EINVAL; }
  *nout = 0;
  if (nrows == 0) return DTHIP_OK;
  if (!stacked || !stacked->data || !cumsizes || !out_indices || nsources < 1) { set_error("null argument"); return DTHIP_EINVAL; }
  if (op < DTHIP_UNION || op > DTHIP_SYMDIFF) { set_error("bad set function %d", op); return DTHIP_EINVAL; }
  if (cumsizes[nsources - 1] != nrows) { set_error("cumsizes[last]=%lld != nrows=%lld", (long long)cumsizes[nsources - 1], (long long)nrows); return DTHIP_EINVAL; }
  // group the stacked column: stable, so row ids ascend inside every group
  dthip_result* g = nullptr;
  DTHIP_TRY(dthip_groupby(ctx, stacked, 1, nrows, DTHIP_NA_FIRST, mem, 1, &g));
  int rc = DTHIP_OK;
  {
    Scratch sc(ctx);
    const int64_t ng = g->ngroups;
    std::vector<int32_t> cum32((size_t)nsources);
    for (int k = 0; k < nsources; k++) cum32[(size_t)k] = (int32_t)cumsizes[k];
    const void* d_cum = nullptr;
    int8_t* mask = nullptr; int32_t *gidx = nullptr, *first = nullptr;
    int32_t* d_out = out_indices;
    int64_t cnt = 0;
    rc = stage_in(ctx, sc, cum32.data(), sizeof(int32_t) * (size_t)nsources, DTHIP_HOST, &d_cum);
    if (rc == DTHIP_OK) rc = sc.get<int8_t>((size_t)ng, &mask);
    if (rc == DTHIP_OK) rc = sc.get<int32_t>((size_t)ng, &gidx);
    if (rc == DTHIP_OK) rc = sc.get<int32_t>((size_t)ng, &first);
    if (rc == DTHIP_OK && mem == DTHIP_HOST) rc = sc.get<int32_t>((size_t)ng, &d_out);
    if (rc == DTHIP_OK) rc = launch_setop_flags(ctx, g->rowindex, g->offsets, ng, op, static_cast<const int32_t*>(d_cum), nsources, mask);
    if (rc == DTHIP_OK) { PredArgs p{mask, DTHIP_BOOL, 0, 0.0, 0, 1}; rc = launch_compact(ctx, p, ng, gidx, &cnt); }
    if (rc == DTHIP_OK && cnt) rc = launch_gather(ctx, g->offsets, DTHIP_INT32, gidx, cnt, first);
    if (rc == DTHIP_OK && cnt) rc = launch_gather(ctx, g->rowindex, DTHIP_INT32, first, cnt, d_out);
    if (rc == DTHIP_OK && cnt && mem == DTHIP_HOST) rc = copy_out(ctx, out_indices, d_out, sizeof(int32_t) * (size_t)cnt, mem);
    if (rc == DTHIP_OK) *nout = cnt;
  }
  result_destroy(ctx, g);
  return rc;
}

int dthip_join_index(dthip_ctx* ctx, const dthip_col* xkeys, const dthip_col* jkeys, int nkeys, int64_t xrows, int64_t jrows,
                     int mem, int32_t* out) {
  DTHIP_TRY(check_common(ctx, xrows, mem));
  DTHIP_TRY(check_common(ctx, jrows, mem));
  if (xrows == 0) return DTHIP_OK;
  if (!xkeys || !jkeys || !out || nkeys < 1 || nkeys > MAX_KEYCOLS) { set_error("bad join arguments (nkeys=%d)", nkeys); return DTHIP_EINVAL; }
  for (int k = 0; k < nkeys; k++) {
    if (!stype_size(xkeys[k].stype) || !stype_size(jkeys[k].stype)) { set_error("join: unsupported key stype"); return DTHIP_ENOTIMPL; }
  }
  Scratch sc(ctx);
  std::vector<dthip_col> xd, jd;
  DTHIP_TRY(stage_cols(ctx, sc, xkeys, nkeys, xrows, mem, &xd));
  DTHIP_TRY(stage_cols(ctx, sc, jkeys, nkeys, jrows, mem, &jd));
  int32_t* d_out = out;
  if (mem == DTHIP_HOST) DTHIP_TRY(sc.get<int32_t>((size_t)xrows, &d_out));
  DTHIP_TRY(launch_join_index(ctx, xd.data(), jd.data(), nkeys, xrows, jrows, d_out));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, sizeof(int32_t) * (size_t)xrows, mem));
  return DTHIP_OK;
}

int dthip_range_bucket(dthip_ctx* ctx, const dthip_col* key, int64_t nrows, const int64_t* bounds, int nbounds, int mem,
                       int8_t* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (nrows == 0) return DTHIP_OK;
  if (!key || !key->data || !out || (nbounds > 0 && !bounds)) { set_error("null argument"); return DTHIP_EINVAL; }
  const int sz = stype_size(key->stype);
  if (!sz) { set_error("unsupported stype %d", key->stype); return DTHIP_ENOTIMPL; }
  Scratch sc(ctx);
  const void* d_key = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, key->data, (size_t)nrows * sz, mem, &d_key));
  int8_t* d_out = out;
  if (mem == DTHIP_HOST) DTHIP_TRY(sc.get<int8_t>((size_t)nrows, &d_out));
  long long b[15];
  for (int j = 0; j < nbounds && j < 15; j++) b[j] = (long long)bounds[j];
  DTHIP_TRY(launch_range_bucket(ctx, d_key, key->stype, nrows, b, nbounds, d_out));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, (size_t)nrows, mem));
  return DTHIP_OK;
}

int dthip_ungroup(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t nrows, int mem, int32_t* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (ngroups < 0 || ngroups > nrows) { set_error("ngroups=%lld inconsistent with nrows=%lld", (long long)ngroups, (long long)nrows); return DTHIP_EINVAL; }
  if (nrows == 0) return DTHIP_OK;
  if (!offsets || !out || ngroups == 0) { set_error("null argument"); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  const void* d_off = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, offsets, sizeof(int32_t) * (size_t)(ngroups + 1), mem, &d_off));
  int32_t* d_out = out;
  if (mem == DTHIP_HOST) DTHIP_TRY(sc.get<int32_t>((size_t)nrows, &d_out));
  DTHIP_TRY(launch_ungroup(ctx, static_cast<const int32_t*>(d_off), ngroups, nrows, d_out));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, sizeof(int32_t) * (size_t)nrows, mem));
  return DTHIP_OK;
}

static int compact_common(dthip_ctx* ctx, const PredArgs& p0, size_t elem, int64_t n, int mem, int32_t* out, int64_t* nout) {
  DTHIP_TRY(check_common(ctx, n, mem));
  if (!nout || (n > 0 && (!p0.data || !out))) { set_error("null argument"); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  PredArgs p = p0;
  DTHIP_TRY(stage_in(ctx, sc, p0.data, (size_t)n * elem, mem, &p.data));
  int32_t* d_out = out;
  if (mem == DTHIP_HOST) DTHIP_TRY(sc.get<int32_t>((size_t)n, &d_out));
  DTHIP_TRY(launch_compact(ctx, p, n, d_out, nout));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, sizeof(int32_t) * (size_t)*nout, mem));
  return DTHIP_OK;
}

int dthip_bool_to_rowindex(dthip_ctx* ctx, const int8_t* mask, int64_t n, int mem, int32_t* out, int64_t* nout) {
  PredArgs p;
  memset(&p, 0, sizeof(p));
  p.data = mask; p.stype = DTHIP_BOOL; p.is_mask = 1;
  return compact_common(ctx, p, 1, n, mem, out, nout);
}

int dthip_filter_cmp(dthip_ctx* ctx, const dthip_col* col, int64_t n, int cmp, double cf, int64_t ci, int mem,
                     int32_t* out, int64_t* nout) {
  if (!col) { set_error("null column"); return DTHIP_EINVAL; }
  if (cmp < DTHIP_GT || cmp > DTHIP_ISNA) { set_error("bad comparison %d", cmp); return DTHIP_EINVAL; }
  const int sz = stype_size(col->stype);
  if (!sz) { set_error("unsupported stype %d", col->stype); return DTHIP_ENOTIMPL; }
  PredArgs p;
  memset(&p, 0, sizeof(p));
  p.data = col->data; p.stype = col->stype; p.cmp = cmp; p.cf = cf; p.ci = ci; p.is_mask = 0;
  return compact_common(ctx, p, sz, n, mem, out, nout);
}

int dthip_filter_take(dthip_ctx* ctx, const dthip_col* col, int cmp, double cf, int64_t ci, const dthip_col* cols, int ncols,
                      int64_t n, int mem, int32_t* out_rowindex, void* const* out_cols, int64_t* nout) {
  DTHIP_TRY(check_common(ctx, n, mem));
  if (!col || !nout || ncols < 0 || ncols > 8 || (ncols > 0 && (!cols || !out_cols))) { set_error("bad filter_take arguments"); return DTHIP_EINVAL; }
  if (cmp < DTHIP_GT || cmp > DTHIP_ISNA) { set_error("bad comparison %d", cmp); return DTHIP_EINVAL; }
  *nout = 0;
  if (n == 0) return DTHIP_OK;
  const int sz = stype_size(col->stype);
  if (!sz || !col->data) { set_error("unsupported predicate column"); return DTHIP_ENOTIMPL; }
  Scratch sc(ctx);
  PredArgs p;
  memset(&p, 0, sizeof(p));
  p.stype = col->stype; p.cmp = cmp; p.cf = cf; p.ci = ci; p.is_mask = 0;
  DTHIP_TRY(stage_in(ctx, sc, col->data, (size_t)n * sz, mem, &p.data));
  TakeCols tc;
  memset(&tc, 0, sizeof(tc));
  tc.n = ncols;
  std::vector<void*> d_out((size_t)ncols, nullptr);
  for (int c = 0; c < ncols; c++) {
    const int w = stype_size(cols[c].stype);
    if (!w || !cols[c].data || !out_cols[c]) { set_error("filter_take: bad column %d", c); return DTHIP_EINVAL; }
    tc.width[c] = w;
    DTHIP_TRY(stage_in(ctx, sc, cols[c].data, (size_t)n * w, mem, &tc.in[c]));
    d_out[(size_t)c] = out_cols[c];
    if (mem == DTHIP_HOST) { unsigned char* t = nullptr; DTHIP_TRY(sc.get<unsigned char>((size_t)n * w, &t)); d_out[(size_t)c] = t; }
    tc.out[c] = d_out[(size_t)c];
  }
  int32_t* d_ri = out_rowindex;
  if (out_rowindex && mem == DTHIP_HOST) DTHIP_TRY(sc.get<int32_t>((size_t)n, &d_ri));
  DTHIP_TRY(launch_compact_take(ctx, p, n, d_ri, tc, nout));
  if (mem == DTHIP_HOST) {
    if (out_rowindex) DTHIP_TRY(copy_out(ctx, out_rowindex, d_ri, sizeof(int32_t) * (size_t)*nout, mem));
    for (int c = 0; c < ncols; c++) DTHIP_TRY(copy_out(ctx, out_cols[c], d_out[(size_t)c], (size_t)*nout * tc.width[c], mem));
  }
  return DTHIP_OK;
}

int dthip_gather(dthip_ctx* ctx, const dthip_col* col, const int32_t* rowindex, int64_t nout, int mem, void* out) {
  DTHIP_TRY(check_common(ctx, nout, mem));
  if (nout == 0) return DTHIP_OK;
  if (!col || !col->data || !rowindex || !out) { set_error("null argument"); return DTHIP_EINVAL; }
  const int sz = stype_size(col->stype);
  if (!sz) { set_error("unsupported stype %d", col->stype); return DTHIP_ENOTIMPL; }
  if (mem == DTHIP_DEVICE) return launch_gather(ctx, col->data, col->stype, rowindex, nout, out);
  Scratch sc(ctx);
  int64_t vrows = 0;
  for (int64_t i = 0; i < nout; i++) if (rowindex[i] >= vrows) vrows = (int64_t)rowindex[i] + 1;
  const void* d_val = nullptr; const void* d_ri = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, col->data, (size_t)vrows * sz, mem, &d_val));
  DTHIP_TRY(stage_in(ctx, sc, rowindex, sizeof(int32_t) * (size_t)nout, mem, &d_ri));
  unsigned char* d_out = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)nout * sz, &d_out));
  DTHIP_TRY(launch_gather(ctx, d_val, col->stype, static_cast<const int32_t*>(d_ri), nout, d_out));
  return copy_out(ctx, out, d_out, (size_t)nout * sz, mem);
}

}  // extern "C"
