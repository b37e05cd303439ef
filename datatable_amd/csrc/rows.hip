// rows.hip -- dthip_groupby (group(), sort.cc:1411-1495), dthip_groupby_rows (DT[:, cols, by(keys)]) and
// dthip_filter_groupby_rows (config 5's two statements in one call: the fused tile-local route) -- split out of api.hip in round 6
#include <algorithm>
#include "host.hpp"

using namespace dthip;

extern "C" {

// ---------------------------------------------------------------------------------
int dthip_groupby(dthip_ctx* ctx, const dthip_col* keys, int nkeys, int64_t nrows, int na_pos, int mem,
                  int want_rowindex, dthip_result** out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  CallScope call_scope(ctx);
  if (!keys || !out) { set_error("null argument"); return DTHIP_EINVAL; }
  const bool remove_na = na_pos == DTHIP_NA_REMOVE;
  if (remove_na) na_pos = DTHIP_NA_FIRST;      // sorted first, then cut off the front (sort.cc:598-608)
  if (na_pos != DTHIP_NA_FIRST && na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", na_pos); return DTHIP_ENOTIMPL; }
  dthip_result* res = new dthip_result();
  res->nkeys = nkeys;
  int rc = DTHIP_OK;
  {
    Scratch sc(ctx);
    std::vector<dthip_col> kd;
    rc = stage_cols(ctx, sc, keys, nkeys, nrows, mem, &kd);
    if (rc == DTHIP_OK) {
      if (nrows == 0) {
        rc = empty_result(ctx, res);   // Groupby::zero_groups(), sort.cc:1428-1431
      } else {
        KeyPlan plan; Grouping g;
        // (round 6, measured and dropped: this call on the TILE-LOCAL levels of dthip_filter_groupby_rows with a predicate every
        // row passes -- no transform / tile-histogram passes, sequential writes -- took 6.6 + 5.4 + 6.2 ms for 1e9 rows against
        // 3.0 + 1.8 + 3.9 + 4.4 + 4.0 here: the gathers cost more than the passes they save, profiles/r06_groupby_tl_ab.txt)
        rc = group_core(ctx, sc, res, kd.data(), nkeys, nrows, na_pos, &plan, &g);
        if (rc == DTHIP_OK) {
          res->nrows = nrows; res->ngroups = g.ngroups; res->offsets = g.offsets;
          if (want_rowindex) {
            // the ordering may alias nothing user-owned here: it is always a scratch buffer
            result_adopt(sc, res, g.rowindex);
            res->rowindex = g.rowindex;
          }
          if (remove_na) {
            // SortContext::get_result_rowindex (sort.cc:598-608) cuts `nacount` rows off the front of the
            // NA-first ordering, where nacount is taken from the column sorted LAST (the context's current
            // column after continue_sort).  With one key that is exactly its NA group; with several keys it
            // is whatever leads the ordering -- reproduced as is.
            int32_t* scratch_idx = nullptr;
            int64_t skip = 0;
            rc = sc.get<int32_t>((size_t)nrows, &scratch_idx);
            if (rc == DTHIP_OK) {
              PredArgs p{kd[nkeys - 1].data, kd[nkeys - 1].stype, DTHIP_ISNA, 0.0, 0, 0};
              rc = launch_compact(ctx, p, nrows, scratch_idx, &skip);
            }
            if (rc == DTHIP_OK && skip > 0) {
              int32_t g0 = 0;              // groups that lie entirely inside the cut
              void* d_g0 = nullptr;
              rc = result_alloc(ctx, res, sizeof(int32_t) * ((size_t)g.ngroups + 2), &d_g0);
              int32_t* off2 = static_cast<int32_t*>(d_g0);
              if (rc == DTHIP_OK) rc = launch_offsets_drop_rows(ctx, g.offsets, g.ngroups, (int32_t)skip, off2, off2 + g.ngroups + 1);
              if (rc == DTHIP_OK) rc = read_back(ctx, &g0, off2 + g.ngroups + 1, sizeof(int32_t));
              if (rc == DTHIP_OK) {
                res->offsets = off2;
                res->ngroups = g.ngroups - g0;
                res->nrows = nrows - skip;
                if (res->rowindex) res->rowindex += skip;
              }
            }
          }
        }
      }
    }
  }
  if (rc != DTHIP_OK) { result_destroy(ctx, res); return rc; }
  *out = res;
  return DTHIP_OK;
}

int dthip_groupby_rows(dthip_ctx* ctx, const dthip_col* keys, int nkeys, const dthip_col* cols, int ncols,
                       int64_t nrows, int na_pos, int mem, int want_rowindex, dthip_result** out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  CallScope call_scope(ctx);
  if (!keys || !out || (ncols > 0 && !cols) || ncols < 0) { set_error("null argument"); return DTHIP_EINVAL; }
  if (na_pos != DTHIP_NA_FIRST && na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", na_pos); return DTHIP_ENOTIMPL; }
  dthip_result* res = new dthip_result();
  res->nkeys = nkeys;
  res->col.assign(ncols, nullptr); res->col_stype.assign(ncols, 0);
  for (int c = 0; c < ncols; c++) res->col_stype[c] = cols[c].stype;
  int rc = DTHIP_OK;
  do {
    Scratch sc(ctx);
    std::vector<dthip_col> kd, cd;
    if ((rc = stage_cols(ctx, sc, keys, nkeys, nrows, mem, &kd)) != DTHIP_OK) break;
    if ((rc = stage_cols(ctx, sc, cols, ncols, nrows, mem, &cd)) != DTHIP_OK) break;
    if (nrows == 0) { rc = empty_result(ctx, res); break; }
    KeyPlan plan; Grouping g;
    if ((rc = plan_keys(ctx, sc, kd.data(), nkeys, nrows, na_pos, &plan, true, true)) != DTHIP_OK) break;   // guessed key range: verified below
    bool ride = plan.nstages == 1 && ncols + (want_rowindex ? 1 : 0) <= MAX_PAYCOLS && ncols > 0;
    for (int c = 0; c < ncols; c++) if (stype_size(cd[c].stype) < 4) ride = false;
    PaySpec ps;
    SortOut so;
    std::vector<int> slot(ncols, -1), is_key(ncols, -1);
    if (ride) {
      // the columns (and the row ids) ride through the radix passes: streaming reads and run-wise
      // writes instead of one random gather per column through the finished RowIndex
      ps.n = 0;
      if (want_rowindex) { ps.in[0] = nullptr; ps.width[0] = 4; ps.iota = true; ps.n = 1; }
      // a requested column that IS a key column does not ride along: the sorted packed keys are
      // turned back into it afterwards (streaming), which saves its bytes in every pass
      for (int c = 0; c < ncols; c++) {
        for (int k = 0; k < nkeys; k++)
          if (cols[c].data == keys[k].data && cols[c].stype == keys[k].stype) is_key[c] = k;
        if (is_key[c] >= 0) continue;
        slot[c] = ps.n;
        ps.in[ps.n] = cd[c].data; ps.width[ps.n] = stype_size(cd[c].stype); ps.n++;
      }
      // one int32 / int64 key that is also a wanted column: the last pass writes its original values (no untransform pass)
      static const bool fuse_ukey = !(getenv("DTHIP_FUSE_UKEY") && atoi(getenv("DTHIP_FUSE_UKEY")) == 0);
      int ukc = -1;
      if (fuse_ukey && nkeys == 1 && (kd[0].stype == DTHIP_INT64 || kd[0].stype == DTHIP_INT32))
        for (int c = 0; c < ncols; c++) if (is_key[c] == 0) { ukc = c; break; }
      if (ukc >= 0) {
        void* q = nullptr;
        if ((rc = result_alloc(ctx, res, (size_t)nrows * stype_size(kd[0].stype), &q)) != DTHIP_OK) break;
        ps.ukey_out = q;
      }
      rc = sort_stage(ctx, sc, plan, 0, nrows, nullptr, ps, &so);
      if (rc == DTHIP_RETRY_EXACT) {
        ctx->call_stats[0]++;
        // the sampled key range did not hold, so real keys lie OUTSIDE it: the exact range is usually WIDER, and with
        // several keys the packed width may now exceed 64 bits (two stages) -- then the columns cannot ride
        if ((rc = plan_keys(ctx, sc, kd.data(), nkeys, nrows, na_pos, &plan)) != DTHIP_OK) break;
        if (plan.nstages != 1) { ride = false; rc = DTHIP_OK; }
        else rc = sort_stage(ctx, sc, plan, 0, nrows, nullptr, ps, &so);
      }
      if (rc != DTHIP_OK) break;
    }
    if (ride) {
      if (so.ukey_done) {
        // groups = runs of equal ORIGINAL key values (the transform is a bijection, NA <-> NA)
        if ((rc = heads_to_offsets(ctx, sc, res, ps.ukey_out, kd[0].stype == DTHIP_INT64, nullptr, nrows, &g)) != DTHIP_OK) break;
      } else
      if ((rc = heads_to_offsets(ctx, sc, res, so.keys, so.key64, nullptr, nrows, &g)) != DTHIP_OK) break;
      if (want_rowindex) { result_adopt(sc, res, so.pay[0]); res->rowindex = static_cast<int32_t*>(so.pay[0]); }
      bool ukey_used = false;
      for (int c = 0; c < ncols && rc == DTHIP_OK; c++) {
        if (is_key[c] >= 0) {
          void* q = nullptr;
          if (so.ukey_done && ps.ukey_out && res->col[c] == nullptr && is_key[c] == 0) {
            // the first copy of the key column is the buffer the last pass filled; further copies are duplicated from it
            bool first = true;
            for (int c2 = 0; c2 < c; c2++) if (is_key[c2] == 0) first = false;
            if (first) { res->col[c] = ps.ukey_out; continue; }
            const size_t bytes = (size_t)nrows * stype_size(cd[c].stype);
            if ((rc = result_alloc(ctx, res, bytes, &q)) != DTHIP_OK) break;
            if (hipMemcpyAsync(q, ps.ukey_out, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; break; }
            res->col[c] = q;
            continue;
          }
          if (so.ukey_done) { set_error("groupby_rows: internal: packed keys missing"); rc = DTHIP_EDEVICE; break; }
          if (ps.ukey_out && is_key[c] == 0 && !ukey_used) { q = ps.ukey_out; ukey_used = true; }      // the buffer set aside for the last pass
          else if ((rc = result_alloc(ctx, res, (size_t)nrows * stype_size(cd[c].stype), &q)) != DTHIP_OK) break;
          rc = launch_untransform_keys(ctx, so.keys, so.key64, nullptr, nrows, plan.col[is_key[c]], plan.nsig[is_key[c]], q);
          res->col[c] = q;
          continue;
        }
        void* p = so.pay[slot[c]];
        if (p == cd[c].data) {       // nothing moved (single group / already ordered passes skipped): copy
          void* q = nullptr;
          const size_t bytes = (size_t)nrows * stype_size(cd[c].stype);
          if ((rc = result_alloc(ctx, res, bytes, &q)) != DTHIP_OK) break;
          if (hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; break; }
          res->col[c] = q;
        } else {
          result_adopt(sc, res, p);
          res->col[c] = p;
        }
      }
      if (rc != DTHIP_OK) break;
    } else {
      if ((rc = group_core(ctx, sc, res, kd.data(), nkeys, nrows, na_pos, &plan, &g)) != DTHIP_OK) break;
      for (int c = 0; c < ncols; c++) {
        void* q = nullptr;
        if ((rc = result_alloc(ctx, res, (size_t)nrows * stype_size(cd[c].stype), &q)) != DTHIP_OK) break;
        if ((rc = launch_gather(ctx, cd[c].data, cd[c].stype, g.rowindex, nrows, q)) != DTHIP_OK) break;
        res->col[c] = q;
      }
      if (rc != DTHIP_OK) break;
      if (want_rowindex) { result_adopt(sc, res, g.rowindex); res->rowindex = g.rowindex; }
    }
    res->nrows = nrows; res->ngroups = g.ngroups; res->offsets = g.offsets;
  } while (0);
  if (rc != DTHIP_OK) { result_destroy(ctx, res); return rc; }
  *out = res;
  return DTHIP_OK;
}

// ---- V = DT[f.x <cmp> c, :]; V[:, cols, by(key)] in one call ----------------------------------------------------------
// The fused route (tlsort.hip): ONE sweep over the unfiltered rows evaluates the predicate, transforms the key and orders
// every tile's passing rows by the top digit inside the tile's own row range (sequential writes + a 16-bit directory);
// level 2 collects every bucket's rows from those segments and scatters them to their final buckets, which the final
// level orders in LDS and writes in place (the last two as in sort_stage's MSD levels).  DTHIP_NOT_APPLICABLE: the query
// does not fit (the caller then runs filter_take + groupby_rows); DTHIP_RETRY_EXACT: a guessed key range was wrong.
static int filter_rows_fused(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const dthip_col& pred, int cmp, double cf, int64_t ci,
                             const dthip_col* keys_orig, const std::vector<dthip_col>& kd, const dthip_col* cols_orig,
                             const std::vector<dthip_col>& cd, int ncols, int64_t n, int na_pos, int want_rowindex, bool speculative) {
  if (ctx->sort_path == 1 || (ctx->sort_path != 2 && n < ctx->msd_min_rows)) return DTHIP_NOT_APPLICABLE;
  if (kd.size() != 1 || (kd[0].stype != DTHIP_INT32 && kd[0].stype != DTHIP_INT64)) return DTHIP_NOT_APPLICABLE;
  if (stype_size(pred.stype) != 8) return DTHIP_NOT_APPLICABLE;
  // riding columns (requested columns that are not the key column) and the row number: <= 2, an 8-byte one first
  std::vector<int> is_key(ncols, 0), slot(ncols, -1);
  int ride[2] = {-1, -1}, nride = 0;
  for (int c = 0; c < ncols; c++) {
    if (cols_orig[c].data == keys_orig[0].data && cols_orig[c].stype == keys_orig[0].stype) { is_key[c] = 1; continue; }
    for (int c2 = 0; c2 < c; c2++) if (!is_key[c2] && cols_orig[c2].data == cols_orig[c].data && cols_orig[c2].stype == cols_orig[c].stype) slot[c] = slot[c2];
    if (slot[c] >= 0) continue;
    if (nride == 2) return DTHIP_NOT_APPLICABLE;
    const int w = stype_size(cd[c].stype);
    if (w != 4 && w != 8) return DTHIP_NOT_APPLICABLE;
    slot[c] = nride; ride[nride++] = c;
  }
  if (nride == 2 && stype_size(cd[ride[0]].stype) == 4 && stype_size(cd[ride[1]].stype) == 8) {
    std::swap(ride[0], ride[1]);
    for (int c = 0; c < ncols; c++) if (slot[c] >= 0) slot[c] ^= 1;
  }
  int npay = nride, rid_slot = -1;
  int payw[2] = {nride > 0 ? stype_size(cd[ride[0]].stype) : 0, nride > 1 ? stype_size(cd[ride[1]].stype) : 0};
  if (want_rowindex) {
    if (npay == 2) return DTHIP_NOT_APPLICABLE;
    rid_slot = npay; payw[npay++] = 4;
  }
  if (npay == 0) return DTHIP_NOT_APPLICABLE;
  if (npay == 2 && payw[0] == 4) return DTHIP_NOT_APPLICABLE;              // (4, 4): not a variant of the gather pass
  KeyPlan plan;
  DTHIP_TRY(plan_keys(ctx, sc, kd.data(), 1, n, na_pos, &plan, speculative, true));
  if (plan.nstages != 1 || plan.stage_bits[0] > 32 || plan.stage_bits[0] < 3) return DTHIP_NOT_APPLICABLE;
  const int bits = plan.stage_bits[0];
  const uint32_t tile = tl_tile_rows();
  PredArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.data = pred.data; pa.stype = pred.stype; pa.cmp = cmp; pa.cf = cf; pa.ci = ci; pa.is_mask = 0;
  // passing rows, estimated from 65536 evenly spaced rows: sizes the digits (a final bucket should hold ~msd_bucket_rows rows)
  uint32_t* d_cnt = nullptr;
  DTHIP_TRY(sc.get<uint32_t>(1, &d_cnt));
  DTHIP_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t), ctx->stream));
  const uint32_t nsamp = (uint32_t)std::min<int64_t>(n, 65536);
  DTHIP_TRY(launch_tl_pred_sample(ctx, pa, (uint32_t)n, nsamp, d_cnt));
  uint32_t scnt = 0;
  DTHIP_TRY(read_back(ctx, &scnt, d_cnt, sizeof(scnt)));
  const int64_t est = std::max<int64_t>(2, (int64_t)((double)n * ((double)scnt + 0.5) / (double)nsamp));
  static const int rbmax = getenv("DTHIP_MSD_RBMAX") ? atoi(getenv("DTHIP_MSD_RBMAX")) : 9;
  const MsdPlan msd = msd_split(est, bits, tile, ctx->msd_bucket_rows, rbmax > 9 ? 9 : rbmax);
  if (!msd.ok) return DTHIP_NOT_APPLICABLE;
  const uint32_t nb1 = 1u << msd.s1, bins2 = 1u << msd.s2;
  // Level 2 writes tile-locally as well (default) -- decided HERE, from the estimate, because level 1's output format
  // depends on it: (the gathering final level works on windows of <= 16 whole buckets whose (bucket, digit) counts fit the
  // exchange buffer: tiny final buckets -- tests forcing the levels onto small inputs -- take the scatter form)
  int maxw_p = 4;
  for (int q = 0; q < npay; q++) maxw_p = std::max(maxw_p, payw[q]);
  const int64_t win_buckets = std::min<int64_t>(16, (int64_t)tile * maxw_p / ((int64_t)8 << msd.rb));
  static const int tl2_env = getenv("DTHIP_TL_LEVEL2") ? atoi(getenv("DTHIP_TL_LEVEL2")) : -1;
  const int tl2_opt = tl2_env >= 0 ? tl2_env : ctx->tl_level2;
  const bool tl2 = tl2_opt == 2 || (tl2_opt == 1 && (est >> (msd.s1 + msd.s2)) * win_buckets * 4 >= (int64_t)tile * 5);
  // ---- level 1: filter + key transform + top digit, tile-local ------------------------------------------------------------
  // first-level tiles: 512 threads x 16 rows, two workgroups per CU -- or (DTHIP_TL_BLOCK=1024, A/B) 16384-row tiles, whose
  // segments are twice as long for the level that gathers them, one workgroup per CU
  static const int tl_block = (getenv("DTHIP_TL_BLOCK") && atoi(getenv("DTHIP_TL_BLOCK")) == 1024) ? 1024 : 512;
  const uint32_t T1 = (uint32_t)tl_block * 16u;
  const uint32_t ntiles1 = (uint32_t)((n + T1 - 1) / T1);
  const uint32_t ntb = (ntiles1 + 63) / 64, dstride = ntb * 64;
  // Level 1 writes RECORDS {key, 4-byte riding value, 8-byte riding value} when a tile-local level 2 will gather them (one
  // 16-byte piece per row instead of three places: level 2's over-fetch 3.2x -> ~1.4x); separate arrays otherwise
  static const bool aos_env = !(getenv("DTHIP_TL_RECORDS") && atoi(getenv("DTHIP_TL_RECORDS")) == 0);
  const bool use_rec = aos_env && tl2 && tl_block == 512 && !(npay == 2 && payw[1] == 8);
  uint32_t* k1 = nullptr; uint16_t* dir = nullptr; uint16_t* dirT = nullptr; uint32_t* cc = nullptr; uint32_t* tot = nullptr;
  unsigned char* rec = nullptr;
  if (use_rec) DTHIP_TRY(sc.get<unsigned char>((size_t)n * 16, &rec));
  else DTHIP_TRY(sc.get<uint32_t>((size_t)n, &k1));
  DTHIP_TRY(sc.get<uint16_t>((size_t)ntiles1 * (nb1 + 1), &dir));
  DTHIP_TRY(sc.get<uint16_t>((size_t)(nb1 + 1) * dstride, &dirT));
  DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 * ntb, &cc));
  DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 + 1, &tot));
  DTHIP_CHECK_HIP(hipMemsetAsync(tot + nb1, 0, sizeof(uint32_t), ctx->stream));
  void* l1[2] = {nullptr, nullptr};
  if (!use_rec)
    for (int q = 0; q < npay; q++) { unsigned char* b = nullptr; DTHIP_TRY(sc.get<unsigned char>((size_t)n * payw[q], &b)); l1[q] = b; }
  TL1Args ta;
  memset(&ta, 0, sizeof(ta));
  ta.pred = pa; ta.key = plan.col[0]; ta.key.shift = 0;
  ta.n = (uint32_t)n; ta.block = tl_block; ta.shift = msd.rb + msd.s2; ta.bits = msd.s1;
  ta.kout = k1; ta.dir = dir; ta.rowid = rid_slot >= 0 ? static_cast<uint32_t*>(l1[rid_slot]) : nullptr;
  ta.keepx = -1; ta.pay.n = nride;
  for (int q = 0; q < nride; q++) {
    ta.pay.in[q] = cd[ride[q]].data; ta.pay.out[q] = l1[q]; ta.pay.width[q] = payw[q];
    static const bool keepx_on = !(getenv("DTHIP_TL_KEEPX") && atoi(getenv("DTHIP_TL_KEEPX")) == 0);     // (A/B: re-read the predicate column instead)
    if (keepx_on && cd[ride[q]].data == pred.data && payw[q] == 8) ta.keepx = q;
  }
  ta.bad = plan.speculative ? tot + nb1 : nullptr;
  ta.rec = rec; ta.rec4 = -1; ta.rec8 = -1;
  if (use_rec)
    for (int q = 0; q < npay; q++) {
      if (payw[q] == 8) ta.rec8 = q;
      else ta.rec4 = (q == rid_slot) ? -2 : q;
    }
  DTHIP_TRY(launch_tl_level1(ctx, ta));
  DTHIP_TRY(launch_tl_directory(ctx, dir, ntiles1, nb1, dirT, dstride, cc, ntb, tot));
  std::vector<uint32_t> htot((size_t)nb1 + 1);
  DTHIP_TRY(read_back(ctx, htot.data(), tot, htot.size() * sizeof(uint32_t)));
  if (plan.speculative && htot[nb1]) return DTHIP_RETRY_EXACT;
  int64_t npass = 0;
  for (uint32_t b = 0; b < nb1; b++) npass += htot[b];
  if (npass == 0) { DTHIP_TRY(empty_result(ctx, res)); return DTHIP_OK; }
  // ---- level 2: ragged tiles inside the level-1 buckets, planned on the host (msd_plan.hpp), rows read through the directory
  BucketGeom hg;
  memset(&hg, 0, sizeof(hg));
  {
    const uint32_t gmax = (uint32_t)ctx->num_cus * 4, nt = (uint32_t)((npass + tile - 1) / tile);
    hg.tpg = (nt + gmax - 1) / gmax; if (hg.tpg == 0) hg.tpg = 1;
  }
  std::vector<uint32_t> tdesc, gdesc, gfirst, pstart((size_t)nb1 + 1, 0);
  msd_level2_tiles(htot.data(), nb1, tile, hg.tpg, &tdesc, &gdesc, &gfirst);
  for (uint32_t b = 0; b < nb1; b++) pstart[b + 1] = pstart[b] + htot[b];
  const uint32_t ntiles2 = (uint32_t)(tdesc.size() / 4), G2 = (uint32_t)(gdesc.size() / 2);
  uint32_t* d_plan = nullptr;
  DTHIP_TRY(sc.get<uint32_t>(tdesc.size() + gdesc.size() + gfirst.size() + pstart.size() + 4, &d_plan));
  uint32_t* d_tdesc = d_plan; uint32_t* d_gdesc = d_tdesc + tdesc.size(); uint32_t* d_gfirst = d_gdesc + gdesc.size();
  uint32_t* d_pstart = d_gfirst + gfirst.size(); uint32_t* d_max = d_pstart + pstart.size();
  DTHIP_CHECK_HIP(hipMemcpyAsync(d_tdesc, tdesc.data(), tdesc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipMemcpyAsync(d_gdesc, gdesc.data(), gdesc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipMemcpyAsync(d_gfirst, gfirst.data(), gfirst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipMemcpyAsync(d_pstart, pstart.data(), pstart.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipMemsetAsync(d_max, 0, 4, ctx->stream));
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));            // (the host vectors above go out of use only at the end; pageable copies)
  // Level 2, tile-local (tl2, decided above): no histogram pass, sequential writes, the final level gathers its buckets'
  // segments.  tl_level2 = 0: level 2 scatters to exact positions (a gathering histogram pass first) and the final level
  // runs in place, as in sort_stage -- kept for A/B runs and for tiny final buckets.

  uint32_t* P = nullptr; uint32_t* gtot2 = nullptr; uint32_t* fstart = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 * bins2 + 1, &fstart));
  const uint32_t nbk = nb1 * bins2;
  unsigned char* kA = nullptr; unsigned char* kB = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)npass * 4, &kA));
  DTHIP_TRY(sc.get<unsigned char>((size_t)npass * 4, &kB));
  void* pb[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  for (int q = 0; q < npay; q++)
    for (int h = 0; h < 2; h++) { unsigned char* b = nullptr; DTHIP_TRY(sc.get<unsigned char>((size_t)npass * payw[q], &b)); pb[h][q] = b; }
  RadixPass rp;
  memset(&rp, 0, sizeof(rp));
  rp.kin = k1; rp.kout = kA; rp.key64 = 0; rp.n = (uint32_t)npass;
  rp.shift = msd.rb; rp.bits = msd.s2; rp.tpg = hg.tpg; rp.iota = 0;
  rp.ntiles = ntiles2; rp.tdesc = d_tdesc;
  rp.pay.n = npay;
  for (int q = 0; q < npay; q++) { rp.pay.in[q] = l1[q]; rp.pay.out[q] = pb[0][q]; rp.pay.width[q] = payw[q]; }
  rp.g_dirT = dirT; rp.g_dstride = dstride; rp.g_cc = cc; rp.g_ntb = ntb; rp.g_ntiles1 = ntiles1; rp.g_T1 = T1; rp.g_pstart = d_pstart;
  rp.g_rec = rec;
  rp.label = "tl_level2_kernel";
  uint16_t* dirT2 = nullptr; uint32_t* d_pfirst = nullptr;
  const uint32_t dstride2 = ((ntiles2 + 63) / 64) * 64;
  if (tl2) {
    uint16_t* dir2 = nullptr;
    DTHIP_TRY(sc.get<uint16_t>((size_t)(ntiles2 ? ntiles2 : 1) * (bins2 + 1), &dir2));
    DTHIP_TRY(sc.get<uint16_t>((size_t)(bins2 + 1) * dstride2, &dirT2));
    DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 + 1, &d_pfirst));
    std::vector<uint32_t> pfirst((size_t)nb1 + 1, ntiles2);           // first level-2 tile of every parent bucket
    for (uint32_t t = ntiles2; t-- > 0;) pfirst[tdesc[4 * (size_t)t + 3]] = t;
    for (uint32_t b = nb1; b-- > 0;) if (htot[b] == 0) pfirst[b] = pfirst[b + 1];
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_pfirst, pfirst.data(), pfirst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    rp.tl_dir2 = dir2;
    DTHIP_TRY(launch_radix_pass(ctx, rp));
    DTHIP_TRY(launch_tl_final_plan(ctx, dir2, ntiles2, nb1, msd.s2, d_pfirst, d_pstart, dirT2, dstride2, fstart, d_max));
  } else {
    DTHIP_TRY(sc.get<uint32_t>((size_t)(ntiles2 ? ntiles2 : 1) * bins2, &P));
    DTHIP_TRY(sc.get<uint32_t>((size_t)(G2 ? G2 : 1) * bins2, &gtot2));
    TLGatherHistArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.keys = k1; ga.shift = msd.rb; ga.bits = msd.s2; ga.tdesc = d_tdesc; ga.gdesc = d_gdesc; ga.pstart = d_pstart;
    ga.dirT = dirT; ga.dstride = dstride; ga.cc = cc; ga.ntb = ntb; ga.ntiles1 = ntiles1; ga.T1 = T1; ga.P = P; ga.gtot = gtot2;
    DTHIP_TRY(launch_tl_gather_hist(ctx, ga, G2));
    DTHIP_TRY(launch_msd_scan(ctx, gtot2, d_gfirst, d_pstart, msd.s2, nb1, (uint32_t)npass, fstart, d_max));
  }
  int maxw_w = 4;
  for (int q = 0; q < npay; q++) maxw_w = std::max(maxw_w, payw[q]);
  WindowPlan wp;
  DTHIP_TRY(plan_windows(ctx, sc, fstart, nb1, bins2, npass, d_max, tile, maxw_w, msd.rb, &wp));
  const bool windows = wp.ok;
  const uint32_t maxsize = wp.maxsize;
  if (getenv("DTHIP_MSD_DEBUG"))
    fprintf(stderr, "[dthip fused] n=%lld est=%lld pass=%lld bits=%d s1=%d s2=%d rb=%d tiles1=%u tiles2=%u groups2=%u largest bucket=%u windows=%u max buckets/window=%u records=%d -> %s\n",
            (long long)n, (long long)est, (long long)npass, bits, msd.s1, msd.s2, msd.rb, ntiles1, ntiles2, G2, maxsize, wp.nwin, wp.span, use_rec ? 1 : 0,
            windows ? "windows" : (maxsize <= tile ? "per bucket" : "not applicable"));
  // a final bucket outgrows a tile (heavy duplicates) -- or, with the gathering final level, there are no windows
  if (!(windows || (!tl2 && maxsize <= tile))) { ctx->call_stats[2]++; return DTHIP_NOT_APPLICABLE; }
  if (!tl2) { rp.P = P; rp.gpre = gtot2; DTHIP_TRY(launch_radix_pass(ctx, rp)); }
  // ---- final level: every bucket (window of buckets) ordered by the remaining bits in LDS, written to its rows of the result
  rp.g_dirT = nullptr; rp.g_cc = nullptr; rp.g_pstart = nullptr; rp.tl_dir2 = nullptr; rp.g_rec = nullptr;
  rp.kin = kA; rp.kout = kB; rp.shift = 0; rp.bits = msd.rb; rp.P = nullptr; rp.gpre = nullptr;
  rp.ntiles = nbk; rp.tdesc = nullptr; rp.bounds = fstart;
  rp.block = (maxsize <= tile / 2) ? 256 : 0;
  if (tl2) {
    rp.tdesc = d_tdesc;
    rp.g2_dirT = dirT2; rp.g2_dstride = dstride2; rp.g2_pfirst = d_pfirst; rp.g2_fstart = fstart; rp.g2_s2bits = msd.s2; rp.g2_nbk = nbk;
  }
  if (windows) {
    rp.ntiles = wp.nwin; rp.bounds = wp.bounds; rp.wfirst = wp.wfirst; rp.block = 0;
    rp.bits2 = wp.bits2; rp.wpairs = wp.pairs;
  }
  for (int q = 0; q < npay; q++) { rp.pay.in[q] = pb[0][q]; rp.pay.out[q] = pb[1][q]; }
  void* ukey_out = nullptr;
  for (int c = 0; c < ncols; c++) if (is_key[c]) { DTHIP_TRY(result_alloc(ctx, res, (size_t)npass * stype_size(kd[0].stype), &ukey_out)); break; }
  if (ukey_out) {
    const KeyColDev& kc = plan.col[0];
    rp.ukout = ukey_out; rp.uk_stype = kc.stype; rp.uk_desc = kc.desc; rp.uk_bits = bits;
    rp.uk_edge = kc.edge; rp.uk_na_repl = kc.na_repl; rp.uk_inc = kc.inc;
  }
  rp.label = "msd_final_kernel";
  DTHIP_TRY(launch_radix_pass(ctx, rp));
  Grouping g;
  if (ukey_out) DTHIP_TRY(heads_to_offsets(ctx, sc, res, ukey_out, kd[0].stype == DTHIP_INT64, nullptr, npass, &g));
  else DTHIP_TRY(heads_to_offsets(ctx, sc, res, kB, 0, nullptr, npass, &g));
  bool first_key = true;
  for (int c = 0; c < ncols; c++) {
    if (is_key[c]) {
      if (first_key) { res->col[c] = ukey_out; first_key = false; continue; }
      void* q = nullptr;
      const size_t bytes = (size_t)npass * stype_size(kd[0].stype);
      DTHIP_TRY(result_alloc(ctx, res, bytes, &q));
      DTHIP_CHECK_HIP(hipMemcpyAsync(q, ukey_out, bytes, hipMemcpyDeviceToDevice, ctx->stream));
      res->col[c] = q;
      continue;
    }
    void* p = pb[1][slot[c]];
    bool dup = false;
    for (int c2 = 0; c2 < c; c2++) if (!is_key[c2] && slot[c2] == slot[c]) dup = true;
    if (!dup) { result_adopt(sc, res, p); res->col[c] = p; continue; }
    void* q = nullptr;
    const size_t bytes = (size_t)npass * stype_size(cd[c].stype);
    DTHIP_TRY(result_alloc(ctx, res, bytes, &q));
    DTHIP_CHECK_HIP(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    res->col[c] = q;
  }
  if (rid_slot >= 0) { result_adopt(sc, res, pb[1][rid_slot]); res->rowindex = static_cast<int32_t*>(pb[1][rid_slot]); }
  res->nrows = npass; res->ngroups = g.ngroups; res->offsets = g.offsets;
  return DTHIP_OK;
}

int dthip_filter_groupby_rows(dthip_ctx* ctx, const dthip_col* pred, int cmp, double cf, int64_t ci, const dthip_col* keys, int nkeys,
                              const dthip_col* cols, int ncols, int64_t nrows, int na_pos, int mem, int want_rowindex,
                              dthip_result** out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  CallScope call_scope(ctx);
  if (!pred || !keys || !out || nkeys < 1 || nkeys > MAX_KEYCOLS || ncols < 0 || (ncols > 0 && !cols)) { set_error("bad filter_groupby_rows arguments"); return DTHIP_EINVAL; }
  if (cmp < DTHIP_GT || cmp > DTHIP_ISNA) { set_error("bad comparison %d", cmp); return DTHIP_EINVAL; }
  if (na_pos != DTHIP_NA_FIRST && na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", na_pos); return DTHIP_ENOTIMPL; }
  if (!stype_size(pred->stype) || (nrows > 0 && !pred->data)) { set_error("unsupported predicate column"); return DTHIP_ENOTIMPL; }
  dthip_result* res = new dthip_result();
  res->nkeys = nkeys;
  res->col.assign(ncols, nullptr); res->col_stype.assign(ncols, 0);
  for (int c = 0; c < ncols; c++) res->col_stype[c] = cols[c].stype;
  int rc = DTHIP_OK;
  do {
    Scratch sc(ctx);
    std::vector<dthip_col> pd, kd, cd;
    if ((rc = stage_cols(ctx, sc, pred, 1, nrows, mem, &pd)) != DTHIP_OK) break;
    if ((rc = stage_cols(ctx, sc, keys, nkeys, nrows, mem, &kd)) != DTHIP_OK) break;
    if ((rc = stage_cols(ctx, sc, cols, ncols, nrows, mem, &cd)) != DTHIP_OK) break;
    if (nrows == 0) { rc = empty_result(ctx, res); break; }
    if (mem == DTHIP_HOST) {
      // staged copies of one host column are different device buffers: the same-column tests below compare the CALLER's
      // pointers, and a staged column that is also the predicate column is mapped back onto it
      for (int c = 0; c < ncols; c++) if (cols[c].data == pred->data && cols[c].stype == pred->stype) cd[c].data = pd[0].data;
    }
    // fused route first (twice at most: a guessed key range, then the exact one)
    rc = DTHIP_NOT_APPLICABLE;
    static const bool fused_on = !(getenv("DTHIP_FILTER_ROWS_FUSED") && atoi(getenv("DTHIP_FILTER_ROWS_FUSED")) == 0);
    if (fused_on && ctx->filter_rows_fused) {
      for (int attempt = 0; attempt < 2; attempt++) {
        Scratch fs(ctx);
        rc = filter_rows_fused(ctx, fs, res, pd[0], cmp, cf, ci, keys, kd, cols, cd, ncols, nrows, na_pos, want_rowindex, attempt == 0);
        if (rc == DTHIP_ENOMEM) {             // 16 B per row of records + double buffers did not fit: the two calls need less
          rc = DTHIP_NOT_APPLICABLE; ctx->call_stats[2]++;
          dev_trim(ctx);
          break;
        }
        if (rc != DTHIP_RETRY_EXACT) break;
        ctx->call_stats[0]++;
      }
      if (rc == DTHIP_RETRY_EXACT) { set_error("filter_groupby_rows: exact key range violated"); rc = DTHIP_EDEVICE; }
      if (rc == DTHIP_OK) ctx->call_stats[3] = 4;
    }
    if (rc != DTHIP_NOT_APPLICABLE) break;
    // ---- the two-call sequence: filter (RowIndex + the view's columns in one sweep), then the rows in grouped order ------
    rc = DTHIP_OK;
    for (void* p : res->owned) dev_release(ctx, p);          // (anything a fused attempt set aside before it gave up)
    res->owned.clear();
    res->col.assign(ncols, nullptr);
    std::vector<const void*> uniq;                          // distinct source columns: keys first, then the requested columns
    std::vector<int> uniq_st, kmap(nkeys, -1), cmap(ncols, -1);
    auto add = [&](const dthip_col& orig, const dthip_col& dev) -> int {
      for (size_t u = 0; u < uniq.size(); u++) if (uniq[u] == dev.data && uniq_st[u] == orig.stype) return (int)u;
      uniq.push_back(dev.data); uniq_st.push_back(orig.stype);
      return (int)uniq.size() - 1;
    };
    for (int k = 0; k < nkeys; k++) kmap[k] = add(keys[k], kd[k]);
    for (int c = 0; c < ncols; c++) {
      bool same_key = false;
      for (int k = 0; k < nkeys; k++) if (cols[c].data == keys[k].data && cols[c].stype == keys[k].stype) { cmap[c] = kmap[k]; same_key = true; break; }
      if (!same_key) cmap[c] = add(cols[c], cd[c]);
    }
    if (uniq.size() > 8) { set_error("filter_groupby_rows: more than 8 distinct columns"); rc = DTHIP_ENOTIMPL; break; }
    PredArgs p;
    memset(&p, 0, sizeof(p));
    p.data = pd[0].data; p.stype = pred->stype; p.cmp = cmp; p.cf = cf; p.ci = ci; p.is_mask = 0;
    TakeCols tc;
    memset(&tc, 0, sizeof(tc));
    tc.n = (int)uniq.size();
    std::vector<void*> fbuf(uniq.size(), nullptr);
    for (size_t u = 0; u < uniq.size() && rc == DTHIP_OK; u++) {
      const int w = stype_size(uniq_st[u]);
      unsigned char* b = nullptr;
      if ((rc = sc.get<unsigned char>((size_t)nrows * w, &b)) != DTHIP_OK) break;
      fbuf[u] = b; tc.in[u] = uniq[u]; tc.out[u] = b; tc.width[u] = w;
    }
    if (rc != DTHIP_OK) break;
    int32_t* fri = nullptr;
    if (want_rowindex && (rc = sc.get<int32_t>((size_t)nrows, &fri)) != DTHIP_OK) break;
    int64_t npass = 0;
    if ((rc = launch_compact_take(ctx, p, nrows, fri, tc, &npass)) != DTHIP_OK) break;
    if (npass == 0) { rc = empty_result(ctx, res); break; }
    std::vector<dthip_col> fk(nkeys), fc(ncols + (want_rowindex ? 1 : 0));
    for (int k = 0; k < nkeys; k++) { fk[k] = keys[k]; fk[k].data = fbuf[kmap[k]]; }
    for (int c = 0; c < ncols; c++) { fc[c] = cols[c]; fc[c].data = fbuf[cmap[c]]; }
    if (want_rowindex) { fc[ncols].data = fri; fc[ncols].stype = DTHIP_INT32; fc[ncols].flags = 0; }
    dthip_result* tmp = nullptr;
    if ((rc = dthip_groupby_rows(ctx, fk.data(), nkeys, fc.data(), (int)fc.size(), npass, na_pos, DTHIP_DEVICE, 0, &tmp)) != DTHIP_OK) break;
    res->owned = tmp->owned; tmp->owned.clear();
    for (int c = 0; c < ncols; c++) res->col[c] = tmp->col[c];
    if (want_rowindex) res->rowindex = static_cast<int32_t*>(tmp->col[ncols]);      // the COMPOSED RowIndex (rowindex_array.cc:258-269)
    res->nrows = tmp->nrows; res->ngroups = tmp->ngroups; res->offsets = tmp->offsets;
    delete tmp;
  } while (0);
  if (rc != DTHIP_OK) { result_destroy(ctx, res); return rc; }
  *out = res;
  return DTHIP_OK;
}

}  // extern "C"
