// plan.hip -- key planning, the sort driver and group heads -> offsets (host side; split out of api.hip in round 6).
// Mirrors sort.cc:728-776 (_initI: key ranges -> transforms), :917-934 (pass parameters), :1128-1353 (radix_psort /
// _radix_recurse: here LSD passes or MSD levels), sort_groups.cc:30-117 (GroupGatherer: heads -> offsets)
#include <algorithm>
#include "host.hpp"

namespace dthip {

// ---- sort planning ---------------------------------------------------------------
static int nbits_u64(unsigned long long v) { int b = 0; while (v) { b++; v >>= 1; } return b; }


static void stype_int_limits(int st, long long* lo, long long* hi) {
  switch (st) {
    case DTHIP_BOOL: case DTHIP_INT8: *lo = INT8_MIN + 1; *hi = INT8_MAX; break;
    case DTHIP_INT16: *lo = INT16_MIN + 1; *hi = INT16_MAX; break;
    case DTHIP_INT32: *lo = (long long)INT32_MIN + 1; *hi = INT32_MAX; break;
    default: *lo = INT64_MIN + 1; *hi = INT64_MAX; break;
  }
}

// min/max of integer keys -> transform parameters (sort.cc:728-776), packing layout.
// speculative: the range of big integer columns is GUESSED from a sample and widened; only the
// bucketed aggregation may use such a plan, because its histogram pass verifies every row.
// tight (sort path): a guess whose 1/64 margin adds a significant bit to a key could cost a whole radix pass, which is
// more than the exact range scan it saves -- such a plan is made again with the exact range at once.
int plan_keys(dthip_ctx* ctx, Scratch& sc, const dthip_col* keys_dev, int nkeys, int64_t n, int na_pos,
                     KeyPlan* plan, bool speculative, bool tight) {
  if (nkeys < 1 || nkeys > MAX_KEYCOLS) { set_error("number of key columns must be 1..%d", MAX_KEYCOLS); return DTHIP_EINVAL; }
  plan->nkeys = nkeys;
  plan->speculative = false;
  if (n < ctx->spec_min_rows) speculative = false;   // below this the exact range scan is cheap enough
  MinMax* d_mm = nullptr;
  DTHIP_TRY(sc.get<MinMax>(nkeys, &d_mm));
  bool any_int = false;
  for (int k = 0; k < nkeys; k++) {
    const int st = keys_dev[k].stype;
    if (stype_size(st) == 0) { set_error("unsupported key stype %d", st); return DTHIP_ENOTIMPL; }
    if (st >= DTHIP_INT8 && st <= DTHIP_INT64) {
      if (speculative) DTHIP_TRY(launch_minmax_sample(ctx, keys_dev[k].data, st, n, SPEC_SAMPLES, d_mm + k));
      else DTHIP_TRY(launch_minmax(ctx, keys_dev[k].data, st, n, d_mm + k));
      any_int = true;
    }
  }
  MinMax mm[MAX_KEYCOLS];
  if (any_int) DTHIP_TRY(read_back(ctx, mm, d_mm, sizeof(MinMax) * nkeys));
  if (speculative) {
    // a sample that met nothing but NAs (a sparse, mostly-NA key column) says nothing about the valid keys in the rows
    // it skipped: such a column gets its exact range (a plan that is not speculative is never verified)
    for (int k = 0; k < nkeys; k++) {
      const int st = keys_dev[k].stype;
      if (st >= DTHIP_INT8 && st <= DTHIP_INT64 && mm[k].nvalid == 0)
        return plan_keys(ctx, sc, keys_dev, nkeys, n, na_pos, plan, false, false);
    }
  }
  for (int k = 0; k < nkeys; k++) {
    KeyColDev& c = plan->col[k];
    const int st = keys_dev[k].stype;
    c.data = keys_dev[k].data;
    c.stype = st;
    c.desc = (keys_dev[k].flags & DTHIP_FLAG_DESCENDING) ? 1 : 0;
    c.shift = 0;
    if (st == DTHIP_BOOL) {
      c.edge = 0; c.inc = 0; c.na_repl = (na_pos == DTHIP_NA_LAST) ? 3 : 0; c.xmax = ~0ULL;
      plan->nsig[k] = 2;
    } else if (st == DTHIP_FLOAT32) {
      c.edge = 0; c.inc = 0; c.na_repl = (na_pos == DTHIP_NA_LAST) ? 0xFFFFFFFFULL : 0; c.xmax = ~0ULL;
      plan->nsig[k] = 32;
    } else if (st == DTHIP_FLOAT64) {
      c.edge = 0; c.inc = 0; c.na_repl = (na_pos == DTHIP_NA_LAST) ? 0xFFFFFFFFFFFFFFFFULL : 0; c.xmax = ~0ULL;
      plan->nsig[k] = 64;
    } else {
      long long mn = mm[k].mn, mx = mm[k].mx;
      if (mm[k].nvalid == 0) { mn = 0; mx = 0; }
      if (speculative && mm[k].nvalid > 0) {
        // widen the sampled range by 1/64 of its width (+64) on both sides, inside the stype's range
        long long tlo, thi;
        stype_int_limits(st, &tlo, &thi);
        const unsigned long long width = (unsigned long long)mx - (unsigned long long)mn;
        const int nb_sample = nbits_u64(width + 1ULL);
        const unsigned long long margin = width / 64 + 64;
        mn = ((unsigned long long)mn - (unsigned long long)tlo > margin) ? (long long)((unsigned long long)mn - margin) : tlo;
        mx = ((unsigned long long)thi - (unsigned long long)mx > margin) ? (long long)((unsigned long long)mx + margin) : thi;
        plan->speculative = true;
        if (tight && nbits_u64((unsigned long long)mx - (unsigned long long)mn + 1ULL) != nb_sample)
          return plan_keys(ctx, sc, keys_dev, nkeys, n, na_pos, plan, false, false);
      }
      const unsigned long long range1 = (unsigned long long)mx - (unsigned long long)mn + 1ULL;
      c.edge = c.desc ? (unsigned long long)mx : (unsigned long long)mn;
      c.inc = (na_pos == DTHIP_NA_LAST) ? 0 : 1;
      c.na_repl = (na_pos == DTHIP_NA_LAST) ? range1 : 0;
      c.xmax = range1 - 1ULL;              // valid keys: [inc, inc + range1 - 1] (range1 == 0: all 2^64 values, wraps to ~0)
      const int nb = nbits_u64(range1);
      plan->nsig[k] = nb ? nb : 64;
    }
  }
  // stages, built from the least significant key backwards
  int stages_rev_first[MAX_KEYCOLS], stages_rev_last[MAX_KEYCOLS], stages_rev_bits[MAX_KEYCOLS];
  int ns = 0;
  int k = nkeys - 1;
  while (k >= 0) {
    int bits = 0, last = k;
    while (k >= 0 && bits + plan->nsig[k] <= 64) { bits += plan->nsig[k]; k--; }
    stages_rev_first[ns] = k + 1; stages_rev_last[ns] = last; stages_rev_bits[ns] = bits;
    ns++;
  }
  plan->nstages = ns;
  for (int s = 0; s < ns; s++) {
    plan->stage_first[s] = stages_rev_first[ns - 1 - s];
    plan->stage_last[s] = stages_rev_last[ns - 1 - s];
    plan->stage_bits[s] = stages_rev_bits[ns - 1 - s];
    int sh = 0;
    for (int j = plan->stage_last[s]; j >= plan->stage_first[s]; j--) { plan->col[j].shift = sh; sh += plan->nsig[j]; }
  }
  return DTHIP_OK;
}


// ---- MSD levels (round 4) --------------------------------------------------------------------------------------------
// The reference sorts most-significant digit first and finishes small buckets with a cheap local sort
// (sort.cc:1206-1353 _radix_recurse, sort_insert.cc:95-141).  Same shape here for big inputs: two STABLE scatter levels
// over the top S1 + S2 bits (the LSD pass kernel with its digit at the top; the second level works inside the buckets
// of the first: ragged tiles that never span two of them, run positions from a scan segmented by parent bucket), then
// every final bucket (<= one radix tile) is ordered by the remaining <= 9 bits in LDS and written back over its own row
// range -- sequential writes, no histogram pass, no run positions.  Stability comes from the passes themselves (every
// level is a stable partition), so no row id has to travel.  Against three LSD passes: the last pass loses its write
// amplification (a (tile, digit) run of 16 rows shares its first and last 64-byte sector with the neighbouring tiles'
// runs: 1.74x the algorithmic bytes reach HBM, 4.55 ms per pass of C5; written in place: 3.1 ms) and one histogram pass.
MsdPlan msd_plan(const dthip_ctx* ctx, int64_t n, int bits, int key64, uint32_t tile) {
  // measured (C5, 5e8 rows, 27 bits, MI355X, one box): levels 4.3 + 4.5 + final 4.4 ms (windows of whole buckets) and two
  // histogram passes against 3 x 4.7 ms of LSD passes and three: ~1 ms per call, more when the final level also writes the
  // original key column.  Below msd_min_rows the LSD passes are quick and the final buckets would be tiny.
  if (ctx->sort_path == 1 || key64 || (ctx->sort_path != 2 && n < ctx->msd_min_rows)) return MsdPlan();
  static const int rbmax = getenv("DTHIP_MSD_RBMAX") ? atoi(getenv("DTHIP_MSD_RBMAX")) : 9;
  return msd_split(n, bits, tile, ctx->msd_bucket_rows, rbmax);       // (host logic: csrc/msd_plan.hpp, tests/test_msd_plan.py)
}

// ---- windows of the final MSD level (whole buckets, together at most one tile of rows), planned on the device ----------
// Packed greedily per parent bucket (radix.hip msd_window_greedy_kernel; round 4's equal-step windows filled 66 % of a tile
// instead of 88 % and went in round 6).  ok = the windowed final level can run; maxsize = the largest final bucket either way.
int plan_windows(dthip_ctx* ctx, Scratch& sc, const uint32_t* fstart, uint32_t nb1, uint32_t bins2, int64_t n, const uint32_t* d_max,
                        uint32_t tile, int maxw, int rb, WindowPlan* wp) {
  // the (bucket, digit) counts of a window and their prefix live in the tile's exchange buffer: 2 x buckets x bins words
  uint32_t maxspan = 16;
  while (maxspan > 1 && (size_t)2 * maxspan * ((size_t)1 << rb) * 4 > (size_t)tile * maxw) maxspan >>= 1;
  const uint32_t nwmax = (uint32_t)(2 * (n / tile)) + nb1 + 8;
  uint32_t* wplan = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)3 * (nwmax + 2) + 4, &wplan));
  uint32_t* wbounds = wplan; uint32_t* wfirst = wplan + 2 * (nwmax + 2); uint32_t* winfo = wfirst + nwmax + 2;
  DTHIP_CHECK_HIP(hipMemsetAsync(winfo, 0, 4 * sizeof(uint32_t), ctx->stream));
  DTHIP_TRY(launch_msd_windows_greedy(ctx, fstart, nb1, bins2, tile, maxspan, nwmax, wbounds, wfirst, winfo));
  uint32_t wi[4] = {0, 0, 0, 0};                     // {~0 = infeasible, -, largest span, windows}
  DTHIP_TRY(read_back(ctx, wi, winfo, sizeof(wi)));
  DTHIP_TRY(read_back(ctx, &wp->maxsize, d_max, sizeof(uint32_t)));
  wp->bounds = wbounds; wp->wfirst = wfirst; wp->pairs = 1; wp->span = wi[2];
  wp->nwin = wi[3];
  wp->step = 0;
  wp->bits2 = 1;
  while ((1u << wp->bits2) < wi[2]) wp->bits2++;
  wp->ok = wp->nwin > 0 && wp->nwin <= nwmax && wi[2] >= 1 && wi[2] <= maxspan && wi[0] != 0xFFFFFFFFu &&
           (size_t)2 * ((size_t)1 << wp->bits2) * ((size_t)1 << rb) * 4 <= (size_t)tile * maxw;
  static const int win_env = getenv("DTHIP_MSD_WINDOWS") ? atoi(getenv("DTHIP_MSD_WINDOWS")) : 1;   // 0: one workgroup per bucket (A/B)
  if (win_env == 0) wp->ok = false;
  return DTHIP_OK;
}

// Stable sort of rows by one stage of packed keys, moving the payload columns along.
// `order` (nullable): the key columns are read through this ordering (later stages).
int sort_stage(dthip_ctx* ctx, Scratch& sc, const KeyPlan& plan, int stage, int64_t n,
                      const int32_t* order, const PaySpec& pay, SortOut* out) {
  const int bits = plan.stage_bits[stage];
  const int key64 = bits > 32;
  const size_t ksz = key64 ? 8 : 4;
  out->key64 = key64;
  // digits of up to 9 bits (512 bins) whenever that saves a pass: 27 significant bits are 3 passes of 9, 63 bits 7 x 9.
  // (Round 1 kept 32-bit keys at 8 bits -- a 9-bit pass measured 2x slower there: its per-wave histograms left LDS
  // for one workgroup per CU only; they are 16-bit words now.)
  int npass = (bits + 7) / 8;
  if ((bits + 8) / 9 < npass) npass = (bits + 8) / 9;
  if (npass > MAX_PASSES) npass = MAX_PASSES;
  XformArgs xa;
  memset(&xa, 0, sizeof(xa));
  xa.ncols = plan.stage_last[stage] - plan.stage_first[stage] + 1;
  for (int j = 0; j < xa.ncols; j++) xa.cols[j] = plan.col[plan.stage_first[stage] + j];
  xa.n = (uint32_t)n;
  xa.order = order;
  xa.out64 = key64;
  xa.npass = npass;
  const uint32_t tile = radix_tile_items(key64, 8);
  const MsdPlan msd = msd_plan(ctx, n, bits, key64, tile);
  if (msd.ok) {
    // digits, least significant first: what the final level orders in LDS, then the two scatter levels (the LSD passes
    // can run the same layout, so giving up on the MSD levels after the histograms costs nothing)
    npass = 3;
    xa.npass = 3;
    xa.pbits[0] = msd.rb > 9 ? 9 : msd.rb; xa.pbits[1] = msd.s2; xa.pbits[2] = msd.s1;      // (a histogram row has 512 bins)
    xa.pshift[0] = 0; xa.pshift[1] = msd.rb; xa.pshift[2] = msd.rb + msd.s2;
  } else {
    const int base = bits / npass, rem = bits % npass;
    int sh = 0;
    for (int p = 0; p < npass; p++) { xa.pbits[p] = base + (p < rem ? 1 : 0); xa.pshift[p] = sh; sh += xa.pbits[p]; }
  }
  unsigned char* kA = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)n * ksz, &kA));
  uint32_t* hist = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)2 * MAX_PASSES * HIST_STRIDE, &hist));
  uint32_t* base = hist + MAX_PASSES * HIST_STRIDE;
  DTHIP_CHECK_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * MAX_PASSES * HIST_STRIDE, ctx->stream));
  xa.out = kA;
  xa.hist = hist;
  // a plan made from a GUESSED key range (plan_keys: sampled min / max) is verified by this very pass: the word after
  // the last histogram row comes back non-zero when some key fell outside, and the caller plans again (exact range)
  static_assert(MAX_PASSES * 8 >= 64 + 8, "a histogram row stays free for the range check");
  xa.bad = plan.speculative ? hist + (size_t)npass * HIST_STRIDE : nullptr;
  DTHIP_TRY(launch_xform_hist(ctx, xa));
  // which passes actually permute anything?
  std::vector<uint32_t> hh((size_t)npass * HIST_STRIDE + 1);
  DTHIP_TRY(read_back(ctx, hh.data(), hist, hh.size() * sizeof(uint32_t)));
  if (plan.speculative && hh[(size_t)npass * HIST_STRIDE]) return DTHIP_RETRY_EXACT;
  int active[MAX_PASSES], nactive = 0;
  for (int p = 0; p < npass; p++) {
    bool constant = false;
    for (int d = 0; d < (1 << xa.pbits[p]); d++) if (hh[(size_t)p * HIST_STRIDE + d] == (uint32_t)n) constant = true;
    if (!constant) active[nactive++] = p;
  }
  out->npasses_run = nactive;
  for (int c = 0; c < pay.n; c++) out->pay[c] = const_cast<void*>(pay.in[c]);
  if (nactive == 0) {
    out->keys = kA;
    if (pay.iota) {
      int32_t* ri = nullptr;
      DTHIP_TRY(sc.get<int32_t>((size_t)n, &ri));
      DTHIP_TRY(launch_iota(ctx, ri, n));
      out->pay[0] = ri;
    }
    return DTHIP_OK;
  }
  DTHIP_TRY(launch_hist_scan(ctx, hist, base, npass));
  const uint32_t ntiles = (uint32_t)((n + tile - 1) / tile);
  // per-pass run positions: per-tile digit counts of the current key order -> P, gpre
  int maxbits = 0;
  for (int i = 0; i < nactive; i++) maxbits = std::max(maxbits, xa.pbits[active[i]]);
  BucketGeom hg;
  memset(&hg, 0, sizeof(hg));
  {
    const uint32_t gmax = (uint32_t)ctx->num_cus * 4;
    hg.ntiles = ntiles;
    hg.tpg = (ntiles + gmax - 1) / gmax; if (hg.tpg == 0) hg.tpg = 1;
    hg.G = (ntiles + hg.tpg - 1) / hg.tpg;
  }
  bool use_msd = msd.ok && nactive == 3;
  if (use_msd) {
    // The levels give up when a final bucket outgrows a tile, AFTER level 1 and two histogram passes.  The digit histograms
    // already on the host say when that is certain or likely, for nothing: rows can only land in (level-1 digit, level-2
    // digit) cells whose two marginal bins are non-empty, so fewer such cells than n / tile means an overflow for sure
    // (few distinct keys over a wide range); and if the two digits were independent the fullest cell would hold
    // max1 * max2 / n rows (a hot key, clustered keys).  Either way the LSD passes run at once.
    if (msd_overflow_expected(&hh[(size_t)2 * HIST_STRIDE], 1 << xa.pbits[2], &hh[(size_t)1 * HIST_STRIDE], 1 << xa.pbits[1], n, tile)) use_msd = false;
    if (getenv("DTHIP_MSD_DEBUG"))
      fprintf(stderr, "[dthip msd] n=%lld bits=%d digits %d+%d+%d -> %s\n", (long long)n, bits, msd.s1, msd.s2, msd.rb,
              use_msd ? "levels" : "LSD passes (overflow certain or likely)");
  } else if (getenv("DTHIP_MSD_DEBUG")) {
    fprintf(stderr, "[dthip msd] n=%lld bits=%d key64=%d: plan %s, active digits %d\n", (long long)n, bits, key64, msd.ok ? "ok" : "not applicable", nactive);
  }
  uint32_t* P = nullptr; uint32_t* gtot = nullptr; uint32_t* tot = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)(ntiles + (use_msd ? (2u << msd.s1) : 0u)) << maxbits, &P));
  DTHIP_TRY(sc.get<uint32_t>((size_t)hg.G << maxbits, &gtot));
  DTHIP_TRY(sc.get<uint32_t>((size_t)1 << maxbits, &tot));
  unsigned char* kB = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)n * ksz, &kB));
  void* pbuf[2][MAX_PAYCOLS];
  for (int c = 0; c < pay.n; c++) {
    unsigned char* b0 = nullptr;
    DTHIP_TRY(sc.get<unsigned char>((size_t)n * pay.width[c], &b0));
    pbuf[0][c] = b0;
    pbuf[1][c] = nullptr;
    if (nactive > 1) {
      unsigned char* b1 = nullptr;
      DTHIP_TRY(sc.get<unsigned char>((size_t)n * pay.width[c], &b1));
      pbuf[1][c] = b1;
    }
  }
  out->ukey_done = false;
  if (use_msd) {
    // ---- level 1: stable scatter by the top s1 bits (regular tiles) ------------------------------------------------
    const int p1 = 2, p2 = 1;
    const uint32_t nb1 = 1u << msd.s1, bins2 = 1u << msd.s2;
    hg.F = nb1;
    DTHIP_TRY(launch_radix_tile_hist(ctx, kA, key64, (uint32_t)n, xa.pshift[p1], xa.pbits[p1], ntiles, hg.tpg, hg.G, P, gtot));
    DTHIP_TRY(launch_bucket_gscan(ctx, hg, gtot, tot, base + (size_t)p1 * HIST_STRIDE, 1));
    RadixPass rp;
    memset(&rp, 0, sizeof(rp));
    rp.kin = kA; rp.kout = kB; rp.key64 = key64; rp.n = (uint32_t)n;
    rp.shift = xa.pshift[p1]; rp.bits = xa.pbits[p1];
    rp.P = P; rp.gpre = gtot; rp.tpg = hg.tpg;
    rp.iota = pay.iota ? 1 : 0;
    rp.pay.n = pay.n;
    for (int c = 0; c < pay.n; c++) { rp.pay.in[c] = pay.in[c]; rp.pay.out[c] = pbuf[0][c]; rp.pay.width[c] = pay.width[c]; }
    rp.label = "msd_level1_kernel";
    DTHIP_TRY(launch_radix_pass(ctx, rp));
    // ---- level 2: the same inside every level-1 bucket: ragged tiles, planned on the host from the level-1 histogram
    std::vector<uint32_t> tdesc, gdesc, gfirst;
    msd_level2_tiles(&hh[(size_t)p1 * HIST_STRIDE], nb1, tile, hg.tpg, &tdesc, &gdesc, &gfirst);
    const uint32_t ntiles2 = (uint32_t)(tdesc.size() / 4), G2 = (uint32_t)(gdesc.size() / 2);
    uint32_t* d_plan = nullptr;
    DTHIP_TRY(sc.get<uint32_t>(tdesc.size() + gdesc.size() + gfirst.size() + 4, &d_plan));
    uint32_t* d_tdesc = d_plan; uint32_t* d_gdesc = d_tdesc + tdesc.size(); uint32_t* d_gfirst = d_gdesc + gdesc.size();
    uint32_t* d_max = d_gfirst + gfirst.size();
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_tdesc, tdesc.data(), tdesc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_gdesc, gdesc.data(), gdesc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_gfirst, gfirst.data(), gfirst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DTHIP_CHECK_HIP(hipMemsetAsync(d_max, 0, 4, ctx->stream));
    uint32_t* gtot2 = nullptr; uint32_t* fstart = nullptr;
    DTHIP_TRY(sc.get<uint32_t>((size_t)(G2 ? G2 : 1) * bins2, &gtot2));
    DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 * bins2 + 1, &fstart));
    DTHIP_TRY(launch_radix_tile_hist(ctx, kB, key64, (uint32_t)n, xa.pshift[p2], xa.pbits[p2], ntiles2, hg.tpg, G2, P, gtot2, d_tdesc, d_gdesc));
    DTHIP_TRY(launch_msd_scan(ctx, gtot2, d_gfirst, base + (size_t)p1 * HIST_STRIDE, msd.s2, nb1, (uint32_t)n, fstart, d_max));
    // windows of the final level (whole buckets, together at most one tile of rows), planned on the device
    int maxw_w = 4;
    for (int c = 0; c < pay.n; c++) maxw_w = std::max(maxw_w, pay.width[c]);
    WindowPlan wp;
    DTHIP_TRY(plan_windows(ctx, sc, fstart, nb1, bins2, n, d_max, tile, maxw_w, msd.rb, &wp));
    const bool windows = wp.ok;
    const uint32_t maxsize = wp.maxsize;
    if (getenv("DTHIP_MSD_DEBUG"))
      fprintf(stderr, "[dthip msd] n=%lld s1=%d s2=%d rb=%d tiles2=%u groups2=%u largest bucket=%u windows=%u step=%u max buckets/window=%u -> %s\n",
              (long long)n, msd.s1, msd.s2, msd.rb, ntiles2, G2, maxsize, wp.nwin, wp.step, wp.span, windows ? "windows" : (maxsize <= tile ? "per bucket" : "LSD"));
    if (windows || maxsize <= tile) {
      rp.kin = kB; rp.kout = kA;
      rp.shift = xa.pshift[p2]; rp.bits = xa.pbits[p2];
      rp.P = P; rp.gpre = gtot2; rp.tpg = hg.tpg; rp.iota = 0;
      rp.ntiles = ntiles2; rp.tdesc = d_tdesc; rp.bounds = nullptr;
      for (int c = 0; c < pay.n; c++) { rp.pay.in[c] = pbuf[0][c]; rp.pay.out[c] = pbuf[1][c]; }
      rp.label = "msd_level2_kernel";
      DTHIP_TRY(launch_radix_pass(ctx, rp));
      // ---- final level: every bucket ordered by the remaining bits in LDS, written over its own rows
      rp.kin = kA; rp.kout = kB;
      rp.shift = 0; rp.bits = msd.rb;
      rp.P = nullptr; rp.gpre = nullptr;
      rp.ntiles = nb1 * bins2; rp.tdesc = nullptr; rp.bounds = fstart;
      static const int fb_env = getenv("DTHIP_MSD_FINAL_BLOCK") ? atoi(getenv("DTHIP_MSD_FINAL_BLOCK")) : 0;
      rp.block = (fb_env != 512 && maxsize <= tile / 2) ? 256 : 0;      // DTHIP_MSD_FINAL_BLOCK=512: A/B against the big workgroup
      if (windows) {
        // a bucket of ~2000 rows per workgroup leaves a CU with too few rows in flight (5.9 ms for C5's 5e8 rows); windows
        // of several whole buckets fill the tile (3.8 ms at ~6000 rows) at the price of a second ranking round in LDS
        rp.ntiles = wp.nwin; rp.bounds = wp.bounds; rp.wfirst = wp.wfirst; rp.block = 0;
        rp.bits2 = wp.bits2; rp.wpairs = wp.pairs;
      }
      for (int c = 0; c < pay.n; c++) { rp.pay.in[c] = pbuf[1][c]; rp.pay.out[c] = pbuf[0][c]; }
      if (pay.ukey_out) {
        const KeyColDev& kc = plan.col[plan.stage_first[stage]];
        rp.ukout = pay.ukey_out; rp.uk_stype = kc.stype; rp.uk_desc = kc.desc; rp.uk_bits = bits;
        rp.uk_edge = kc.edge; rp.uk_na_repl = kc.na_repl; rp.uk_inc = kc.inc;
        out->ukey_done = true;
      }
      rp.label = "msd_final_kernel";
      DTHIP_TRY(launch_radix_pass(ctx, rp));
      out->keys = kB;
      for (int c = 0; c < pay.n; c++) out->pay[c] = pbuf[0][c];
      return DTHIP_OK;
    }
    // a final bucket does not fit a tile (heavy duplicates / clustered keys): the LSD passes below start over from kA
    // and the caller's payload columns, which level 1 only read
    if (msd.rb > 9) { set_error("MSD levels with a 10-bit final digit (experiment) cannot fall back"); return DTHIP_ENOTIMPL; }
  }
  unsigned char* kin = kA; unsigned char* kout = kB;
  out->ukey_done = false;
  for (int i = 0; i < nactive; i++) {
    const int p = active[i];
    hg.F = 1u << xa.pbits[p];
    DTHIP_TRY(launch_radix_tile_hist(ctx, kin, key64, (uint32_t)n, xa.pshift[p], xa.pbits[p], ntiles, hg.tpg, hg.G, P, gtot));
    DTHIP_TRY(launch_bucket_gscan(ctx, hg, gtot, tot, base + (size_t)p * HIST_STRIDE, 1));
    RadixPass rp;
    memset(&rp, 0, sizeof(rp));
    rp.kin = kin; rp.kout = kout; rp.key64 = key64; rp.n = (uint32_t)n;
    rp.shift = xa.pshift[p]; rp.bits = xa.pbits[p];
    rp.P = P; rp.gpre = gtot; rp.tpg = hg.tpg;
    rp.iota = (i == 0 && pay.iota) ? 1 : 0;
    // (measured on C5: in an LSD pass the 8-byte key values are scattered runs like every other column -- the last pass
    // got 1.05 ms slower and the group scan reads 8 instead of 4 bytes (+0.43), which eats the 1.52 ms of the untransform
    // pass; the final MSD level writes in place and keeps 0.7 ms of it.  DTHIP_FUSE_UKEY=2 forces it here for A/B runs)
    static const bool fuse_lsd = getenv("DTHIP_FUSE_UKEY") && atoi(getenv("DTHIP_FUSE_UKEY")) == 2;
    if (i == nactive - 1 && pay.ukey_out && fuse_lsd) {
      const KeyColDev& kc = plan.col[plan.stage_first[stage]];
      rp.ukout = pay.ukey_out; rp.uk_stype = kc.stype; rp.uk_desc = kc.desc; rp.uk_bits = bits;
      rp.uk_edge = kc.edge; rp.uk_na_repl = kc.na_repl; rp.uk_inc = kc.inc;
      out->ukey_done = true;
    }
    rp.pay.n = pay.n;
    for (int c = 0; c < pay.n; c++) {
      rp.pay.in[c] = (i == 0) ? pay.in[c] : pbuf[(i - 1) & 1][c];
      rp.pay.out[c] = pbuf[i & 1][c];
      rp.pay.width[c] = pay.width[c];
    }
    DTHIP_TRY(launch_radix_pass(ctx, rp));
    std::swap(kin, kout);
  }
  out->keys = kin;
  for (int c = 0; c < pay.n; c++) out->pay[c] = pbuf[(nactive - 1) & 1][c];
  return DTHIP_OK;
}

// (rounds 4-5 carried a build flavour in which the final MSD level also marked the group heads: bit-exact, measured a LOSS --
// the level 4.75 -> 6.04 ms for a count_heads pass of 0.73 ms saved -- and removed in round 6)

int heads_to_offsets(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const void* keys, int key64,
                            const uint8_t* heads, int64_t n, Grouping* g) {
  const uint32_t nt = (uint32_t)((n + SEG_TILE - 1) / SEG_TILE);
  uint32_t* tile_counts = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)nt + 4 + nt / 8192, &tile_counts));
  unsigned long long* bitmap = nullptr;
  DTHIP_TRY(sc.get<unsigned long long>((size_t)((n + 63) / 64) + 1, &bitmap));
  int64_t ng = 0;
  DTHIP_TRY(launch_count_heads(ctx, keys, key64, heads, n, tile_counts, bitmap, tile_counts + nt, &ng));
  void* off = nullptr;
  DTHIP_TRY(result_alloc(ctx, res, sizeof(int32_t) * (size_t)(ng + 1), &off));
  DTHIP_TRY(launch_write_offsets(ctx, bitmap, n, tile_counts, ng, static_cast<int32_t*>(off)));
  g->n = n; g->ngroups = ng; g->offsets = static_cast<int32_t*>(off);
  g->bitmap = bitmap; g->tile_first = tile_counts;
  return DTHIP_OK;
}

// full group(): ordering + offsets (+ head bitmap) for any number of keys
int group_core(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const dthip_col* keys_dev, int nkeys,
                      int64_t n, int na_pos, KeyPlan* plan, Grouping* g) {
  // integer key ranges of big columns are guessed from a sample first (saves the exact min / max scan: 0.8 ms per 1e9-row
  // int64 column); the key-transform pass of every stage verifies the guess, a wrong one costs one more round
  const int32_t* order = nullptr;
  SortOut so;
  for (int attempt = 0; attempt < 2; attempt++) {
    DTHIP_TRY(plan_keys(ctx, sc, keys_dev, nkeys, n, na_pos, plan, attempt == 0, true));
    order = nullptr;
    int rc = DTHIP_OK;
    for (int s = plan->nstages - 1; s >= 0; s--) {
      PaySpec ps;
      ps.n = 1; ps.width[0] = 4;
      if (order) { ps.in[0] = order; ps.iota = false; } else { ps.in[0] = nullptr; ps.iota = true; }
      rc = sort_stage(ctx, sc, *plan, s, n, order, ps, &so);
      if (rc != DTHIP_OK) break;
      order = static_cast<const int32_t*>(so.pay[0]);
    }
    if (rc == DTHIP_RETRY_EXACT && attempt == 0) { ctx->call_stats[0]++; continue; }
    if (rc == DTHIP_RETRY_EXACT) { set_error("group: exact key range violated"); return DTHIP_EDEVICE; }
    DTHIP_TRY(rc);
    break;
  }
  g->rowindex = const_cast<int32_t*>(order);
  g->sorted_keys = so.keys; g->key64 = so.key64;
  if (plan->nstages == 1) {
    DTHIP_TRY(heads_to_offsets(ctx, sc, res, so.keys, so.key64, nullptr, n, g));
  } else {
    uint8_t* heads = nullptr;
    DTHIP_TRY(sc.get<uint8_t>((size_t)n, &heads));
    DTHIP_CHECK_HIP(hipMemsetAsync(heads, 0, (size_t)n, ctx->stream));
    for (int s = 0; s < plan->nstages; s++) {
      const int bits = plan->stage_bits[s];
      XformArgs xa;
      memset(&xa, 0, sizeof(xa));
      xa.ncols = plan->stage_last[s] - plan->stage_first[s] + 1;
      for (int j = 0; j < xa.ncols; j++) xa.cols[j] = plan->col[plan->stage_first[s] + j];
      xa.n = (uint32_t)n; xa.order = order; xa.out64 = bits > 32; xa.npass = 0;
      unsigned char* kk = nullptr;
      DTHIP_TRY(sc.get<unsigned char>((size_t)n * (xa.out64 ? 8 : 4), &kk));
      uint32_t* dummy = nullptr;
      DTHIP_TRY(sc.get<uint32_t>(16, &dummy));
      xa.out = kk; xa.hist = dummy;
      DTHIP_TRY(launch_xform_hist(ctx, xa));
      DTHIP_TRY(launch_mark_heads(ctx, kk, xa.out64, n, heads));
    }
    DTHIP_TRY(heads_to_offsets(ctx, sc, res, nullptr, 0, heads, n, g));
  }
  return DTHIP_OK;
}

}  // namespace dthip
