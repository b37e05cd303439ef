// agg.hip -- dthip_groupby_agg: DT[:, {reducers}, by(keys)] (EvalContext::evaluate, eval_context.cc:144-172,249-288,473-516)
// on three paths: bucketed aggregation (no sort), hash combiner (sparse keys), sort path (split out of api.hip in round 6)
#include <algorithm>
#include "host.hpp"

namespace dthip {

// ---- bucketed aggregation (bucket.hip): DT[:, aggs, by(keys)] without a sort -------------
// Accumulators each value column needs for the requested reducers.

// guess_nona: the column is believed to hold no NA (sampled): its valid count IS the group size, so the per-column
// counter (one DS atomic per row) is dropped and the kernels verify the belief on every row instead (ACC_CHKNA)
static int acc_flags_for(const dthip_agg* aggs, int naggs, int col, int vstype, int colflags = 0, bool guess_nona = false) {
  int f = (colflags & DTHIP_FLAG_NONA) ? ACC_NONA : 0;
  const bool isf = stype_is_float(vstype);
  for (int a = 0; a < naggs; a++) {
    if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != col) continue;
    switch (aggs[a].op) {
      case DTHIP_SUM: f |= ACC_SUM; break;
      case DTHIP_MEAN: f |= ACC_VCNT | (isf ? ACC_SUM : ACC_FSUM); break;
      case DTHIP_MIN: f |= ACC_MIN | ACC_VCNT; break;
      case DTHIP_MAX: f |= ACC_MAX | ACC_VCNT; break;
      case DTHIP_COUNT: f |= ACC_VCNT; break;
      default: break;
    }
  }
  if (guess_nona && (f & ACC_VCNT) && !(f & ACC_NONA)) f = (f & ~ACC_VCNT) | ACC_CHKNA;
  return f;
}

static int floor_log2_sz(size_t v) { int b = -1; while (v) { b++; v >>= 1; } return b; }

constexpr size_t BUCKET_LDS_TABLE = 144 * 1024;   // LDS bytes one aggregation table may take
constexpr size_t BUCKET_LDS_TABLE_HALF = 78 * 1024;   // ... when two workgroups are to share a CU
constexpr int BUCKET_MAX_R = 14;                  // slot keys are uint16
constexpr int BUCKET_MAX_D = 11;                  // <= 2048 buckets in one partition pass

static bool bucket_need_counts(const dthip_ctx* ctx, const dthip_agg* aggs, int naggs) {
  if (ctx->agg_offsets) return true;
  for (int a = 0; a < naggs; a++) if (aggs[a].op == DTHIP_COUNT0) return true;
  return false;
}

// Decides whether the bucket path applies; fills the slot-bit count r.
static bool bucket_eligible(const dthip_ctx* ctx, const KeyPlan& plan, const std::vector<dthip_col>& vd,
                            const std::vector<int>& used, const dthip_agg* aggs, int naggs, int64_t n, int* r_out, bool guess_nona = false) {
  if (ctx->agg_path == 1) return false;
  if (plan.nstages != 1) return false;
  const int B = plan.stage_bits[0];
  if (B > 32 || B < 1) return false;
  const int first_flag = (bucket_need_counts(ctx, aggs, naggs) || guess_nona) ? ACC_CNT : ACC_PRES;
  int r = BUCKET_MAX_R;
  bool first = true;
  // A table with three or more 8-byte accumulators per slot (sum + min + max: BASELINE C2) is bound by its LDS atomics, not by
  // HBM: capped at half the LDS, two workgroups share a CU and one's table set-up / flush hides behind the other's rows
  // (C2 table_agg 0.24 -> 0.22 ms per column, profiles/r06_c2_tab_ab.txt); DTHIP_TAB_KB overrides (A/B)
  static const size_t tab_cap_env = getenv("DTHIP_TAB_KB") ? (size_t)atoi(getenv("DTHIP_TAB_KB")) * 1024 : 0;
  auto fit = [&](int f, bool half_ok) {
    const size_t cap = tab_cap_env ? tab_cap_env : (half_ok && table_agg_slot_bytes(f) >= 24 ? BUCKET_LDS_TABLE_HALF : BUCKET_LDS_TABLE);
    int rc = BUCKET_MAX_R;
    while (rc > 0 && table_agg_lds_bytes(f, 1u << rc) > cap) rc--;
    return rc;
  };
  for (int pass = 0; pass < 2; pass++) {          // (the smaller tables must not cost a wide key range its place on this path)
    r = BUCKET_MAX_R; first = true;
    for (int c : used) {
      const int sz = stype_size(vd[c].stype);
      if (sz != 4 && sz != 8) return false;
      const int f = acc_flags_for(aggs, naggs, c, vd[c].stype, vd[c].flags, guess_nona) | (first ? first_flag : 0);
      first = false;
      r = std::min(r, fit(f, pass == 0));
    }
    if (first) r = std::min(r, fit(first_flag, pass == 0));
    if (B - std::min(r, B) <= BUCKET_MAX_D) break;
  }
  if (r > B) r = B;
  if (B - r > BUCKET_MAX_D) return false;
  // the dense accumulator arrays have 2^B slots: only worth it when the key range is dense enough
  if (ctx->agg_path != 2 && (1ULL << B) > 16ULL * (unsigned long long)n + 4096ULL) return false;
  if (ctx->agg_path != 2 && n < 4096) return false;
  *r_out = r;
  return true;
}

// ---- outlier keys of a guessed range (round 6) ----------------------------------------------------------------------------
// The range of a big integer key column is guessed from a sample (SPEC_SAMPLES pieces, widened by 1/64) and every row is
// checked.  Rounds 2-5 answered ONE key outside the guess with a second sweep over all rows (C3, one outlier: 2.1 x).  Now
// the partition lists such rows (at most OUTLIER_CAP; more than that means the guess was wrong, not that there are
// outliers) and leaves them out; here they are gathered, grouped by a nested call, and their groups -- all of them below or
// above every group of the range -- are spliced in: [NA group if NA comes first] [outliers before the range] [range]
// [outliers behind it] [NA group if NA comes last].  One key column (with several, an outlier in one column interleaves).
constexpr uint32_t OUTLIER_CAP = 1u << 16;

static int splice_outlier_groups(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const KeyPlan& plan, const std::vector<dthip_col>& kd,
                                 const std::vector<dthip_col>& vd, const dthip_agg* aggs, int naggs, const uint32_t* ovf_rows,
                                 uint32_t n_out, const uint32_t* d_cnt, bool cnt_is_counts, bool want_offsets) {
  const KeyColDev& kc = plan.col[0];
  const int kst = kd[0].stype, ksz = stype_size(kst);
  // the listed rows' key and value columns (their order in the list is arbitrary: the nested call orders them)
  unsigned char* ok = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)n_out * ksz + 16, &ok));
  DTHIP_TRY(launch_gather(ctx, kd[0].data, kst, reinterpret_cast<const int32_t*>(ovf_rows), n_out, ok));
  std::vector<dthip_col> ov(vd.size());
  for (size_t c = 0; c < vd.size(); c++) {
    ov[c] = vd[c];
    unsigned char* b = nullptr;
    DTHIP_TRY(sc.get<unsigned char>((size_t)n_out * stype_size(vd[c].stype) + 16, &b));
    DTHIP_TRY(launch_gather(ctx, vd[c].data, vd[c].stype, reinterpret_cast<const int32_t*>(ovf_rows), n_out, b));
    ov[c].data = b;
  }
  dthip_col k2 = kd[0]; k2.data = ok;
  const int na_pos = kc.na_repl == 0 ? DTHIP_NA_FIRST : DTHIP_NA_LAST;
  dthip_result* r2 = nullptr;
  const int saved_off = ctx->agg_offsets;
  const bool saved_merge = ctx->in_merge;
  ctx->agg_offsets = want_offsets ? 1 : 0; ctx->in_merge = true;      // (in_merge: no hash combiner, no list of its own)
  int rc = dthip_groupby_agg(ctx, &k2, 1, ov.empty() ? nullptr : ov.data(), (int)ov.size(), aggs, naggs, n_out, na_pos, DTHIP_DEVICE, &r2);
  ctx->agg_offsets = saved_off; ctx->in_merge = saved_merge;
  if (rc != DTHIP_OK) return rc;
  const int64_t ng2 = r2->ngroups, ng = res->ngroups;
  do {
    // groups of the list that come BEFORE the range: ascending keys below the minimum, descending keys above the maximum
    std::vector<unsigned char> hk((size_t)ng2 * ksz);
    if ((rc = read_back(ctx, hk.data(), r2->key[0], hk.size())) != DTHIP_OK) break;
    int64_t nA = 0;
    for (int64_t i = 0; i < ng2; i++) {
      long long v;
      switch (kst) {
        case DTHIP_BOOL: case DTHIP_INT8: v = reinterpret_cast<const int8_t*>(hk.data())[i]; break;
        case DTHIP_INT16: v = reinterpret_cast<const int16_t*>(hk.data())[i]; break;
        case DTHIP_INT32: v = reinterpret_cast<const int32_t*>(hk.data())[i]; break;
        default: v = reinterpret_cast<const long long*>(hk.data())[i]; break;
      }
      const bool before = kc.desc ? v > (long long)kc.edge : v < (long long)kc.edge;
      if (before) nA++; else break;          // (r2 is ordered the same way: the groups before the range lead it)
    }
    // is the first / last group of the range's result the NA group?
    bool has_na = false;
    if (ng > 0) {
      const size_t na_slot = (size_t)kc.na_repl;
      uint32_t w = 0;
      if ((rc = read_back(ctx, &w, d_cnt + (cnt_is_counts ? na_slot : na_slot / 32), sizeof(w))) != DTHIP_OK) break;
      has_na = cnt_is_counts ? w != 0 : ((w >> (na_slot & 31)) & 1u) != 0;
    }
    const int64_t naF = (has_na && na_pos == DTHIP_NA_FIRST) ? 1 : 0, naL = (has_na && na_pos == DTHIP_NA_LAST) ? 1 : 0;
    const int64_t mid = ng - naF - naL, nB = ng2 - nA, ngT = ng + ng2;
    // pieces of the spliced result: (source 0 = range result / 1 = outliers, first group, groups)
    struct Piece { int src; int64_t first, count; };
    const Piece pieces[5] = {{0, 0, naF}, {1, 0, nA}, {0, naF, mid}, {1, nA, nB}, {0, naF + mid, naL}};
    auto splice = [&](const void* a0, const void* a1, int elem, void** out) -> int {
      void* q = nullptr;
      DTHIP_TRY(result_alloc(ctx, res, (size_t)ngT * elem + 16, &q));
      size_t at = 0;
      for (const Piece& pc : pieces) {
        if (pc.count <= 0) continue;
        const unsigned char* src = static_cast<const unsigned char*>(pc.src ? a1 : a0) + (size_t)pc.first * elem;
        DTHIP_CHECK_HIP(hipMemcpyAsync(static_cast<unsigned char*>(q) + at, src, (size_t)pc.count * elem, hipMemcpyDeviceToDevice, ctx->stream));
        at += (size_t)pc.count * elem;
      }
      *out = q;
      return DTHIP_OK;
    };
    void* q = nullptr;
    if ((rc = splice(res->key[0], r2->key[0], ksz, &q)) != DTHIP_OK) break;
    res->key[0] = q;
    for (int a = 0; a < naggs && rc == DTHIP_OK; a++) {
      if ((rc = splice(res->agg[a], r2->agg[a], stype_size(res->agg_stype[a]), &q)) != DTHIP_OK) break;
      res->agg[a] = q;
    }
    if (rc != DTHIP_OK) break;
    if (res->offsets && r2->offsets) {
      // offsets: every piece keeps its group sizes, shifted to where the piece starts.  Piece starts in ROWS need the two
      // offset arrays at five places: read back those few words
      int32_t e0[4] = {0, 0, 0, 0}, e1[3] = {0, 0, 0};     // range: at naF, naF + mid, ng; outliers: at nA, ng2
      const int64_t i0[3] = {naF, naF + mid, ng}, i1[2] = {nA, ng2};
      for (int t = 0; t < 3 && rc == DTHIP_OK; t++) rc = read_back(ctx, &e0[t], res->offsets + i0[t], sizeof(int32_t));
      for (int t = 0; t < 2 && rc == DTHIP_OK; t++) rc = read_back(ctx, &e1[t], r2->offsets + i1[t], sizeof(int32_t));
      if (rc != DTHIP_OK) break;
      void* no = nullptr;
      if ((rc = result_alloc(ctx, res, sizeof(int32_t) * ((size_t)ngT + 2), &no)) != DTHIP_OK) break;
      int32_t* out = static_cast<int32_t*>(no);
      // rows before each piece in the spliced order
      const int32_t rows_naF = e0[0], rowsA = e1[0], rows_mid = e0[1] - e0[0], rowsB = e1[1] - e1[0];
      int64_t gat = 0; int32_t rat = 0;
      // piece 0: NA group (first)
      if ((rc = launch_offsets_piece(ctx, res->offsets, naF, 0, out)) != DTHIP_OK) break;
      gat += naF; rat += rows_naF;
      if ((rc = launch_offsets_piece(ctx, r2->offsets, nA, rat, out + gat)) != DTHIP_OK) break;
      gat += nA; rat += rowsA;
      if ((rc = launch_offsets_piece(ctx, res->offsets + naF, mid, rat - e0[0], out + gat)) != DTHIP_OK) break;
      gat += mid; rat += rows_mid;
      if ((rc = launch_offsets_piece(ctx, r2->offsets + nA, nB, rat - e1[0], out + gat)) != DTHIP_OK) break;
      gat += nB; rat += rowsB;
      // NA group (last) and the closing entry
      if ((rc = launch_offsets_piece(ctx, res->offsets + naF + mid, naL + 1, rat - e0[1], out + gat)) != DTHIP_OK) break;
      res->offsets = out;
    }
    res->ngroups = ngT;
  } while (0);
  result_destroy(ctx, r2);
  return rc;
}

static int bucket_groupby_agg(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const KeyPlan& plan,
                              const std::vector<dthip_col>& kd, const std::vector<dthip_col>& vd,
                              const std::vector<int>& used, const dthip_agg* aggs, int naggs, int64_t n, int r, bool guess_nona = false) {
  // guess_nona: value columns whose sample showed no NA keep no valid counter in LDS; an NA row that turns up after all is
  // skipped and counted apart by a global atomic (AggTable::nacnt) -- round 6: no second aggregation (rounds 4-5 aggregated
  // again, counting: C3 with mean() and one planted NaN 1.4 - 2.0x), DESIGN 6 "adversarial inputs"
  const int nkeys = plan.nkeys;
  const int B = plan.stage_bits[0];
  KeyXform kx;
  memset(&kx, 0, sizeof(kx));
  kx.ncols = nkeys;
  for (int k = 0; k < nkeys; k++) kx.cols[k] = plan.col[k];
  // vector key loads: up to 4 aligned key columns, all int64 or all int32, and aligned value columns
  int km = 0;
  if (nkeys <= 4) {
    bool all64 = true, all32 = true, aligned = true;
    for (int k = 0; k < nkeys; k++) {
      all64 &= kx.cols[k].stype == DTHIP_INT64;
      all32 &= kx.cols[k].stype == DTHIP_INT32;
      aligned &= (reinterpret_cast<uintptr_t>(kx.cols[k].data) & 15) == 0;
    }
    if (aligned && all64) km = 1;
    else if (aligned && all32) km = 2;
  }
  for (int c : used) if (reinterpret_cast<uintptr_t>(vd[c].data) & 15) km = 0;
  BucketGeom g;
  bucket_geometry(ctx, n, B, r, km, &g);
  const size_t nslots = (size_t)g.F * g.S;

  // want_offsets: group sizes are part of the result; need_cnt: rows per slot are COUNTED -- also when value columns are
  // guessed NA-free, whose valid counts the row counts then stand for
  const bool want_offsets = bucket_need_counts(ctx, aggs, naggs);
  bool need_cnt = want_offsets || guess_nona;
  int first_flag = need_cnt ? ACC_CNT : ACC_PRES;
  // --- partition (skipped when one table holds the whole key range) ---
  uint16_t* kpart = nullptr;
  std::vector<const void*> vsrc(vd.size(), nullptr);
  for (int c : used) vsrc[c] = vd[c].data;
  uint32_t* bbase = nullptr; WorkItem* items = nullptr; uint32_t* nitems = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)g.F + 6, &bbase));
  nitems = bbase + g.F + 1;
  uint32_t* d_bad = bbase + g.F + 2;
  uint32_t* d_clustered = bbase + g.F + 3;        // [2]
  uint32_t* nitems2 = bbase + g.F + 5;            // tile-local layout: the second work list (seg_plan_kernel)
  // SMALL path (one table of <= SMALL_SLOTS slots: BASELINE C1, 1e6 rows / 100 groups, is bound by its ~17 launches, not by
  // bytes): the plan kernel also initialises every table, and one single-workgroup kernel turns the slot counts into the
  // group list, the offsets and the group count
  const bool small = ctx->small_path != 0 && g.d == 0 && nslots <= SMALL_SLOTS;
  FillList fills;
  fills.n = 0;
  BigFill big;
  big.n = 0;
  bool defer_fills = false;
  auto fill = [&](void* p, size_t bytes, int byte) -> int {
    if (small && fills.n < 12 && (bytes & 3) == 0) {
      fills.p[fills.n] = static_cast<uint32_t*>(p); fills.words[fills.n] = (uint32_t)(bytes / 4); fills.val[fills.n] = byte ? 0xFFFFFFFFu : 0u;
      fills.n++;
      return DTHIP_OK;
    }
    // the general sequence: collected, ONE launch before the aggregation (what must be clear before the PARTITION runs --
    // the status word -- is set at once)
    // (buffers above 4 MB stay with the runtime's memset: C3's 134 MB table took 0.05-0.1 ms longer in this kernel)
    if (!small && defer_fills && bytes <= (4u << 20) && (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      if (big.n == 24) { DTHIP_TRY(launch_fill_list(ctx, big)); big.n = 0; }
      big.p[big.n] = static_cast<uint32_t*>(p); big.words[big.n] = bytes / 4; big.val[big.n] = byte ? 0xFFFFFFFFu : 0u;
      big.n++;
      return DTHIP_OK;
    }
    DTHIP_CHECK_HIP(hipMemsetAsync(p, byte, bytes, ctx->stream));
    return DTHIP_OK;
  };
  DTHIP_TRY(fill(d_bad, sizeof(uint32_t), 0));
  static const bool fill_one_launch = !(getenv("DTHIP_FILL_LIST") && atoi(getenv("DTHIP_FILL_LIST")) == 0);
  defer_fills = fill_one_launch;
  uint32_t M;
  {
    // with fewer buckets than CUs (BASELINE C2: 32) the aggregation is bound by its DS atomics, one 1024-thread workgroup
    // per CU: many small parts even out the tail (measured on 1e8 rows x 4 columns: 512 parts 0.29 ms per column,
    // 1500 parts 0.26; C2 3.07 -> 2.89 ms); with >= 1024 buckets the parts are whole buckets anyway
    static const int part_div_env = getenv("DTHIP_PART_DIV") ? atoi(getenv("DTHIP_PART_DIV")) : 0;
    const int part_div = part_div_env > 0 ? part_div_env : (g.F < (uint32_t)ctx->num_cus ? 16 : 4);
    const uint64_t denom = std::max<uint64_t>(g.F, (uint64_t)ctx->num_cus * part_div);
    uint64_t m = (2 * (uint64_t)n + denom - 1) / denom;
    // a part must amortise the set-up and the flush of its LDS table (S slots): small tables allow small parts, so a
    // 1e6-row frame with 100 groups (BASELINE C1) still spreads over a few hundred workgroups instead of 16
    const uint64_t m_min = std::min<uint64_t>(65536, std::max<uint64_t>(4096, 16ull * g.S));
    if (m < m_min) m = m_min;
    m = (m + 7) & ~7ULL;
    M = (uint32_t)std::min<uint64_t>(m, 0x7FFFFFF8ULL);
  }
  const uint32_t max_items = g.F + (uint32_t)((uint64_t)n / M) + 1 + 16;       // (+ the padding between seg_plan_kernel's two lists)
  DTHIP_TRY(sc.get<WorkItem>(max_items, &items));
  // sorted / clustered / constant keys? (decides which kernel variants run; one tiny read-back)
  bool clustered = ctx->cluster_mode == 2;
  bool even = ctx->bucket_variant == 3;           // rows spread evenly over the buckets (decides the tile-local layout)
  if (ctx->cluster_mode == 0 && n >= (1 << 20)) DTHIP_TRY(launch_bucket_cluster_sample(ctx, kx, n, g.r, d_clustered, &clustered, g.F, &even));
  int src = 1;
  // TILE-LOCAL layout: no histogram pass.  Every partition tile writes its rows, ordered by bucket, into its own row
  // range plus a 2-byte directory entry per bucket; the aggregation walks one short segment per tile.  Keys are read
  // once (16 B/row less HBM traffic for C3).  Random row order only: for sorted / clustered keys a bucket's rows sit in
  // few tiles and the exact-position layout (with its clustered kernel variants and row-range work items) is better.
  const uint16_t* dirT = nullptr; uint32_t dstride = 0;
  uint32_t* ovf_rows = nullptr; uint32_t* ovf_n = nullptr; uint32_t n_outliers = 0;
  // Worth it when the segments are short and alike: >= 1024 buckets (<= 12 rows of a tile per bucket) and no hot bucket
  // (sampled).  Measured on 1e9 rows: C3 9.5 -> 9.1 ms, C4 11.0 -> 9.4; but 4 x float64 columns over 32 buckets 2.8 -> 3.9
  // and a heavily skewed key 10.6 -> 14.1, which therefore keep the exact-position layout.
  // Round 6, third part: also with FEW buckets (<= 128: a tile's segment per bucket is >= 128 rows, streamed by whole waves --
  // table_agg_seg_kernel's long mode; a hot bucket only makes them longer): the histogram pass goes (C2 2.62 -> see DESIGN 6).
  // In between (256 / 512 buckets: 32- to 64-row segments suit neither mode) the exact-position layout stays, and so it does
  // for few buckets with a hot one: there table_agg_kernel combines the lanes that share a slot in registers, which the
  // all-long instance cannot do profitably (C2 shape, 30 % of the rows in one key: 3.23 against 3.75 ms, profiles/r06_long_combine_ab.txt).
  static const bool tl_few = !(getenv("DTHIP_TL_FEW") && atoi(getenv("DTHIP_TL_FEW")) == 0);
  const bool tile_local = g.d > 0 && !clustered && ctx->bucket_variant != 2 && g.block == 1024 &&
                          ((n >= (1 << 22) && ((g.F >= 1024 && even) || (g.F <= 128 && tl_few && even))) || ctx->bucket_variant == 3);   // variant 3: forced (tests)
  if (tile_local) {
    // round 6: 1024 x 16-row tiles (segments of 16 instead of 12 rows: fewer partly used sectors for the aggregation);
    // DTHIP_TL_ITEMS=12 keeps round 5's tiles (A/B)
    static const int tl_items = getenv("DTHIP_TL_ITEMS") ? atoi(getenv("DTHIP_TL_ITEMS")) : 16;
    // (few buckets keep the 12-row tiles: C2 2.67 -> 2.58 ms against the 16-row ones, profiles/r06_c2_tl_few_ab.txt)
    if (tl_items == 16 && g.F > 128) {
      int maxw = 4;
      for (int c : used) maxw = std::max(maxw, stype_size(vd[c].stype));
      (void)bucket_tl16_geometry(ctx, n, maxw, &g);
    }
    uint16_t* dir = nullptr; uint16_t* dT = nullptr; uint32_t* tot = nullptr;
    dstride = (g.ntiles + 63u) & ~63u;
    DTHIP_TRY(sc.get<uint16_t>((size_t)g.ntiles * (g.F + 1) + 8, &dir));
    DTHIP_TRY(sc.get<uint16_t>((size_t)dstride * (g.F + 2) + 8, &dT));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.F + 1, &tot));
    PayCols pc;
    memset(&pc, 0, sizeof(pc));
    DTHIP_TRY(sc.get<uint16_t>((size_t)g.ntiles * g.tile + 8, &kpart));
    for (int c : used) {
      unsigned char* vb = nullptr;
      const int w = stype_size(vd[c].stype);
      DTHIP_TRY(sc.get<unsigned char>((size_t)g.ntiles * g.tile * w + 64, &vb));
      pc.in[pc.n] = vd[c].data; pc.out[pc.n] = vb; pc.width[pc.n] = w; pc.n++;
      vsrc[c] = vb;
    }
    src = 2;
    // a guessed key range (one key column): rows outside it are LISTED instead of making the query start over
    // (DTHIP_OUTLIER_LIST=0: the second sweep of rounds 2-5, A/B)
    static const bool ovf_ok = !(getenv("DTHIP_OUTLIER_LIST") && atoi(getenv("DTHIP_OUTLIER_LIST")) == 0);
    if (ovf_ok && plan.speculative && nkeys == 1 && g.items == 16 && !ctx->in_merge) {
      DTHIP_TRY(sc.get<uint32_t>((size_t)OUTLIER_CAP + 4, &ovf_rows));
      ovf_n = ovf_rows + OUTLIER_CAP;
      DTHIP_CHECK_HIP(hipMemsetAsync(ovf_n, 0, sizeof(uint32_t), ctx->stream));
    }
    DTHIP_TRY(launch_bucket_partition(ctx, kx, n, g, nullptr, nullptr, kpart, pc, false, dir, d_bad, ovf_rows, ovf_n, OUTLIER_CAP));
    DTHIP_TRY(launch_dir_prepare(ctx, dir, g.ntiles, g.F, dT, dstride, tot, M, items, nitems, nitems2));
    dirT = dT;
  } else if (g.d > 0) {
    uint32_t* P = nullptr; uint32_t* gtot = nullptr; uint32_t* tot = nullptr;
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.ntiles * g.F, &P));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.G * g.F, &gtot));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.F, &tot));
    DTHIP_TRY(launch_bucket_hist(ctx, kx, n, g, P, gtot, d_bad, clustered));
    DTHIP_TRY(launch_bucket_gscan(ctx, g, gtot, tot, nullptr, 0));
    DTHIP_TRY(launch_bucket_plan(ctx, tot, g.F, 0, M, bbase, items, nitems));
    DTHIP_TRY(launch_bucket_gscan(ctx, g, gtot, tot, bbase, 1));
    PayCols pc;
    memset(&pc, 0, sizeof(pc));
    DTHIP_TRY(sc.get<uint16_t>((size_t)n + 8, &kpart));
    for (int c : used) {
      unsigned char* vb = nullptr;
      const int w = stype_size(vd[c].stype);
      DTHIP_TRY(sc.get<unsigned char>((size_t)n * w + 64, &vb));
      pc.in[pc.n] = vd[c].data; pc.out[pc.n] = vb; pc.width[pc.n] = w; pc.n++;
      vsrc[c] = vb;
    }
    src = 0;
    DTHIP_TRY(launch_bucket_partition(ctx, kx, n, g, P, gtot, kpart, pc, clustered));
  } else if (!small) {
    DTHIP_TRY(launch_bucket_plan(ctx, nullptr, 1, (uint32_t)n, M, bbase, items, nitems));
  }

  // --- dense accumulators + one aggregation launch per value column ---
  uint32_t* d_cnt = nullptr;      // rows per slot, or (no counts wanted) one presence bit per slot
  std::vector<AggTable> tabs(vd.size());
  std::vector<int> tflags(vd.size(), 0);
  int32_t* idx = nullptr;
  int64_t ng = 0;
  DTHIP_TRY(sc.get<int32_t>(std::min<size_t>(nslots, (size_t)n) + 1, &idx));
  const size_t cnt_words = need_cnt ? nslots : (nslots + 31) / 32;
  DTHIP_TRY(sc.get<uint32_t>(cnt_words, &d_cnt));
  DTHIP_TRY(fill(d_cnt, cnt_words * 4, 0));
  for (auto& t : tabs) t = AggTable();
  bool first = true;
  for (int c : used) {              // tables first (all of them: the small path initialises them in ONE kernel) ...
    AggTable& t = tabs[c];
    int f = acc_flags_for(aggs, naggs, c, vd[c].stype, vd[c].flags, guess_nona);
    if (first) { f |= first_flag; if (need_cnt) t.cnt = d_cnt; else t.pres = d_cnt; }
    first = false;
    tflags[c] = f;
    if (f & ACC_SUM) { DTHIP_TRY(sc.get<unsigned long long>(nslots, &t.sum)); DTHIP_TRY(fill(t.sum, nslots * 8, 0)); }
    if (f & ACC_MIN) { DTHIP_TRY(sc.get<unsigned long long>(nslots, &t.mn)); DTHIP_TRY(fill(t.mn, nslots * 8, 0xFF)); }
    if (f & ACC_MAX) { DTHIP_TRY(sc.get<unsigned long long>(nslots, &t.mx)); DTHIP_TRY(fill(t.mx, nslots * 8, 0)); }
    if (f & ACC_FSUM) { DTHIP_TRY(sc.get<double>(nslots, &t.fsum)); DTHIP_TRY(fill(t.fsum, nslots * 8, 0)); }
    if (f & ACC_VCNT) { DTHIP_TRY(sc.get<uint32_t>(nslots, &t.vcnt)); DTHIP_TRY(fill(t.vcnt, nslots * 4, 0)); }
    if (f & ACC_CHKNA) { DTHIP_TRY(sc.get<uint32_t>(nslots, &t.nacnt)); DTHIP_TRY(fill(t.nacnt, nslots * 4, 0)); }
  }
  if (small) DTHIP_TRY(launch_bucket_plan(ctx, nullptr, 1, (uint32_t)n, M, bbase, items, nitems, &fills));
  if (big.n) { DTHIP_TRY(launch_fill_list(ctx, big)); big.n = 0; }
  first = true;
  for (int c : used) {              // ... then one aggregation launch per value column
    const AggTable& t = tabs[c];
    const int f = tflags[c];
    if (src == 2) {
      TableAggSegArgs sa;
      memset(&sa, 0, sizeof(sa));
      sa.items = items; sa.nitems = nitems; sa.nitems2 = nitems2; sa.max_items = max_items; sa.kpart = kpart; sa.val = vsrc[c]; sa.vstype = vd[c].stype;
      sa.dirT = dirT; sa.dstride = dstride; sa.tile_rows = g.tile; sa.S = g.S; sa.flags = f; sa.tab = t; sa.bad = d_bad;
      sa.all_long = g.F <= 128;
      DTHIP_TRY(launch_table_agg_seg(ctx, sa));
      first = false;
      continue;
    }
    TableAggArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.items = items; ta.nitems = nitems; ta.max_items = max_items; ta.src = src;
    ta.kpart = kpart; ta.kx = kx; ta.val = vsrc[c]; ta.vstype = vd[c].stype; ta.S = g.S; ta.flags = f; ta.tab = t; ta.bad = d_bad; ta.clustered = clustered;
    DTHIP_TRY(launch_table_agg(ctx, ta));
    first = false;
  }
  if (first && src == 2) {
    TableAggSegArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.items = items; sa.nitems = nitems; sa.nitems2 = nitems2; sa.max_items = max_items; sa.kpart = kpart; sa.val = nullptr; sa.vstype = DTHIP_INT32;
    sa.dirT = dirT; sa.dstride = dstride; sa.tile_rows = g.tile; sa.S = g.S; sa.flags = first_flag; sa.all_long = g.F <= 128;
    if (need_cnt) sa.tab.cnt = d_cnt; else sa.tab.pres = d_cnt;
    DTHIP_TRY(launch_table_agg_seg(ctx, sa));
  } else if (first) {   // no value column at all: row counts (or key presence) alone
    TableAggArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.items = items; ta.nitems = nitems; ta.max_items = max_items; ta.src = src;
    ta.kpart = kpart; ta.kx = kx; ta.val = nullptr; ta.vstype = DTHIP_INT32; ta.S = g.S; ta.flags = first_flag;
    if (need_cnt) ta.tab.cnt = d_cnt; else ta.tab.pres = d_cnt;
    ta.bad = d_bad; ta.clustered = clustered;
    DTHIP_TRY(launch_table_agg(ctx, ta));
  }

  // --- groups = non-empty slots in slot order ---
  PredArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.data = d_cnt; pa.stype = DTHIP_INT32; pa.cmp = DTHIP_GT; pa.ci = 0; pa.is_mask = need_cnt ? 0 : 2;
  ng = 0;
  if (small) {
    void* off = nullptr;
    if (want_offsets) DTHIP_TRY(result_alloc(ctx, res, sizeof(int32_t) * (nslots + 2), &off));
    SmallGroupsArgs ga;
    ga.cnt = d_cnt; ga.bits = need_cnt ? 0 : 1; ga.nslots = (uint32_t)nslots; ga.idx = idx;
    ga.off = static_cast<uint32_t*>(off); ga.bad = d_bad;
    uint32_t w[2] = {0, 0};
    if (ctx->small_path == 2 && host_words(ctx)) {
      // the kernel writes its two words straight into mapped host memory: no copy command, one stream wait
      ga.out = ctx->host_words_dev;
      DTHIP_TRY(launch_small_groups(ctx, ga));
      DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      w[0] = reinterpret_cast<volatile uint32_t*>(ctx->host_words)[0];
      w[1] = reinterpret_cast<volatile uint32_t*>(ctx->host_words)[1];
    } else {
      ga.out = d_clustered;                        // its two words were read before the partition and are free now
      DTHIP_TRY(launch_small_groups(ctx, ga));
      DTHIP_TRY(read_back(ctx, w, d_clustered, sizeof(w)));
    }
    if (plan.speculative && (w[1] & 1u)) return DTHIP_RETRY_EXACT;
    ng = w[0];
    res->offsets = static_cast<int32_t*>(off);
  } else {
    DTHIP_TRY(launch_compact(ctx, pa, (int64_t)nslots, idx, &ng));
    if (plan.speculative) {
      uint32_t bad = 0;
      DTHIP_TRY(read_back(ctx, &bad, d_bad, sizeof(bad)));
      if (bad & 1u) return DTHIP_RETRY_EXACT;
      if (ovf_rows) {
        DTHIP_TRY(read_back(ctx, &n_outliers, ovf_n, sizeof(n_outliers)));
        if (n_outliers > OUTLIER_CAP) return DTHIP_RETRY_EXACT;      // not a few outliers: the guess was simply wrong
      }
    }
  }
  res->nrows = n; res->ngroups = ng;
  if (want_offsets && !small) {
    // offsets = exclusive scan of the group sizes (Groupby offsets, groupby.h:54-91)
    void* off = nullptr;
    DTHIP_TRY(result_alloc(ctx, res, sizeof(int32_t) * ((size_t)ng + 2 + (size_t)ng / 8192 + 1), &off));
    DTHIP_TRY(launch_gather(ctx, d_cnt, DTHIP_INT32, idx, ng, off));
    DTHIP_TRY(launch_scan_tiles(ctx, static_cast<uint32_t*>(off), (uint32_t)ng, static_cast<uint32_t*>(off) + ng));
    res->offsets = static_cast<int32_t*>(off);
  }
  // group-key columns: the slot index is the packed transformed key
  for (int k = 0; k < nkeys; k++) {
    void* kp = nullptr;
    DTHIP_TRY(result_alloc(ctx, res, (size_t)ng * stype_size(kd[k].stype), &kp));
    res->key[k] = kp;
    DTHIP_TRY(launch_untransform_keys(ctx, idx, 0, nullptr, ng, plan.col[k], plan.nsig[k], kp));
  }
  for (int a = 0; a < naggs; a++) {
    void* ap = nullptr;
    DTHIP_TRY(result_alloc(ctx, res, (size_t)ng * stype_size(res->agg_stype[a]), &ap));
    res->agg[a] = ap;
  }
  for (int c : used) {
    TableFinArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.idx = idx; fa.ng = (uint32_t)ng; fa.tab = tabs[c]; fa.vstype = vd[c].stype;
    if (tflags[c] & ACC_CHKNA) fa.tab.vcnt = d_cnt;        // guessed NA-free: the valid count of a group is its size minus tab.nacnt
    std::vector<std::pair<int, int>> dups;
    int first_of_op[6] = {-1, -1, -1, -1, -1, -1};
    for (int a = 0; a < naggs; a++) {
      if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != c) continue;
      if (first_of_op[aggs[a].op] >= 0) { dups.push_back({a, first_of_op[aggs[a].op]}); continue; }
      first_of_op[aggs[a].op] = a;
      switch (aggs[a].op) {
        case DTHIP_SUM: fa.o_sum = res->agg[a]; break;
        case DTHIP_MEAN: fa.o_mean = res->agg[a]; break;
        case DTHIP_MIN: fa.o_min = res->agg[a]; break;
        case DTHIP_MAX: fa.o_max = res->agg[a]; break;
        default: fa.o_count = static_cast<int64_t*>(res->agg[a]); break;
      }
    }
    DTHIP_TRY(launch_table_finalize(ctx, fa));
    for (auto& d : dups)
      DTHIP_CHECK_HIP(hipMemcpyAsync(res->agg[d.first], res->agg[d.second], (size_t)ng * stype_size(res->agg_stype[d.first]),
                                     hipMemcpyDeviceToDevice, ctx->stream));
  }
  for (int a = 0; a < naggs; a++)
    if (aggs[a].op == DTHIP_COUNT0) DTHIP_TRY(launch_count0(ctx, res->offsets, ng, static_cast<int64_t*>(res->agg[a])));
  if (n_outliers) {
    DTHIP_TRY(splice_outlier_groups(ctx, sc, res, plan, kd, vd, aggs, naggs, ovf_rows, n_outliers, d_cnt, need_cnt, want_offsets));
    ctx->call_stats[4] += n_outliers;
  }
  return DTHIP_OK;
}


// a path gave up after it had started to fill `res` (the hash combiner's pass i > 0 on a table overflow): its buffers go
// back to the cache AND every pointer into them is cleared, so the path that takes over cannot hand out a dangling one
static void drop_partial_result(dthip_ctx* ctx, dthip_result* res) {
  for (void* p : res->owned) dev_release(ctx, p);
  res->owned.clear();
  res->offsets = nullptr; res->rowindex = nullptr; res->ngroups = 0;
  for (auto& k : res->key) k = nullptr;
  for (auto& a : res->agg) a = nullptr;
}

// ---- hash combiner for sparse keys (bucket.hip): partial groups + merge ------------------------
constexpr int HASH_PK_BITS = 24, HASH_R = 13;          // pseudo key: 2048 buckets by hash

// distinct-key estimate from a strided sample of m rows: group the sample with the ordinary path,
// invert  u = N (1 - exp(-m / N))  (u distinct keys among m draws from N equally likely keys)
static int estimate_distinct(dthip_ctx* ctx, const std::vector<dthip_col>& kd, int nkeys, int64_t n, int na_pos, double* est) {
  const int64_t m = std::min<int64_t>(n, 1 << 21);
  Scratch sc(ctx);
  int32_t* ri = nullptr;
  DTHIP_TRY(sc.get<int32_t>((size_t)m, &ri));
  DTHIP_TRY(launch_sample_rows(ctx, ri, m, n));
  std::vector<dthip_col> sk(nkeys);
  for (int k = 0; k < nkeys; k++) {
    unsigned char* b = nullptr;
    DTHIP_TRY(sc.get<unsigned char>((size_t)m * stype_size(kd[k].stype), &b));
    DTHIP_TRY(launch_gather(ctx, kd[k].data, kd[k].stype, ri, m, b));
    sk[k] = kd[k];
    sk[k].data = b;
  }
  dthip_result* r = nullptr;
  DTHIP_TRY(dthip_groupby(ctx, sk.data(), nkeys, m, na_pos, DTHIP_DEVICE, 0, &r));
  const double u = (double)dthip_result_ngroups(r);
  dthip_result_free(ctx, r);
  if (m == n) { *est = u; return DTHIP_OK; }
  if (u > 0.97 * (double)m) { *est = 1e300; return DTHIP_OK; }     // (nearly) all distinct in the sample
  double lo = u, hi = 1e15;
  for (int it = 0; it < 200; it++) {
    const double mid = 0.5 * (lo + hi);
    const double f = mid * (1.0 - exp(-(double)m / mid));
    if (f < u) lo = mid; else hi = mid;
  }
  *est = hi;
  return DTHIP_OK;
}

static int hash_groupby_agg(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const KeyPlan& plan,
                            const std::vector<dthip_col>& kd, const std::vector<dthip_col>& vd,
                            const std::vector<int>& used, const dthip_agg* aggs, int naggs, int64_t n, int na_pos) {
  const int nkeys = plan.nkeys;
  if (ctx->hash_mode == 1 || ctx->in_merge || ctx->agg_path == 1) return DTHIP_NOT_APPLICABLE;
  // Several value columns (round 2): the rows are partitioned ONCE with every value column as payload; each column
  // then gets its own pass of LDS hash tables over the partitioned (key, value) rows and its own merge.  Every merge
  // orders the same set of keys, so the per-column results line up group by group.
  if (plan.nstages != 1 || (int)used.size() > MAX_PAYCOLS - 1) return DTHIP_NOT_APPLICABLE;
  for (int c : used) {
    if (vd[c].flags & DTHIP_FLAG_NONA) return DTHIP_NOT_APPLICABLE;
    if (stype_size(vd[c].stype) != 4 && stype_size(vd[c].stype) != 8) return DTHIP_NOT_APPLICABLE;
  }
  if (ctx->hash_mode < 2 && n < (1 << 22)) return DTHIP_NOT_APPLICABLE;
  const bool need_cnt = bucket_need_counts(ctx, aggs, naggs);
  const uint32_t F = 1u << (HASH_PK_BITS - HASH_R);
  double est = 0;
  DTHIP_TRY(estimate_distinct(ctx, kd, nkeys, n, na_pos, &est));
  // Passes of hash tables: (payload column, accumulator set).  A column's accumulators share one pass when an entry
  // (8-byte key + accumulators) is small enough for 2048 tables of load <= 0.75 to hold the estimated distinct keys;
  // otherwise they are split into {sum / mean / count}, {min}, {max} passes over the same partitioned rows.
  struct HPass { int col; int slot; int flags; uint32_t C; };
  auto table_entries = [](int flags) {
    const size_t entry = hash_agg_entry_bytes(flags);
    uint32_t P = ((uint32_t)((158 * 1024 - hash_agg_queue_bytes()) / entry) - 1) / 2;      // the whole LDS of a CU for one table (+ the waves' queues) ...
    for (;; P--) {                                                     // ... of P pairs of entries, P prime (double hashing over the pairs)
      bool prime = P % 2 != 0;
      for (uint32_t q = 3; prime && q * q <= P; q += 2) prime = P % q != 0;
      if (prime) break;
    }
    const uint32_t C = 2 * P;
    return C;
  };
  auto fits = [&](int flags) { return est * 1.05 <= 0.75 * (double)F * (double)table_entries(flags); };
  std::vector<HPass> passes;
  std::vector<int> agg_pass(naggs, 0);           // which pass computes aggregate a (count() rides with pass 0)
  if (used.empty()) {
    passes.push_back(HPass{-1, -1, need_cnt ? ACC_CNT : 0, 0});
  } else {
    for (size_t i = 0; i < used.size(); i++) {
      const int c = used[i];
      const int fl = acc_flags_for(aggs, naggs, c, vd[c].stype);
      const int first = (need_cnt && passes.empty()) ? ACC_CNT : 0;
      if (fits(fl | first)) {
        for (int a = 0; a < naggs; a++) if (aggs[a].op != DTHIP_COUNT0 && aggs[a].col == c) agg_pass[a] = (int)passes.size();
        passes.push_back(HPass{c, (int)i, fl | first, 0});
        continue;
      }
      // split: {sum / mean / count}, {min}, {max}; min and max keep the valid count their NA rule needs
      bool want[3] = {false, false, false};
      for (int a = 0; a < naggs; a++) {
        if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != c) continue;
        want[aggs[a].op == DTHIP_MIN ? 1 : aggs[a].op == DTHIP_MAX ? 2 : 0] = true;
      }
      const int parts[3] = {fl & (ACC_SUM | ACC_FSUM | ACC_VCNT), ACC_MIN | ACC_VCNT, ACC_MAX | ACC_VCNT};
      for (int q = 0; q < 3; q++) {
        if (!want[q]) continue;
        const int f2 = parts[q] | ((need_cnt && passes.empty()) ? ACC_CNT : 0);
        if (!fits(f2)) return DTHIP_NOT_APPLICABLE;
        for (int a = 0; a < naggs; a++) {
          if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != c) continue;
          if ((aggs[a].op == DTHIP_MIN ? 1 : aggs[a].op == DTHIP_MAX ? 2 : 0) == q) agg_pass[a] = (int)passes.size();
        }
        passes.push_back(HPass{c, (int)i, f2, 0});
      }
    }
  }
  for (auto& hp : passes) hp.C = table_entries(hp.flags);
  if (!fits(passes[0].flags)) return DTHIP_NOT_APPLICABLE;
  const int ncolpass = (int)passes.size();
  KeyXform kx;
  memset(&kx, 0, sizeof(kx));
  kx.ncols = nkeys;
  for (int k = 0; k < nkeys; k++) kx.cols[k] = plan.col[k];
  // one int64 key: the raw key IS a usable 64-bit image (round 3): no packed-key array is written (8 of the 20 bytes
  // per row hash_xform moved), the key column itself is payload 0 of the partition, and the partial groups' keys come
  // out typed already
  static const bool raw_ok = !(getenv("DTHIP_HASH_RAW") && atoi(getenv("DTHIP_HASH_RAW")) == 0);
  // (round 6: a float64 key column too -- its BITS are the image; the NaN patterns, one NA group in the reference, meet in
  // the merge, which groups the typed keys; -0.0 and 0.0 stay two groups as in the reference)
  const bool raw_key = raw_ok && nkeys == 1 && (kd[0].stype == DTHIP_INT64 || kd[0].stype == DTHIP_FLOAT64);
  // round 6: with a raw key the histogram and partition kernels HASH THE KEY COLUMN ON THE FLY (a two-multiply 24-bit hash,
  // keyxform.hpp hash_pk24) instead of reading a pseudo-key array that a pass of its own wrote: 12 of the 80 bytes per row
  // and one sweep less (DTHIP_HASH_FUSED=0: the pseudo-key pass of rounds 3-5, A/B)
  static const bool fused_ok = !(getenv("DTHIP_HASH_FUSED") && atoi(getenv("DTHIP_HASH_FUSED")) == 0);
  const bool fused_pk = raw_key && fused_ok;
  unsigned long long* xs = nullptr; int32_t* pk = nullptr;
  if (!fused_pk) DTHIP_TRY(sc.get<int32_t>((size_t)n + 4, &pk));
  if (fused_pk) {
  } else if (raw_key) {
    DTHIP_TRY(launch_hash_pk_raw(ctx, kd[0].data, n, pk));
  } else {
    DTHIP_TRY(sc.get<unsigned long long>((size_t)n + 2, &xs));
    DTHIP_TRY(launch_hash_xform(ctx, kx, n, xs, pk));
  }
  // the bucket machinery, driven by the pseudo key pk in [0, 2^24)
  KeyXform pkx;
  memset(&pkx, 0, sizeof(pkx));
  pkx.ncols = 1;
  pkx.cols[0].data = pk; pkx.cols[0].stype = DTHIP_INT32; pkx.cols[0].desc = 0; pkx.cols[0].edge = 0;
  pkx.cols[0].na_repl = 0; pkx.cols[0].inc = 0; pkx.cols[0].xmax = ~0ULL; pkx.cols[0].shift = 0;
  int km = 2;
  if (fused_pk) { pkx.cols[0].data = kd[0].data; pkx.cols[0].stype = DTHIP_KEY_HASH64; km = 1; }
  for (int c : used) if (reinterpret_cast<uintptr_t>(vd[c].data) & 15) km = 0;
  if (raw_key && (reinterpret_cast<uintptr_t>(kd[0].data) & 15)) km = 0;
  BucketGeom g;
  bucket_geometry(ctx, n, HASH_PK_BITS, HASH_R, km, &g);
  uint32_t* bbase = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)g.F + 8, &bbase));
  uint32_t* nitems = bbase + g.F + 1;
  uint32_t* d_bad = bbase + g.F + 2;
  uint32_t* d_outn = bbase + g.F + 3;
  uint32_t* d_ovf = bbase + g.F + 4;
  DTHIP_CHECK_HIP(hipMemsetAsync(bbase + g.F + 1, 0, 7 * sizeof(uint32_t), ctx->stream));
  uint32_t M;
  {
    const uint64_t denom = std::max<uint64_t>(g.F, (uint64_t)ctx->num_cus * 4);
    uint64_t m = (2 * (uint64_t)n + denom - 1) / denom;
    // a part must amortise the set-up and the flush of its LDS table (S slots): small tables allow small parts, so a
    // 1e6-row frame with 100 groups (BASELINE C1) still spreads over a few hundred workgroups instead of 16
    const uint64_t m_min = std::min<uint64_t>(65536, std::max<uint64_t>(4096, 16ull * g.S));
    if (m < m_min) m = m_min;
    m = (m + 7) & ~7ULL;
    M = (uint32_t)std::min<uint64_t>(m, 0x7FFFFFF8ULL);
  }
  const uint32_t max_items = g.F + (uint32_t)((uint64_t)n / M) + 1;
  WorkItem* items = nullptr;
  DTHIP_TRY(sc.get<WorkItem>(max_items, &items));
  // round 6: TILE-LOCAL partition (aligned columns: a raw int64 key hashed on the fly, or the int32 pseudo keys) -- 16384-row tiles written sequentially (each bucket one ~8-row
  // segment per tile + a 2-byte directory entry), no histogram pass; hash_agg_seg_kernel walks the segments
  // (DTHIP_HASH_TL=0: histogram + exact scatter positions as in rounds 2-5, A/B)
  static const bool tl_ok = !(getenv("DTHIP_HASH_TL") && atoi(getenv("DTHIP_HASH_TL")) == 0);
  bool tile_local = tl_ok && ctx->hash_mode != 3 && km != 0 && g.block == 1024 && (n >= (1 << 22) || ctx->hash_mode == 2);
  if (tile_local) {
    int maxw = 8;
    for (int c : used) maxw = std::max(maxw, stype_size(vd[c].stype));
    tile_local = bucket_tl16_geometry(ctx, n, maxw, &g);
  }
  const size_t part_rows = tile_local ? (size_t)g.ntiles * g.tile : (size_t)n;
  // one 8-byte value column: key and value travel as ONE 16-byte record (hash_partition_rec_kernel; DTHIP_HASH_REC=0: columns, A/B)
  static const bool rec_env = !(getenv("DTHIP_HASH_REC") && atoi(getenv("DTHIP_HASH_REC")) == 0);
  const bool rec_mode = tile_local && rec_env && fused_pk && used.size() == 1 && stype_size(vd[used[0]].stype) == 8;
  uint16_t* kslot = nullptr;         // not written: the packed key itself travels as payload 0
  unsigned long long* xs_part = nullptr;
  DTHIP_TRY(sc.get<unsigned long long>((rec_mode ? 2 * part_rows : part_rows) + 8, &xs_part));
  PayCols pc;
  memset(&pc, 0, sizeof(pc));
  pc.in[0] = raw_key ? kd[0].data : static_cast<const void*>(xs); pc.out[0] = xs_part; pc.width[0] = 8; pc.n = 1;
  std::vector<unsigned char*> v_part(std::max<size_t>(used.size(), 1), nullptr);
  for (size_t i = 0; i < used.size() && !rec_mode; i++) {
    const int w = stype_size(vd[used[i]].stype);
    DTHIP_TRY(sc.get<unsigned char>(part_rows * w + 64, &v_part[i]));
    pc.in[pc.n] = vd[used[i]].data; pc.out[pc.n] = v_part[i]; pc.width[pc.n] = w; pc.n++;
  }
  const uint16_t* dirT = nullptr; uint32_t dstride = 0;
  if (tile_local) {
    uint16_t* dir = nullptr; uint16_t* dT = nullptr; uint32_t* tot = nullptr;
    dstride = (g.ntiles + 63u) & ~63u;
    DTHIP_TRY(sc.get<uint16_t>((size_t)g.ntiles * (g.F + 1) + 8, &dir));
    DTHIP_TRY(sc.get<uint16_t>((size_t)dstride * (g.F + 2) + 8, &dT));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.F + 1, &tot));
    if (rec_mode) DTHIP_TRY(launch_hash_partition_rec(ctx, kd[0].data, vd[used[0]].data, n, g.r, g.F, g.ntiles, xs_part, dir));
    else DTHIP_TRY(launch_bucket_partition(ctx, pkx, n, g, nullptr, nullptr, kslot, pc, false, dir, d_bad, nullptr, nullptr, 0));
    DTHIP_TRY(launch_dir_prepare(ctx, dir, g.ntiles, g.F, dT, dstride, tot, M, items, nitems));
    dirT = dT;
  } else {
    uint32_t* P = nullptr; uint32_t* gtot = nullptr; uint32_t* tot = nullptr;
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.ntiles * g.F, &P));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.G * g.F, &gtot));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.F, &tot));
    DTHIP_TRY(launch_bucket_hist(ctx, pkx, n, g, P, gtot, d_bad, false));
    DTHIP_TRY(launch_bucket_gscan(ctx, g, gtot, tot, nullptr, 0));
    DTHIP_TRY(launch_bucket_plan(ctx, tot, g.F, 0, M, bbase, items, nitems));
    DTHIP_TRY(launch_bucket_gscan(ctx, g, gtot, tot, bbase, 1));
    DTHIP_TRY(launch_bucket_partition(ctx, pkx, n, g, P, gtot, kslot, pc, false));
  }

  int64_t ng_all = -1;
  for (int i = 0; i < ncolpass; i++) {
    Scratch sci(ctx);                        // this pass's partial groups
    const int c0 = passes[i].col;
    const int vst = c0 >= 0 ? vd[c0].stype : DTHIP_INT32;
    const int flags = passes[i].flags;
    const uint32_t C = passes[i].C;
    DTHIP_CHECK_HIP(hipMemsetAsync(d_outn, 0, 2 * sizeof(uint32_t), ctx->stream));
    // partial groups
    const size_t out_cap = std::min<size_t>((size_t)n, (size_t)max_items * (C + 1));
    HashAggArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.items = items; ha.nitems = nitems; ha.max_items = max_items; ha.xs = xs_part; ha.val = c0 >= 0 ? v_part[passes[i].slot] : nullptr; ha.vstype = vst;
    ha.C = C; ha.flags = flags; ha.out_n = d_outn; ha.out_cap = (uint32_t)out_cap; ha.overflow = d_ovf;
    DTHIP_TRY(sci.get<unsigned long long>(out_cap, &ha.o_key));
    if (flags & ACC_CNT) DTHIP_TRY(sci.get<uint32_t>(out_cap, &ha.o_tab.cnt));
    if (flags & ACC_VCNT) DTHIP_TRY(sci.get<uint32_t>(out_cap, &ha.o_tab.vcnt));
    if (flags & ACC_SUM) DTHIP_TRY(sci.get<unsigned long long>(out_cap, &ha.o_tab.sum));
    if (flags & ACC_MIN) DTHIP_TRY(sci.get<unsigned long long>(out_cap, &ha.o_tab.mn));
    if (flags & ACC_MAX) DTHIP_TRY(sci.get<unsigned long long>(out_cap, &ha.o_tab.mx));
    if (flags & ACC_FSUM) DTHIP_TRY(sci.get<double>(out_cap, &ha.o_tab.fsum));
    if (dirT) DTHIP_TRY(launch_hash_agg_seg(ctx, ha, dirT, dstride, g.tile, rec_mode));
    else DTHIP_TRY(launch_hash_agg(ctx, ha));
    uint32_t hn[2] = {0, 0};
    DTHIP_TRY(read_back(ctx, hn, d_outn, sizeof(hn)));       // {number of partial groups, overflow bits}
    if (hn[1]) { ctx->call_stats[2]++; return DTHIP_NOT_APPLICABLE; }      // a table filled up: the sort path takes over
    const int64_t np = hn[0];

    // typed columns of the partial groups
    std::vector<dthip_col> k2(nkeys);
    for (int k = 0; k < nkeys; k++) {
      k2[k] = kd[k];
      if (raw_key) { k2[k].data = ha.o_key; continue; }
      unsigned char* bb = nullptr;
      DTHIP_TRY(sci.get<unsigned char>((size_t)np * stype_size(kd[k].stype) + 16, &bb));
      DTHIP_TRY(launch_untransform_keys(ctx, ha.o_key, 1, nullptr, np, plan.col[k], plan.nsig[k], bb));
      k2[k].data = bb;
    }
    const bool isf = stype_is_float(vst);
    PartialColsArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.tab = ha.o_tab; pa.n = (uint32_t)np; pa.vstype = vst;
    std::vector<dthip_col> v2;
    std::vector<dthip_agg> a2;
    int iSUM = -1, iFSUM = -1, iMIN = -1, iMAX = -1, iVCNT = -1, iCNT = -1;
    // partial SUMS are merged with DTHIP_FLAG_NONA: a partial that is NaN (inf - inf) or wrapped to INT64_MIN is a value
    auto add_col = [&](void* data, int st, int op) { v2.push_back(dthip_col{data, st, op == DTHIP_SUM ? DTHIP_FLAG_NONA : 0}); a2.push_back(dthip_agg{op, (int32_t)v2.size() - 1}); return (int)a2.size() - 1; };
    if (flags & ACC_SUM) { DTHIP_TRY(sci.get<unsigned long long>((size_t)np + 2, &pa.o_sum)); iSUM = add_col(pa.o_sum, isf ? DTHIP_FLOAT64 : DTHIP_INT64, DTHIP_SUM); }
    if (flags & ACC_FSUM) { DTHIP_TRY(sci.get<double>((size_t)np + 2, &pa.o_fsum)); iFSUM = add_col(pa.o_fsum, DTHIP_FLOAT64, DTHIP_SUM); }
    if (flags & ACC_MIN) { unsigned char* bb = nullptr; DTHIP_TRY(sci.get<unsigned char>((size_t)np * 8 + 16, &bb)); pa.o_min = bb; iMIN = add_col(bb, vst, DTHIP_MIN); }
    if (flags & ACC_MAX) { unsigned char* bb = nullptr; DTHIP_TRY(sci.get<unsigned char>((size_t)np * 8 + 16, &bb)); pa.o_max = bb; iMAX = add_col(bb, vst, DTHIP_MAX); }
    if (flags & ACC_VCNT) { DTHIP_TRY(sci.get<int64_t>((size_t)np + 2, &pa.o_vcnt)); iVCNT = add_col(pa.o_vcnt, DTHIP_INT64, DTHIP_SUM); }
    if (flags & ACC_CNT) { DTHIP_TRY(sci.get<int64_t>((size_t)np + 2, &pa.o_cnt)); iCNT = add_col(pa.o_cnt, DTHIP_INT64, DTHIP_SUM); }
    DTHIP_TRY(launch_partial_columns(ctx, pa));

    // merge: the ordinary path on the partial groups (few rows), keys in their own stypes and flags
    dthip_result* r2 = nullptr;
    const int saved_off = ctx->agg_offsets;
    ctx->in_merge = true; ctx->agg_offsets = 0;
    int rc = dthip_groupby_agg(ctx, k2.data(), nkeys, v2.empty() ? nullptr : v2.data(), (int)v2.size(),
                               a2.empty() ? nullptr : a2.data(), (int)a2.size(), np, na_pos, DTHIP_DEVICE, &r2);
    ctx->in_merge = false; ctx->agg_offsets = saved_off;
    if (rc != DTHIP_OK) return rc;
    const int64_t ng = dthip_result_ngroups(r2);
    if (ng_all >= 0 && ng != ng_all) { dthip_result_free(ctx, r2); set_error("hash combiner: columns disagree on the number of groups"); return DTHIP_EDEVICE; }
    ng_all = ng;
    res->nrows = n; res->ngroups = ng;
    do {
      if (i == 0) {
        for (int k = 0; k < nkeys && rc == DTHIP_OK; k++) {
          void* kp = nullptr;
          const size_t bytes = (size_t)ng * stype_size(kd[k].stype);
          if ((rc = result_alloc(ctx, res, bytes, &kp)) != DTHIP_OK) break;
          res->key[k] = kp;
          if (bytes && hipMemcpyAsync(kp, dthip_result_key(r2, k), bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; }
        }
        if (rc != DTHIP_OK) break;
        if (need_cnt) {
          void* off = nullptr;
          if ((rc = result_alloc(ctx, res, sizeof(int32_t) * ((size_t)ng + 2 + (size_t)ng / 8192 + 1), &off)) != DTHIP_OK) break;
          if ((rc = launch_narrow_i64_u32(ctx, static_cast<const long long*>(dthip_result_agg(r2, iCNT)), ng, static_cast<uint32_t*>(off))) != DTHIP_OK) break;
          if ((rc = launch_scan_tiles(ctx, static_cast<uint32_t*>(off), (uint32_t)ng, static_cast<uint32_t*>(off) + ng)) != DTHIP_OK) break;
          res->offsets = static_cast<int32_t*>(off);
        }
      }
      for (int a = 0; a < naggs && rc == DTHIP_OK; a++) {
        // this pass fills the aggregates assigned to it; count() (no column) goes with the first pass
        const bool mine = aggs[a].op == DTHIP_COUNT0 ? i == 0 : agg_pass[a] == i;
        if (!mine) continue;
        void* ap = nullptr;
        const size_t bytes = (size_t)ng * stype_size(res->agg_stype[a]);
        if ((rc = result_alloc(ctx, res, bytes, &ap)) != DTHIP_OK) break;
        res->agg[a] = ap;
        if (ng == 0) continue;
        const void* src = nullptr;
        switch (aggs[a].op) {
          case DTHIP_SUM:
            if (vst == DTHIP_FLOAT32) rc = launch_cast_f64_f32(ctx, static_cast<const double*>(dthip_result_agg(r2, iSUM)), ng, static_cast<float*>(ap));
            else src = dthip_result_agg(r2, iSUM);
            break;
          case DTHIP_MEAN:
            rc = launch_mean_div(ctx, static_cast<const double*>(dthip_result_agg(r2, isf ? iSUM : iFSUM)),
                                 static_cast<const long long*>(dthip_result_agg(r2, iVCNT)), ng, ap, vst == DTHIP_FLOAT32);
            break;
          case DTHIP_MIN: src = dthip_result_agg(r2, iMIN); break;
          case DTHIP_MAX: src = dthip_result_agg(r2, iMAX); break;
          case DTHIP_COUNT: src = dthip_result_agg(r2, iVCNT); break;
          default: src = dthip_result_agg(r2, iCNT); break;      // COUNT0
        }
        if (rc == DTHIP_OK && src && hipMemcpyAsync(ap, src, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; }
      }
    } while (0);
    dthip_result_free(ctx, r2);
    if (rc != DTHIP_OK) return rc;
  }
  return DTHIP_OK;
}


}  // namespace dthip

using namespace dthip;

extern "C" {

int dthip_groupby_agg(dthip_ctx* ctx, const dthip_col* keys, int nkeys, const dthip_col* values, int nvalues,
                      const dthip_agg* aggs, int naggs, int64_t nrows, int na_pos, int mem, dthip_result** out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  CallScope call_scope(ctx);
  if (!keys || !out || (naggs > 0 && !aggs) || (nvalues > 0 && !values)) { set_error("null argument"); return DTHIP_EINVAL; }
  if (na_pos != DTHIP_NA_FIRST && na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", na_pos); return DTHIP_ENOTIMPL; }
  for (int a = 0; a < naggs; a++) {
    if (aggs[a].op < DTHIP_SUM || aggs[a].op > DTHIP_LAST) { set_error("bad reducer op %d", aggs[a].op); return DTHIP_EINVAL; }
    if (aggs[a].op != DTHIP_COUNT0 && (aggs[a].col < 0 || aggs[a].col >= nvalues)) {
      set_error("agg %d refers to value column %d of %d", a, aggs[a].col, nvalues); return DTHIP_EINVAL;
    }
  }
  dthip_result* res = new dthip_result();
  res->nkeys = nkeys; res->naggs = naggs;
  res->agg.assign(naggs, nullptr); res->agg_stype.assign(naggs, 0);
  for (int a = 0; a < naggs; a++)
    res->agg_stype[a] = dthip_reduce_out_stype(aggs[a].op, aggs[a].op == DTHIP_COUNT0 ? DTHIP_INT64 : values[aggs[a].col].stype);
  for (int k = 0; k < nkeys && k < MAX_KEYCOLS; k++) res->key_stype[k] = keys[k].stype;
  int rc = DTHIP_OK;
  do {
    Scratch sc(ctx);
    std::vector<dthip_col> kd, vd;
    if ((rc = stage_cols(ctx, sc, keys, nkeys, nrows, mem, &kd)) != DTHIP_OK) break;
    if ((rc = stage_cols(ctx, sc, values, nvalues, nrows, mem, &vd)) != DTHIP_OK) break;
    if (nrows == 0) { rc = empty_result(ctx, res); break; }
    // value columns actually referenced
    std::vector<int> used;
    for (int a = 0; a < naggs; a++)
      if (aggs[a].op != DTHIP_COUNT0 && std::find(used.begin(), used.end(), aggs[a].col) == used.end()) used.push_back(aggs[a].col);
    bool fused = (int)used.size() <= MAX_PAYCOLS;
    for (int c : used) if (stype_size(vd[c].stype) < 4) fused = false;
    for (int a = 0; a < naggs; a++) if (aggs[a].op == DTHIP_FIRST || aggs[a].op == DTHIP_LAST) fused = false;   // need the row order
    bool f32_seq = false;       // option "f32_sum": the reference's float32 accumulation needs the rows of a group in order
    for (int a = 0; a < naggs; a++)
      if (ctx->f32_sum_ref && aggs[a].op == DTHIP_SUM && vd[aggs[a].col].stype == DTHIP_FLOAT32) { f32_seq = true; fused = false; }
    KeyPlan plan; Grouping g;
    std::vector<const void*> sorted_val(nvalues, nullptr);
    const int32_t* gather_ri = nullptr;
    // round 6: rows that ARE in key order already (a time-ordered log grouped by day, the output of an earlier sort) need no
    // grouping pass at all.  One ascending int32 / int64 key, NA first: "non-descending as signed integers" is the grouped
    // order (NA = INT*_MIN leads).  16384 sampled pairs decide whether the full check is worth its pass; that pass
    // (count_heads_kernel over the raw column: group heads + a flag for any descent) IS the grouping when the flag stays
    // clear; the reducers then read the value columns in place (C3 with sorted keys: 13.4 -> see DESIGN 6).  DTHIP_PRESORTED=0: off
    bool presorted = false;
    static const bool presorted_ok = !(getenv("DTHIP_PRESORTED") && atoi(getenv("DTHIP_PRESORTED")) == 0);
    if (presorted_ok && fused && nkeys == 1 && (kd[0].stype == DTHIP_INT64 || kd[0].stype == DTHIP_INT32) &&
        !(kd[0].flags & DTHIP_FLAG_DESCENDING) && na_pos == DTHIP_NA_FIRST && nrows >= ((int64_t)1 << 22) && ctx->agg_path == 0 && !ctx->in_merge) {
      uint32_t* d_fl = nullptr;
      if ((rc = sc.get<uint32_t>(4, &d_fl)) != DTHIP_OK) break;
      bool maybe = false;
      if ((rc = launch_sorted_sample(ctx, kd[0].data, kd[0].stype, nrows, d_fl, &maybe)) != DTHIP_OK) break;
      if (maybe) {
        const uint32_t nt = (uint32_t)((nrows + SEG_TILE - 1) / SEG_TILE);
        uint32_t* tile_counts = nullptr; unsigned long long* bitmap = nullptr;
        if ((rc = sc.get<uint32_t>((size_t)nt + 4 + nt / 8192, &tile_counts)) != DTHIP_OK) break;
        if ((rc = sc.get<unsigned long long>((size_t)((nrows + 63) / 64) + 1, &bitmap)) != DTHIP_OK) break;
        int64_t ngs = 0; bool sorted = false;
        if ((rc = launch_count_heads_presorted(ctx, kd[0].data, kd[0].stype, nrows, tile_counts, bitmap, d_fl, &ngs, &sorted)) != DTHIP_OK) break;
        if (sorted) {
          void* off = nullptr;
          if ((rc = result_alloc(ctx, res, sizeof(int32_t) * (size_t)(ngs + 1), &off)) != DTHIP_OK) break;
          if ((rc = launch_write_offsets(ctx, bitmap, nrows, tile_counts, ngs, static_cast<int32_t*>(off))) != DTHIP_OK) break;
          g.n = nrows; g.ngroups = ngs; g.offsets = static_cast<int32_t*>(off); g.bitmap = bitmap; g.tile_first = tile_counts;
          for (int c : used) sorted_val[c] = vd[c].data;
          presorted = true;
          ctx->call_stats[3] = 6;
        } else {
          ctx->call_stats[2]++;          // the sample looked sorted, the column is not: one pass over the keys spent for nothing
        }
      }
    }
    if (presorted) {
    } else
    if (fused) {
      // first attempt: key ranges guessed from a sample (verified by the bucketed path); if that
      // path does not apply, or the guess was wrong, plan again with the exact ranges
      int slot_bits = 0;
      bool done = false, hash_tried = false;
      // value columns whose reducers need a valid count: guess from a sample that they hold no NA; the bucketed path then
      // drops their per-row counter and verifies the guess on every row (DTHIP_RETRY_NA: aggregate again, counting)
      bool guess_nona = false;
      if (ctx->nona_guess && nrows >= ((int64_t)1 << 20)) {
        uint32_t* d_na = nullptr;
        std::vector<int> cand;
        for (int c : used)
          if (!(vd[c].flags & DTHIP_FLAG_NONA) && (acc_flags_for(aggs, naggs, c, vd[c].stype) & ACC_VCNT)) cand.push_back(c);
        if (!cand.empty()) {
          if ((rc = sc.get<uint32_t>(1, &d_na)) != DTHIP_OK) break;
          if (hipMemsetAsync(d_na, 0, sizeof(uint32_t), ctx->stream) != hipSuccess) { set_error("memset failed"); rc = DTHIP_EDEVICE; break; }
          std::vector<const void*> cdata; std::vector<int> cst;
          for (int c : cand) { cdata.push_back(vd[c].data); cst.push_back(vd[c].stype); }
          if ((rc = launch_value_na_sample(ctx, cdata.data(), cst.data(), (int)cand.size(), nrows, d_na)) != DTHIP_OK) break;
          uint32_t seen = 0;
          if ((rc = read_back(ctx, &seen, d_na, sizeof(seen))) != DTHIP_OK) break;
          guess_nona = seen == 0;
        }
      }
      for (int attempt = (ctx->agg_path == 1 ? 1 : 0); attempt < 2 && !done; attempt++) {
        if ((rc = plan_keys(ctx, sc, kd.data(), nkeys, nrows, na_pos, &plan, attempt == 0)) != DTHIP_OK) break;
        if (attempt == 0 && !plan.speculative) attempt = 1;      // nothing was guessed: this IS the exact plan
        if (attempt == 0 && nkeys == 1 && kd[0].stype == DTHIP_INT64 && !(kd[0].flags & DTHIP_FLAG_DESCENDING) &&
            na_pos == DTHIP_NA_FIRST && plan.stage_bits[0] >= 36 && !hash_tried) {
          // one wide int64 key: the guessed range already rules the bucketed path out (the exact range is at most a
          // bit narrower), and the hash combiner needs no range at all -- x = key - (INT64_MIN + 1) + 1 covers every
          // valid key in 64 bits -- so the exact min/max scan of the whole column (1.6 ms per 1e9 rows) is skipped
          // unless the hash path turns the query down
          KeyPlan full = plan;
          full.speculative = false;
          full.col[0].edge = (unsigned long long)(INT64_MIN + 1); full.col[0].inc = 1; full.col[0].na_repl = 0;
          full.col[0].xmax = ~0ULL; full.col[0].shift = 0;
          full.nsig[0] = 64; full.nstages = 1; full.stage_first[0] = 0; full.stage_last[0] = 0; full.stage_bits[0] = 64;
          hash_tried = true;
          rc = hash_groupby_agg(ctx, sc, res, full, kd, vd, used, aggs, naggs, nrows, na_pos);
          if (rc == DTHIP_OK) { done = true; ctx->call_stats[3] = 3; break; }
          if (rc != DTHIP_NOT_APPLICABLE) break;
          rc = DTHIP_OK;
          drop_partial_result(ctx, res);
          continue;
        }
        if (bucket_eligible(ctx, plan, vd, used, aggs, naggs, nrows, &slot_bits, guess_nona)) {
          rc = bucket_groupby_agg(ctx, sc, res, plan, kd, vd, used, aggs, naggs, nrows, slot_bits, guess_nona);
          if (rc == DTHIP_RETRY_EXACT && attempt == 0) { ctx->call_stats[0]++; rc = DTHIP_OK; drop_partial_result(ctx, res); continue; }
          if (rc == DTHIP_RETRY_EXACT) { set_error("bucketed aggregation: exact key range violated"); rc = DTHIP_EDEVICE; }
          done = true;
          if (rc == DTHIP_OK) ctx->call_stats[3] = 2;
        }
      }
      if (rc != DTHIP_OK || done) break;
      // sparse keys (exact plan at this point): hash combiner + merge, when its tables are large enough
      if (!hash_tried) rc = hash_groupby_agg(ctx, sc, res, plan, kd, vd, used, aggs, naggs, nrows, na_pos);
      else rc = DTHIP_NOT_APPLICABLE;
      if (rc == DTHIP_OK) { ctx->call_stats[3] = 3; break; }
      if (rc != DTHIP_NOT_APPLICABLE) break;
      rc = DTHIP_OK;
      drop_partial_result(ctx, res);                       // nothing of a half-built attempt survives
      if (plan.nstages != 1) fused = false;
    }
    if (presorted) {
    } else if (fused) {
      // values ride through the sort; the RowIndex is never materialised
      PaySpec ps;
      ps.n = (int)used.size();
      for (int i = 0; i < ps.n; i++) { ps.in[i] = vd[used[i]].data; ps.width[i] = stype_size(vd[used[i]].stype); }
      SortOut so;
      ctx->call_stats[3] = 1;
      if ((rc = sort_stage(ctx, sc, plan, 0, nrows, nullptr, ps, &so)) != DTHIP_OK) break;
      for (int i = 0; i < ps.n; i++) sorted_val[used[i]] = so.pay[i];
      g.sorted_keys = so.keys; g.key64 = so.key64;
      if ((rc = heads_to_offsets(ctx, sc, res, so.keys, so.key64, nullptr, nrows, &g)) != DTHIP_OK) break;
    } else {
      ctx->call_stats[3] = 1;
      if ((rc = group_core(ctx, sc, res, kd.data(), nkeys, nrows, na_pos, &plan, &g)) != DTHIP_OK) break;
      for (int c : used) sorted_val[c] = vd[c].data;
      gather_ri = g.rowindex;
    }
    res->nrows = nrows; res->ngroups = g.ngroups; res->offsets = g.offsets;
    const int64_t ng = g.ngroups;
    // group-key columns: value of each key at the first row of its group
    for (int k = 0; k < nkeys; k++) {
      void* kp = nullptr;
      if ((rc = result_alloc(ctx, res, (size_t)ng * stype_size(kd[k].stype), &kp)) != DTHIP_OK) break;
      res->key[k] = kp;
      if (presorted) {
        rc = launch_gather(ctx, kd[k].data, kd[k].stype, g.offsets, ng, kp);       // the key at the first row of every group
      } else if (fused) {
        rc = launch_untransform_keys(ctx, g.sorted_keys, g.key64, g.offsets, ng, plan.col[k], plan.nsig[k], kp);
      } else {
        int32_t* firstrow = nullptr;
        if ((rc = sc.get<int32_t>((size_t)ng, &firstrow)) != DTHIP_OK) break;
        if ((rc = launch_gather(ctx, g.rowindex, DTHIP_INT32, g.offsets, ng, firstrow)) != DTHIP_OK) break;
        rc = launch_gather(ctx, kd[k].data, kd[k].stype, firstrow, ng, kp);
      }
      if (rc != DTHIP_OK) break;
    }
    if (rc != DTHIP_OK) break;
    // aggregates
    for (int a = 0; a < naggs && rc == DTHIP_OK; a++) {
      void* ap = nullptr;
      rc = result_alloc(ctx, res, (size_t)ng * stype_size(res->agg_stype[a]), &ap);
      res->agg[a] = ap;
    }
    if (rc != DTHIP_OK) break;
    for (int c : used) {
      ReduceOuts ro;
      std::vector<std::pair<int, int>> dups;   // (agg index, first agg index with same op)
      int first_of_op[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
      bool any_seg = false;
      for (int a = 0; a < naggs; a++) {
        if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != c) continue;
        if (aggs[a].op == DTHIP_FIRST || aggs[a].op == DTHIP_LAST) {
          if ((rc = launch_firstlast(ctx, sorted_val[c], vd[c].stype, gather_ri, g.offsets, ng, aggs[a].op == DTHIP_LAST,
                                     res->agg[a])) != DTHIP_OK) break;
          continue;
        }
        if (first_of_op[aggs[a].op] >= 0) { dups.push_back({a, first_of_op[aggs[a].op]}); continue; }
        first_of_op[aggs[a].op] = a;
        any_seg = true;
        if ((rc = reduce_outs_for(aggs[a].op, res->agg[a], &ro)) != DTHIP_OK) break;
      }
      if (rc != DTHIP_OK) break;
      if (any_seg)
        rc = launch_reduce(ctx, sorted_val[c], vd[c].stype, gather_ri, reinterpret_cast<const uint8_t*>(g.bitmap),
                           g.tile_first, nrows, ro, (vd[c].flags & DTHIP_FLAG_NONA) ? 1 : 0);
      if (rc != DTHIP_OK) break;
      for (auto& d : dups) {
        if (hipMemcpyAsync(res->agg[d.first], res->agg[d.second], (size_t)ng * stype_size(res->agg_stype[d.first]),
                           hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; break; }
      }
      if (rc != DTHIP_OK) break;
    }
    if (rc != DTHIP_OK) break;
    for (int a = 0; a < naggs; a++) {
      if (aggs[a].op != DTHIP_COUNT0) continue;
      if ((rc = launch_count0(ctx, g.offsets, ng, static_cast<int64_t*>(res->agg[a]))) != DTHIP_OK) break;
    }
    if (rc != DTHIP_OK) break;
    for (int a = 0; a < naggs && f32_seq; a++) {
      if (aggs[a].op != DTHIP_SUM || vd[aggs[a].col].stype != DTHIP_FLOAT32) continue;
      if ((rc = launch_sum_f32_seq(ctx, vd[aggs[a].col].data, gather_ri, g.offsets, ng, res->agg[a])) != DTHIP_OK) break;
    }
  } while (0);
  if (rc != DTHIP_OK) { result_destroy(ctx, res); return rc; }
  *out = res;
  return DTHIP_OK;
}

}  // extern "C"
