// bucket.hip -- DT[:, {sum,mean,min,max,count}, by(keys)] for DENSE key ranges without a sort.
//
// What the reference computes (src/core/expr/eval_context.cc:144-172, sort.cc:1411-1495,
// column/{sumprod,mean,minmax,count}.h): groups = distinct transformed keys in ascending
// order (NA first), one reducer value per group.  The reference gets there with an MSD
// radix sort producing an ordering vector (sort.cc:1128-1353) and then gathers every value
// through it.  When the transformed key x = key - min + 1 has few significant bits
// (B <= ~25: config C3 has 24, C2 17, C4 24), the groups can be produced with ONE scatter:
//
//   hist       per-tile digit counts of the top d = B - r bits (LDS atomics) -> the exact
//              global position of every (tile, bucket) run: no look-back, no spinning
//   partition  each 12288-row tile is ordered by bucket in LDS and written as runs; the
//              slot key (low r bits, uint16) and the value columns move together.
//              Tiles are dealt to XCDs in contiguous ranges so that the partially written
//              128-B lines of a bucket are completed by the same XCD's L2.
//   table_agg  one workgroup per bucket (or per <= M-row part of a big bucket) accumulates
//              count / sum / min / max into an LDS table of S = 2^r slots with DS atomics
//              (ds_add_u32, ds_add_f64 / ds_add_u64, ds_min_u64 / ds_max_u64 on an order-
//              preserving image), then stores the table to the dense accumulator arrays
//              (plain stores when the bucket has a single part, global atomics otherwise).
//   finalize   non-empty slots, in slot order, are the groups in key order (the slot index
//              IS the transformed key): compaction + gathers produce keys / offsets / aggs.
//
// HBM traffic for C3 (int64 key + float64 value): hist 8 B/row, partition 16 R + 10 W,
// table_agg 10 R  => 44 B/row against 3 x 24 + 16 for the LSD sort path.
// Integer results (counts, int sums, min/max, group keys/sizes) are exact and order
// independent; float sums are accumulated in a data-dependent order (<= 1e-6 relative,
// the tolerance BASELINE.json states).  Not usable when a RowIndex is requested.
#include <algorithm>
#include <functional>
#include "agg_dev.hpp"
#include "keyxform.hpp"

namespace dthip {


// ---------------------------------------------------------------------------------------
// tile loaders.  KM = 1: 16-B aligned int64 key column(s), 2 consecutive rows per lane per
// load; KM = 2: aligned int32 key column(s), 4 consecutive rows; KM = 0: anything, 1 row.
// Item j of thread tid is row  tile_base + ((j / VW) * BLOCK + tid) * VW + j % VW.
// ---------------------------------------------------------------------------------------
template <int KM> struct KmVW { static constexpr int value = KM == 1 ? 2 : KM == 2 ? 4 : 1; };

template <int BLOCK, int KM>
__device__ __forceinline__ uint32_t item_row(int j, int tid) {
  constexpr int VW = KmVW<KM>::value;
  return ((uint32_t)(j / VW) * BLOCK + (uint32_t)tid) * VW + (uint32_t)(j % VW);
}

// integer key transform (sort.cc:728-776) without branches: ascending u - edge, descending edge - u
__device__ __forceinline__ uint32_t xf_int(const KeyColDev& c, long long v, long long na, bool& bad) {
  if (c.stype == DTHIP_KEY_HASH64) return hash_pk24((u64)v);     // (uniform: the descriptor is a kernel argument)
  const u64 m = 0ULL - (u64)c.desc;              // all-ones when descending
  const u64 d0 = (((u64)v - c.edge) ^ m) - m;    // distance from the range's edge: [0, xmax] for a key inside the range
  const u64 t = (v == na) ? c.na_repl : d0 + c.inc;
  const bool b = v != na && d0 > c.xmax;         // only possible when the key range was guessed from a sample
  bad |= b;
  return b ? 0u : (uint32_t)t;
}

constexpr int VEC_KEYCOLS = 4;      // KM 1 / 2 handle up to this many key columns, all of the mode's type

template <int BLOCK, int ITEMS, int KM>
__device__ __forceinline__ void load_tile_x(const KeyXform& kx, uint32_t tile_base, uint32_t nvalid, bool full,
                                            int tid, uint32_t (&x)[ITEMS], bool& bad) {
  if (KM == 1 && full) {
#pragma unroll
    for (int j = 0; j < ITEMS; j++) x[j] = 0;
#pragma unroll
    for (int ci = 0; ci < VEC_KEYCOLS; ci++) {
      if (ci < kx.ncols) {
        const KeyColDev& c = kx.cols[ci];
        const long long* src = static_cast<const long long*>(c.data) + tile_base;
        bu32x4 w[ITEMS / 2];
#pragma unroll
        for (int q = 0; q < ITEMS / 2; q++) w[q] = *reinterpret_cast<const bu32x4*>(src + ((uint32_t)q * BLOCK + tid) * 2);
#pragma unroll
        for (int q = 0; q < ITEMS / 2; q++) {
          x[2 * q] |= xf_int(c, (long long)((u64)w[q].x | ((u64)w[q].y << 32)), INT64_MIN, bad) << c.shift;
          x[2 * q + 1] |= xf_int(c, (long long)((u64)w[q].z | ((u64)w[q].w << 32)), INT64_MIN, bad) << c.shift;
        }
      }
    }
  } else if (KM == 2 && full) {
#pragma unroll
    for (int j = 0; j < ITEMS; j++) x[j] = 0;
#pragma unroll
    for (int ci = 0; ci < VEC_KEYCOLS; ci++) {
      if (ci < kx.ncols) {
        const KeyColDev& c = kx.cols[ci];
        const int32_t* src = static_cast<const int32_t*>(c.data) + tile_base;
        bu32x4 w[ITEMS / 4];
#pragma unroll
        for (int q = 0; q < ITEMS / 4; q++) w[q] = *reinterpret_cast<const bu32x4*>(src + ((uint32_t)q * BLOCK + tid) * 4);
#pragma unroll
        for (int q = 0; q < ITEMS / 4; q++) {
          x[4 * q] |= xf_int(c, (long long)(int32_t)w[q].x, INT32_MIN, bad) << c.shift;
          x[4 * q + 1] |= xf_int(c, (long long)(int32_t)w[q].y, INT32_MIN, bad) << c.shift;
          x[4 * q + 2] |= xf_int(c, (long long)(int32_t)w[q].z, INT32_MIN, bad) << c.shift;
          x[4 * q + 3] |= xf_int(c, (long long)(int32_t)w[q].w, INT32_MIN, bad) << c.shift;
        }
      }
    }
    // a row with any out-of-range key must be key 0 as a whole (what the generic path does)
  } else {
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
      const uint32_t rel = item_row<BLOCK, KM>(j, tid);
      x[j] = 0u;
      if (rel < nvalid) x[j] = (uint32_t)packed_key_checked(kx.cols, kx.ncols, tile_base + rel, bad);
    }
  }
}

// The same for ONE key column of a vector-load mode, split into the loads (raw registers) and the transform, so that
// a persistent kernel can have the next tile's loads in flight while it works on the current one.
template <int BLOCK, int ITEMS, int KM>
__device__ __forceinline__ void load_raw1(const KeyColDev& c, uint32_t tile_base, int tid, bu32x4 (&w)[ITEMS / KmVW<KM>::value]) {
  constexpr int VW = KmVW<KM>::value;
  const bu32x4* src = reinterpret_cast<const bu32x4*>(static_cast<const unsigned char*>(c.data) + (size_t)tile_base * (16 / VW));
#pragma unroll
  for (int q = 0; q < ITEMS / VW; q++) w[q] = src[(uint32_t)q * BLOCK + tid];
}
template <int BLOCK, int ITEMS, int KM>
__device__ __forceinline__ void xform_raw1(const KeyColDev& c, const bu32x4 (&w)[ITEMS / KmVW<KM>::value], uint32_t (&x)[ITEMS], bool& bad) {
  constexpr int VW = KmVW<KM>::value;
#pragma unroll
  for (int q = 0; q < ITEMS / VW; q++) {
    if (KM == 1) {
      x[2 * q] = xf_int(c, (long long)((u64)w[q].x | ((u64)w[q].y << 32)), INT64_MIN, bad) << c.shift;
      x[2 * q + 1] = xf_int(c, (long long)((u64)w[q].z | ((u64)w[q].w << 32)), INT64_MIN, bad) << c.shift;
    } else {
      x[4 * q] = xf_int(c, (long long)(int32_t)w[q].x, INT32_MIN, bad) << c.shift;
      x[4 * q + 1] = xf_int(c, (long long)(int32_t)w[q].y, INT32_MIN, bad) << c.shift;
      x[4 * q + 2] = xf_int(c, (long long)(int32_t)w[q].z, INT32_MIN, bad) << c.shift;
      x[4 * q + 3] = xf_int(c, (long long)(int32_t)w[q].w, INT32_MIN, bad) << c.shift;
    }
  }
}

// payload column values of the thread's items (PT = uint32_t / u64), same row assignment
template <int BLOCK, int ITEMS, int KM, typename PT>
__device__ __forceinline__ void load_tile_vals(const PT* __restrict__ src, uint32_t nvalid, bool full, int tid,
                                               PT (&v)[ITEMS]) {
  constexpr int VW = KmVW<KM>::value;
  if (full && VW > 1) {
    constexpr int BYTES = VW * (int)sizeof(PT);       // contiguous bytes per lane per row group: 8, 16 or 32
#pragma unroll
    for (int q = 0; q < ITEMS / VW; q++) {
      const PT* p = src + ((uint32_t)q * BLOCK + tid) * VW;
      if (BYTES == 8) {
        const bu32x2 w = *reinterpret_cast<const bu32x2*>(p);
        v[VW * q] = (PT)w.x; v[VW * q + 1] = (PT)w.y;
      } else if (BYTES == 16 && sizeof(PT) == 8) {
        const bu32x4 w = *reinterpret_cast<const bu32x4*>(p);
        v[VW * q] = (PT)((u64)w.x | ((u64)w.y << 32)); v[VW * q + 1] = (PT)((u64)w.z | ((u64)w.w << 32));
      } else if (BYTES == 16) {
        const bu32x4 w = *reinterpret_cast<const bu32x4*>(p);
        v[VW * q] = (PT)w.x; v[VW * q + 1] = (PT)w.y; v[VW * q + 2] = (PT)w.z; v[VW * q + 3] = (PT)w.w;
      } else {
        const bu32x4 w0 = reinterpret_cast<const bu32x4*>(p)[0];
        const bu32x4 w1 = reinterpret_cast<const bu32x4*>(p)[1];
        v[VW * q] = (PT)((u64)w0.x | ((u64)w0.y << 32)); v[VW * q + 1] = (PT)((u64)w0.z | ((u64)w0.w << 32));
        v[VW * q + 2] = (PT)((u64)w1.x | ((u64)w1.y << 32)); v[VW * q + 3] = (PT)((u64)w1.z | ((u64)w1.w << 32));
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
      const uint32_t rel = item_row<BLOCK, KM>(j, tid);
      v[j] = rel < nvalid ? src[rel] : PT(0);
    }
  }
}

// Do neighbouring rows tend to fall into the same bucket (sorted / clustered / constant keys)?
// 65536 evenly spaced row pairs decide; the histogram and partition kernels then count a
// wave-uniform bucket with one DS atomic per wave instead of 64 serialised ones.  For random keys
// the per-row uniformity test would cost ~8 % of those kernels, hence the switch.
__global__ void __launch_bounds__(256) bucket_cluster_sample_kernel(KeyXform kx, uint32_t n, int r, uint32_t nsamp, uint32_t* acc,
                                                                    uint32_t* bcnt, uint32_t F) {
  // bucket counts of the block's 256 samples first in LDS: with few buckets (F = 32 for BASELINE C2) 65536 global
  // atomics on 32 addresses took 0.15 ms
  __shared__ uint32_t lh[2048];
  if (bcnt) for (uint32_t b = threadIdx.x; b < F; b += 256) lh[b] = 0;
  __syncthreads();
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  bool same = false;
  if (gid < nsamp && n > 1) {
    const uint32_t p = (uint32_t)(((unsigned long long)gid * (n - 1)) / nsamp);
    const unsigned long long x0 = packed_key(kx.cols, kx.ncols, p), x1 = packed_key(kx.cols, kx.ncols, p + 1);
    same = (x0 >> r) == (x1 >> r);
    // how evenly do the rows spread over the buckets?  (keys outside a guessed range land in bucket 0 like everywhere)
    if (bcnt) { const unsigned long long b = x0 >> r; atomicAdd(&lh[b < F ? (uint32_t)b : 0u], 1u); }
  }
  const unsigned long long b = __ballot(same);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&acc[1], (uint32_t)__popcll(b));
  __syncthreads();
  if (bcnt) for (uint32_t q = threadIdx.x; q < F; q += 256) { const uint32_t c = lh[q]; if (c) atomicAdd(&bcnt[q], c); }
}

// *clustered: neighbouring rows mostly share a bucket.  *even (nullable): no bucket holds more than ~2.5x its fair share
// of the sampled rows (the tile-local layout is only worth it then: its segments are short and all alike).
int launch_bucket_cluster_sample(dthip_ctx* ctx, const KeyXform& kx, int64_t n, int r, uint32_t* flag2, bool* clustered,
                                 uint32_t F, bool* even) {
  const uint32_t nsamp = 65536;
  Scratch sc(ctx);
  uint32_t* bcnt = nullptr;
  if (even) {
    DTHIP_TRY(sc.get<uint32_t>((size_t)F + 2, &bcnt));
    DTHIP_CHECK_HIP(hipMemsetAsync(bcnt, 0, sizeof(uint32_t) * ((size_t)F + 2), ctx->stream));
    flag2 = bcnt + F;
  } else {
    DTHIP_CHECK_HIP(hipMemsetAsync(flag2, 0, 2 * sizeof(uint32_t), ctx->stream));
  }
  DTHIP_LAUNCH(ctx, "bucket_cluster_sample_kernel", bucket_cluster_sample_kernel, nsamp / 256, 256, 0, kx, (uint32_t)n, r, nsamp, flag2,
               bcnt, F);
  if (!even) {
    uint32_t same = 0;
    DTHIP_TRY(read_back(ctx, &same, flag2 + 1, sizeof(same)));
    *clustered = same * 4u > nsamp;
    return DTHIP_OK;
  }
  std::vector<uint32_t> h((size_t)F + 2);
  DTHIP_TRY(read_back(ctx, h.data(), bcnt, sizeof(uint32_t) * h.size()));
  *clustered = h[F + 1] * 4u > nsamp;
  // fair share = sampled rows / buckets that got any (a key range that is not a power of two leaves the top buckets empty)
  uint32_t mx = 0, used = 0;
  for (uint32_t b = 0; b < F; b++) { mx = h[b] > mx ? h[b] : mx; used += h[b] ? 1u : 0u; }
  *even = (unsigned long long)mx * used * 2ull <= 5ull * nsamp + 64ull * used;
  if (!*even && F >= 1024) {
    // round 6: a FEW hot buckets (one hot key, the NA group, a handful of them) among otherwise even ones are fine too -- their
    // segments are long, table_agg_seg_kernel streams them with whole waves and their parts are dealt over all XCDs (seg_plan's
    // list A).  The same test without the four largest buckets; DTHIP_TL_HOT=0: as before (A/B)
    static const bool hot_ok = !(getenv("DTHIP_TL_HOT") && atoi(getenv("DTHIP_TL_HOT")) == 0);
    std::vector<uint32_t> c(h.begin(), h.begin() + F);
    std::partial_sort(c.begin(), c.begin() + 5, c.end(), std::greater<uint32_t>());
    unsigned long long rest = nsamp; uint32_t used2 = used;
    for (int i = 0; i < 4; i++) { rest -= c[i]; used2 -= c[i] ? 1u : 0u; }
    if (hot_ok && used2 >= 512 && (unsigned long long)c[4] * used2 * 2ull <= 5ull * rest + 64ull * used2) *even = true;
  }
  return DTHIP_OK;
}

// Does a value column hold an NA?  65536 evenly spaced elements answer "probably not"; the aggregation kernels then drop
// the per-column valid count (one DS atomic per row and column: BASELINE C2 runs 16 -> 13 of them) and VERIFY the guess on
// every row (ACC_CHKNA): a wrong guess costs a second aggregation, never a wrong result.
struct NaSampleCols { int n; const void* data[8]; int stype[8]; };
__global__ void __launch_bounds__(256) value_na_sample_kernel(NaSampleCols c, uint32_t n, uint32_t nsamp, uint32_t* flag) {
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  bool na = false;
  if (gid < nsamp) {
    const uint32_t p = (uint32_t)(((unsigned long long)gid * n) / nsamp);
#pragma unroll
    for (int i = 0; i < 8; i++) {            // (constant indices: the descriptors stay in scalar registers)
      if (i < c.n) {
        const void* data = c.data[i];
        switch (c.stype[i]) {
          case DTHIP_INT32: na |= static_cast<const int32_t*>(data)[p] == INT32_MIN; break;
          case DTHIP_INT64: na |= static_cast<const long long*>(data)[p] == INT64_MIN; break;
          case DTHIP_FLOAT32: { const float v = static_cast<const float*>(data)[p]; na |= v != v; break; }
          case DTHIP_FLOAT64: { const double v = static_cast<const double*>(data)[p]; na |= v != v; break; }
          default: na = true; break;             // a type the tables do not take: nothing is guessed
        }
      }
    }
  }
  if (__ballot(na) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// up to 8 columns per launch (round 6: C2's four columns in one launch instead of four)
int launch_value_na_sample(dthip_ctx* ctx, const void* const* data, const int* stype, int ncols, int64_t n, uint32_t* flag) {
  const uint32_t nsamp = 65536;
  for (int at = 0; at < ncols; at += 8) {
    NaSampleCols c;
    c.n = std::min(8, ncols - at);
    for (int i = 0; i < 8; i++) { c.data[i] = i < c.n ? data[at + i] : nullptr; c.stype[i] = i < c.n ? stype[at + i] : 0; }
    DTHIP_LAUNCH(ctx, "value_na_sample_kernel", value_na_sample_kernel, nsamp / 256, 256, 0, c, (uint32_t)n, nsamp, flag);
  }
  return DTHIP_OK;
}

// ---------------------------------------------------------------------------------------
// hist: per-tile bucket counts -> P[tile][b] = rows of bucket b in the earlier tiles of the
// same group; gtot[group][b] = rows of bucket b in the group.  One workgroup per group.
// ---------------------------------------------------------------------------------------
struct HistArgs {
  KeyXform kx; uint32_t n; int r; uint32_t F;
  uint32_t ntiles, tpg;
  uint32_t* P; uint32_t* gtot; uint32_t* bad;
};

// PIPE (one key column, vector-load key mode): the next tile's loads are issued before the current tile is counted.
template <int BLOCK, int ITEMS, int KM, bool CL, bool PIPE>
__global__ void __launch_bounds__(BLOCK) bucket_hist_kernel(HistArgs a) {
  constexpr uint32_t TILE = BLOCK * ITEMS;
  constexpr int NB = 2048 / BLOCK;          // bins per thread (F <= 2048)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // two counter arrays used alternately: the tile after next re-uses an array only after the
  // barrier of the tile in between, so ONE barrier per tile separates counting from reading
  uint32_t* cnt2 = reinterpret_cast<uint32_t*>(smem);
  const uint32_t Fp = (a.F + 3u) & ~3u;
  const int tid = threadIdx.x;
  uint32_t run[NB];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NB; k++) {
    run[k] = 0;
    const uint32_t b = (uint32_t)k * BLOCK + tid;
    if (b < a.F) { cnt2[b] = 0; cnt2[Fp + b] = 0; }
  }
  __syncthreads();
  const uint32_t t0 = blockIdx.x * a.tpg;
  const uint32_t t1 = (t0 + a.tpg < a.ntiles) ? t0 + a.tpg : a.ntiles;
  constexpr int NRAW = PIPE ? ITEMS / KmVW<KM>::value : 1;
  bu32x4 raw[NRAW];
  bool have = false;
  if (PIPE && t0 < t1 && a.n - t0 * TILE >= TILE) { load_raw1<BLOCK, ITEMS, KM>(a.kx.cols[0], t0 * TILE, tid, reinterpret_cast<bu32x4(&)[ITEMS / KmVW<KM>::value]>(raw)); have = true; }
  for (uint32_t t = t0; t < t1; t++) {
    uint32_t* cnt = cnt2 + ((t - t0) & 1u) * Fp;
    const uint32_t tile_base = t * TILE;
    const uint32_t nvalid = (a.n - tile_base < TILE) ? (a.n - tile_base) : TILE;
    const bool full = nvalid == TILE;
    uint32_t x[ITEMS];
    if (PIPE && have) {
      xform_raw1<BLOCK, ITEMS, KM>(a.kx.cols[0], reinterpret_cast<const bu32x4(&)[ITEMS / KmVW<KM>::value]>(raw), x, bad);
      have = false;
      if (t + 1 < t1 && a.n - (tile_base + TILE) >= TILE) {
        load_raw1<BLOCK, ITEMS, KM>(a.kx.cols[0], tile_base + TILE, tid, reinterpret_cast<bu32x4(&)[ITEMS / KmVW<KM>::value]>(raw));
        have = true;
      }
    } else {
      load_tile_x<BLOCK, ITEMS, KM>(a.kx, tile_base, nvalid, full, tid, x, bad);
    }
#pragma unroll
    for (int j = 0; j < ITEMS; j++)
      if (full || item_row<BLOCK, KM>(j, tid) < nvalid) {
        if (CL) (void)lds_count_rank(cnt, x[j] >> a.r);
        else atomicAdd(&cnt[x[j] >> a.r], 1u);
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NB; k++) {
      const uint32_t b = (uint32_t)k * BLOCK + tid;
      if (b < a.F) {
        const uint32_t c = cnt[b];
        cnt[b] = 0;
        a.P[(size_t)t * a.F + b] = run[k];
        run[k] += c;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NB; k++) {
    const uint32_t b = (uint32_t)k * BLOCK + tid;
    if (b < a.F) a.gtot[(size_t)blockIdx.x * a.F + b] = run[k];
  }
  if (__ballot(bad) && (tid & 63) == 0) atomicOr(a.bad, 1u);
}

// phase 0: tot[b] = rows of bucket b (column totals of gtot[g][b]).
// phase 1: gtot[g][b] <- bbase[b] + rows of bucket b in the groups before g (in place): the global
//          position of group g's first row of bucket b.
// One workgroup per 64 buckets: lane = bucket, each of the 16 waves owns a range of groups.
__global__ void __launch_bounds__(1024) bucket_gscan_kernel(uint32_t* gtot, uint32_t G, uint32_t F, uint32_t* tot,
                                                            const uint32_t* bbase, int phase) {
  __shared__ uint32_t part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t b = blockIdx.x * 64 + lane;
  const uint32_t per = (G + 15) / 16;
  const uint32_t g0 = wave * per, g1 = (g0 + per < G) ? g0 + per : G;
  uint32_t s = 0;
  if (b < F) for (uint32_t g = g0; g < g1; g++) s += gtot[(size_t)g * F + b];
  part[wave][lane] = s;
  __syncthreads();
  uint32_t off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { const uint32_t v = part[w][lane]; if (w < wave) off += v; total += v; }
  if (b < F) {
    if (phase == 0) {
      if (wave == 0) tot[b] = total;
    } else {
      off += bbase[b];
      for (uint32_t g = g0; g < g1; g++) {
        const uint32_t v = gtot[(size_t)g * F + b];
        gtot[(size_t)g * F + b] = off;
        off += v;
      }
    }
  }
}

// bucket sizes -> bucket starts + the aggregation work list (parts of <= M rows)
__global__ void __launch_bounds__(1024) bucket_plan_kernel(const uint32_t* tot, uint32_t F, uint32_t n_raw, uint32_t M,
                                                           uint32_t* bbase, WorkItem* items, uint32_t* nitems, FillList fl) {
  __shared__ uint32_t scratch[16];
  const int tid = threadIdx.x;
  for (int i = 0; i < fl.n; i++)
    for (uint32_t w = tid; w < fl.words[i]; w += 1024) fl.p[i][w] = fl.val[i];
  if (!tot && F == 1) {
    // ONE table for all rows (no partition): its parts are written by all threads -- one thread writing 245 items in a
    // row made this kernel the second longest (14 us) of a 1e6-row call (BASELINE C1)
    const uint32_t np1 = (n_raw + M - 1) / M;
    for (uint32_t i = tid; i < np1; i += 1024) {
      WorkItem it;
      it.bucket = 0; it.begin = i * M; it.end = (i + 1 == np1) ? n_raw : (i + 1) * M; it.single = np1 == 1 ? 1u : 0u;
      items[i] = it;
    }
    if (tid == 0) { bbase[0] = 0; bbase[1] = n_raw; *nitems = np1; }
    return;
  }
  uint32_t sz[2], np[2], s = 0, ps = 0;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t b = (uint32_t)tid * 2 + k;
    sz[k] = b < F ? (tot ? tot[b] : n_raw) : 0u;
    np[k] = (sz[k] + M - 1) / M;
    s += sz[k]; ps += np[k];
  }
  uint32_t stot, ptot;
  uint32_t e = block_excl_scan_u32<1024>(s, scratch, &stot);
  uint32_t pe = block_excl_scan_u32<1024>(ps, scratch, &ptot);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t b = (uint32_t)tid * 2 + k;
    if (b < F) {
      bbase[b] = e;
      for (uint32_t i = 0; i < np[k]; i++) {
        WorkItem it;
        it.bucket = b;
        it.begin = e + i * M;
        it.end = (i + 1 == np[k]) ? e + sz[k] : e + (i + 1) * M;
        // bit 0: the bucket has a single part (plain stores of the table); bit 2: it holds more than 3x its fair share of the
        // rows -- a hot key: table_agg_kernel combines the lanes that share a slot in registers (agg_dev.hpp acc_row)
        it.single = (np[k] == 1 ? 1u : 0u) | ((unsigned long long)sz[k] * F > 3ull * stot ? 4u : 0u);
        items[pe + i] = it;
      }
      e += sz[k]; pe += np[k];
    }
  }
  if (tid == 0) { bbase[F] = stot; *nitems = ptot; }
}

// ---------------------------------------------------------------------------------------
// partition
// ---------------------------------------------------------------------------------------
struct PartArgs {
  KeyXform kx; uint32_t n; int r; uint32_t F;
  uint32_t tpg;
  const uint32_t* P; const uint32_t* gpre;      // gpre[g][b]: global position of group g's first row of bucket b
  uint16_t* kout;
  PayCols pay;
  // TILE-LOCAL layout (no histogram pass): the tile's rows, ordered by bucket, are written to the tile's OWN row
  // range of the outputs; dir[tile][b] (b <= F) = first position of bucket b inside the tile.  With no histogram pass
  // before it, this kernel also reports keys outside a guessed key range (*bad).
  uint16_t* dir; uint32_t* bad;
  // 16-row tile-local instance, ONE key column with a guessed range: a row whose key lies outside the range is not a reason
  // to start over -- its row number goes to this list (*ovf_n counts; rows past ovf_cap are dropped: the caller retries then)
  // and the row into the extra bin behind the tile's valid rows, i.e. nowhere.  The caller aggregates the listed rows
  // separately and splices their groups in front of / behind the others (agg.hip splice_outlier_groups)
  uint32_t* ovf_rows; uint32_t* ovf_n; uint32_t ovf_cap;
};

// SEQ: the tile writes over ONE contiguous row range starting at seq_base (tile-local layout), so the global position of
// staged row s is seq_base + s and no position array is kept in registers (gpos is a dummy then)
template <int BLOCK, int ITEMS, int KM, typename PT, bool SEQ = false, int NG = ITEMS>
__device__ __forceinline__ void place_payload(const PT (&v)[ITEMS], PT* __restrict__ pout, unsigned char* stage,
                                              uint32_t nvalid, bool full, int tid, const uint32_t (&lpos)[ITEMS],
                                              const uint32_t (&gpos)[NG], uint32_t seq_base = 0) {
  PT* st = reinterpret_cast<PT*>(stage);
  __syncthreads();                                   // previous users of `stage` are done
#pragma unroll
  for (int j = 0; j < ITEMS; j++)
    if (SEQ || full || item_row<BLOCK, KM>(j, tid) < nvalid) st[lpos[j]] = v[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const uint32_t s = (uint32_t)j * BLOCK + tid;
    if (s < nvalid) pout[SEQ ? seq_base + s : gpos[SEQ ? 0 : j]] = st[s];
  }
}

template <int BLOCK, int ITEMS, int KM, typename PT, bool SEQ = false, int NG = ITEMS>
__device__ __forceinline__ void move_payload(const PT* __restrict__ pin, PT* __restrict__ pout, unsigned char* stage,
                                             uint32_t nvalid, bool full, int tid, const uint32_t (&lpos)[ITEMS],
                                             const uint32_t (&gpos)[NG], uint32_t seq_base = 0) {
  PT v[ITEMS];
  load_tile_vals<BLOCK, ITEMS, KM, PT>(pin, nvalid, full, tid, v);
  PT* st = reinterpret_cast<PT*>(stage);
  __syncthreads();                                   // previous users of `stage` are done
#pragma unroll
  for (int j = 0; j < ITEMS; j++)
    if (SEQ || full || item_row<BLOCK, KM>(j, tid) < nvalid) st[lpos[j]] = v[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const uint32_t s = (uint32_t)j * BLOCK + tid;
    if (s < nvalid) pout[SEQ ? seq_base + s : gpos[SEQ ? 0 : j]] = st[s];
  }
}

// PF: the first payload column is 8 bytes wide and its tile is loaded together with the keys, BEFORE the LDS
// phases -- one workgroup fills a CU (LDS), so nothing else hides the latency of a load issued after them.
template <int BLOCK, int ITEMS, int KM, bool CL, bool PF>
__global__ void __launch_bounds__(BLOCK) bucket_partition_kernel(PartArgs a) {
  constexpr uint32_t TILE = BLOCK * ITEMS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // TLS: 16-row items (16384-row tiles) exist for the tile-local layout only: the position array goes (compile time), and with
  // it the registers that kept 1024 x 16 rows from fitting 128 VGPRs; a bucket's segment is 16 rows instead of 12 (C3).
  // The rows past the end of a ragged last tile are ranked into an extra bin F behind the valid rows (its start is
  // dir[tile][F] = nvalid anyway) and never written: no per-item "is this row valid" predicate lives across the phases
  // (16 of them, 64-bit each, overflowed the scalar registers into VGPR lanes and scratch)
  constexpr bool TLS = ITEMS == 16;
  const uint32_t F = a.F, Fp = TLS ? ((F + 4u) & ~3u) : ((F + 3u) & ~3u);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);     // [Fp] bucket counts, then tile-local exclusive starts
  uint32_t* delta = cnt + Fp;                            // [Fp] global start - local start
  uint32_t* misc = delta + Fp;                           // [32]
  unsigned char* stage = reinterpret_cast<unsigned char*>(misc + 32);
  const int tid = threadIdx.x;

  // tiles are dealt to XCDs (block b runs on XCD b % 8: speed only) in contiguous ranges
  const uint32_t nt = gridDim.x, bi = blockIdx.x;
  const uint32_t xq = nt / 8, xr = nt % 8, xc = bi % 8, q = bi / 8;
  const uint32_t tile = xc * xq + (xc < xr ? xc : xr) + q;
  const uint32_t tile_base = tile * TILE;
  const uint32_t nvalid = (a.n - tile_base < TILE) ? (a.n - tile_base) : TILE;
  const bool full = nvalid == TILE;

  for (uint32_t b = tid; b < F + (TLS ? 1u : 0u); b += BLOCK) cnt[b] = 0;
  // global start of this tile's run of every bucket this thread scans below: issued first, needed after the ranking
  const uint32_t K = (F + BLOCK - 1) / BLOCK;            // consecutive bins per thread
  uint32_t gstart[2048 / BLOCK];
  const bool tl = TLS || a.dir != nullptr;
  if (!tl) {
    const uint32_t g = tile / a.tpg;
#pragma unroll
    for (int k = 0; k < 2048 / BLOCK; k++) {
      const uint32_t b = (uint32_t)tid * K + k;
      gstart[k] = ((uint32_t)k < K && b < F) ? a.gpre[(size_t)g * F + b] + a.P[(size_t)tile * F + b] : 0u;
    }
  }
  uint32_t x[ITEMS];
  bool bad = false;     // out-of-range keys were reported by the histogram pass; here they are just key 0 again
  load_tile_x<BLOCK, ITEMS, KM>(a.kx, tile_base, nvalid, full, tid, x, bad);
  if (TLS && !full) {
#pragma unroll
    for (int j = 0; j < ITEMS; j++) if (item_row<BLOCK, KM>(j, tid) >= nvalid) x[j] = F << a.r;
  }
  if (TLS && a.ovf_rows && __ballot(bad)) {        // (rare: a wave that holds an outlier key)
    if (bad) {
#pragma unroll 1
      for (int j = 0; j < ITEMS; j++) {
        const uint32_t rel = item_row<BLOCK, KM>(j, tid);
        if (rel >= nvalid) continue;
        bool b = false;
        (void)packed_key_checked(a.kx.cols, a.kx.ncols, tile_base + rel, b);
        if (b) {
          const uint32_t at = atomicAdd(a.ovf_n, 1u);
          if (at < a.ovf_cap) a.ovf_rows[at] = tile_base + rel;
          x[j] = F << a.r;
        }
      }
    }
    bad = false;
  }
  if (tl && __ballot(bad) && (tid & 63) == 0) atomicOr(a.bad, 1u);
  u64 pv[PF ? ITEMS : 1];
  if (PF) load_tile_vals<BLOCK, ITEMS, KM, u64>(static_cast<const u64*>(a.pay.in[0]) + tile_base, nvalid, full, tid,
                                                 reinterpret_cast<u64(&)[ITEMS]>(pv));
  __syncthreads();

  // rank of every row inside its bucket (arrival order: buckets are unordered sets)
  uint32_t lpos[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    lpos[j] = 0;
    if (TLS || full || item_row<BLOCK, KM>(j, tid) < nvalid)
      lpos[j] = CL ? lds_count_rank(cnt, x[j] >> a.r) : atomicAdd(&cnt[x[j] >> a.r], 1u);
  }
  // TLS: the transformed keys wait in the upper half of the stage (the lower half receives them in bucket order) instead of
  // in 16 registers across the scan: the 16-row instance stays inside 128 VGPRs without spilling
  uint32_t* xkeep = reinterpret_cast<uint32_t*>(stage) + TILE;
  if (TLS) {
#pragma unroll
    for (int j = 0; j < ITEMS; j++) xkeep[(uint32_t)j * BLOCK + tid] = x[j];
  }
  __syncthreads();

  // exclusive scan of the bucket counts; global position of every bucket's run
  {
    uint32_t c[2048 / BLOCK], s = 0;
#pragma unroll
    for (int k = 0; k < 2048 / BLOCK; k++) {
      const uint32_t b = (uint32_t)tid * K + k;
      c[k] = ((uint32_t)k < K && b < F) ? cnt[b] : 0u;
      s += c[k];
    }
    uint32_t e = block_excl_scan_u32<BLOCK>(s, misc, nullptr);
#pragma unroll
    for (int k = 0; k < 2048 / BLOCK; k++) {
      const uint32_t b = (uint32_t)tid * K + k;
      if ((uint32_t)k < K && b < F) {
        cnt[b] = e;
        if (tl) { delta[b] = tile_base; a.dir[(size_t)tile * (F + 1) + b] = (uint16_t)e; }
        else delta[b] = gstart[k] - e;
        e += c[k];
      }
    }
    if (TLS) {
      // the extra bin (rows past the end of a ragged tile, listed outlier keys) starts behind the rows that count: its
      // rows take the last places of the tile's stage and are never part of a segment
      if (tid == 0) { const uint32_t ngood = TILE - cnt[F]; a.dir[(size_t)tile * (F + 1) + F] = (uint16_t)ngood; cnt[F] = ngood; }
    } else if (tl && tid == 0) a.dir[(size_t)tile * (F + 1) + F] = (uint16_t)nvalid;
  }
  __syncthreads();

  // keys: registers -> LDS in bucket order -> global runs of uint16 slot keys
  uint32_t* st32 = reinterpret_cast<uint32_t*>(stage);
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    if (TLS || full || item_row<BLOCK, KM>(j, tid) < nvalid) {
      const uint32_t xj = TLS ? xkeep[(uint32_t)j * BLOCK + tid] : x[j];
      lpos[j] += cnt[xj >> a.r];
      st32[lpos[j]] = xj;
    }
  }
  __syncthreads();
  constexpr int NG = TLS ? 1 : ITEMS;
  uint32_t gpos[NG];
  const uint32_t smask = (1u << a.r) - 1u;
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const uint32_t s = (uint32_t)j * BLOCK + tid;
    if (!TLS) gpos[TLS ? 0 : j] = 0;
    if (s < nvalid) {
      const uint32_t xs = st32[s];
      const uint32_t gp = TLS ? tile_base + s : delta[xs >> a.r] + s;
      if (!TLS) gpos[TLS ? 0 : j] = gp;
      if (a.kout) a.kout[gp] = (uint16_t)(xs & smask);      // (hash combiner: no slot keys, the packed key is a payload)
    }
  }
  if (TLS) gpos[0] = 0;

  // value columns follow the same permutation
  if (PF) place_payload<BLOCK, ITEMS, KM, u64, TLS, NG>(reinterpret_cast<const u64(&)[ITEMS]>(pv), static_cast<u64*>(a.pay.out[0]), stage,
                                                        nvalid, full, tid, lpos, gpos, tile_base);
  for (int c = PF ? 1 : 0; c < a.pay.n; c++) {
    if (a.pay.width[c] == 8)
      move_payload<BLOCK, ITEMS, KM, u64, TLS, NG>(static_cast<const u64*>(a.pay.in[c]) + tile_base,
                                                   static_cast<u64*>(a.pay.out[c]), stage, nvalid, full, tid, lpos, gpos, tile_base);
    else
      move_payload<BLOCK, ITEMS, KM, uint32_t, TLS, NG>(static_cast<const uint32_t*>(a.pay.in[c]) + tile_base,
                                                        static_cast<uint32_t*>(a.pay.out[c]), stage, nvalid, full, tid, lpos, gpos, tile_base);
  }
}

// ---------------------------------------------------------------------------------------
// geometry + launchers
// ---------------------------------------------------------------------------------------
struct BkGeomV { uint32_t block, items; };
static const BkGeomV BK_GEOMS[] = {{1024, 12}, {512, 12}};

void bucket_geometry(dthip_ctx* ctx, int64_t n, int B, int r, int km, BucketGeom* g) {
  int v = ctx->bucket_variant;
  if (v < 0 || v >= (int)(sizeof(BK_GEOMS) / sizeof(BK_GEOMS[0]))) v = 0;
  if (r > B) r = B;
  g->B = B; g->r = r; g->d = B - r;
  g->F = 1u << g->d; g->S = 1u << r;
  g->block = BK_GEOMS[v].block; g->items = BK_GEOMS[v].items;
  g->tile = g->block * g->items;
  g->ntiles = (uint32_t)((n + g->tile - 1) / g->tile);
  const uint32_t gmax = (uint32_t)ctx->num_cus * (g->block == 1024 ? 2u : 4u);
  g->tpg = (g->ntiles + gmax - 1) / gmax;
  if (g->tpg == 0) g->tpg = 1;
  g->G = (g->ntiles + g->tpg - 1) / g->tpg;
  g->km = km;
}

#define BK_DISPATCH_KM(FN, B_, I_, CL_, g, ...)                                                  \
  do {                                                                                            \
    if ((g).km == 1) return FN<B_, I_, 1, CL_>(__VA_ARGS__);                                      \
    if ((g).km == 2) return FN<B_, I_, 2, CL_>(__VA_ARGS__);                                      \
    return FN<B_, I_, 0, CL_>(__VA_ARGS__);                                                       \
  } while (0)
#define BK_DISPATCH(FN, g, cl, ...)                                                               \
  do {                                                                                            \
    if ((g).block == 1024) {                                                                      \
      if (cl) BK_DISPATCH_KM(FN, 1024, 12, true, g, __VA_ARGS__);                                 \
      BK_DISPATCH_KM(FN, 1024, 12, false, g, __VA_ARGS__);                                        \
    }                                                                                             \
    if (cl) BK_DISPATCH_KM(FN, 512, 12, true, g, __VA_ARGS__);                                    \
    BK_DISPATCH_KM(FN, 512, 12, false, g, __VA_ARGS__);                                           \
  } while (0)

template <int BLOCK, int ITEMS, int KM, bool CL>
static int hist_t(dthip_ctx* ctx, const HistArgs& a, uint32_t G) {
  const size_t lds = (size_t)((a.F + 3u) & ~3u) * 8 + 16;
  if (KM != 0 && !CL && a.kx.ncols == 1 && ctx->bucket_variant != 2) {
    DTHIP_LAUNCH(ctx, "bucket_hist_kernel", (bucket_hist_kernel<BLOCK, ITEMS, KM, CL, (KM != 0 && !CL)>), G, BLOCK, lds, a);
    return DTHIP_OK;
  }
  DTHIP_LAUNCH(ctx, "bucket_hist_kernel", (bucket_hist_kernel<BLOCK, ITEMS, KM, CL, false>), G, BLOCK, lds, a);
  return DTHIP_OK;
}

int launch_bucket_hist(dthip_ctx* ctx, const KeyXform& kx, int64_t n, const BucketGeom& g, uint32_t* P, uint32_t* gtot,
                       uint32_t* bad, bool clustered) {
  HistArgs a;
  a.kx = kx; a.n = (uint32_t)n; a.r = g.r; a.F = g.F; a.ntiles = g.ntiles; a.tpg = g.tpg; a.P = P; a.gtot = gtot; a.bad = bad;
  BK_DISPATCH(hist_t, g, clustered, ctx, a, g.G);
}

int launch_bucket_gscan(dthip_ctx* ctx, const BucketGeom& g, uint32_t* gtot, uint32_t* tot, const uint32_t* bbase, int phase) {
  DTHIP_LAUNCH(ctx, "bucket_gscan_kernel", bucket_gscan_kernel, (g.F + 63) / 64, 1024, 0, gtot, g.G, g.F, tot, bbase, phase);
  return DTHIP_OK;
}

int launch_bucket_plan(dthip_ctx* ctx, const uint32_t* tot, uint32_t F, uint32_t n_raw, uint32_t M,
                       uint32_t* bbase, WorkItem* items, uint32_t* nitems, const FillList* fills) {
  if (F > 2048) { set_error("bucket plan: F=%u > 2048", F); return DTHIP_EINVAL; }
  FillList fl;
  if (fills) fl = *fills; else fl.n = 0;
  DTHIP_LAUNCH(ctx, "bucket_plan_kernel", bucket_plan_kernel, 1, 1024, 0, tot, F, n_raw, M, bbase, items, nitems, fl);
  return DTHIP_OK;
}

// ---------------------------------------------------------------------------------------
// small slot tables: the group list in one launch.  Chunks of 1024 slots, two block scans per chunk (non-empty slots,
// rows); slot order == key order, so idx is ascending and off the Groupby offsets (groupby.h:54-91) of the result.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) small_groups_kernel(SmallGroupsArgs a) {
  __shared__ uint32_t scratch[16];
  const uint32_t tid = threadIdx.x;
  uint32_t g0 = 0, r0 = 0;
  for (uint32_t base = 0; base < a.nslots; base += 1024) {
    const uint32_t s = base + tid;
    uint32_t c = 0;
    if (s < a.nslots) c = a.bits ? (a.cnt[s >> 5] >> (s & 31)) & 1u : a.cnt[s];
    const uint32_t f = c ? 1u : 0u;
    uint32_t gt = 0, rt = 0, re = 0;
    const uint32_t ge = block_excl_scan_u32<1024>(f, scratch, &gt);
    if (a.off) re = block_excl_scan_u32<1024>(c, scratch, &rt);
    if (f) {
      a.idx[g0 + ge] = (int32_t)s;
      if (a.off) a.off[g0 + ge] = r0 + re;
    }
    g0 += gt; r0 += rt;
  }
  if (tid == 0) {
    if (a.off) a.off[g0] = r0;
    a.out[0] = g0;
    a.out[1] = a.bad ? *a.bad : 0u;
  }
}

__global__ void __launch_bounds__(256) fill_list_kernel(BigFill f) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x, nt = (unsigned long long)gridDim.x * 256;
#pragma unroll 1
  for (int i = 0; i < f.n; i++) {
    uint32_t* p = f.p[i];
    const unsigned long long w = f.words[i], q4 = w / 4;
    const uint32_t v = f.val[i];
    u32x4* p4 = reinterpret_cast<u32x4*>(p);
    for (unsigned long long q = t; q < q4; q += nt) p4[q] = u32x4{v, v, v, v};
    if (t < w - q4 * 4) p[q4 * 4 + t] = v;
  }
}

int launch_fill_list(dthip_ctx* ctx, const BigFill& f) {
  if (f.n == 0) return DTHIP_OK;
  unsigned long long words = 0;
  for (int i = 0; i < f.n; i++) words += f.words[i];
  const unsigned grid = (unsigned)std::min<unsigned long long>(std::max<unsigned long long>(words / (4 * 256 * 4), 1), 4096);
  DTHIP_LAUNCH(ctx, "fill_list_kernel", fill_list_kernel, grid, 256, 0, f);
  return DTHIP_OK;
}

int launch_small_groups(dthip_ctx* ctx, const SmallGroupsArgs& a) {
  DTHIP_LAUNCH(ctx, "small_groups_kernel", small_groups_kernel, 1, 1024, 0, a);
  return DTHIP_OK;
}

template <int BLOCK, int ITEMS, int KM, bool CL, bool PF>
static int part_launch(dthip_ctx* ctx, const PartArgs& a, uint32_t ntiles, size_t lds) {
  auto kfn = bucket_partition_kernel<BLOCK, ITEMS, KM, CL, PF>;
  DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 256));
  DTHIP_LAUNCH(ctx, "bucket_partition_kernel", kfn, ntiles, BLOCK, lds, a);
  return DTHIP_OK;
}

template <int BLOCK, int ITEMS, int KM, bool CL>
static int part_t(dthip_ctx* ctx, const PartArgs& a, uint32_t ntiles, size_t lds) {
  // early load of payload 0: vector-load key modes only (their tiles are read with 16-byte loads)
  if (KM != 0 && !CL && a.pay.n >= 1 && a.pay.width[0] == 8 && ctx->bucket_variant != 2)
    return part_launch<BLOCK, ITEMS, KM, CL, true>(ctx, a, ntiles, lds);
  return part_launch<BLOCK, ITEMS, KM, CL, false>(ctx, a, ntiles, lds);
}

int launch_bucket_partition(dthip_ctx* ctx, const KeyXform& kx, int64_t n, const BucketGeom& g, const uint32_t* P,
                            const uint32_t* gpre, uint16_t* kout, const PayCols& pay, bool clustered, uint16_t* dir, uint32_t* bad,
                            uint32_t* ovf_rows, uint32_t* ovf_n, uint32_t ovf_cap) {
  PartArgs a;
  a.ovf_rows = (g.items == 16 && kx.ncols == 1) ? ovf_rows : nullptr; a.ovf_n = ovf_n; a.ovf_cap = ovf_cap;
  a.kx = kx; a.n = (uint32_t)n; a.r = g.r; a.F = g.F; a.tpg = g.tpg; a.P = P; a.gpre = gpre;
  a.kout = kout; a.pay = pay; a.dir = dir; a.bad = bad;
  int maxw = 4;
  for (int c = 0; c < pay.n; c++) maxw = pay.width[c] > maxw ? pay.width[c] : maxw;
  const uint32_t Fp = (g.F + 3u) & ~3u;
  size_t lds = (size_t)(2 * Fp + 32) * 4 + (size_t)g.tile * maxw;
  if (g.items == 16) {          // tile-local layout only (bucket_tl16_geometry); the stage's upper half keeps the keys
    lds = (size_t)(2 * ((g.F + 4u) & ~3u) + 32) * 4 + (size_t)g.tile * 8;
    if (!dir || g.block != 1024 || clustered || lds > 160 * 1024 - 256) { set_error("bucket partition: 16-row items need the tile-local layout"); return DTHIP_EINVAL; }
    BK_DISPATCH_KM(part_t, 1024, 16, false, g, ctx, a, g.ntiles, lds);
  }
  BK_DISPATCH(part_t, g, clustered, ctx, a, g.ntiles, lds);
}

// 1024 threads x 16 rows: the tile-local layout's own geometry (its partition instance keeps no position array); false when
// the staged tile of the widest value column would not fit the LDS
bool bucket_tl16_geometry(dthip_ctx* ctx, int64_t n, int maxw, BucketGeom* g) {
  if (g->block != 1024) return false;
  const uint32_t Fp = (g->F + 4u) & ~3u;
  if (maxw > 8 || (size_t)(2 * Fp + 32) * 4 + (size_t)16384 * 8 > 160 * 1024 - 256) return false;
  g->items = 16; g->tile = 16384;
  g->ntiles = (uint32_t)((n + g->tile - 1) / g->tile);
  const uint32_t gmax = (uint32_t)ctx->num_cus * 2u;
  g->tpg = (g->ntiles + gmax - 1) / gmax;
  if (g->tpg == 0) g->tpg = 1;
  g->G = (g->ntiles + g->tpg - 1) / g->tpg;
  return true;
}

// ---------------------------------------------------------------------------------------
// table aggregation
// ---------------------------------------------------------------------------------------

size_t table_agg_slot_bytes(int flags) {
  size_t b = 0;
  if (flags & ACC_CNT) b += 4;
  if (flags & ACC_VCNT) b += 4;
  if (flags & ACC_SUM) b += 8;
  if (flags & ACC_MIN) b += 8;
  if (flags & ACC_MAX) b += 8;
  if (flags & ACC_FSUM) b += 8;
  return b;
}

// LDS bytes of one aggregation table of S slots (ACC_PRES: one presence bit per slot)
size_t table_agg_lds_bytes(int flags, uint32_t S) {
  return (size_t)S * table_agg_slot_bytes(flags) + ((flags & ACC_PRES) ? (size_t)((S + 31) / 32) * 4 : 0) + 16;
}

// LDS table -> dense accumulator arrays: plain stores when the bucket has a single part, global atomics otherwise
__device__ __forceinline__ void flush_table(const LdsTab& t, const AggTable& tab, uint32_t bucket, uint32_t S, int flags,
                                            bool single, int isfloat, int tid) {
  const size_t base = (size_t)bucket * S;
  if (single) {
    for (uint32_t s = tid; s < S; s += TA_BLOCK) {
      if (flags & ACC_CNT) tab.cnt[base + s] = t.cnt[s];
      if (flags & ACC_VCNT) tab.vcnt[base + s] = t.vcnt[s];
      if (flags & ACC_SUM) tab.sum[base + s] = t.sum[s];
      if (flags & ACC_MIN) tab.mn[base + s] = t.mn[s];
      if (flags & ACC_MAX) tab.mx[base + s] = t.mx[s];
      if (flags & ACC_FSUM) tab.fsum[base + s] = t.fsum[s];
    }
    if (flags & ACC_PRES) for (uint32_t s = tid; s < (S + 31) / 32; s += TA_BLOCK) tab.pres[base / 32 + s] = t.pres[s];
  } else {
    if (flags & ACC_PRES)
      for (uint32_t s = tid; s < (S + 31) / 32; s += TA_BLOCK) { const uint32_t w = t.pres[s]; if (w) atomicOr(&tab.pres[base / 32 + s], w); }
    for (uint32_t s = tid; s < S; s += TA_BLOCK) {
      if (flags & ACC_CNT) { const uint32_t c = t.cnt[s]; if (c) atomicAdd(&tab.cnt[base + s], c); }
      if (flags & ACC_VCNT) { const uint32_t c = t.vcnt[s]; if (c) atomicAdd(&tab.vcnt[base + s], c); }
      if (flags & ACC_SUM) {
        const u64 w = t.sum[s];
        if (w) {
          if (isfloat) atomicAdd(reinterpret_cast<double*>(&tab.sum[base + s]), __longlong_as_double((long long)w));
          else atomicAdd(&tab.sum[base + s], w);
        }
      }
      if (flags & ACC_MIN) { const u64 w = t.mn[s]; if (w != ~0ULL) atomicMin(&tab.mn[base + s], w); }
      if (flags & ACC_MAX) { const u64 w = t.mx[s]; if (w) atomicMax(&tab.mx[base + s], w); }
      if (flags & ACC_FSUM) { const double w = t.fsum[s]; if (w != 0.0) atomicAdd(&tab.fsum[base + s], w); }
    }
  }
}

__device__ __forceinline__ void init_table(const LdsTab& t, uint32_t S, int flags, int tid) {
  for (uint32_t s = tid; s < S; s += TA_BLOCK) {
    if (flags & ACC_SUM) t.sum[s] = 0;
    if (flags & ACC_MIN) t.mn[s] = ~0ULL;
    if (flags & ACC_MAX) t.mx[s] = 0;
    if (flags & ACC_FSUM) t.fsum[s] = 0.0;
    if (flags & ACC_CNT) t.cnt[s] = 0;
    if (flags & ACC_VCNT) t.vcnt[s] = 0;
  }
  if (flags & ACC_PRES) for (uint32_t s = tid; s < (S + 31) / 32; s += TA_BLOCK) t.pres[s] = 0;
}

struct TableAggDev {
  const WorkItem* items; const uint32_t* nitems;
  const uint16_t* kpart; KeyXform kx;
  const void* val;
  uint32_t S; int flags; int isfloat;
  AggTable tab;
  uint32_t* bad;
  int hot_ok;                 // DTHIP_HOT_COMBINE=0: hot buckets like the others (A/B)
};

// SRC: 0 = slot keys (uint16) + value column of the partitioned rows, 1 = raw rows (keys transformed
// on the fly; one table holds the whole key range)
template <typename VT, int SRC, bool UNI>
__global__ void __launch_bounds__(TA_BLOCK) __attribute__((amdgpu_waves_per_eu(SRC == 0 ? 8 : 1))) table_agg_kernel(TableAggDev a) {
  constexpr bool RAW = SRC == 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x >= *a.nitems) return;
  const WorkItem it = a.items[blockIdx.x];
  const int tid = threadIdx.x;
  const int flags = a.flags;
  const uint32_t S = a.S;
  LdsTab t = carve_tab(smem, S, flags);
  t.g_na = a.tab.nacnt ? a.tab.nacnt + (size_t)it.bucket * S : nullptr;
  init_table(t, S, flags, tid);
  __syncthreads();
  const VT* __restrict__ val = static_cast<const VT*>(a.val);
  const bool hasval = (flags & (ACC_VALUE_MASK)) != 0;
  if (RAW) {
    bool bad = false;
    for (uint32_t row = it.begin + tid; row < it.end; row += TA_BLOCK) {
      const uint32_t slot = (uint32_t)packed_key_checked(a.kx.cols, a.kx.ncols, row, bad);
      const VT v = hasval ? val[row] : VT(0);
      acc_row<VT, UNI>(t, flags, slot, v);
    }
    if (__ballot(bad) && (tid & 63) == 0) atomicOr(a.bad, 1u);
  } else {
    const uint16_t* __restrict__ kp = a.kpart;
    const bool hot = (it.single & 4u) != 0 && a.hot_ok;
    uint32_t a0 = (it.begin + 7u) & ~7u; if (a0 > it.end) a0 = it.end;
    uint32_t a1 = it.end & ~7u; if (a1 < a0) a1 = a0;
    // ragged head [begin, a0) and tail [a1, end): fewer than 8 rows each
    {
      const uint32_t nh = a0 - it.begin, ntl = it.end - a1;
      if ((uint32_t)tid < nh) {
        const uint32_t row = it.begin + tid;
        acc_row<VT, UNI>(t, flags, kp[row], hasval ? val[row] : VT(0));
      } else if ((uint32_t)tid >= 64u && (uint32_t)tid - 64u < ntl) {
        const uint32_t row = a1 + ((uint32_t)tid - 64u);
        acc_row<VT, UNI>(t, flags, kp[row], hasval ? val[row] : VT(0));
      }
    }
    const uint32_t ngr = (a1 - a0) >> 3;
    for (uint32_t g = tid; g < ngr; g += TA_BLOCK) {
      const uint32_t row = a0 + g * 8u;
      const bu32x4 kw = *reinterpret_cast<const bu32x4*>(kp + row);
      uint32_t slot[8];
      slot[0] = kw.x & 0xFFFFu; slot[1] = kw.x >> 16; slot[2] = kw.y & 0xFFFFu; slot[3] = kw.y >> 16;
      slot[4] = kw.z & 0xFFFFu; slot[5] = kw.z >> 16; slot[6] = kw.w & 0xFFFFu; slot[7] = kw.w >> 16;
      VT v[8];
      if (hasval) {
        constexpr int NV = (int)sizeof(VT) / 2;     // 16-B loads for 8 values: 2 (4-byte) or 4 (8-byte)
        bu32x4 w[NV];
        const bu32x4* vp = reinterpret_cast<const bu32x4*>(val + row);
#pragma unroll
        for (int j = 0; j < NV; j++) w[j] = vp[j];
        const VT* wv = reinterpret_cast<const VT*>(w);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = wv[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = VT(0);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) acc_row<VT, UNI>(t, flags, slot[j], v[j], hot);
    }
  }
  __syncthreads();
  flush_table(t, a.tab, it.bucket, S, flags, (it.single & 1u) != 0, a.isfloat, tid);
}

template <typename VT, int SRC, bool UNI>
static int table_agg_t(dthip_ctx* ctx, const TableAggDev& d, uint32_t grid, size_t lds) {
  auto kfn = table_agg_kernel<VT, SRC, UNI>;
  DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 256));
  DTHIP_LAUNCH(ctx, "table_agg_kernel", kfn, grid, TA_BLOCK, lds, d);
  return DTHIP_OK;
}

int launch_table_agg(dthip_ctx* ctx, const TableAggArgs& a) {
  if (a.max_items == 0) return DTHIP_OK;
  TableAggDev d;
  d.items = a.items; d.nitems = a.nitems; d.kpart = a.kpart; d.kx = a.kx; d.val = a.val;
  d.S = a.S; d.flags = a.flags; d.isfloat = stype_is_float(a.vstype) ? 1 : 0; d.tab = a.tab; d.bad = a.bad;
  static const int hot_ok = !(getenv("DTHIP_HOT_COMBINE") && atoi(getenv("DTHIP_HOT_COMBINE")) == 0);
  d.hot_ok = hot_ok;
  const size_t lds = table_agg_lds_bytes(a.flags, a.S);
  if (lds > 160 * 1024 - 256) { set_error("table_agg: table of %zu bytes exceeds LDS", lds); return DTHIP_EINVAL; }
  if ((a.flags & ACC_PRES) && (a.S & 31) && a.src != 1) { set_error("table_agg: presence bitmaps need S %% 32 == 0"); return DTHIP_EINVAL; }
  const int st = a.val ? a.vstype : DTHIP_INT32;
#define TA_GO(VT)                                                                 \
  do {                                                                            \
    if (a.src == 1) { if (a.clustered) return table_agg_t<VT, 1, true>(ctx, d, a.max_items, lds);   \
                      return table_agg_t<VT, 1, false>(ctx, d, a.max_items, lds); }                 \
    if (a.clustered) return table_agg_t<VT, 0, true>(ctx, d, a.max_items, lds);   \
    return table_agg_t<VT, 0, false>(ctx, d, a.max_items, lds);                   \
  } while (0)
  switch (st) {
    case DTHIP_INT32: TA_GO(int32_t);
    case DTHIP_INT64: TA_GO(long long);
    case DTHIP_FLOAT32: TA_GO(float);
    case DTHIP_FLOAT64: TA_GO(double);
    default: set_error("table_agg: unsupported value stype %d", a.vstype); return DTHIP_ENOTIMPL;
  }
#undef TA_GO
}


// ---------------------------------------------------------------------------------------
// The same aggregation over the TILE-LOCAL layout: bucket b's rows are one short segment per partition tile
// (12 rows on average for C3), located by the transposed directory dirT[b][tile].  An item = (bucket, tile
// range).  A wave takes 64 tiles at a time (lane = tile: coalesced directory reads), then walks their segments four
// at a time, 16 lanes per segment.  Items are dealt to XCDs in contiguous bucket ranges: the 64-byte sectors that
// straddle two neighbouring buckets' segments are then fetched from HBM once and hit in that XCD's L2 the second time.
// ---------------------------------------------------------------------------------------
struct TableAggSegDev {
  const WorkItem* items; const uint32_t* nitems; const uint32_t* nitems2;
  const uint16_t* kpart; const void* val;
  const uint16_t* dirT; uint32_t dstride; uint32_t tile_rows;
  uint32_t S; int flags; int isfloat;
  AggTable tab;
  uint32_t* bad;
};

// LONG: every item is walked in the long-segment mode (few buckets); the instance keeps none of the short mode's register
// arrays and two workgroups share a CU when the table allows it
template <typename VT, bool LONG>
__global__ void __launch_bounds__(TA_BLOCK) __attribute__((amdgpu_waves_per_eu(LONG ? 8 : 1))) table_agg_seg_kernel(TableAggSegDev a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // list A (parts of multi-part buckets: first, round robin over the XCDs), then list B (one item per bucket: XCD c takes a
  // contiguous range of buckets) -- seg_plan_kernel
  const uint32_t na = a.nitems2 ? *a.nitems : 0u, na_pad = (na + 7u) & ~7u;
  uint32_t bi = blockIdx.x, item_at;
  if (bi < na_pad) {
    if (bi >= na) return;
    item_at = bi;
  } else {
    bi -= na_pad;
    const uint32_t nit = a.nitems2 ? *a.nitems2 : *a.nitems;
    const uint32_t xq = nit / 8, xr = nit % 8, xc = bi % 8, q0 = bi / 8;
    if (q0 >= xq + (xc < xr ? 1u : 0u)) return;
    item_at = na_pad + xc * xq + (xc < xr ? xc : xr) + q0;
  }
  const WorkItem it = a.items[item_at];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, sub = lane & 15;
  const int flags = a.flags;
  const uint32_t S = a.S;
  LdsTab t = carve_tab(smem, S, flags);
  t.g_na = a.tab.nacnt ? a.tab.nacnt + (size_t)it.bucket * S : nullptr;
  init_table(t, S, flags, tid);
  __syncthreads();
  const VT* __restrict__ val = static_cast<const VT*>(a.val);
  const uint16_t* __restrict__ kp = a.kpart;
  const bool hasval = (flags & (ACC_VALUE_MASK)) != 0;
  const uint16_t* __restrict__ ds = a.dirT + (size_t)it.bucket * a.dstride;
  const uint16_t* __restrict__ de = ds + a.dstride;
  const uint32_t t0 = it.begin, t1 = it.end;
  const uint32_t tr = a.tile_rows;
  // directory entries of the wave's first 64 tiles; the next chunk's are loaded while the current one is processed.
  // Long segments (few buckets: hundreds of rows of a tile per bucket; or a hot bucket): the waves take turns FOUR tiles at
  // a time -- a part of such a bucket spans a few hundred tiles at most, and chunks of 64 left 13 of the 16 waves idle
  // (BASELINE C2 on this layout: 1.20 -> see DESIGN 3.1 "Round 6, third part")
  const bool long_mode = LONG || (it.single & 2u) != 0;
  const uint32_t cw = long_mode ? 4u : 64u;                 // tiles per wave and turn
  const uint32_t cstep = (TA_BLOCK / 64) * cw;
  uint32_t c = t0 + (uint32_t)wave * cw;
  uint32_t st_n = 0, ln_n = 0;
  if ((uint32_t)lane < cw && c + (uint32_t)lane < t1) { st_n = ds[c + lane]; ln_n = (uint32_t)de[c + lane] - st_n; }
  for (; c < t1; c += cstep) {
    const uint32_t st = st_n, ln = ln_n;
    {
      const uint32_t tn = c + cstep + (uint32_t)lane;
      st_n = 0; ln_n = 0;
      if ((uint32_t)lane < cw && tn < t1) { st_n = ds[tn]; ln_n = (uint32_t)de[tn] - st_n; }
    }
    if (long_mode) {
      // whole waves stream the segments (coalesced 512-byte loads).  The first 192 rows of the turn's FOUR segments are loaded
      // before the first DS atomic (twelve loads per lane in flight: a segment of 64 buckets x 12288-row tiles is ~192 rows),
      // what is left of longer segments follows four loads at a time
      const uint32_t nseg = (t1 - c < cw) ? (t1 - c) : cw;
      const bool hot = !LONG;          // a long bucket among short ones holds a hot key: lanes that share a slot are combined (acc_row)
      uint32_t sst[4], sln[4];
#pragma unroll
      for (int sg = 0; sg < 4; sg++) {
        sst[sg] = (uint32_t)__builtin_amdgcn_readlane((int)st, sg);
        sln[sg] = (uint32_t)sg < nseg ? (uint32_t)__builtin_amdgcn_readlane((int)ln, sg) : 0u;
      }
      {
        uint32_t slot[4][3];
        VT v[4][3];
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
          const uint32_t base = (c + (uint32_t)sg) * tr + sst[sg];
#pragma unroll
          for (int u = 0; u < 3; u++) {
            const uint32_t j = (uint32_t)u * 64u + (uint32_t)lane;
            slot[sg][u] = 0; v[sg][u] = VT(0);
            if (j < sln[sg]) { slot[sg][u] = kp[base + j]; if (hasval) v[sg][u] = val[base + j]; }
          }
        }
#pragma unroll
        for (int sg = 0; sg < 4; sg++)
#pragma unroll
          for (int u = 0; u < 3; u++)
            if ((uint32_t)u * 64u + (uint32_t)lane < sln[sg]) acc_row<VT, false>(t, flags, slot[sg][u], v[sg][u], hot);
      }
#pragma unroll
      for (int sg = 0; sg < 4; sg++) {
        if (sln[sg] <= 192u) continue;
        const uint32_t base = (c + (uint32_t)sg) * tr + sst[sg];
        for (uint32_t j0 = 192u; j0 < sln[sg]; j0 += 256u) {
          uint32_t slot[4];
          VT v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t j = j0 + (uint32_t)u * 64u + (uint32_t)lane;
            slot[u] = 0; v[u] = VT(0);
            if (j < sln[sg]) { slot[u] = kp[base + j]; if (hasval) v[u] = val[base + j]; }
          }
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (j0 + (uint32_t)u * 64u + (uint32_t)lane < sln[sg]) acc_row<VT, false>(t, flags, slot[u], v[u], hot);
        }
      }
      continue;
    }
    if (LONG) continue;
    // 16 lanes per segment, four segments per wave instruction.  The first 32 rows of 32 segments are loaded (two
    // loads per lane and segment) before their DS atomics start, so the usual segment (12 .. 25 rows) never waits on a
    // second, dependent round of loads; longer ones finish in a plain loop.
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t slot[8][2];
      VT v[8][2];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int seg = (h * 8 + q) * 4 + grp;
        const uint32_t sst = (uint32_t)__shfl((int)st, seg, 64), sln = (uint32_t)__shfl((int)ln, seg, 64);
        const uint32_t row = (c + (uint32_t)seg) * tr + sst + (uint32_t)sub;
#pragma unroll
        for (int k = 0; k < 2; k++) {
          slot[q][k] = 0; v[q][k] = VT(0);
          if ((uint32_t)sub + 16u * k < sln) {
            slot[q][k] = kp[row + 16u * k];
            if (hasval) v[q][k] = val[row + 16u * k];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int seg = (h * 8 + q) * 4 + grp;
        const uint32_t sln = (uint32_t)__shfl((int)ln, seg, 64);
#pragma unroll
        for (int k = 0; k < 2; k++)
          if ((uint32_t)sub + 16u * k < sln) acc_row<VT, false>(t, flags, slot[q][k], v[q][k]);
        if (__ballot(sln > 32u)) {
          const uint32_t sst = (uint32_t)__shfl((int)st, seg, 64);
          const uint32_t base = (c + (uint32_t)seg) * tr + sst;
          for (uint32_t j = (uint32_t)sub + 32u; j < sln; j += 16u)
            acc_row<VT, false>(t, flags, kp[base + j], hasval ? val[base + j] : VT(0));
        }
      }
    }
  }
  __syncthreads();
  flush_table(t, a.tab, it.bucket, S, flags, (it.single & 1u) != 0, a.isfloat, tid);
}

template <typename VT>
static int table_agg_seg_t(dthip_ctx* ctx, const TableAggSegDev& d, uint32_t grid, size_t lds, bool all_long) {
  if (all_long) {
    auto kfn = table_agg_seg_kernel<VT, true>;
    DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 256));
    DTHIP_LAUNCH(ctx, "table_agg_seg_kernel", kfn, grid, TA_BLOCK, lds, d);
    return DTHIP_OK;
  }
  auto kfn = table_agg_seg_kernel<VT, false>;
  DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 256));
  DTHIP_LAUNCH(ctx, "table_agg_seg_kernel", kfn, grid, TA_BLOCK, lds, d);
  return DTHIP_OK;
}

int launch_table_agg_seg(dthip_ctx* ctx, const TableAggSegArgs& a) {
  if (a.max_items == 0) return DTHIP_OK;
  TableAggSegDev d;
  d.items = a.items; d.nitems = a.nitems; d.nitems2 = a.nitems2; d.kpart = a.kpart; d.val = a.val; d.dirT = a.dirT; d.dstride = a.dstride;
  d.tile_rows = a.tile_rows; d.S = a.S; d.flags = a.flags; d.isfloat = stype_is_float(a.vstype) ? 1 : 0; d.tab = a.tab; d.bad = a.bad;
  const size_t lds = table_agg_lds_bytes(a.flags, a.S);
  if (lds > 160 * 1024 - 256) { set_error("table_agg_seg: table of %zu bytes exceeds LDS", lds); return DTHIP_EINVAL; }
  const uint32_t grid = (a.max_items + 7u) & ~7u;
  switch (a.val ? a.vstype : DTHIP_INT32) {
    case DTHIP_INT32: return table_agg_seg_t<int32_t>(ctx, d, grid, lds, a.all_long);
    case DTHIP_INT64: return table_agg_seg_t<long long>(ctx, d, grid, lds, a.all_long);
    case DTHIP_FLOAT32: return table_agg_seg_t<float>(ctx, d, grid, lds, a.all_long);
    case DTHIP_FLOAT64: return table_agg_seg_t<double>(ctx, d, grid, lds, a.all_long);
    default: set_error("table_agg_seg: unsupported value stype %d", a.vstype); return DTHIP_ENOTIMPL;
  }
}

// dir[tile][F+1] -> dirT[F+1][dstride] (64 x 64 blocks through LDS)
__global__ void __launch_bounds__(256) dir_transpose_kernel(const uint16_t* __restrict__ dir, uint32_t ntiles, uint32_t F1,
                                                            uint16_t* __restrict__ dirT, uint32_t dstride) {
  __shared__ uint16_t blk[64][66];
  const uint32_t tb = blockIdx.x * 64u, bb = blockIdx.y * 64u;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const uint32_t t = tb + (uint32_t)r, b = bb + (uint32_t)tx;
    blk[r][tx] = (t < ntiles && b < F1) ? dir[(size_t)t * F1 + b] : (uint16_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const uint32_t b = bb + (uint32_t)r, t = tb + (uint32_t)tx;
    if (b < F1 && t < dstride) dirT[(size_t)b * dstride + t] = blk[tx][r];
  }
}

// rows of every bucket: tot[b] = sum over tiles of dirT[b+1][t] - dirT[b][t]
__global__ void __launch_bounds__(256) dir_totals_kernel(const uint16_t* __restrict__ dirT, uint32_t ntiles, uint32_t dstride, uint32_t* tot) {
  __shared__ uint32_t part[4];
  const uint32_t b = blockIdx.x;
  const uint16_t* r0 = dirT + (size_t)b * dstride; const uint16_t* r1 = r0 + dstride;
  uint32_t s = 0;
  for (uint32_t t = threadIdx.x; t < ntiles; t += 256) s += (uint32_t)r1[t] - (uint32_t)r0[t];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tot[b] = part[0] + part[1] + part[2] + part[3];
}

// work list over the tile-local layout: bucket b in ceil(tot[b] / M) parts of equal TILE ranges.
// nitems2 != nullptr (the bucketed aggregation; round 6): TWO lists.  A = the parts of buckets that need several (a hot key's
// bucket; every bucket when there are few): items[0 .. nA), dispatched first and dealt to the XCDs round robin -- as one
// contiguous run of the list below they all landed on ONE XCD, which then had 1.5x the rows of the others (C3 with 7 % of the
// rows in one key: table_agg_seg 2.8 -> 3.9 ms).  B = the buckets that are one item each: items[pad8(nA) .. + nB), dealt to the
// XCDs in contiguous bucket ranges (neighbouring segments share sectors in that XCD's L2).  *nitems = nA, *nitems2 = nB.
__global__ void __launch_bounds__(1024) seg_plan_kernel(const uint32_t* tot, uint32_t F, uint32_t ntiles, uint32_t M,
                                                        WorkItem* items, uint32_t* nitems, uint32_t* nitems2) {
  __shared__ uint32_t scratch[16];
  const int tid = threadIdx.x;
  uint32_t np[2], ps = 0, pb = 0;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t b = (uint32_t)tid * 2 + k;
    np[k] = b < F ? (tot[b] + M - 1) / M : 0u;
    if (np[k] > ntiles) np[k] = ntiles;
    if (nitems2 && np[k] == 1) pb += 1; else ps += np[k];
  }
  uint32_t ptot, btot = 0;
  uint32_t pe = block_excl_scan_u32<1024>(ps, scratch, &ptot);
  uint32_t be = 0;
  if (nitems2) { __syncthreads(); be = block_excl_scan_u32<1024>(pb, scratch, &btot); }
  const uint32_t bbase_ = (ptot + 7u) & ~7u;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t b = (uint32_t)tid * 2 + k;
    const bool lone = nitems2 && np[k] == 1;
    for (uint32_t i = 0; i < np[k]; i++) {
      WorkItem it;
      it.bucket = b;
      it.begin = (uint32_t)(((unsigned long long)i * ntiles) / np[k]);
      it.end = (uint32_t)(((unsigned long long)(i + 1) * ntiles) / np[k]);
      // bit 0: the bucket has a single part (plain stores of the table); bit 1: its segments average more than 48
      // rows per tile (few buckets, or a hot bucket): whole-wave streaming instead of 16 lanes per segment
      it.single = (np[k] == 1 ? 1u : 0u) | ((unsigned long long)tot[b] > 48ull * ntiles ? 2u : 0u);
      items[lone ? bbase_ + be : pe + i] = it;
    }
    if (lone) be += 1; else pe += np[k];
  }
  if (tid == 0) { *nitems = ptot; if (nitems2) *nitems2 = btot; }
}

int launch_dir_prepare(dthip_ctx* ctx, const uint16_t* dir, uint32_t ntiles, uint32_t F, uint16_t* dirT, uint32_t dstride,
                       uint32_t* tot, uint32_t M, WorkItem* items, uint32_t* nitems, uint32_t* nitems2) {
  if (F > 2048) { set_error("seg plan: F=%u > 2048", F); return DTHIP_EINVAL; }
  dim3 grid((dstride + 63) / 64, (F + 1 + 63) / 64);
  DTHIP_LAUNCH(ctx, "dir_transpose_kernel", dir_transpose_kernel, grid, 256, 0, dir, ntiles, F + 1, dirT, dstride);
  DTHIP_LAUNCH(ctx, "dir_totals_kernel", dir_totals_kernel, F, 256, 0, dirT, ntiles, dstride, tot);
  DTHIP_LAUNCH(ctx, "seg_plan_kernel", seg_plan_kernel, 1, 1024, 0, tot, F, ntiles, M, items, nitems, nitems2);
  return DTHIP_OK;
}

// ---------------------------------------------------------------------------------------
// finalize: dense accumulators at the non-empty slots -> reducer columns
// (output stypes: fexpr_sumprod.cc:47-66, fexpr_mean.cc:45-74, fexpr_minmax.cc:47-68)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) table_finalize_kernel(TableFinArgs a) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= a.ng) return;
  const uint32_t s = (uint32_t)a.idx[g];
  const uint32_t vc = a.tab.vcnt ? a.tab.vcnt[s] - (a.tab.nacnt ? a.tab.nacnt[s] : 0u) : 0u;
  const int st = a.vstype;
  const bool isf = st == DTHIP_FLOAT32 || st == DTHIP_FLOAT64;
  if (a.o_sum) {
    const u64 w = a.tab.sum[s];
    if (st == DTHIP_FLOAT64) static_cast<double*>(a.o_sum)[g] = __longlong_as_double((long long)w);
    else if (st == DTHIP_FLOAT32) static_cast<float*>(a.o_sum)[g] = (float)__longlong_as_double((long long)w);
    else static_cast<long long*>(a.o_sum)[g] = (long long)w;
  }
  if (a.o_mean) {
    double m = __builtin_nan("");
    if (vc) m = (isf ? __longlong_as_double((long long)a.tab.sum[s]) : a.tab.fsum[s]) / (double)vc;
    if (st == DTHIP_FLOAT32) static_cast<float*>(a.o_mean)[g] = vc ? (float)m : __builtin_nanf("");
    else static_cast<double*>(a.o_mean)[g] = m;
  }
  for (int which = 0; which < 2; which++) {
    void* o = which ? a.o_max : a.o_min;
    if (!o) continue;
    const u64 k = which ? a.tab.mx[s] : a.tab.mn[s];
    if (isf) {
      const double d = vc ? unsortable_f64(k) : __builtin_nan("");
      if (st == DTHIP_FLOAT64) static_cast<double*>(o)[g] = d;
      else static_cast<float*>(o)[g] = vc ? (float)d : __builtin_nanf("");
    } else {
      const long long v = (long long)(k ^ 0x8000000000000000ULL);
      if (st == DTHIP_INT64) static_cast<long long*>(o)[g] = vc ? v : INT64_MIN;
      else static_cast<int32_t*>(o)[g] = vc ? (int32_t)v : INT32_MIN;
    }
  }
  if (a.o_count) a.o_count[g] = (int64_t)vc;
}

int launch_table_finalize(dthip_ctx* ctx, const TableFinArgs& a) {
  if (a.ng == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "table_finalize_kernel", table_finalize_kernel, (a.ng + 255) / 256, 256, 0, a);
  return DTHIP_OK;
}

}  // namespace dthip
