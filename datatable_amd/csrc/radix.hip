// radix.hip -- key transform + digit histograms, and the stable one-sweep
// LSD radix pass (run positions precomputed per tile, no inter-workgroup communication) for gfx950.
//
// Reference behaviour being reproduced (not its algorithm): the ordering of
// SortContext -- stable, ascending in the transformed unsigned key, NA first
// (src/core/sort.cc:24-104, :728-845).  The reference does an MSD radix sort
// with per-chunk histograms on CPU threads (sort.cc:950-1074, 1128-1353); on
// the GPU the same permutation is produced by LSD passes over the significant
// bits of the same transformed key:
//   pass kernel   one launch per digit, each tile = 512 threads x 16 keys,
//                 reads every key/payload once and writes it once (HBM-bound);
//                 per-wave match-any ranking with 64-wide ballots keeps it
//                 stable; the global position of every (tile, digit) run comes
//                 from a per-tile digit count taken before the pass (no look-back);
//                 keys and payloads are re-ordered through LDS so that global
//                 stores are runs of consecutive addresses per digit.
#include <cstdlib>
#include "common.hpp"
#include "device_utils.hpp"
#include "keyxform.hpp"
#include "radix_dev.hpp"

namespace dthip {

constexpr int XH_BLOCK = 256;

// Packed transformed key of every row + the digit histogram of every radix
// pass, in one streaming read of the key column(s).
__global__ void __launch_bounds__(XH_BLOCK) xform_hist_kernel(XformArgs a) {
  __shared__ uint32_t lhist[MAX_PASSES * HIST_STRIDE];
  for (int i = threadIdx.x; i < a.npass * HIST_STRIDE; i += XH_BLOCK) lhist[i] = 0;
  __syncthreads();
  const uint32_t stride = gridDim.x * XH_BLOCK;
  bool oob = false;            // a key outside the (guessed) range of its column: the caller plans again with the exact one
  for (uint32_t row = blockIdx.x * XH_BLOCK + threadIdx.x; row < a.n; row += stride) {
    const uint32_t src = a.order ? (uint32_t)a.order[row] : row;
    unsigned long long k = 0;
    for (int j = 0; j < a.ncols; j++) {
      bool na;
      const unsigned long long x = xform_key_na(a.cols[j], src, na);
      if (!na && !xform_in_range(a.cols[j], x)) oob = true;
      k |= x << a.cols[j].shift;
    }
    if (a.out64) static_cast<unsigned long long*>(a.out)[row] = k;
    else static_cast<uint32_t*>(a.out)[row] = (uint32_t)k;
    for (int p = 0; p < a.npass; p++) {
      const uint32_t d = (uint32_t)(k >> a.pshift[p]) & ((1u << a.pbits[p]) - 1u);
      // wave-uniform digit (constant / heavily skewed keys): one lane adds for all
      const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
      const unsigned long long same = __ballot(d == d0);
      const unsigned long long act = __ballot(1);
      if (same == act) {
        if (mbcnt64(act) == 0) atomicAdd(&lhist[p * HIST_STRIDE + d], (uint32_t)__popcll(act));
      } else {
        atomicAdd(&lhist[p * HIST_STRIDE + d], 1u);
      }
    }
  }
  if (oob && a.bad) atomicOr(a.bad, 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < a.npass * HIST_STRIDE; i += XH_BLOCK) {
    const uint32_t c = lhist[i];
    if (c) atomicAdd(&a.hist[i], c);
  }
}

typedef uint32_t xu32x4 __attribute__((ext_vector_type(4)));

// Fast path of the above for ONE integer key column (int32 / int64) read in row order:
// 4 consecutive rows per thread, 16-byte loads and stores.
template <typename T, bool OUT64>
__global__ void __launch_bounds__(XH_BLOCK) xform_hist_int_kernel(XformArgs a) {
  typedef unsigned long long u64;
  __shared__ uint32_t lhist[MAX_PASSES * HIST_STRIDE];
  for (int i = threadIdx.x; i < a.npass * HIST_STRIDE; i += XH_BLOCK) lhist[i] = 0;
  __syncthreads();
  const KeyColDev c = a.cols[0];
  const T* __restrict__ src = static_cast<const T*>(c.data);
  const T na = (T)((u64)1 << (8 * sizeof(T) - 1));
  const uint32_t ngrp = (a.n + 3) / 4;
  const uint32_t stride = gridDim.x * XH_BLOCK;
  bool oob = false;
  for (uint32_t g = blockIdx.x * XH_BLOCK + threadIdx.x; g < ngrp; g += stride) {
    const uint32_t r0 = g * 4;
    const uint32_t nv = a.n - r0 < 4u ? a.n - r0 : 4u;
    T v[4];
    if (nv == 4) {
      constexpr int NV = (int)sizeof(T) / 4;
      xu32x4 w[NV];
      const xu32x4* p = reinterpret_cast<const xu32x4*>(src + r0);
#pragma unroll
      for (int j = 0; j < NV; j++) w[j] = p[j];
      const T* wt = reinterpret_cast<const T*>(w);
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = wt[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = (uint32_t)j < nv ? src[r0 + j] : na;
    }
    u64 k[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u64 u = (u64)(long long)v[j];
      k[j] = (v[j] == na) ? c.na_repl : (c.desc ? c.edge - u + c.inc : u - c.edge + c.inc);
      if ((uint32_t)j < nv && v[j] != na && !xform_in_range(c, k[j])) oob = true;
    }
    if (nv == 4) {
      if (OUT64) {
        xu32x4* o = reinterpret_cast<xu32x4*>(static_cast<u64*>(a.out) + r0);
        xu32x4 w0, w1;
        w0.x = (uint32_t)k[0]; w0.y = (uint32_t)(k[0] >> 32); w0.z = (uint32_t)k[1]; w0.w = (uint32_t)(k[1] >> 32);
        w1.x = (uint32_t)k[2]; w1.y = (uint32_t)(k[2] >> 32); w1.z = (uint32_t)k[3]; w1.w = (uint32_t)(k[3] >> 32);
        o[0] = w0; o[1] = w1;
      } else {
        xu32x4 w0;
        w0.x = (uint32_t)k[0]; w0.y = (uint32_t)k[1]; w0.z = (uint32_t)k[2]; w0.w = (uint32_t)k[3];
        *reinterpret_cast<xu32x4*>(static_cast<uint32_t*>(a.out) + r0) = w0;
      }
    } else {
      for (uint32_t j = 0; j < nv; j++) {
        if (OUT64) static_cast<u64*>(a.out)[r0 + j] = k[j];
        else static_cast<uint32_t*>(a.out)[r0 + j] = (uint32_t)k[j];
      }
    }
    for (int p = 0; p < a.npass; p++) {
      const int sh = a.pshift[p];
      const uint32_t msk = (1u << a.pbits[p]) - 1u;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if ((uint32_t)j < nv) (void)lds_count_peel_rank(&lhist[p * HIST_STRIDE], (uint32_t)(k[j] >> sh) & msk);
      }
    }
  }
  if (oob && a.bad) atomicOr(a.bad, 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < a.npass * HIST_STRIDE; i += XH_BLOCK) {
    const uint32_t cnt = lhist[i];
    if (cnt) atomicAdd(&a.hist[i], cnt);
  }
}

int launch_xform_hist(dthip_ctx* ctx, const XformArgs& a) {
  if (a.n == 0) return DTHIP_OK;
  const long long maxb = (long long)ctx->num_cus * 8;
  const bool fast = a.ncols == 1 && a.order == nullptr && a.cols[0].shift == 0 &&
                    (a.cols[0].stype == DTHIP_INT32 || a.cols[0].stype == DTHIP_INT64) &&
                    (reinterpret_cast<uintptr_t>(a.cols[0].data) & 15) == 0;
  if (fast) {
    long long blocks = ((long long)a.n + XH_BLOCK * 16 - 1) / (XH_BLOCK * 16);
    if (blocks > maxb) blocks = maxb;
    const unsigned g = (unsigned)blocks;
    if (a.cols[0].stype == DTHIP_INT64) {
      if (a.out64) { DTHIP_LAUNCH(ctx, "xform_hist_kernel", (xform_hist_int_kernel<long long, true>), g, XH_BLOCK, 0, a); }
      else { DTHIP_LAUNCH(ctx, "xform_hist_kernel", (xform_hist_int_kernel<long long, false>), g, XH_BLOCK, 0, a); }
    } else {
      if (a.out64) { DTHIP_LAUNCH(ctx, "xform_hist_kernel", (xform_hist_int_kernel<int32_t, true>), g, XH_BLOCK, 0, a); }
      else { DTHIP_LAUNCH(ctx, "xform_hist_kernel", (xform_hist_int_kernel<int32_t, false>), g, XH_BLOCK, 0, a); }
    }
    return DTHIP_OK;
  }
  long long blocks = ((long long)a.n + XH_BLOCK * 8 - 1) / (XH_BLOCK * 8);
  if (blocks > maxb) blocks = maxb;
  DTHIP_LAUNCH(ctx, "xform_hist_kernel", xform_hist_kernel, (unsigned)blocks, XH_BLOCK, 0, a);
  return DTHIP_OK;
}

// exclusive scan of each pass' histogram -> bucket start offsets
__global__ void __launch_bounds__(HIST_STRIDE) hist_scan_kernel(const uint32_t* hist, uint32_t* base) {
  __shared__ uint32_t scratch[HIST_STRIDE / 64];
  const uint32_t v = hist[blockIdx.x * HIST_STRIDE + threadIdx.x];
  const uint32_t e = block_excl_scan_u32<HIST_STRIDE>(v, scratch, nullptr);
  base[blockIdx.x * HIST_STRIDE + threadIdx.x] = e;
}

int launch_hist_scan(dthip_ctx* ctx, const uint32_t* hist, uint32_t* base, int npass) {
  if (npass == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "hist_scan_kernel", hist_scan_kernel, npass, HIST_STRIDE, 0, hist, base);
  return DTHIP_OK;
}

// ---------------------------------------------------------------------------
// one-sweep radix pass
// ---------------------------------------------------------------------------
// Tile = 512 threads x 16 keys.  The global position of every (tile, digit) run is known before
// the pass starts -- radix_tile_hist_kernel counts the digits of every tile of the CURRENT key
// order (one extra 4/8-byte read per row and pass) and bucket_gscan_kernel turns the counts into
// positions -- so the pass needs no inter-workgroup communication: no decoupled look-back, no
// tickets, no spinning (measured at 1e9 rows: 7.2 ms per pass with look-back, 5.6 + 0.8 ms without).

template <typename KeyT>
struct PassArgsT {
  const KeyT* kin; KeyT* kout;
  uint32_t n; int shift; int bits;
  const uint32_t* P;        // [ntiles][bins] rows of digit d in the earlier tiles of the same group
  const uint32_t* gpre;     // [G][bins]      global position of group g's first row of digit d
  uint32_t tpg;             // tiles per group
  int iota;
  // MSD levels (sort_stage_msd in api.hip).  A level below the first works INSIDE the buckets of the level above, so its
  // tiles are ragged: tdesc[t] = {first row, rows, histogram group, -} never spans two parent buckets.  The final
  // level sorts every bucket of the last scatter level in LDS: tile t = rows [bounds[t], bounds[t+1]) and the tile's
  // rows, ordered by the remaining digit, go back to the tile's own row range (seq): sequential writes, no histogram.
  const uint32_t* tdesc;
  const uint32_t* bounds;
  int seq;
  const uint32_t* wfirst;   // R2: first bucket of every window
  int wpairs;               // R2: bounds holds (start, end) pairs (greedily packed windows)
  int bits2;                // R2: bits of the bucket number inside a window
  // last pass of a single-key sort whose key column is wanted in sorted order: the ORIGINAL key values (int32 / int64)
  // are written instead of the packed transformed keys (saves the untransform pass: 4 B read + 8 B written per row)
  void* ukout; int uk_stype; int uk_desc; int uk_bits;
  unsigned long long uk_edge, uk_na_repl, uk_inc;
  // gather mode (GATH): see RadixPass::g_dirT
  const uint16_t* dirT; uint32_t dstride; const uint32_t* cc; uint32_t ntb, ntiles1, T1; const uint32_t* pstart;
  uint16_t* dir2;           // GATH == 3 / 4: tile-local output + directory instead of P / gpre (RadixPass::tl_dir2)
  const u32x4* rec;         // GATH == 4: the level above wrote 16-byte records {key, 4-byte value, 8-byte value} (RadixPass::g_rec)
  // GATH == 2 (final level over windows of a tile-local level): RadixPass::g2_*
  const uint16_t* dirT2; uint32_t dstride2; const uint32_t* pfirst; const uint32_t* gfstart; int s2bits; uint32_t nbk;
  PayCols pay;
};

// per-tile digit counts of a key array: P and group totals (same scheme as bucket_hist_kernel).
// tdesc / gdesc (nullable): ragged tiles {first row, rows, group, -} and groups {first tile, tiles} of an MSD level
template <typename KeyT>
__global__ void __launch_bounds__(RP_BLOCK) radix_tile_hist_kernel(const KeyT* __restrict__ keys, uint32_t n, int shift, int bits,
                                                                 uint32_t ntiles, uint32_t tpg, uint32_t* __restrict__ P,
                                                                 uint32_t* __restrict__ gtot, const uint32_t* __restrict__ tdesc,
                                                                 const uint32_t* __restrict__ gdesc) {
  __shared__ uint32_t cnt[HIST_STRIDE];
  const int tid = threadIdx.x;
  const uint32_t bins = 1u << bits, dmask = bins - 1u;
  constexpr int HK = (HIST_STRIDE + RP_BLOCK - 1) / RP_BLOCK;      // bins per thread (1 unless the tile has fewer threads than bins)
  uint32_t run[HK];
#pragma unroll
  for (int k = 0; k < HK; k++) { run[k] = 0; if (tid + k * RP_BLOCK < HIST_STRIDE) cnt[tid + k * RP_BLOCK] = 0; }
  __syncthreads();
  uint32_t t0 = blockIdx.x * tpg, t1 = (t0 + tpg < ntiles) ? t0 + tpg : ntiles;
  if (gdesc) { t0 = gdesc[2 * blockIdx.x]; t1 = t0 + gdesc[2 * blockIdx.x + 1]; }
  typedef uint32_t hu32x4 __attribute__((ext_vector_type(4), aligned(4)));      // ragged tiles start at any row
  constexpr int KPV = 16 / (int)sizeof(KeyT);          // keys per 16-byte load
  constexpr int NV = RP_ITEMS / KPV;
  for (uint32_t t = t0; t < t1; t++) {
    uint32_t tile_base = t * (uint32_t)RP_TILE;
    uint32_t nvalid = (n - tile_base < (uint32_t)RP_TILE) ? (n - tile_base) : (uint32_t)RP_TILE;
    if (tdesc) { tile_base = tdesc[4 * t]; nvalid = tdesc[4 * t + 1]; }
    if (nvalid == (uint32_t)RP_TILE) {
      const hu32x4* src = reinterpret_cast<const hu32x4*>(keys + tile_base);
      hu32x4 w[NV];
#pragma unroll
      for (int j = 0; j < NV; j++) w[j] = src[j * RP_BLOCK + tid];
      const KeyT* k = reinterpret_cast<const KeyT*>(w);
#pragma unroll
      for (int j = 0; j < RP_ITEMS; j++) (void)lds_count_rank(cnt, (uint32_t)(k[j] >> shift) & dmask);
    } else {
      for (uint32_t i = tid; i < nvalid; i += RP_BLOCK) (void)lds_count_rank(cnt, (uint32_t)(keys[tile_base + i] >> shift) & dmask);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HK; k++) {
      const uint32_t b = (uint32_t)tid + (uint32_t)k * RP_BLOCK;
      if (b < bins) {
        const uint32_t c = cnt[b];
        cnt[b] = 0;
        P[(size_t)t * bins + b] = run[k];
        run[k] += c;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < HK; k++) {
    const uint32_t b = (uint32_t)tid + (uint32_t)k * RP_BLOCK;
    if (b < bins) gtot[(size_t)blockIdx.x * bins + b] = run[k];
  }
}

// MSD level below the first: the groups of parent bucket b are [gfirst[b], gfirst[b+1]).  One workgroup per parent
// bucket, thread = digit: gtot[g][d] (rows of digit d in group g) -> the global position of group g's first row of digit
// d; fstart[b * bins + d] = first row of the child bucket (b, d) (and fstart[nb * bins] = n); *maxsize = largest child.
__global__ void __launch_bounds__(HIST_STRIDE) msd_scan_kernel(uint32_t* __restrict__ gtot, const uint32_t* __restrict__ gfirst,
                                                               const uint32_t* __restrict__ pstart, int bits, uint32_t nb, uint32_t n,
                                                               uint32_t* __restrict__ fstart, uint32_t* __restrict__ maxsize) {
  __shared__ uint32_t scratch[HIST_STRIDE / 64];
  const uint32_t bins = 1u << bits, b = blockIdx.x, d = threadIdx.x;
  const uint32_t g0 = gfirst[b], g1 = gfirst[b + 1];
  uint32_t run = 0;
  if (d < bins) {
    for (uint32_t g = g0; g < g1; g++) {
      const uint32_t c = gtot[(size_t)g * bins + d];
      gtot[(size_t)g * bins + d] = run;
      run += c;
    }
  }
  const uint32_t excl = block_excl_scan_u32<HIST_STRIDE>(run, scratch, nullptr);
  if (d < bins) {
    const uint32_t fs = pstart[b] + excl;
    fstart[(size_t)b * bins + d] = fs;
    for (uint32_t g = g0; g < g1; g++) gtot[(size_t)g * bins + d] += fs;
  }
  // largest child bucket: the wave maximum is taken by ALL lanes (lanes past the last digit contribute 0 -- run is 0
  // there), never inside the divergent branch
  uint32_t m = run;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)m, o, 64); m = x > m ? x : m; }
  if ((d & 63u) == 0 && m) atomicMax(maxsize, m);
  if (b + 1 == nb && d == 0) fstart[(size_t)nb * bins] = n;
}

// Round 5: windows packed GREEDILY -- whole buckets are added while they fit the tile (and the window spans <= maxspan bucket
// numbers): ~88 % of a tile instead of the 66 % of round 4's equal-step rule leaves when the largest bucket is 1.5x the average
// (config 5: 69k windows instead of 92k).  A greedy scan is sequential, so it runs per PARENT bucket (pb = buckets per parent;
// a window never spans two parents): one wave per parent, lane 0 walks the parent's pb bucket starts in LDS, the windows of a
// parent take a block of slots from an atomic counter.  Windows are (start, end) PAIRS: wbounds[2 w], wbounds[2 w + 1] -- their
// order in the list is arbitrary.  info = {number of windows, 0, largest span}; a bucket bigger than a tile: info[0] = ~0.
__global__ void __launch_bounds__(64) msd_window_greedy_kernel(const uint32_t* __restrict__ fstart, uint32_t pb, uint32_t tile, uint32_t maxspan,
                                                               uint32_t nwmax, uint32_t* __restrict__ wbounds, uint32_t* __restrict__ wfirst,
                                                               uint32_t* __restrict__ info) {
  extern __shared__ uint32_t fs[];                      // [pb + 1] bucket starts of this parent, then [2 * pb] windows found
  const uint32_t p = blockIdx.x;
  for (uint32_t i = threadIdx.x; i <= pb; i += 64) fs[i] = fstart[(size_t)p * pb + i];
  __syncthreads();
  uint32_t* wl = fs + pb + 1;
  __shared__ uint32_t nw_s, base_s;
  if (threadIdx.x == 0) {
    uint32_t nw = 0, span = 0, bad = 0;
    uint32_t c = 0;
    while (c < pb) {
      if (fs[c + 1] == fs[c]) { c++; continue; }        // empty buckets before a window do not count
      const uint32_t c0 = c, row0 = fs[c];
      if (fs[c + 1] - row0 > tile) { bad = 1; break; }
      uint32_t e = c + 1;                               // the window holds buckets [c0, e)
      while (e < pb && e - c0 < maxspan && fs[e + 1] - row0 <= tile) e++;
      while (e > c0 + 1 && fs[e] == fs[e - 1]) e--;     // ... nor do empty buckets behind its last row
      wl[2 * nw] = c0; wl[2 * nw + 1] = e;
      nw++;
      span = e - c0 > span ? e - c0 : span;
      c = e;
    }
    if (bad) { atomicMax(&info[0], 0xFFFFFFFFu); nw = 0; }
    else if (span) atomicMax(&info[2], span);
    nw_s = nw;
    base_s = nw ? atomicAdd(&info[3], nw) : 0u;
  }
  __syncthreads();
  const uint32_t nw = nw_s, base = base_s;
  for (uint32_t i = threadIdx.x; i < nw; i += 64) {
    if (base + i >= nwmax) break;
    const uint32_t c0 = wl[2 * i], e = wl[2 * i + 1];
    wbounds[2 * (base + i)] = fs[c0];
    wbounds[2 * (base + i) + 1] = fs[e];
    wfirst[base + i] = p * pb + c0;
  }
}

// info[4] zeroed by the caller; afterwards info[3] = number of windows (info[0] = ~0: a bucket outgrows a tile), info[2] = span
int launch_msd_windows_greedy(dthip_ctx* ctx, const uint32_t* fstart, uint32_t nparents, uint32_t pb, uint32_t tile, uint32_t maxspan,
                              uint32_t nwmax, uint32_t* wbounds, uint32_t* wfirst, uint32_t* info) {
  DTHIP_LAUNCH(ctx, "msd_window_greedy_kernel", msd_window_greedy_kernel, nparents, 64, (size_t)(3 * pb + 1) * 4, fstart, pb, tile, maxspan,
               nwmax, wbounds, wfirst, info);
  return DTHIP_OK;
}

int launch_msd_scan(dthip_ctx* ctx, uint32_t* gtot, const uint32_t* gfirst, const uint32_t* pstart, int bits, uint32_t nb,
                    uint32_t n, uint32_t* fstart, uint32_t* maxsize) {
  DTHIP_LAUNCH(ctx, "msd_scan_kernel", msd_scan_kernel, nb, HIST_STRIDE, 0, gtot, gfirst, pstart, bits, nb, n, fstart, maxsize);
  return DTHIP_OK;
}

// RB   = number of ballot rounds (>= bits of every pass run with this instance)
// P0W  = byte width of payload column 0 when it is prefetched with the keys (0: none / iota)
// P1W  = the same for payload column 1 (only with P0W == 8): both columns' loads are in flight before the ranking
// RK   = how a key finds the lanes of its wave that hold the same digit (the stable "match any"):
//        0  RB ballot rounds, one per digit bit (6 VALU instructions per bit and key: 154 per key at 9 bits, which kept
//           every SIMD busy 37 % of the pass -- profiles/r03_sq_tcc_counters.txt);
//        1  through LDS: every lane ORs its lane bit into a wave-private 64-bit word per digit (ds_or_b64), reads the
//           word back -- the wave's DS instructions execute in order, so the word then holds exactly the lanes of this
//           item with this digit -- and the lowest of them clears it for the next item.  popcount below the lane = the
//           stable rank inside the item.  Three DS instructions and ~10 VALU per key, whatever the digit width.
// BLK  = threads per workgroup: RP_BLOCK, or 256 for the final MSD level over small buckets (a bucket of ~2000 rows gives
//        a 512-thread workgroup four items per wave: too few bytes in flight per CU to cover the HBM latency of its
//        load -> rank -> store chain; four 256-thread workgroups per CU hold twice as many)
// R2   = final MSD level over WINDOWS (whole consecutive buckets of the last scatter level, together at most one tile):
//        two ranking rounds in LDS -- by the low a.bits bits, then by the bucket (key >> a.bits) - wfirst[tile], a.bits2 bits
//        -- give every row its place in the window; rows, ordered, go back over the window's own row range
// GATH = the tile's rows are gathered from the segments of a tile-local level above (RadixPass::g_dirT): their global rows
//        are listed in LDS first (in the exchange buffer, free until the ranking), every load goes through that list
//        (1: a level reading the first, tile-local level and scattering to exact positions; 3: the same level writing its
//        rows tile-locally again + a directory; 2: the final level over windows reading a tile-local second level)
// A tile whose rows go back over ONE contiguous row range -- the windowed final level (R2) and GATH == 3 -- knows its global
// positions at compile time (first row + slot): no per-row position array, no digit recomputation for the stores.  PMC showed
// why that matters (profiles/r05_c5_pmc.txt): the 26-28 spilled VGPRs of these instances were 3.6 + 5.1 GB of scratch
// WRITES per config-5 query, a quarter of what the two kernels wrote.
template <typename KeyT, int RB, int P0W, int P1W = 0, int RK = 0, int BLK = RP_BLOCK, bool R2 = false, int GATH = 0>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(DTHIP_RP_WAVES, DTHIP_RP_WAVES))) radix_pass_kernel(PassArgsT<KeyT> a) {
  constexpr int BLOCK = BLK, ITEMS = RP_ITEMS;
  constexpr int WAVES = BLOCK / 64, TILE = BLOCK * ITEMS;
  constexpr int GROUPS = ITEMS / 4;     // each thread owns GROUPS groups of 4 consecutive tile-sorted slots
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bins = 1 << a.bits;
  const uint32_t dmask = (uint32_t)bins - 1u;
  const int cbins = R2 ? (1 << ((a.bits > a.bits2 || a.bits2 > 31) ? a.bits : a.bits2)) : bins;      // the LDS arrays hold the wider of the two rounds
  // per-wave digit counts as 16-bit words (a wave holds 64 x ITEMS = 1024 keys, a tile 8192): at 512 bins the
  // 32-bit form took the LDS a second workgroup per CU needs, which is what made a 9-bit pass twice as slow
  uint16_t* wh = reinterpret_cast<uint16_t*>(smem);   // [WAVES][bins] per-wave digit counts
  uint32_t* bin_excl = reinterpret_cast<uint32_t*>(wh + WAVES * cbins);   // [bins] tile-local exclusive digit start
  uint32_t* bin_delta = bin_excl + cbins;              // [bins] global start - local start
  uint32_t* misc = bin_delta + cbins;                  // [16]
  unsigned char* exch = reinterpret_cast<unsigned char*>(misc + 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // tiles are dealt to XCDs (block b runs on XCD b % 8: speed only) in contiguous ranges, so the
  // neighbouring runs of a digit are completed in one XCD's L2
  const uint32_t nt = gridDim.x, bi = blockIdx.x;
  const uint32_t xq = nt / 8, xr = nt % 8, xc = bi % 8;
  const uint32_t tile = xc * xq + (xc < xr ? xc : xr) + bi / 8;
  uint32_t tile_base = tile * (uint32_t)TILE;
  uint32_t nvalid = (a.n - tile_base < (uint32_t)TILE) ? (a.n - tile_base) : (uint32_t)TILE;
  uint32_t grp = tile / a.tpg;
  if (R2 && a.wpairs) {                                          // greedily packed windows: (start, end) pairs
    tile_base = a.bounds[2 * tile];
    nvalid = a.bounds[2 * tile + 1] - tile_base;
    if (nvalid > (uint32_t)TILE) nvalid = (uint32_t)TILE;
    if (nvalid == 0) return;
  } else if (a.bounds) {
    tile_base = a.bounds[tile];
    nvalid = a.bounds[tile + 1] - tile_base;
    if (nvalid > (uint32_t)TILE) nvalid = (uint32_t)TILE;     // the host never launches this level over a bigger bucket
    if (nvalid == 0) return;                                   // (uniform: before the first barrier)
  } else if (a.tdesc) {
    tile_base = a.tdesc[4 * tile]; nvalid = a.tdesc[4 * tile + 1]; grp = a.tdesc[4 * tile + 2];
  }
  for (int i = tid; i < WAVES * bins / 2; i += BLOCK) reinterpret_cast<uint32_t*>(wh)[i] = 0;
  __syncthreads();
  const bool full = nvalid == (uint32_t)TILE;
  // wave w owns the `chunk` consecutive rows from w * chunk on (wave-striped: item i at wbase + 64 * i); a short tile
  // (the last one, a ragged one, a final bucket) is shared out evenly, so that all waves rank and move rows
  const uint32_t chunk = full ? 64u * ITEMS : ((((nvalid + WAVES - 1) / WAVES) + 63u) & ~63u);
  const uint32_t wbase = (uint32_t)wave * chunk + (uint32_t)lane;
#define RP_VALID(i) (64u * (uint32_t)(i) < chunk && wbase + 64u * (uint32_t)(i) < nvalid)
  const uint32_t* gsrc_rows = reinterpret_cast<const uint32_t*>(exch);       // GATH: global row of the tile's v-th row
  constexpr bool SEQOUT = R2 || GATH >= 3;       // the tile's rows go back, in tile-sorted order, over [tile_base, tile_base + nvalid)
  if (GATH == 1 || GATH >= 3) {
    const uint32_t bkt = a.tdesc[4 * tile + 3];
    tl_build_src<BLOCK>(reinterpret_cast<uint32_t*>(exch), a.dirT, a.dstride, a.cc, a.ntb, a.ntiles1, a.T1, bkt,
                        tile_base - a.pstart[bkt], nvalid);
    __syncthreads();
  }
  if (GATH == 2) {
    tl_build_src_final<BLOCK>(reinterpret_cast<uint32_t*>(exch), a.dirT2, a.dstride2, a.tdesc, a.pfirst, a.gfstart, a.s2bits, a.nbk,
                              a.wfirst ? a.wfirst[tile] : tile, tile_base, nvalid);
    __syncthreads();
  }
#define RP_SRC(loc) (GATH ? gsrc_rows[(loc)] : tile_base + (loc))

  // ---- load keys (and payload column 0) -----------------------------------
  // Full tiles: 16-byte coalesced loads of the wave's 1024 consecutive keys, transposed
  // through LDS into the wave-striped arrangement the stable ranking needs.
  KeyT key[ITEMS];
  typedef typename std::conditional<P0W == 8, unsigned long long, uint32_t>::type P0T;
  P0T pay0[P0W ? ITEMS : 1];
  if (P0W && GATH != 4) {
    const P0T* pin = static_cast<const P0T*>(a.pay.in[0]);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const uint32_t loc = wbase + 64u * i;
      pay0[i] = RP_VALID(i) ? RP_LD(&pin[RP_SRC(loc)]) : P0T(0);
    }
  }
  typedef typename std::conditional<P1W == 8, unsigned long long, uint32_t>::type P1T;
  P1T pay1[P1W ? ITEMS : 1];
  if (P1W && GATH != 4) {
    const P1T* pin = static_cast<const P1T*>(a.pay.in[1]);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const uint32_t loc = wbase + 64u * i;
      pay1[i] = RP_VALID(i) ? RP_LD(&pin[RP_SRC(loc)]) : P1T(0);
    }
  }
  if (GATH == 4) {
    // one 16-byte record per row: key, the 4-byte riding value (payload 1, or payload 0 when that is the 4-byte one), the 8-byte one
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const uint32_t loc = wbase + 64u * i;
      u32x4 r; r.x = 0; r.y = 0; r.z = 0; r.w = 0;
      if (RP_VALID(i)) r = RP_LD(&a.rec[gsrc_rows[loc]]);
      key[i] = (KeyT)r.x;
      if (P0W == 8) pay0[i] = (P0T)(((unsigned long long)r.w << 32) | r.z);
      else if (P0W == 4) pay0[i] = (P0T)r.y;
      if (P1W == 4) pay1[i] = (P1T)r.y;
    }
    __syncthreads();
  } else if (full && !GATH) {
    constexpr int NV = ITEMS * (int)sizeof(KeyT) / 16;
    const u32x4_u* gsrc = reinterpret_cast<const u32x4_u*>(a.kin + tile_base + (uint32_t)wave * 64u * ITEMS);   // ragged tiles start at any row
    u32x4* wl = reinterpret_cast<u32x4*>(exch) + (size_t)wave * 64 * NV;
    u32x4 tmp[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) tmp[j] = RP_LD(&gsrc[j * 64 + lane]);
#pragma unroll
    for (int j = 0; j < NV; j++) wl[j * 64 + lane] = tmp[j];
    __syncthreads();
    const KeyT* wk = reinterpret_cast<const KeyT*>(wl);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) key[i] = wk[i * 64 + lane];
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const uint32_t loc = wbase + 64u * i;
      key[i] = RP_VALID(i) ? RP_LD(&a.kin[RP_SRC(loc)]) : KeyT(0);
    }
    if (GATH) __syncthreads();      // every wave has read its rows' places: the list makes room for the lane-mask tables
  }

  uint32_t pos[ITEMS];
  if (R2) {
    // ---- final MSD level over a window: two stable rounds, all in LDS ----------------------------------------------
    constexpr uint32_t SLICE = 64u * ITEMS * (uint32_t)sizeof(KeyT);
    uint32_t vmask = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) vmask |= (RP_VALID(i) ? 1u : 0u) << i;
    __syncthreads();                                        // (full tiles: every wave has read its transposed keys)
    rank_round<BLOCK, ITEMS, RB>([&](int i) { return (uint32_t)key[i] & dmask; }, vmask, bins, wh, bin_excl, misc, exch, SLICE, pos);
    __syncthreads();
    // The window's rows arrive ordered by bucket (the scatter levels put them there), so the round above -- stable by
    // the low digit d -- leaves every d-group ordered by bucket.  The wanted order is (bucket, d): a row's place is
    //   first row of its (bucket, d) group  +  its rank inside the group
    //   = exclusive prefix of the counts cnt[bucket][d] in (bucket, d) order
    //   + (rank among all rows with digit d) - (rows with digit d in EARLIER buckets of the window)
    // One DS atomic per row for the counts and one scan over buckets x bins words: no second ranking round, no exchange
    // of rows (the first version of this level did both: 1.15 ms of its 5.5).
    const uint32_t c0 = a.wfirst[tile];
    const uint32_t nseg = 1u << a.bits2;
    const uint32_t total2 = nseg * (uint32_t)bins;
    uint32_t* cnt2 = reinterpret_cast<uint32_t*>(exch);              // [nseg][bins] rows of (bucket, digit); then: in earlier buckets
    uint32_t* base2 = cnt2 + total2;                                 // [nseg][bins] first place of the (bucket, digit) group
    for (uint32_t f = tid; f < total2; f += BLOCK) cnt2[f] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++)
      if ((vmask >> i) & 1u) atomicAdd(&cnt2[(((uint32_t)(key[i] >> a.bits)) - c0) * (uint32_t)bins + ((uint32_t)key[i] & dmask)], 1u);
    __syncthreads();
    {
      const uint32_t per = (total2 + BLOCK - 1) / BLOCK, f0 = (uint32_t)tid * per;
      uint32_t sum = 0;
      for (uint32_t j = 0; j < per; j++) if (f0 + j < total2) sum += cnt2[f0 + j];
      uint32_t e = block_excl_scan_u32<BLOCK>(sum, misc, nullptr);
      for (uint32_t j = 0; j < per; j++) if (f0 + j < total2) { base2[f0 + j] = e; e += cnt2[f0 + j]; }
    }
    __syncthreads();
    for (int b = tid; b < bins; b += BLOCK) {
      uint32_t run = 0;
      for (uint32_t sg = 0; sg < nseg; sg++) { const uint32_t c = cnt2[sg * bins + b]; cnt2[sg * bins + b] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      if ((vmask >> i) & 1u) {
        const uint32_t d = (uint32_t)key[i] & dmask;
        const uint32_t f = (((uint32_t)(key[i] >> a.bits)) - c0) * (uint32_t)bins + d;
        pos[i] = base2[f] + (pos[i] - bin_excl[d] - cnt2[f]);
      }
    }
    __syncthreads();
    for (int b = tid; b < bins; b += BLOCK) bin_delta[b] = tile_base;
    __syncthreads();
  } else {
  // ---- stable rank of every key among equal digits of its wave --------------
  // (the cross-lane traffic below goes through wavefront-scope relaxed atomics, not `volatile`: volatile accesses lose
  // the LDS address space and become flat_load / flat_store ... sc0 sc1 with a full vmcnt(0) wait each -- rounds 1-3
  // paid that for the 16-bit counters, 32 flat accesses per lane and tile)
  uint16_t* mywh = wh + wave * bins;
  if (RK == 1) {
    // the wave's own slice of `exch` (its transposed keys, all read by now: DS instructions of a wave run in order)
    // becomes its table of lane masks, one 64-bit word per digit
    constexpr uint32_t SLICE = 64u * ITEMS * (uint32_t)sizeof(KeyT);
    if (SLICE < (8u << RB)) __syncthreads();        // narrow slices: the table spills into the neighbours' keys
    const uint32_t stride = SLICE < (8u << RB) ? (8u << RB) : SLICE;
    unsigned long long* mk = reinterpret_cast<unsigned long long*>(exch + (size_t)wave * stride);
    for (int b = lane; b < bins; b += 64) mk[b] = 0ULL;
    const unsigned long long mybit = 1ULL << lane;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const bool valid = RP_VALID(i);
      const uint32_t d = (uint32_t)(key[i] >> a.shift) & dmask;
      pos[i] = 0;
      __builtin_amdgcn_wave_barrier();
      if (valid) {
        __hip_atomic_fetch_or(&mk[d], mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const unsigned long long m = __hip_atomic_load(&mk[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t prev = __hip_atomic_load(&mywh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t below = mbcnt64(m);
        pos[i] = prev + below;
        if (below == 0) {
          __hip_atomic_store(&mk[d], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          __hip_atomic_store(&mywh[d], (uint16_t)(prev + (uint32_t)__popcll(m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const bool valid = RP_VALID(i);
      const uint32_t d = (uint32_t)(key[i] >> a.shift) & dmask;
      unsigned long long m = __ballot(valid);
#pragma unroll
      for (int b = 0; b < RB; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      const uint32_t below = mbcnt64(m);
      const uint32_t cnt = (uint32_t)__popcll(m);
      uint32_t prev = 0;
      __builtin_amdgcn_wave_barrier();
      if (valid) prev = __hip_atomic_load(&mywh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      pos[i] = prev + below;
      if (valid && below == 0) __hip_atomic_store(&mywh[d], (uint16_t)(prev + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  __syncthreads();

  // ---- per-digit: wave offsets, tile count, global position of the run ----------
  // (KB consecutive digits per thread: 1 unless the workgroup has fewer threads than bins)
  constexpr int KB = ((1 << RB) + BLOCK - 1) / BLOCK;
  uint32_t tc[KB], tsum = 0;
#pragma unroll
  for (int k = 0; k < KB; k++) {
    const int b = tid * KB + k;
    tc[k] = 0;
    if (b < bins) {
      uint32_t s = 0;
#pragma unroll
      for (int w = 0; w < WAVES; w++) {
        const uint32_t c = wh[w * bins + b];
        wh[w * bins + b] = (uint16_t)s;
        s += c;
      }
      tc[k] = s;
    }
    tsum += tc[k];
  }
  uint32_t excl = block_excl_scan_u32<BLOCK>(tsum, misc, nullptr);
#pragma unroll
  for (int k = 0; k < KB; k++) {
    const int b = tid * KB + k;
    if (b < bins) {
      bin_excl[b] = excl;
      if (GATH >= 3) {
        // second tile-local level: the tile's rows go over its own rows of the outputs, the directory says where digit b starts
        a.dir2[(size_t)tile * (uint32_t)(bins + 1) + b] = (uint16_t)excl;
        if (b == bins - 1) a.dir2[(size_t)tile * (uint32_t)(bins + 1) + bins] = (uint16_t)nvalid;
      } else
      bin_delta[b] = a.seq ? tile_base : a.gpre[(size_t)grp * bins + b] + a.P[(size_t)tile * bins + b] - excl;
      excl += tc[k];
    }
  }
  __syncthreads();

  }     // !R2

  // ---- keys: registers -> LDS in tile-sorted order -> global ---------------
  KeyT* ek = reinterpret_cast<KeyT*>(exch);
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    if (RP_VALID(i)) {
      if (!R2) {
        const uint32_t d = (uint32_t)(key[i] >> a.shift) & dmask;
        pos[i] += bin_excl[d] + wh[wave * bins + d];
      }
      ek[pos[i]] = key[i];
    }
  }
  __syncthreads();
  // thread owns slots (g*BLOCK + tid)*4 .. +3 for g in [0, GROUPS)
  uint32_t gpos[SEQOUT ? 1 : ITEMS];
#pragma unroll
  for (int g = 0; g < GROUPS; g++) {
    const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
    const uint32_t nv = s0 < nvalid ? (nvalid - s0 < 4u ? nvalid - s0 : 4u) : 0u;
    KeyT kk[4];
    uint32_t gp[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      kk[j] = ek[s0 + j];     // slots beyond nvalid hold stale data, never stored
      if (SEQOUT) {
        gp[j] = tile_base + s0 + j;
      } else {
        const uint32_t d = (uint32_t)(kk[j] >> a.shift) & dmask;
        gp[j] = bin_delta[d] + s0 + j;
        gpos[g * 4 + j] = gp[j];
      }
    }
    if (a.ukout) {
      // inverse of the integer key transform (sort.cc:728-776), as in group.hip's untransform_kernel
      long long uv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        unsigned long long k = (unsigned long long)kk[j];
        if (a.uk_bits < 64) k &= (1ULL << a.uk_bits) - 1ULL;
        const unsigned long long u = a.uk_desc ? a.uk_edge - (k - a.uk_inc) : (k - a.uk_inc) + a.uk_edge;
        uv[j] = (k == a.uk_na_repl) ? (a.uk_stype == DTHIP_INT64 ? (long long)INT64_MIN : (long long)INT32_MIN) : (long long)u;
      }
      if (nv) {
        if (a.uk_stype == DTHIP_INT64) store_group4<long long>(static_cast<long long*>(a.ukout), gp, uv, nv);
        else {
          uint32_t u4[4];
#pragma unroll
          for (int j = 0; j < 4; j++) u4[j] = (uint32_t)(int32_t)uv[j];
          store_group4<uint32_t>(static_cast<uint32_t*>(a.ukout), gp, u4, nv);
        }
      }
    } else if (nv) store_group4<KeyT>(a.kout, gp, kk, nv);
  }

  // ---- payload columns follow the same permutation --------------------------
  // KIND 1 / 2: the column waits in pay0 / pay1 (prefetched with the keys), 3: it is the row numbers, 0: it is loaded here --
  // ALL of the thread's loads first (rows past the end of a short tile read the tile's first row and drop it), then the LDS
  // writes.  (Rounds 1-5 had `if (valid) e[pos] = load` per item: sixteen load -> s_waitcnt vmcnt(0) -> ds_write round trips
  // per thread, tile and column.)  Columns 0 and 1 are peeled off the loop so that pay0 / pay1 are dead in the other branches.
#ifndef RP_INPL
#define RP_INPL 4         // loads of a column that is loaded here in flight per thread
#endif
  auto paycol = [&](int c, auto kindc) {
    constexpr int KIND = decltype(kindc)::value;
    const bool w4 = KIND == 3 ? true : KIND == 1 ? P0W == 4 : KIND == 2 ? P1W == 4 : a.pay.width[c] == 4;
    __syncthreads();
    if (w4) {
      uint32_t* e4 = reinterpret_cast<uint32_t*>(exch);
      const uint32_t* pin = static_cast<const uint32_t*>(a.pay.in[c]);
      uint32_t* pout = static_cast<uint32_t*>(a.pay.out[c]);
      if (KIND == 0) {
#pragma unroll
        for (int h = 0; h < ITEMS; h += RP_INPL) {
          uint32_t v4[RP_INPL];
#pragma unroll
          for (int i = 0; i < RP_INPL; i++) v4[i] = RP_LD(&pin[tile_base + (RP_VALID(h + i) ? wbase + 64u * (h + i) : 0u)]);
#pragma unroll
          for (int i = 0; i < RP_INPL; i++) if (RP_VALID(h + i)) e4[pos[h + i]] = v4[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
          if (RP_VALID(i)) {
            uint32_t v;
            if (KIND == 3) v = tile_base + wbase + 64u * i;
            else if (KIND == 1) v = (uint32_t)pay0[i];
            else v = (uint32_t)pay1[i];
            e4[pos[i]] = v;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < GROUPS; g++) {
        const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
        const uint32_t nv = s0 < nvalid ? (nvalid - s0 < 4u ? nvalid - s0 : 4u) : 0u;
        uint32_t vv[4], gp[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { vv[j] = e4[s0 + j]; gp[j] = SEQOUT ? tile_base + s0 + j : gpos[g * 4 + j]; }
        if (nv) store_group4<uint32_t>(pout, gp, vv, nv);
      }
    } else {
      unsigned long long* e8 = reinterpret_cast<unsigned long long*>(exch);
      const unsigned long long* pin = static_cast<const unsigned long long*>(a.pay.in[c]);
      unsigned long long* pout = static_cast<unsigned long long*>(a.pay.out[c]);
      if (KIND == 0) {
#pragma unroll
        for (int h = 0; h < ITEMS; h += RP_INPL) {
          unsigned long long v8[RP_INPL];
#pragma unroll
          for (int i = 0; i < RP_INPL; i++) v8[i] = RP_LD(&pin[tile_base + (RP_VALID(h + i) ? wbase + 64u * (h + i) : 0u)]);
#pragma unroll
          for (int i = 0; i < RP_INPL; i++) if (RP_VALID(h + i)) e8[pos[h + i]] = v8[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
          if (RP_VALID(i)) e8[pos[i]] = KIND == 1 ? (unsigned long long)pay0[i] : (unsigned long long)pay1[i];
        }
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < GROUPS; g++) {
        const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
        const uint32_t nv = s0 < nvalid ? (nvalid - s0 < 4u ? nvalid - s0 : 4u) : 0u;
        unsigned long long vv[4];
        uint32_t gp[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { vv[j] = e8[s0 + j]; gp[j] = SEQOUT ? tile_base + s0 + j : gpos[g * 4 + j]; }
        if (nv) store_group4<unsigned long long>(pout, gp, vv, nv);
      }
    }
  };
  if (a.pay.n > 0) {
    if (a.iota) paycol(0, std::integral_constant<int, 3>());
    else if (P0W) paycol(0, std::integral_constant<int, 1>());
    else paycol(0, std::integral_constant<int, 0>());
  }
  if (a.pay.n > 1) {
    if (P1W) paycol(1, std::integral_constant<int, 2>());
    else paycol(1, std::integral_constant<int, 0>());
  }
  for (int c = 2; c < a.pay.n; c++) paycol(c, std::integral_constant<int, 0>());
}

#undef RP_VALID
#undef RP_SRC

uint32_t radix_tile_items(int, int) { return (uint32_t)RP_TILE; }

int launch_radix_tile_hist(dthip_ctx* ctx, const void* keys, int key64, uint32_t n, int shift, int bits,
                           uint32_t ntiles, uint32_t tpg, uint32_t G, uint32_t* P, uint32_t* gtot, const uint32_t* tdesc,
                           const uint32_t* gdesc) {
  if (key64) {
    DTHIP_LAUNCH(ctx, "radix_tile_hist_kernel", radix_tile_hist_kernel<unsigned long long>, G, RP_BLOCK, 0,
                 static_cast<const unsigned long long*>(keys), n, shift, bits, ntiles, tpg, P, gtot, tdesc, gdesc);
  } else {
    DTHIP_LAUNCH(ctx, "radix_tile_hist_kernel", radix_tile_hist_kernel<uint32_t>, G, RP_BLOCK, 0,
                 static_cast<const uint32_t*>(keys), n, shift, bits, ntiles, tpg, P, gtot, tdesc, gdesc);
  }
  return DTHIP_OK;
}

template <typename KeyT, int RB, int P0W, int P1W, int RK, int BLK = RP_BLOCK, bool R2 = false, int GATH = 0>
static int launch_pass_r(dthip_ctx* ctx, const RadixPass& p) {
  PassArgsT<KeyT> a;
  a.dirT = p.g_dirT; a.dstride = p.g_dstride; a.cc = p.g_cc; a.ntb = p.g_ntb; a.ntiles1 = p.g_ntiles1; a.T1 = p.g_T1; a.pstart = p.g_pstart;
  a.dir2 = p.tl_dir2; a.rec = static_cast<const u32x4*>(p.g_rec);
  a.dirT2 = p.g2_dirT; a.dstride2 = p.g2_dstride; a.pfirst = p.g2_pfirst; a.gfstart = p.g2_fstart; a.s2bits = p.g2_s2bits; a.nbk = p.g2_nbk;
  a.kin = static_cast<const KeyT*>(p.kin); a.kout = static_cast<KeyT*>(p.kout);
  a.n = p.n; a.shift = p.shift; a.bits = p.bits; a.P = p.P; a.gpre = p.gpre; a.tpg = p.tpg;
  a.iota = p.iota; a.pay = p.pay;
  a.tdesc = p.tdesc; a.bounds = p.bounds; a.seq = p.bounds ? 1 : 0;
  a.wfirst = p.wfirst; a.bits2 = p.bits2; a.wpairs = p.wpairs;
  a.ukout = p.ukout; a.uk_stype = p.uk_stype; a.uk_desc = p.uk_desc; a.uk_bits = p.uk_bits;
  a.uk_edge = p.uk_edge; a.uk_na_repl = p.uk_na_repl; a.uk_inc = p.uk_inc;
  int maxw = (int)sizeof(KeyT);
  for (int c = 0; c < p.pay.n; c++) maxw = p.pay.width[c] > maxw ? p.pay.width[c] : maxw;
  const int bins = 1 << (R2 && p.bits2 > p.bits && p.bits2 < 32 ? p.bits2 : p.bits);
  size_t lds = (size_t)(BLK / 64) * bins * 2 + (size_t)(2 * bins + 16) * 4 + (size_t)(BLK * RP_ITEMS) * maxw;
  auto kfn = radix_pass_kernel<KeyT, RB, P0W, P1W, RK, BLK, R2, GATH>;
  DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 1024));
  const uint32_t ntiles = p.ntiles ? p.ntiles : (p.n + RP_TILE - 1) / RP_TILE;
  DTHIP_LAUNCH(ctx, (p.label ? p.label : "radix_pass_kernel"), kfn, ntiles, BLK, lds, a);
  return DTHIP_OK;
}

// ranking through the LDS lane masks (RK = 1); the ballot ranking of rounds 1-3 (RK = 0) serves the 10-bit final level only
template <typename KeyT, int RB, int P0W, int P1W = 0>
static int launch_pass_t(dthip_ctx* ctx, const RadixPass& p) {
  if (p.wfirst) {        // final MSD level over windows: two rounds in LDS (4-byte keys, digits of <= 9 bits)
    if (sizeof(KeyT) != 4 || RB != 9) { set_error("radix pass: the windowed final level takes 4-byte keys"); return DTHIP_EINVAL; }
    return launch_pass_r<uint32_t, 9, P0W, P1W, 1, RP_BLOCK, true>(ctx, p);
  }
  if (p.block == 256 && sizeof(KeyT) == 4) return launch_pass_r<uint32_t, RB, P0W, P1W, 1, 256>(ctx, p);   // final MSD level, small buckets
  return launch_pass_r<KeyT, RB, P0W, P1W, 1>(ctx, p);
}

template <typename KeyT>
static int launch_pass_k(dthip_ctx* ctx, const RadixPass& p) {
  static const int prefetch = getenv("DTHIP_RP_PREFETCH") ? atoi(getenv("DTHIP_RP_PREFETCH")) : 2;   // payload columns loaded with the keys
  int p0w = (p.pay.n > 0 && !p.iota) ? p.pay.width[0] : 0;
  int p1w = (p0w == 8 && p.pay.n > 1) ? p.pay.width[1] : 0;
  if (prefetch < 2) p1w = 0;
  if (prefetch < 1) p0w = 0;
  // TWO 8-byte columns prefetched next to the keys cost the kernel 96 spilled VGPRs (scripts/kernel_resources.sh) and lose
  // to loading the second one after the ranking: 5e8 rows, int64 key + two float64 columns riding, MSD levels 8.7 + 8.7 + 9.1
  // -> 6.0 + 6.3 + 5.4 ms (29.9 -> 21.1 ms; scripts/sort88_bench.py).  8 + 4 bytes (config 5: value + RowIndex) stay
  // prefetched: there the spills (24) are cheaper than the exposed load (profiles/r04_rank_ab.txt).  3: the old dispatch
  if (p1w == 8 && prefetch != 3) p1w = 0;
  if (p.bits > 9) {
    // final MSD level over 10 bits: 1024 bins rule the per-wave mask tables out (8 KB each), the ballot ranking needs none
    if (sizeof(KeyT) != 4 || !p.bounds) { set_error("radix pass: a 10-bit digit is for the final MSD level only"); return DTHIP_EINVAL; }
    if (p0w == 8 && p1w == 8) return launch_pass_r<uint32_t, 10, 8, 8, 0>(ctx, p);
    if (p0w == 8 && p1w == 4) return launch_pass_r<uint32_t, 10, 8, 4, 0>(ctx, p);
    if (p0w == 8) return launch_pass_r<uint32_t, 10, 8, 0, 0>(ctx, p);
    if (p0w == 4) return launch_pass_r<uint32_t, 10, 4, 0, 0>(ctx, p);
    return launch_pass_r<uint32_t, 10, 0, 0, 0>(ctx, p);
  }
  if (p.bits > 8 || p.wfirst) {
    if (p0w == 8 && p1w == 8) return launch_pass_t<KeyT, 9, 8, 8>(ctx, p);
    if (p0w == 8 && p1w == 4) return launch_pass_t<KeyT, 9, 8, 4>(ctx, p);
    if (p0w == 8) return launch_pass_t<KeyT, 9, 8>(ctx, p);
    if (p0w == 4) return launch_pass_t<KeyT, 9, 4>(ctx, p);
    return launch_pass_t<KeyT, 9, 0>(ctx, p);
  }
  if (p0w == 8 && p1w == 8) return launch_pass_t<KeyT, 8, 8, 8>(ctx, p);
  if (p0w == 8 && p1w == 4) return launch_pass_t<KeyT, 8, 8, 4>(ctx, p);
  if (p0w == 8) return launch_pass_t<KeyT, 8, 8>(ctx, p);
  if (p0w == 4) return launch_pass_t<KeyT, 8, 4>(ctx, p);
  return launch_pass_t<KeyT, 8, 0>(ctx, p);
}

// gather mode: 4-byte keys, ragged tiles, every payload column prefetched (the list of rows lives in the exchange buffer
// only until the ranking starts)
static int launch_pass_gather(dthip_ctx* ctx, const RadixPass& p) {
  const int w0 = p.pay.n > 0 ? p.pay.width[0] : 0, w1 = p.pay.n > 1 ? p.pay.width[1] : 0;
  if (p.key64 || !p.tdesc || p.bounds || p.wfirst || p.iota || p.pay.n < 1 || p.pay.n > 2 || p.bits > 9 || !p.g_cc || !p.g_pstart ||
      (!p.tl_dir2 && (!p.P || !p.gpre)) ||
      !((w0 == 8 && (w1 == 0 || w1 == 4 || w1 == 8)) || (w0 == 4 && w1 == 0))) {
    set_error("radix pass: gather mode takes 4-byte keys, ragged tiles and payload widths 8 / 8+4 / 8+8 / 4");
    return DTHIP_EINVAL;
  }
  if (p.tl_dir2 && p.g_rec) {     // records in, tile-local output + directory out
    if (w1 == 8) { set_error("radix pass: records carry one 8-byte and one 4-byte riding value"); return DTHIP_EINVAL; }
    if (p.bits > 8) {
      if (w0 == 8 && w1 == 4) return launch_pass_r<uint32_t, 9, 8, 4, 1, RP_BLOCK, false, 4>(ctx, p);
      if (w0 == 8) return launch_pass_r<uint32_t, 9, 8, 0, 1, RP_BLOCK, false, 4>(ctx, p);
      return launch_pass_r<uint32_t, 9, 4, 0, 1, RP_BLOCK, false, 4>(ctx, p);
    }
    if (w0 == 8 && w1 == 4) return launch_pass_r<uint32_t, 8, 8, 4, 1, RP_BLOCK, false, 4>(ctx, p);
    if (w0 == 8) return launch_pass_r<uint32_t, 8, 8, 0, 1, RP_BLOCK, false, 4>(ctx, p);
    return launch_pass_r<uint32_t, 8, 4, 0, 1, RP_BLOCK, false, 4>(ctx, p);
  }
  if (p.tl_dir2) {                // tile-local output + directory
    if (p.bits > 8) {
      if (w0 == 8 && w1 == 4) return launch_pass_r<uint32_t, 9, 8, 4, 1, RP_BLOCK, false, 3>(ctx, p);
      if (w0 == 8 && w1 == 8) return launch_pass_r<uint32_t, 9, 8, 8, 1, RP_BLOCK, false, 3>(ctx, p);
      if (w0 == 8) return launch_pass_r<uint32_t, 9, 8, 0, 1, RP_BLOCK, false, 3>(ctx, p);
      return launch_pass_r<uint32_t, 9, 4, 0, 1, RP_BLOCK, false, 3>(ctx, p);
    }
    if (w0 == 8 && w1 == 4) return launch_pass_r<uint32_t, 8, 8, 4, 1, RP_BLOCK, false, 3>(ctx, p);
    if (w0 == 8 && w1 == 8) return launch_pass_r<uint32_t, 8, 8, 8, 1, RP_BLOCK, false, 3>(ctx, p);
    if (w0 == 8) return launch_pass_r<uint32_t, 8, 8, 0, 1, RP_BLOCK, false, 3>(ctx, p);
    return launch_pass_r<uint32_t, 8, 4, 0, 1, RP_BLOCK, false, 3>(ctx, p);
  }
  if (p.bits > 8) {
    if (w0 == 8 && w1 == 4) return launch_pass_r<uint32_t, 9, 8, 4, 1, RP_BLOCK, false, 1>(ctx, p);
    if (w0 == 8 && w1 == 8) return launch_pass_r<uint32_t, 9, 8, 8, 1, RP_BLOCK, false, 1>(ctx, p);
    if (w0 == 8) return launch_pass_r<uint32_t, 9, 8, 0, 1, RP_BLOCK, false, 1>(ctx, p);
    return launch_pass_r<uint32_t, 9, 4, 0, 1, RP_BLOCK, false, 1>(ctx, p);
  }
  if (w0 == 8 && w1 == 4) return launch_pass_r<uint32_t, 8, 8, 4, 1, RP_BLOCK, false, 1>(ctx, p);
  if (w0 == 8 && w1 == 8) return launch_pass_r<uint32_t, 8, 8, 8, 1, RP_BLOCK, false, 1>(ctx, p);
  if (w0 == 8) return launch_pass_r<uint32_t, 8, 8, 0, 1, RP_BLOCK, false, 1>(ctx, p);
  return launch_pass_r<uint32_t, 8, 4, 0, 1, RP_BLOCK, false, 1>(ctx, p);
}

// the final level over windows, reading a tile-local second level (RadixPass::g2_dirT)
static int launch_pass_gather_final(dthip_ctx* ctx, const RadixPass& p) {
  const int w0 = p.pay.n > 0 ? p.pay.width[0] : 0, w1 = p.pay.n > 1 ? p.pay.width[1] : 0;
  if (p.key64 || !p.tdesc || !p.bounds || !p.wfirst || p.iota || p.pay.n < 1 || p.pay.n > 2 || p.bits > 9 || !p.g2_pfirst || !p.g2_fstart ||
      !((w0 == 8 && (w1 == 0 || w1 == 4 || w1 == 8)) || (w0 == 4 && w1 == 0))) {
    set_error("radix pass: the gathering final level takes windows, 4-byte keys and payload widths 8 / 8+4 / 8+8 / 4");
    return DTHIP_EINVAL;
  }
  if (w0 == 8 && w1 == 4) return launch_pass_r<uint32_t, 9, 8, 4, 1, RP_BLOCK, true, 2>(ctx, p);
  if (w0 == 8 && w1 == 8) return launch_pass_r<uint32_t, 9, 8, 8, 1, RP_BLOCK, true, 2>(ctx, p);
  if (w0 == 8) return launch_pass_r<uint32_t, 9, 8, 0, 1, RP_BLOCK, true, 2>(ctx, p);
  return launch_pass_r<uint32_t, 9, 4, 0, 1, RP_BLOCK, true, 2>(ctx, p);
}

int launch_radix_pass(dthip_ctx* ctx, const RadixPass& p) {
  if (p.n == 0) return DTHIP_OK;
  if (p.bits < 1 || p.bits > 10) { set_error("radix pass: bad digit width %d", p.bits); return DTHIP_EINVAL; }
  if (p.g_dirT) return launch_pass_gather(ctx, p);
  if (p.g2_dirT) return launch_pass_gather_final(ctx, p);
  if (p.key64) return launch_pass_k<unsigned long long>(ctx, p);
  return launch_pass_k<uint32_t>(ctx, p);
}

}  // namespace dthip
