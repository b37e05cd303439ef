// tlsort.hip -- the first level of `V = DT[f.x <cmp> c, :]; V[:, cols, by(key)]` (BASELINE config 5) as ONE sweep over the
// unfiltered rows, with a TILE-LOCAL output layout.
//
// Reference path being replaced (behaviour, not algorithm): init_from_boolean_column (src/core/rowindex_array.cc:130-170)
// builds the filter's RowIndex, the view's columns are read through it (column/view.cc:140-145), group() transforms the
// key (sort.cc:728-776 _initI), histograms and reorders by the most significant digit (sort.cc:950-1074), recurses
// (sort.cc:1206-1353).  Round 4 ran that as: count pass + take pass (filter), key-transform pass, tile-histogram pass,
// level-1 scatter -- five sweeps, 13 ms of config 5's 24.  Here ONE kernel does all of it for every tile of 8192 input
// rows: evaluate the predicate, transform the key of the passing rows, rank them stably by the top digit (the lane-mask
// ranking of radix_dev.hpp), and write them -- transformed key, original row number (the composed RowIndex of filter and
// grouping), riding columns -- ordered by that digit into the tile's OWN row range, plus a 16-bit directory of where every
// digit starts inside the tile.  All writes are sequential, nothing needs a histogram or a run position, a tile whose rows
// pass only in part simply leaves the rest of its range unused (no compaction, no global count).  The second level
// collects a bucket's rows from the tiles' segments (tl_build_src: read amplification instead of write amplification) and
// goes on exactly like the MSD levels of api.hip sort_stage: exact-position scatter by the next digit, final buckets
// ordered in LDS.
#include "common.hpp"
#include "device_utils.hpp"
#include "keyxform.hpp"
#include "pred.hpp"
#include "radix_dev.hpp"

namespace dthip {

// KT = the key column's element type (int32_t / long long); KEEPX = the predicate column (8 bytes wide) is one of the
// riding columns: its loaded values stay in registers instead of being read a second time
// BLK = threads per workgroup: 512 (8192-row tiles, two workgroups per CU) or 1024 (16384-row tiles, one per CU: segments
// twice as long for the level that gathers them)
// AOS = the passing rows are written as 16-byte records {key, 4-byte riding value, 8-byte riding value} (TL1Args::rec)
template <typename KT, int RB, bool KEEPX, int BLK, bool AOS = false>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(DTHIP_RP_WAVES, DTHIP_RP_WAVES))) tl_level1_kernel(TL1Args a) {
  typedef unsigned long long u64;
  constexpr int BLOCK = BLK, ITEMS = RP_ITEMS, WAVES = BLOCK / 64, TILE = BLOCK * ITEMS, GROUPS = ITEMS / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bins = 1 << a.bits;
  const uint32_t dmask = (uint32_t)bins - 1u;
  uint16_t* wh = reinterpret_cast<uint16_t*>(smem);                         // [WAVES][bins] per-wave digit counts
  uint32_t* bin_excl = reinterpret_cast<uint32_t*>(wh + WAVES * bins);      // [bins] first place of every digit inside the tile
  uint32_t* misc = bin_excl + bins;                                         // [16]
  unsigned char* exch = reinterpret_cast<unsigned char*>(misc + 16);        // [TILE x 8 bytes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tiles dealt to the XCDs in contiguous ranges (speed only), as in radix_pass_kernel
  const uint32_t nt = gridDim.x, bi = blockIdx.x;
  const uint32_t xq = nt / 8, xr = nt % 8, xc = bi % 8;
  const uint32_t tile = xc * xq + (xc < xr ? xc : xr) + bi / 8;
  const uint32_t tile_base = tile * (uint32_t)TILE;
  const uint32_t nvalid = (a.n - tile_base < (uint32_t)TILE) ? (a.n - tile_base) : (uint32_t)TILE;
  const bool full = nvalid == (uint32_t)TILE;
  const uint32_t chunk = full ? 64u * ITEMS : ((((nvalid + WAVES - 1) / WAVES) + 63u) & ~63u);
  const uint32_t wbase = (uint32_t)wave * chunk + (uint32_t)lane;
#define TL_VALID(i) (64u * (uint32_t)(i) < chunk && wbase + 64u * (uint32_t)(i) < nvalid)

  // ---- predicate column and key column of the tile's rows (wave-striped: item i of wave w = row w * chunk + 64 i + lane)
  const u64* px = static_cast<const u64*>(a.pred.data);
  const KT* pk = static_cast<const KT*>(a.key.data);
  u64 xv[ITEMS];
  KT kv[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) xv[i] = TL_VALID(i) ? RP_LD(&px[tile_base + wbase + 64u * i]) : 0ULL;
#pragma unroll
  for (int i = 0; i < ITEMS; i++) kv[i] = TL_VALID(i) ? RP_LD(&pk[tile_base + wbase + 64u * i]) : KT(0);
  const KT kna = (KT)((u64)1 << (8 * sizeof(KT) - 1));
  uint32_t key[ITEMS];
  uint32_t vmask = 0;
  bool oob = false;
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const bool pass = TL_VALID(i) && pred_val8(a.pred, xv[i]);
    // the key transform of sort.cc:728-776 (NA -> na_repl, else value - min + 1 / max - value + 1)
    const u64 u = (u64)(long long)kv[i];
    const u64 x = (kv[i] == kna) ? a.key.na_repl : (a.key.desc ? a.key.edge - u + a.key.inc : u - a.key.edge + a.key.inc);
    if (pass && kv[i] != kna && !xform_in_range(a.key, x)) oob = true;     // (a guessed key range: the caller plans again)
    key[i] = (uint32_t)x;
    vmask |= (pass ? 1u : 0u) << i;
  }
  if (oob && a.bad) atomicOr(a.bad, 1u);

  // ---- stable rank of the passing rows by the top digit --------------------------------------------------------------
  uint32_t pos[ITEMS];
  uint32_t total = 0;
  rank_round<BLOCK, ITEMS, RB>([&](int i) { return (key[i] >> a.shift) & dmask; }, vmask, bins, wh, bin_excl, misc, exch,
                               8u * (uint32_t)bins, pos, 0, &total);
  __syncthreads();
  {
    uint16_t* drow = a.dir + (size_t)tile * (uint32_t)(bins + 1);
    for (int b = tid; b < bins; b += BLOCK) drow[b] = (uint16_t)bin_excl[b];
    if (tid == 0) drow[bins] = (uint16_t)total;
  }

  if (AOS) {
    // ---- records: {key, 4-byte value} pairs through LDS first, kept in registers; then the 8-byte values; every thread owns 4
    // consecutive places per group = 64 contiguous bytes of records, written with four 16-byte stores
    u64* e8 = reinterpret_cast<u64*>(exch);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      if ((vmask >> i) & 1u) {
        uint32_t v4 = 0;
        if (a.rec4 == -2) v4 = tile_base + wbase + 64u * i;
        else if (a.rec4 >= 0) v4 = RP_LD(&static_cast<const uint32_t*>(a.pay.in[a.rec4])[tile_base + wbase + 64u * i]);
        e8[pos[i]] = (u64)key[i] | ((u64)v4 << 32);
      }
    }
    __syncthreads();
    u64 kr[ITEMS];
#pragma unroll
    for (int g = 0; g < GROUPS; g++) {
      const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
#pragma unroll
      for (int j = 0; j < 4; j++) kr[g * 4 + j] = e8[s0 + j];
    }
    __syncthreads();
    if (a.rec8 >= 0) {
      const u64* pin = static_cast<const u64*>(a.pay.in[a.rec8]);
      const bool isx = KEEPX && a.rec8 == a.keepx;
#pragma unroll
      for (int i = 0; i < ITEMS; i++)
        if ((vmask >> i) & 1u) e8[pos[i]] = (KEEPX && isx) ? xv[i] : RP_LD(&pin[tile_base + wbase + 64u * i]);
      __syncthreads();
    }
    u32x4* rec = static_cast<u32x4*>(a.rec);
#pragma unroll
    for (int g = 0; g < GROUPS; g++) {
      const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (s0 + j < total) {
          const u64 x8 = a.rec8 >= 0 ? e8[s0 + j] : 0ULL;
          u32x4 r;
          r.x = (uint32_t)kr[g * 4 + j]; r.y = (uint32_t)(kr[g * 4 + j] >> 32); r.z = (uint32_t)x8; r.w = (uint32_t)(x8 >> 32);
          RP_ST(&rec[tile_base + s0 + j], r);
        }
      }
    }
    return;
  }
  // ---- transformed keys, row numbers, riding columns: registers -> LDS in (digit, row) order -> the tile's own rows -----
  uint32_t* e4 = reinterpret_cast<uint32_t*>(exch);
#pragma unroll
  for (int i = 0; i < ITEMS; i++) if ((vmask >> i) & 1u) e4[pos[i]] = key[i];
  __syncthreads();
#pragma unroll
  for (int g = 0; g < GROUPS; g++) {
    const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
    const uint32_t nv = s0 < total ? (total - s0 < 4u ? total - s0 : 4u) : 0u;
    uint32_t vv[4], gp[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { vv[j] = e4[s0 + j]; gp[j] = tile_base + s0 + j; }
    if (nv) store_group4<uint32_t>(a.kout, gp, vv, nv);
  }
  if (a.rowid) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) if ((vmask >> i) & 1u) e4[pos[i]] = tile_base + wbase + 64u * i;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GROUPS; g++) {
      const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
      const uint32_t nv = s0 < total ? (total - s0 < 4u ? total - s0 : 4u) : 0u;
      uint32_t vv[4], gp[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { vv[j] = e4[s0 + j]; gp[j] = tile_base + s0 + j; }
      if (nv) store_group4<uint32_t>(a.rowid, gp, vv, nv);
    }
  }
  for (int c = 0; c < a.pay.n; c++) {
    __syncthreads();
    if (a.pay.width[c] == 4) {
      const uint32_t* pin = static_cast<const uint32_t*>(a.pay.in[c]);
      uint32_t* pout = static_cast<uint32_t*>(a.pay.out[c]);
#pragma unroll
      for (int i = 0; i < ITEMS; i++) if ((vmask >> i) & 1u) e4[pos[i]] = RP_LD(&pin[tile_base + wbase + 64u * i]);
      __syncthreads();
#pragma unroll
      for (int g = 0; g < GROUPS; g++) {
        const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
        const uint32_t nv = s0 < total ? (total - s0 < 4u ? total - s0 : 4u) : 0u;
        uint32_t vv[4], gp[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { vv[j] = e4[s0 + j]; gp[j] = tile_base + s0 + j; }
        if (nv) store_group4<uint32_t>(pout, gp, vv, nv);
      }
    } else {
      u64* e8 = reinterpret_cast<u64*>(exch);
      const u64* pin = static_cast<const u64*>(a.pay.in[c]);
      u64* pout = static_cast<u64*>(a.pay.out[c]);
      const bool isx = KEEPX && c == a.keepx;
#pragma unroll
      for (int i = 0; i < ITEMS; i++)
        if ((vmask >> i) & 1u) e8[pos[i]] = (KEEPX && isx) ? xv[i] : RP_LD(&pin[tile_base + wbase + 64u * i]);
      __syncthreads();
#pragma unroll
      for (int g = 0; g < GROUPS; g++) {
        const uint32_t s0 = ((uint32_t)g * BLOCK + tid) * 4u;
        const uint32_t nv = s0 < total ? (total - s0 < 4u ? total - s0 : 4u) : 0u;
        u64 vv[4];
        uint32_t gp[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { vv[j] = e8[s0 + j]; gp[j] = tile_base + s0 + j; }
        if (nv) store_group4<u64>(pout, gp, vv, nv);
      }
    }
  }
#undef TL_VALID
}

// how many of `nsamp` evenly spaced rows pass the predicate (sizes the digits of the levels; correctness never depends on it)
__global__ void __launch_bounds__(256) tl_pred_sample_kernel(PredArgs p, uint32_t n, uint32_t nsamp, uint32_t* count) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  bool f = false;
  if (i < nsamp) {
    const uint32_t row = (uint32_t)(((unsigned long long)i * n) / nsamp);
    f = pred_at(p, row);
  }
  const unsigned long long m = __ballot(f);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (uint32_t)__popcll(m));
}

int launch_tl_pred_sample(dthip_ctx* ctx, const PredArgs& p, uint32_t n, uint32_t nsamp, uint32_t* count) {
  DTHIP_LAUNCH(ctx, "tl_pred_sample_kernel", tl_pred_sample_kernel, (nsamp + 255) / 256, 256, 0, p, n, nsamp, count);
  return DTHIP_OK;
}

template <typename KT, int RB, int BLK>
static int launch_tl1_t(dthip_ctx* ctx, const TL1Args& a, uint32_t ntiles, size_t lds) {
  if (a.rec && BLK == 512) {
    if (a.keepx >= 0) {
      auto kfn = tl_level1_kernel<KT, RB, true, 512, true>;
      DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 1024));
      DTHIP_LAUNCH(ctx, "tl_level1_kernel", kfn, ntiles, 512, lds, a);
    } else {
      auto kfn = tl_level1_kernel<KT, RB, false, 512, true>;
      DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 1024));
      DTHIP_LAUNCH(ctx, "tl_level1_kernel", kfn, ntiles, 512, lds, a);
    }
    return DTHIP_OK;
  }
  if (a.rec) { set_error("tile-local level: records need 512-thread tiles"); return DTHIP_EINVAL; }
  if (a.keepx >= 0) {
    auto kfn = tl_level1_kernel<KT, RB, true, BLK>;
    DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 1024));
    DTHIP_LAUNCH(ctx, "tl_level1_kernel", kfn, ntiles, BLK, lds, a);
  } else {
    auto kfn = tl_level1_kernel<KT, RB, false, BLK>;
    DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 1024));
    DTHIP_LAUNCH(ctx, "tl_level1_kernel", kfn, ntiles, BLK, lds, a);
  }
  return DTHIP_OK;
}

uint32_t tl_tile_rows() { return (uint32_t)RP_TILE; }

// a.block = 512 | 1024 threads: tiles of a.block * 16 rows
int launch_tl_level1(dthip_ctx* ctx, const TL1Args& a) {
  if (a.n == 0) return DTHIP_OK;
  if (a.bits < 1 || a.bits > 9) { set_error("tile-local level: bad digit width %d", a.bits); return DTHIP_EINVAL; }
  if (a.block != 512 && a.block != 1024) { set_error("tile-local level: 512 or 1024 threads"); return DTHIP_EINVAL; }
  if (stype_size(a.pred.stype) != 8 || a.pred.is_mask) { set_error("tile-local level: the predicate column must be 8 bytes wide"); return DTHIP_EINVAL; }
  if (a.key.stype != DTHIP_INT32 && a.key.stype != DTHIP_INT64) { set_error("tile-local level: int32 / int64 keys"); return DTHIP_EINVAL; }
  for (int c = 0; c < a.pay.n; c++)
    if (a.pay.width[c] != 4 && a.pay.width[c] != 8) { set_error("tile-local level: riding columns are 4 or 8 bytes wide"); return DTHIP_EINVAL; }
  const int bins = 1 << a.bits;
  const uint32_t trows = (uint32_t)a.block * RP_ITEMS;
  const uint32_t ntiles = (a.n + trows - 1) / trows;
  const size_t lds = (size_t)(a.block / 64) * bins * 2 + (size_t)(bins + 16) * 4 + (size_t)trows * 8;
  const bool k64 = a.key.stype == DTHIP_INT64, b9 = a.bits > 8;
  if (a.block == 1024) {
    if (k64) return b9 ? launch_tl1_t<long long, 9, 1024>(ctx, a, ntiles, lds) : launch_tl1_t<long long, 8, 1024>(ctx, a, ntiles, lds);
    return b9 ? launch_tl1_t<int32_t, 9, 1024>(ctx, a, ntiles, lds) : launch_tl1_t<int32_t, 8, 1024>(ctx, a, ntiles, lds);
  }
  if (k64) return b9 ? launch_tl1_t<long long, 9, 512>(ctx, a, ntiles, lds) : launch_tl1_t<long long, 8, 512>(ctx, a, ntiles, lds);
  return b9 ? launch_tl1_t<int32_t, 9, 512>(ctx, a, ntiles, lds) : launch_tl1_t<int32_t, 8, 512>(ctx, a, ntiles, lds);
}

// ---- directory: transpose, rows per (digit, tile block), their prefix over the blocks, rows per digit ------------------
__global__ void __launch_bounds__(256) tl_dir_transpose_kernel(const uint16_t* __restrict__ dir, uint32_t ntiles, uint32_t F1,
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

// bs[b][tb] = rows of digit b in the 64 tiles of block tb (thread = digit: the directory rows are read coalesced)
__global__ void __launch_bounds__(256) tl_blocksum_kernel(const uint16_t* __restrict__ dir, uint32_t ntiles, uint32_t F, uint32_t ntb,
                                                          uint32_t* __restrict__ bs) {
  const uint32_t tb = blockIdx.x;
  const uint32_t t0 = tb * 64u, t1 = t0 + 64u < ntiles ? t0 + 64u : ntiles;
  for (uint32_t b = threadIdx.x; b < F; b += 256) {
    uint32_t s = 0;
    for (uint32_t t = t0; t < t1; t++) {
      const uint16_t* row = dir + (size_t)t * (F + 1);
      s += (uint32_t)row[b + 1] - (uint32_t)row[b];
    }
    bs[(size_t)b * ntb + tb] = s;
  }
}

// one workgroup per digit: bs[b][.] -> its exclusive prefix (in place), tot[b] = rows of digit b
__global__ void __launch_bounds__(256) tl_blockscan_kernel(uint32_t* __restrict__ bs, uint32_t ntb, uint32_t* __restrict__ tot) {
  __shared__ uint32_t scratch[4];
  uint32_t* row = bs + (size_t)blockIdx.x * ntb;
  const uint32_t per = (ntb + 255u) / 256u, i0 = threadIdx.x * per;
  uint32_t s = 0;
  for (uint32_t j = 0; j < per; j++) if (i0 + j < ntb) s += row[i0 + j];
  uint32_t total = 0;
  uint32_t e = block_excl_scan_u32<256>(s, scratch, &total);
  for (uint32_t j = 0; j < per; j++) if (i0 + j < ntb) { const uint32_t c = row[i0 + j]; row[i0 + j] = e; e += c; }
  if (threadIdx.x == 0) tot[blockIdx.x] = total;
}

int launch_tl_directory(dthip_ctx* ctx, const uint16_t* dir, uint32_t ntiles, uint32_t F, uint16_t* dirT, uint32_t dstride, uint32_t* cc,
                        uint32_t ntb, uint32_t* tot) {
  dim3 grid((dstride + 63) / 64, (F + 1 + 63) / 64);
  DTHIP_LAUNCH(ctx, "tl_dir_transpose_kernel", tl_dir_transpose_kernel, grid, 256, 0, dir, ntiles, F + 1, dirT, dstride);
  DTHIP_LAUNCH(ctx, "tl_blocksum_kernel", tl_blocksum_kernel, ntb, 256, 0, dir, ntiles, F, ntb, cc);
  DTHIP_LAUNCH(ctx, "tl_blockscan_kernel", tl_blockscan_kernel, F, 256, 0, cc, ntb, tot);
  return DTHIP_OK;
}

// ---- second tile-local level -> the final level's plan ------------------------------------------------------------------
// fstart[b1 * bins2 + d2] = first row of final bucket (b1, d2) in the result = first row of the parent (pstart[b1], known on
// the host from the first level's directory) + the rows of the parent's smaller digits; a bucket's rows = the sum over the
// parent's tiles of their d2 segments.  One workgroup per parent, thread = digit (directory rows read coalesced).
__global__ void __launch_bounds__(512) tl_final_starts_kernel(const uint16_t* __restrict__ dir2, const uint32_t* __restrict__ pfirst,
                                                              const uint32_t* __restrict__ pstart, uint32_t bins2, uint32_t nb1,
                                                              uint32_t* __restrict__ fstart, uint32_t* __restrict__ maxsize) {
  __shared__ uint32_t scratch[8];
  const uint32_t b1 = blockIdx.x, p0 = pfirst[b1], p1 = pfirst[b1 + 1];
  const uint32_t d = threadIdx.x;
  uint32_t s = 0;
  if (d < bins2)
    for (uint32_t p = p0; p < p1; p++) {
      const uint16_t* row = dir2 + (size_t)p * (bins2 + 1);
      s += (uint32_t)row[d + 1] - (uint32_t)row[d];
    }
  const uint32_t e = block_excl_scan_u32<512>(s, scratch, nullptr);
  if (d < bins2) fstart[(size_t)b1 * bins2 + d] = pstart[b1] + e;
  if (b1 + 1 == nb1 && d == 0) fstart[(size_t)nb1 * bins2] = pstart[nb1];
  uint32_t mx = s;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)mx, o, 64); mx = x > mx ? x : mx; }
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(maxsize, mx);
}

int launch_tl_final_plan(dthip_ctx* ctx, const uint16_t* dir2, uint32_t ntiles2, uint32_t nb1, int s2bits, const uint32_t* pfirst,
                         const uint32_t* pstart, uint16_t* dirT2, uint32_t dstride2, uint32_t* fstart, uint32_t* maxsize) {
  const uint32_t bins2 = 1u << s2bits;
  dim3 grid((dstride2 + 63) / 64, (bins2 + 1 + 63) / 64);
  DTHIP_LAUNCH(ctx, "tl_dir_transpose_kernel", tl_dir_transpose_kernel, grid, 256, 0, dir2, ntiles2, bins2 + 1, dirT2, dstride2);
  DTHIP_LAUNCH(ctx, "tl_final_starts_kernel", tl_final_starts_kernel, nb1, 512, 0, dir2, pfirst, pstart, bins2, nb1, fstart, maxsize);
  return DTHIP_OK;
}

// ---- digit counts of the second level's tiles, read through the directory -------------------------------------------
// The counterpart of radix_tile_hist_kernel with ragged tiles (tdesc / gdesc of msd_plan.hpp) for rows that still sit in
// the tile-local layout: P[t][d] = rows of digit d in the earlier tiles of t's group, gtot[g][d] = the group's rows.
__global__ void __launch_bounds__(RP_BLOCK) tl_gather_hist_kernel(TLGatherHistArgs a) {
  constexpr int BLOCK = RP_BLOCK;
  __shared__ uint32_t cnt[HIST_STRIDE];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* src = reinterpret_cast<uint32_t*>(smem);
  const int tid = threadIdx.x;
  const uint32_t bins = 1u << a.bits, dmask = bins - 1u;
  constexpr int HK = (HIST_STRIDE + BLOCK - 1) / BLOCK;
  uint32_t run[HK];
#pragma unroll
  for (int k = 0; k < HK; k++) { run[k] = 0; if (tid + k * BLOCK < HIST_STRIDE) cnt[tid + k * BLOCK] = 0; }
  __syncthreads();
  const uint32_t t0 = a.gdesc[2 * blockIdx.x], t1 = t0 + a.gdesc[2 * blockIdx.x + 1];
  for (uint32_t t = t0; t < t1; t++) {
    const uint32_t first = a.tdesc[4 * t], rows = a.tdesc[4 * t + 1], bkt = a.tdesc[4 * t + 3];
    tl_build_src<BLOCK>(src, a.dirT, a.dstride, a.cc, a.ntb, a.ntiles1, a.T1, bkt, first - a.pstart[bkt], rows);
    __syncthreads();
    for (uint32_t v0 = (uint32_t)tid; v0 < rows; v0 += 4u * BLOCK) {
      uint32_t k[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { const uint32_t v = v0 + (uint32_t)j * BLOCK; k[j] = v < rows ? a.keys[src[v]] : 0u; }
#pragma unroll
      for (int j = 0; j < 4; j++) if (v0 + (uint32_t)j * BLOCK < rows) (void)lds_count_rank(cnt, (k[j] >> a.shift) & dmask);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HK; k++) {
      const uint32_t b = (uint32_t)tid + (uint32_t)k * BLOCK;
      if (b < bins) {
        const uint32_t c = cnt[b];
        cnt[b] = 0;
        a.P[(size_t)t * bins + b] = run[k];
        run[k] += c;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < HK; k++) {
    const uint32_t b = (uint32_t)tid + (uint32_t)k * BLOCK;
    if (b < bins) a.gtot[(size_t)blockIdx.x * bins + b] = run[k];
  }
}

int launch_tl_gather_hist(dthip_ctx* ctx, const TLGatherHistArgs& a, uint32_t G) {
  if (G == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "tl_gather_hist_kernel", tl_gather_hist_kernel, G, RP_BLOCK, (size_t)RP_TILE * 4, a);
  return DTHIP_OK;
}

}  // namespace dthip
