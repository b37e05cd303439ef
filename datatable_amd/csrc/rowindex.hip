// rowindex.hip -- RowIndex construction and application.
//
//   compact:  boolean mask (or  col <cmp> scalar  predicate) -> ascending ARR32
//             of passing rows.  Reference: ArrayRowIndexImpl::init_from_boolean_column
//             (src/core/rowindex_array.cc:130-170): parallel count, then a SERIAL
//             compaction loop; here count + scan + ordered scatter, all parallel.
//   gather:   out[i] = col[ri[i]] (NA for negative indices) -- the loop behind
//             ArrayView_ColumnImpl + _materialize_fw (src/core/column/view.cc:140-145,
//             column_impl.cc:78-101).
#include "common.hpp"
#include "device_utils.hpp"
#include "pred.hpp"

namespace dthip {

constexpr int CP_BLOCK = 256;
constexpr int CP_ITEMS = 8;
constexpr int CP_TILE = CP_BLOCK * CP_ITEMS;

__device__ __forceinline__ uint32_t sweep_pred(const PredArgs& p, uint32_t n, uint32_t wave_base, uint32_t* cnt) {
  const int lane = lane_id();
  uint32_t bits = 0, c = 0;
#pragma unroll
  for (int k = 0; k < CP_ITEMS; k++) {
    const uint32_t idx = wave_base + 64u * k + lane;
    const bool f = idx < n && pred_at(p, idx);
    bits |= (uint32_t)f << k;
    c += (uint32_t)__popcll(__ballot(f));
  }
  *cnt = c;
  return bits;
}

typedef uint32_t cu32x4 __attribute__((ext_vector_type(4)));

// passing rows of one full tile of an 8-byte column, counted with 16-byte loads (the order of the rows does not matter
// for a count; the generic sweep reads 8 bytes per lane and instruction: 3.9 instead of 5+ TB/s)
__device__ __forceinline__ bool pred_fast8(const PredArgs& p, uint32_t n, uint32_t tile_base) {
  return !p.is_mask && (p.stype == DTHIP_FLOAT64 || p.stype == DTHIP_INT64) && n - tile_base >= (uint32_t)CP_TILE &&
         (reinterpret_cast<uintptr_t>(p.data) & 15) == 0;
}

template <typename T>
__device__ __forceinline__ uint32_t count_tile16(const PredArgs& p, uint32_t tile_base) {
  const cu32x4* src = reinterpret_cast<const cu32x4*>(static_cast<const T*>(p.data) + tile_base);
  cu32x4 w[CP_ITEMS / 2];
#pragma unroll
  for (int q = 0; q < CP_ITEMS / 2; q++) w[q] = src[q * CP_BLOCK + threadIdx.x];
  const T* v = reinterpret_cast<const T*>(w);
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < CP_ITEMS; j++) c += pred_val8(p, v[j]) ? 1u : 0u;
  return c;
}

__global__ void __launch_bounds__(CP_BLOCK) compact_count_kernel(PredArgs p, uint32_t n, uint32_t* tile_counts) {
  __shared__ uint32_t wc[CP_BLOCK / 64];
  uint32_t cnt;
  const uint32_t tile_base = blockIdx.x * CP_TILE;
  if (pred_fast8(p, n, tile_base)) {
    uint32_t c = count_tile16<unsigned long long>(p, tile_base);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
    cnt = c;
  } else
  sweep_pred(p, n, blockIdx.x * CP_TILE + wave_id() * (64 * CP_ITEMS), &cnt);
  if (lane_id() == 0) wc[wave_id()] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
    for (int w = 0; w < CP_BLOCK / 64; w++) s += wc[w];
    tile_counts[blockIdx.x] = s;
  }
}

__global__ void __launch_bounds__(CP_BLOCK) compact_write_kernel(PredArgs p, uint32_t n, const uint32_t* tile_base,
                                                                 int32_t* out) {
  __shared__ uint32_t wc[CP_BLOCK / 64];
  const int lane = lane_id(), wave = wave_id();
  const uint32_t wave_base = blockIdx.x * CP_TILE + wave * (64 * CP_ITEMS);
  uint32_t cnt;
  const uint32_t bits = sweep_pred(p, n, wave_base, &cnt);
  if (lane == 0) wc[wave] = cnt;
  __syncthreads();
  uint32_t running = tile_base[blockIdx.x];
  for (int w = 0; w < wave; w++) running += wc[w];
#pragma unroll
  for (int k = 0; k < CP_ITEMS; k++) {
    const bool f = (bits >> k) & 1u;
    const unsigned long long bal = __ballot(f);
    if (f) out[running + mbcnt64(bal)] = (int32_t)(wave_base + 64u * k + lane);
    running += (uint32_t)__popcll(bal);
  }
}

static int launch_compact1(dthip_ctx* ctx, const PredArgs& p, int64_t n, int32_t* out_ri, const TakeCols& tc, int64_t* nout_host);

int launch_compact(dthip_ctx* ctx, const PredArgs& p, int64_t n, int32_t* out, int64_t* nout_host) {
  *nout_host = 0;
  if (n == 0) return DTHIP_OK;
  if (ctx->filter_path != 1) { TakeCols none; none.n = 0; return launch_compact1(ctx, p, n, out, none, nout_host); }
  const uint32_t nt = (uint32_t)((n + CP_TILE - 1) / CP_TILE);
  Scratch sc(ctx);
  uint32_t* tile_counts = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)nt + 4 + nt / 8192, &tile_counts));
  DTHIP_LAUNCH(ctx, "compact_count_kernel", compact_count_kernel, nt, CP_BLOCK, 0, p, (uint32_t)n, tile_counts);
  DTHIP_TRY(launch_scan_tiles(ctx, tile_counts, nt, tile_counts + nt));
  DTHIP_LAUNCH(ctx, "compact_write_kernel", compact_write_kernel, nt, CP_BLOCK, 0, p, (uint32_t)n, tile_counts, out);
  uint32_t total = 0;
  DTHIP_TRY(read_back(ctx, &total, tile_counts + nt, sizeof(total)));
  *nout_host = total;
  return DTHIP_OK;
}

// compact + take: the passing rows' index AND their values of up to 8 columns in one sweep --
// DT[f.x > 0, cols] materialised (init_from_boolean_column + _materialize_fw of every column through
// the new RowIndex, rowindex_array.cc:130-170, column_impl.cc:78-101).  The columns are read
// sequentially next to the predicate column instead of being gathered through the finished RowIndex.
// PV8: the predicate column is an 8-byte column (float64 / int64, not a mask): its values stay in registers after the
// predicate is evaluated, and a taken column that IS the predicate column (DT[f.x > 0, :] takes x) is written from them --
// round 3 loaded it a second time, and the PMC passes showed those 8 GB per 1e9 rows coming from HBM again, not from L2
// (compact_take_kernel: 24 GB fetched for 16 GB of input, profiles/r04_c5_pmc.txt)
template <bool PV8>
__global__ void __launch_bounds__(CP_BLOCK) compact_take_kernel(PredArgs p, uint32_t n, const uint32_t* tile_base,
                                                                int32_t* out_ri, TakeCols tc) {
  __shared__ uint32_t wc[CP_BLOCK / 64];
  const int lane = lane_id(), wave = wave_id();
  const uint32_t wave_base = blockIdx.x * CP_TILE + wave * (64 * CP_ITEMS);
  uint32_t cnt = 0, bits = 0;
  unsigned long long pv[PV8 ? CP_ITEMS : 1];
  if (PV8) {
    const unsigned long long* px = static_cast<const unsigned long long*>(p.data);
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++) {
      const uint32_t idx = wave_base + 64u * k + lane;
      pv[k] = idx < n ? px[idx] : 0ULL;
    }
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++) {
      const uint32_t idx = wave_base + 64u * k + lane;
      const bool f = idx < n && pred_val8(p, pv[k]);
      bits |= (uint32_t)f << k;
      cnt += (uint32_t)__popcll(__ballot(f));
    }
  } else {
    bits = sweep_pred(p, n, wave_base, &cnt);
  }
  if (lane == 0) wc[wave] = cnt;
  __syncthreads();
  uint32_t running = tile_base[blockIdx.x];
  for (int w = 0; w < wave; w++) running += wc[w];
#pragma unroll
  for (int k = 0; k < CP_ITEMS; k++) {
    const bool f = (bits >> k) & 1u;
    const unsigned long long bal = __ballot(f);
    if (f) {
      const uint32_t src = wave_base + 64u * k + lane, dst = running + mbcnt64(bal);
      if (out_ri) out_ri[dst] = (int32_t)src;
      for (int c = 0; c < tc.n; c++) {
        if (PV8 && tc.in[c] == p.data) { static_cast<unsigned long long*>(tc.out[c])[dst] = pv[k]; continue; }
        switch (tc.width[c]) {
          case 8: static_cast<unsigned long long*>(tc.out[c])[dst] = static_cast<const unsigned long long*>(tc.in[c])[src]; break;
          case 4: static_cast<uint32_t*>(tc.out[c])[dst] = static_cast<const uint32_t*>(tc.in[c])[src]; break;
          case 2: static_cast<uint16_t*>(tc.out[c])[dst] = static_cast<const uint16_t*>(tc.in[c])[src]; break;
          default: static_cast<uint8_t*>(tc.out[c])[dst] = static_cast<const uint8_t*>(tc.in[c])[src]; break;
        }
      }
    }
    running += (uint32_t)__popcll(bal);
  }
}

int launch_compact_take(dthip_ctx* ctx, const PredArgs& p, int64_t n, int32_t* out_ri, const TakeCols& tc, int64_t* nout_host) {
  *nout_host = 0;
  if (n == 0) return DTHIP_OK;
  if (ctx->filter_path != 1) return launch_compact1(ctx, p, n, out_ri, tc, nout_host);
  const uint32_t nt = (uint32_t)((n + CP_TILE - 1) / CP_TILE);
  Scratch sc(ctx);
  uint32_t* tile_counts = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)nt + 4 + nt / 8192, &tile_counts));
  DTHIP_LAUNCH(ctx, "compact_count_kernel", compact_count_kernel, nt, CP_BLOCK, 0, p, (uint32_t)n, tile_counts);
  DTHIP_TRY(launch_scan_tiles(ctx, tile_counts, nt, tile_counts + nt));
  const bool pv8 = !p.is_mask && (p.stype == DTHIP_FLOAT64 || p.stype == DTHIP_INT64);
  if (pv8) { DTHIP_LAUNCH(ctx, "compact_take_kernel", compact_take_kernel<true>, nt, CP_BLOCK, 0, p, (uint32_t)n, tile_counts, out_ri, tc); }
  else { DTHIP_LAUNCH(ctx, "compact_take_kernel", compact_take_kernel<false>, nt, CP_BLOCK, 0, p, (uint32_t)n, tile_counts, out_ri, tc); }
  uint32_t total = 0;
  DTHIP_TRY(read_back(ctx, &total, tile_counts + nt, sizeof(total)));
  *nout_host = total;
  return DTHIP_OK;
}

// ---- single pass (round 4): the tile offsets come from a decoupled look-back instead of a count pass -----------------
// MEASURED SLOWER than the two-pass form on MI355X (C5's filter: 8.6-9.2 ms against 1.3 + 5.8) and therefore not the default
// (option filter_path = 0 / DTHIP_FILTER_PATH=0 selects it).
// The two-pass form reads the predicate column twice (count, then write: 1.3 + 5.7 ms per 1e9 float64 rows).  Here every
// workgroup takes a TICKET (its tile number: tiles are numbered in the order workgroups start, so every lower tile
// belongs to a workgroup that is already running and will publish without waiting for anyone behind it -- no assumption
// about dispatch order or co-residency), counts its tile's passing rows, publishes {flag, count} as ONE 64-bit word and
// looks back over the words of the tiles before it: an "aggregate" word adds its count and the walk goes on, an
// "inclusive" word ends it.  Wave 0 inspects 64 predecessors per step.  state[] must be zero when the kernel starts.
constexpr int C1_BLOCK = 256, C1_ITEMS = 16, C1_TILE = C1_BLOCK * C1_ITEMS;
constexpr unsigned long long C1_AGG = 1ULL << 62, C1_INC = 2ULL << 62, C1_VAL = (1ULL << 62) - 1ULL;

__device__ __forceinline__ unsigned long long c1_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void c1_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(C1_BLOCK) compact_take1_kernel(PredArgs p, uint32_t n, unsigned long long* __restrict__ state,
                                                                 uint32_t* __restrict__ ticket, int32_t* __restrict__ out_ri,
                                                                 TakeCols tc, uint32_t* __restrict__ total) {
  __shared__ uint32_t wc[C1_BLOCK / 64];
  __shared__ uint32_t s_tile, s_excl;
  const int lane = lane_id(), wave = wave_id();
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t wave_base = tile * C1_TILE + wave * (64 * C1_ITEMS);
  // predicate bits of the lane's 16 rows (wave-striped: row = wave_base + 64 k + lane)
  uint32_t bits = 0, cnt = 0;
#pragma unroll
  for (int k = 0; k < C1_ITEMS; k++) {
    const uint32_t idx = wave_base + 64u * k + lane;
    const bool f = idx < n && pred_at(p, idx);
    bits |= (uint32_t)f << k;
    cnt += (uint32_t)__popcll(__ballot(f));
  }
  if (lane == 0) wc[wave] = cnt;
  __syncthreads();
  uint32_t agg = 0;
#pragma unroll
  for (int w = 0; w < C1_BLOCK / 64; w++) agg += wc[w];
  if (wave == 0) {
    uint32_t excl = 0;
    if (tile == 0) {
      if (lane == 0) c1_store(&state[0], C1_INC | agg);
    } else {
      if (lane == 0) c1_store(&state[tile], C1_AGG | agg);
      int32_t look = (int32_t)tile - 1;                  // the nearest predecessor not yet accounted for
      while (true) {
        const int32_t idx = look - lane;
        unsigned long long w = C1_INC;                   // tiles before tile 0: an empty inclusive prefix
        if (idx >= 0) w = c1_load(&state[idx]);
        const unsigned long long inc = __ballot((w >> 62) == 2ULL);
        const unsigned long long inv = __ballot((w >> 62) == 0ULL);
        // lanes up to (and including) the first inclusive word must all be published
        const int first_inc = inc ? __ffsll((long long)inc) - 1 : 64;
        const unsigned long long need = first_inc >= 63 ? ~0ULL : ((2ULL << first_inc) - 1ULL);
        if (inv & need) { __builtin_amdgcn_s_sleep(1); continue; }       // somebody in the window is still counting
        uint32_t v = (lane <= first_inc) ? (uint32_t)(w & C1_VAL) : 0u;
        excl += wave_reduce_sum_u32(v);
        if (first_inc < 64) break;
        look -= 64;
      }
      if (lane == 0) c1_store(&state[tile], C1_INC | (unsigned long long)(excl + agg));
    }
    if (lane == 0) {
      s_excl = excl;
      if ((uint64_t)(tile + 1) * C1_TILE >= n) *total = excl + agg;      // the last tile knows the number of passing rows
    }
  }
  __syncthreads();
  uint32_t running = s_excl;
  for (int w = 0; w < wave; w++) running += wc[w];
#pragma unroll
  for (int k = 0; k < C1_ITEMS; k++) {
    const bool f = (bits >> k) & 1u;
    const unsigned long long bal = __ballot(f);
    if (f) {
      const uint32_t src = wave_base + 64u * k + lane, dst = running + mbcnt64(bal);
      if (out_ri) out_ri[dst] = (int32_t)src;
      for (int c = 0; c < tc.n; c++) {
        switch (tc.width[c]) {
          case 8: static_cast<unsigned long long*>(tc.out[c])[dst] = static_cast<const unsigned long long*>(tc.in[c])[src]; break;
          case 4: static_cast<uint32_t*>(tc.out[c])[dst] = static_cast<const uint32_t*>(tc.in[c])[src]; break;
          case 2: static_cast<uint16_t*>(tc.out[c])[dst] = static_cast<const uint16_t*>(tc.in[c])[src]; break;
          default: static_cast<uint8_t*>(tc.out[c])[dst] = static_cast<const uint8_t*>(tc.in[c])[src]; break;
        }
      }
    }
    running += (uint32_t)__popcll(bal);
  }
}

static int launch_compact1(dthip_ctx* ctx, const PredArgs& p, int64_t n, int32_t* out_ri, const TakeCols& tc, int64_t* nout_host) {
  const uint32_t nt = (uint32_t)((n + C1_TILE - 1) / C1_TILE);
  Scratch sc(ctx);
  unsigned long long* state = nullptr;
  DTHIP_TRY(sc.get<unsigned long long>((size_t)nt + 2, &state));
  uint32_t* ticket = reinterpret_cast<uint32_t*>(state + nt);       // {ticket, total} in the words after the tile states
  DTHIP_CHECK_HIP(hipMemsetAsync(state, 0, sizeof(unsigned long long) * ((size_t)nt + 2), ctx->stream));
  DTHIP_LAUNCH(ctx, "compact_take1_kernel", compact_take1_kernel, nt, C1_BLOCK, 0, p, (uint32_t)n, state, ticket, out_ri, tc, ticket + 1);
  uint32_t total = 0;
  DTHIP_TRY(read_back(ctx, &total, ticket + 1, sizeof(total)));
  *nout_host = total;
  return DTHIP_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) gather_kernel(const T* __restrict__ data, const int32_t* __restrict__ ri,
                                                     uint32_t n, T* __restrict__ out, T na) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int32_t r = ri[i];
    out[i] = r >= 0 ? data[r] : na;
  }
}

// destination of every row in a key-range partition (multi-GPU exchange of rows): out[i] = number of
// boundaries <= key[i] (NA keys -> 0: the NA group lives on rank 0, like the smallest keys)
struct BoundsArg { long long b[15]; int n; };
template <typename T>
__global__ void __launch_bounds__(256) range_bucket_kernel(const T* __restrict__ keys, uint32_t n, T na, BoundsArg bd,
                                                           int8_t* __restrict__ out) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const T k = keys[i];
    int d = 0;
    if (k != na) {
#pragma unroll
      for (int j = 0; j < 15; j++) d += (j < bd.n && (long long)k >= bd.b[j]) ? 1 : 0;
    }
    out[i] = (int8_t)d;
  }
}

int launch_range_bucket(dthip_ctx* ctx, const void* keys, int stype, int64_t n, const long long* bounds, int nbounds,
                        int8_t* out) {
  if (n == 0) return DTHIP_OK;
  if (nbounds < 0 || nbounds > 15) { set_error("range_bucket: at most 15 boundaries (16 destinations)"); return DTHIP_EINVAL; }
  BoundsArg bd;
  memset(&bd, 0, sizeof(bd));
  bd.n = nbounds;
  for (int j = 0; j < nbounds; j++) bd.b[j] = bounds[j];
  long long blocks = (n + 2047) / 2048;
  if (blocks > ctx->num_cus * 8) blocks = ctx->num_cus * 8;
  const unsigned g = (unsigned)blocks;
  switch (stype) {
    case DTHIP_INT8:
      DTHIP_LAUNCH(ctx, "range_bucket_kernel", range_bucket_kernel<int8_t>, g, 256, 0, static_cast<const int8_t*>(keys), (uint32_t)n, (int8_t)INT8_MIN, bd, out); break;
    case DTHIP_INT16:
      DTHIP_LAUNCH(ctx, "range_bucket_kernel", range_bucket_kernel<int16_t>, g, 256, 0, static_cast<const int16_t*>(keys), (uint32_t)n, (int16_t)INT16_MIN, bd, out); break;
    case DTHIP_INT32:
      DTHIP_LAUNCH(ctx, "range_bucket_kernel", range_bucket_kernel<int32_t>, g, 256, 0, static_cast<const int32_t*>(keys), (uint32_t)n, (int32_t)INT32_MIN, bd, out); break;
    case DTHIP_INT64:
      DTHIP_LAUNCH(ctx, "range_bucket_kernel", range_bucket_kernel<long long>, g, 256, 0, static_cast<const long long*>(keys), (uint32_t)n, (long long)INT64_MIN, bd, out); break;
    default: set_error("range_bucket: integer key columns only (stype %d)", stype); return DTHIP_ENOTIMPL;
  }
  return DTHIP_OK;
}

// first(A) / last(A): the element at the first / last grouped position of every group, NA included
// (FirstLast_ColumnImpl::_get, src/core/expr/head_reduce_unary.cc:116-160)
template <typename T>
__global__ void __launch_bounds__(256) firstlast_kernel(const T* __restrict__ data, const int32_t* __restrict__ ri,
                                                        const int32_t* __restrict__ offsets, uint32_t ngroups, int last,
                                                        T* __restrict__ out, T na) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const int32_t p = last ? offsets[g + 1] - 1 : offsets[g];
  const int32_t r = ri ? ri[p] : p;
  out[g] = r >= 0 ? data[r] : na;
}

int launch_firstlast(dthip_ctx* ctx, const void* data, int stype, const int32_t* ri, const int32_t* offsets,
                     int64_t ngroups, int last, void* out) {
  if (ngroups == 0) return DTHIP_OK;
  const unsigned g = (unsigned)((ngroups + 255) / 256);
  const uint32_t ng = (uint32_t)ngroups;
  switch (stype_size(stype)) {
    case 1:
      DTHIP_LAUNCH(ctx, "firstlast_kernel", firstlast_kernel<uint8_t>, g, 256, 0, static_cast<const uint8_t*>(data), ri, offsets,
                   ng, last, static_cast<uint8_t*>(out), (uint8_t)0x80);
      break;
    case 2:
      DTHIP_LAUNCH(ctx, "firstlast_kernel", firstlast_kernel<uint16_t>, g, 256, 0, static_cast<const uint16_t*>(data), ri, offsets,
                   ng, last, static_cast<uint16_t*>(out), (uint16_t)0x8000);
      break;
    case 4:
      DTHIP_LAUNCH(ctx, "firstlast_kernel", firstlast_kernel<uint32_t>, g, 256, 0, static_cast<const uint32_t*>(data), ri, offsets,
                   ng, last, static_cast<uint32_t*>(out), stype == DTHIP_FLOAT32 ? 0x7FC00000u : 0x80000000u);
      break;
    case 8:
      DTHIP_LAUNCH(ctx, "firstlast_kernel", firstlast_kernel<unsigned long long>, g, 256, 0,
                   static_cast<const unsigned long long*>(data), ri, offsets, ng, last, static_cast<unsigned long long*>(out),
                   stype == DTHIP_FLOAT64 ? 0x7FF8000000000000ULL : 0x8000000000000000ULL);
      break;
    default: set_error("first/last: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
  return DTHIP_OK;
}

int launch_gather(dthip_ctx* ctx, const void* data, int stype, const int32_t* ri, int64_t nout, void* out) {
  if (nout == 0) return DTHIP_OK;
  long long blocks = (nout + 1023) / 1024;
  if (blocks > ctx->num_cus * 16) blocks = ctx->num_cus * 16;
  const unsigned g = (unsigned)blocks;
  const uint32_t n = (uint32_t)nout;
  switch (stype_size(stype)) {
    case 1:
      DTHIP_LAUNCH(ctx, "gather_kernel", gather_kernel<uint8_t>, g, 256, 0, static_cast<const uint8_t*>(data), ri, n,
                   static_cast<uint8_t*>(out), (uint8_t)0x80);
      break;
    case 2:
      DTHIP_LAUNCH(ctx, "gather_kernel", gather_kernel<uint16_t>, g, 256, 0, static_cast<const uint16_t*>(data), ri, n,
                   static_cast<uint16_t*>(out), (uint16_t)0x8000);
      break;
    case 4:
      DTHIP_LAUNCH(ctx, "gather_kernel", gather_kernel<uint32_t>, g, 256, 0, static_cast<const uint32_t*>(data), ri, n,
                   static_cast<uint32_t*>(out), stype == DTHIP_FLOAT32 ? 0x7FC00000u : 0x80000000u);
      break;
    case 8:
      DTHIP_LAUNCH(ctx, "gather_kernel", gather_kernel<unsigned long long>, g, 256, 0,
                   static_cast<const unsigned long long*>(data), ri, n, static_cast<unsigned long long*>(out),
                   stype == DTHIP_FLOAT64 ? 0x7FF8000000000000ULL : 0x8000000000000000ULL);
      break;
    default: set_error("gather: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
  return DTHIP_OK;
}

}  // namespace dthip
