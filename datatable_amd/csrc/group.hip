// group.hip -- group boundaries of a sorted key sequence -> Groupby offsets.
//
// Reproduces what GroupGatherer builds on the CPU (src/core/sort_groups.cc:30-117,
// src/core/sort.h:118-147): offsets[0]=0, offsets[g] = first sorted position of
// group g, offsets[ngroups] = nrows (src/core/groupby.h:54-91).  The reference
// compacts per-thread group lists serially; here a run head is any position
// whose transformed key differs from its predecessor's, found with 64-wide
// ballots, ranked with a workgroup scan, and scanned across tiles.
//
// The same sweep also emits the head BITMAP (1 bit per sorted position) and
// per-tile head counts that the segmented reducers (reduce.hip) consume.
#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

constexpr int GB_BLOCK = 256;
constexpr int GB_ITEMS = 8;                         // consecutive positions per thread = one bitmap byte
constexpr int GB_TILE = GB_BLOCK * GB_ITEMS;        // 2048 positions per workgroup (== reduce tile)
static_assert(GB_TILE == SEG_TILE, "reduce.hip assumes 2048-position tiles");

typedef uint32_t gu32x4 __attribute__((ext_vector_type(4)));

// Head flags of 8 consecutive sorted positions -> one byte of the head bitmap, plus the
// number of heads of every 2048-position tile.  Keys are read with 16-byte loads.
//   KeyT = uint32_t / unsigned long long: head(i) = i==0 || K[i] != K[i-1]
//   KeyT = uint8_t: K is a byte-per-position flag array (multi-stage sorts)
//   unsorted (nullable; KeyT = int32_t / long long, the RAW key column of a frame whose rows may already be in key order):
//   *unsorted <- 1 when some K[i] < K[i-1] -- NA is the smallest value of a signed integer stype, so "ascending as signed
//   integers" IS the grouped order of an ascending key with NA first, and the heads found here are the group heads
template <typename KeyT>
__global__ void __launch_bounds__(GB_BLOCK) count_heads_kernel(const KeyT* __restrict__ K, uint32_t n,
                                                               uint32_t* __restrict__ tile_counts,
                                                               uint8_t* __restrict__ bitmap, uint32_t* __restrict__ unsorted = nullptr) {
  __shared__ uint32_t wc[GB_BLOCK / 64];
  bool down = false;
  const uint32_t p0 = blockIdx.x * GB_TILE + threadIdx.x * GB_ITEMS;
  uint32_t hb = 0;
  if (p0 < n) {
    KeyT k[GB_ITEMS];
    if (p0 + GB_ITEMS <= n) {
      constexpr int NV = GB_ITEMS * (int)sizeof(KeyT) / 16;      // 0 for bytes, 2 for u32, 4 for u64
      if (NV > 0) {
        const gu32x4* src = reinterpret_cast<const gu32x4*>(K + p0);
        gu32x4 v[NV > 0 ? NV : 1];
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = src[j];
        const KeyT* vk = reinterpret_cast<const KeyT*>(v);
#pragma unroll
        for (int j = 0; j < GB_ITEMS; j++) k[j] = vk[j];
      } else {
        const unsigned long long w = *reinterpret_cast<const unsigned long long*>(K + p0);
#pragma unroll
        for (int j = 0; j < GB_ITEMS; j++) k[j] = (KeyT)((w >> (8 * j)) & 0xFF);
      }
    } else {
#pragma unroll
      for (int j = 0; j < GB_ITEMS; j++) k[j] = (p0 + j < n) ? K[p0 + j] : KeyT(0);
    }
    if (sizeof(KeyT) == 1) {
#pragma unroll
      for (int j = 0; j < GB_ITEMS; j++) hb |= (uint32_t)((p0 + j < n) && k[j] != 0) << j;
    } else {
      const KeyT kp = p0 ? K[p0 - 1] : k[0];
      const bool h0 = (p0 == 0) || (kp != k[0]);
      hb = h0 ? 1u : 0u;
      down = k[0] < kp;
#pragma unroll
      for (int j = 1; j < GB_ITEMS; j++) {
        hb |= (uint32_t)((p0 + j < n) && k[j] != k[j - 1]) << j;
        down |= (p0 + j < n) && k[j] < k[j - 1];
      }
    }
    bitmap[p0 >> 3] = (uint8_t)hb;
  }
  if (unsorted && __ballot(down) && lane_id() == 0) atomicOr(unsorted, 1u);
  const uint32_t c = wave_reduce_sum_u32((uint32_t)__popc(hb));
  if (lane_id() == 0) wc[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
    for (int w = 0; w < GB_BLOCK / 64; w++) s += wc[w];
    tile_counts[blockIdx.x] = s;
  }
}

// In-place exclusive scan of counts[0..m) in three small launches: every workgroup scans
// its own chunk of 8192 and records the chunk total; one workgroup scans the chunk totals;
// every workgroup adds its chunk base.
constexpr int SC_CHUNK = 1024 * 8;

__global__ void __launch_bounds__(1024) scan_chunk_kernel(uint32_t* counts, uint32_t m, uint32_t* chunk_tot) {
  __shared__ uint32_t scratch[16];
  const uint32_t i0 = blockIdx.x * SC_CHUNK + threadIdx.x * 8;
  uint32_t v[8], s = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { v[j] = (i0 + j < m) ? counts[i0 + j] : 0; s += v[j]; }
  uint32_t tot;
  uint32_t e = block_excl_scan_u32<1024>(s, scratch, &tot);
#pragma unroll
  for (int j = 0; j < 8; j++) { if (i0 + j < m) counts[i0 + j] = e; e += v[j]; }
  if (threadIdx.x == 0) chunk_tot[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(1024) scan_chunk_totals_kernel(uint32_t* chunk_tot, uint32_t nchunks, uint32_t* total) {
  __shared__ uint32_t scratch[16];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nchunks; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nchunks ? chunk_tot[i] : 0;
    uint32_t tot;
    const uint32_t e = block_excl_scan_u32<1024>(v, scratch, &tot);
    if (i < nchunks) chunk_tot[i] = carry + e;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(1024) scan_add_base_kernel(uint32_t* counts, uint32_t m, const uint32_t* chunk_base) {
  const uint32_t b = chunk_base[blockIdx.x];
  if (b == 0) return;
  const uint32_t i0 = blockIdx.x * SC_CHUNK + threadIdx.x * 8;
#pragma unroll
  for (int j = 0; j < 8; j++) if (i0 + j < m) counts[i0 + j] += b;
}

// counts must have room for m + 1 + ceil(m / 8192) words: [0,m) data, [m] total, then scratch
int launch_scan_tiles(dthip_ctx* ctx, uint32_t* counts, uint32_t m, uint32_t* total) {
  const uint32_t nchunks = (m + SC_CHUNK - 1) / SC_CHUNK;
  uint32_t* chunk_tot = counts + m + 1;
  if (total != counts + m) { set_error("scan_tiles: total must be counts + m"); return DTHIP_EINVAL; }
  if (m == 0) { DTHIP_CHECK_HIP(hipMemsetAsync(total, 0, sizeof(uint32_t), ctx->stream)); return DTHIP_OK; }
  DTHIP_LAUNCH(ctx, "scan_tiles_kernel", scan_chunk_kernel, nchunks, 1024, 0, counts, m, chunk_tot);
  DTHIP_LAUNCH(ctx, "scan_tiles_kernel", scan_chunk_totals_kernel, 1, 1024, 0, chunk_tot, nchunks, total);
  if (nchunks > 1) DTHIP_LAUNCH(ctx, "scan_tiles_kernel", scan_add_base_kernel, nchunks, 1024, 0, counts, m, chunk_tot);
  return DTHIP_OK;
}

// offsets[g] = sorted position of the g-th head, read back from the head bitmap
__global__ void __launch_bounds__(GB_BLOCK) write_offsets_kernel(const uint8_t* __restrict__ bitmap, uint32_t n,
                                                                 const uint32_t* __restrict__ tile_base,
                                                                 uint32_t ngroups, int32_t* __restrict__ offsets) {
  __shared__ uint32_t scratch[GB_BLOCK / 64];
  const uint32_t p0 = blockIdx.x * GB_TILE + threadIdx.x * GB_ITEMS;
  uint32_t hb = (p0 < n) ? bitmap[p0 >> 3] : 0u;
  uint32_t r = tile_base[blockIdx.x] + block_excl_scan_u32<GB_BLOCK>((uint32_t)__popc(hb), scratch, nullptr);
  while (hb) {
    const int j = __ffs((int)hb) - 1;
    offsets[r++] = (int32_t)(p0 + j);
    hb &= hb - 1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[ngroups] = (int32_t)n;
}

static uint32_t ntiles_of(int64_t n) { return (uint32_t)((n + GB_TILE - 1) / GB_TILE); }

// tile_counts[ntiles] <- exclusive scan of heads per tile (= index of the first head of each
// tile); bitmap <- 1 bit per position (ceil(n/8) bytes, padded to 8); d_total <- ngroups
int launch_count_heads(dthip_ctx* ctx, const void* keys, int key64, const uint8_t* heads, int64_t n,
                       uint32_t* tile_counts, unsigned long long* bitmap, uint32_t* d_total,
                       int64_t* ngroups_host) {
  const uint32_t nt = ntiles_of(n);
  uint8_t* bm = reinterpret_cast<uint8_t*>(bitmap);
  if (heads) {
    DTHIP_LAUNCH(ctx, "count_heads_kernel", count_heads_kernel<uint8_t>, nt, GB_BLOCK, 0,
                 heads, (uint32_t)n, tile_counts, bm);
  } else if (key64) {
    DTHIP_LAUNCH(ctx, "count_heads_kernel", count_heads_kernel<unsigned long long>, nt, GB_BLOCK, 0,
                 static_cast<const unsigned long long*>(keys), (uint32_t)n, tile_counts, bm);
  } else {
    DTHIP_LAUNCH(ctx, "count_heads_kernel", count_heads_kernel<uint32_t>, nt, GB_BLOCK, 0,
                 static_cast<const uint32_t*>(keys), (uint32_t)n, tile_counts, bm);
  }
  DTHIP_TRY(launch_scan_tiles(ctx, tile_counts, nt, d_total));
  if (ngroups_host) {
    uint32_t t = 0;
    DTHIP_TRY(read_back(ctx, &t, d_total, sizeof(t)));
    *ngroups_host = t;
  }
  return DTHIP_OK;
}

// the same over the RAW int32 / int64 key column of rows that may already be in key order: *d_flags (zeroed here) <- 1 if some
// key is smaller than its predecessor; read back, and with it the number of groups when the column is in order
int launch_count_heads_presorted(dthip_ctx* ctx, const void* keys, int stype, int64_t n, uint32_t* tile_counts,
                                 unsigned long long* bitmap, uint32_t* d_flags, int64_t* ngroups_host, bool* sorted_host) {
  const uint32_t nt = ntiles_of(n);
  uint8_t* bm = reinterpret_cast<uint8_t*>(bitmap);
  DTHIP_CHECK_HIP(hipMemsetAsync(d_flags, 0, sizeof(uint32_t), ctx->stream));
  if (stype == DTHIP_INT64)
    DTHIP_LAUNCH(ctx, "count_heads_kernel", count_heads_kernel<long long>, nt, GB_BLOCK, 0, static_cast<const long long*>(keys), (uint32_t)n, tile_counts, bm, d_flags);
  else if (stype == DTHIP_INT32)
    DTHIP_LAUNCH(ctx, "count_heads_kernel", count_heads_kernel<int32_t>, nt, GB_BLOCK, 0, static_cast<const int32_t*>(keys), (uint32_t)n, tile_counts, bm, d_flags);
  else { set_error("presorted heads: stype %d", stype); return DTHIP_ENOTIMPL; }
  uint32_t w[2] = {0, 0};
  DTHIP_TRY(read_back(ctx, &w[1], d_flags, sizeof(uint32_t)));
  if (w[1] == 0) {                                    // (a column that is not in order needs no scan)
    DTHIP_TRY(launch_scan_tiles(ctx, tile_counts, nt, tile_counts + nt));
    DTHIP_TRY(read_back(ctx, &w[0], tile_counts + nt, sizeof(uint32_t)));
  }
  *ngroups_host = w[0];
  *sorted_host = w[1] == 0;
  return DTHIP_OK;
}

// 8192 sampled neighbour pairs and 8192 pairs one stratum apart: *flag |= 1 when any of them descends (the cheap test
// before the full pass above is spent on a column)
template <typename T>
__global__ void __launch_bounds__(256) sorted_sample_kernel(const T* __restrict__ K, uint32_t n, uint32_t nsamp, uint32_t* flag) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  bool down = false;
  if (t < nsamp) {
    const uint32_t i = (uint32_t)(((unsigned long long)t * n) / nsamp), step = n / nsamp;
    const T a = K[i];
    if (i + 1 < n) down |= K[i + 1] < a;
    if (step && i + step < n) down |= K[i + step] < a;
  }
  if (__ballot(down) && lane_id() == 0) atomicOr(flag, 1u);
}

int launch_sorted_sample(dthip_ctx* ctx, const void* keys, int stype, int64_t n, uint32_t* d_flag, bool* maybe_sorted) {
  const uint32_t nsamp = (uint32_t)std::min<int64_t>(n, 8192);
  DTHIP_CHECK_HIP(hipMemsetAsync(d_flag, 0, sizeof(uint32_t), ctx->stream));
  if (stype == DTHIP_INT64)
    DTHIP_LAUNCH(ctx, "sorted_sample_kernel", sorted_sample_kernel<long long>, (nsamp + 255) / 256, 256, 0, static_cast<const long long*>(keys), (uint32_t)n, nsamp, d_flag);
  else if (stype == DTHIP_INT32)
    DTHIP_LAUNCH(ctx, "sorted_sample_kernel", sorted_sample_kernel<int32_t>, (nsamp + 255) / 256, 256, 0, static_cast<const int32_t*>(keys), (uint32_t)n, nsamp, d_flag);
  else { *maybe_sorted = false; return DTHIP_OK; }
  uint32_t w = 0;
  DTHIP_TRY(read_back(ctx, &w, d_flag, sizeof(w)));
  *maybe_sorted = w == 0;
  return DTHIP_OK;
}

int launch_write_offsets(dthip_ctx* ctx, const unsigned long long* bitmap, int64_t n,
                         const uint32_t* tile_base, int64_t ngroups, int32_t* offsets) {
  const uint32_t nt = ntiles_of(n);
  DTHIP_LAUNCH(ctx, "write_offsets_kernel", write_offsets_kernel, nt, GB_BLOCK, 0,
               reinterpret_cast<const uint8_t*>(bitmap), (uint32_t)n, tile_base, (uint32_t)ngroups, offsets);
  return DTHIP_OK;
}

// OR "key differs from predecessor" into a byte-per-position head array
// (multi-stage sorts whose packed keys exceed 64 bits)
template <typename KeyT>
__global__ void __launch_bounds__(256) mark_heads_kernel(const KeyT* K, uint32_t n, uint8_t* heads) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    if (i == 0 || K[i] != K[i - 1]) heads[i] = 1;
}

int launch_mark_heads(dthip_ctx* ctx, const void* keys, int key64, int64_t n, uint8_t* heads) {
  if (n == 0) return DTHIP_OK;
  long long blocks = (n + 2047) / 2048;
  if (blocks > ctx->num_cus * 8) blocks = ctx->num_cus * 8;
  if (key64) {
    DTHIP_LAUNCH(ctx, "mark_heads_kernel", mark_heads_kernel<unsigned long long>, (unsigned)blocks, 256, 0,
                 static_cast<const unsigned long long*>(keys), (uint32_t)n, heads);
  } else {
    DTHIP_LAUNCH(ctx, "mark_heads_kernel", mark_heads_kernel<uint32_t>, (unsigned)blocks, 256, 0,
                 static_cast<const uint32_t*>(keys), (uint32_t)n, heads);
  }
  return DTHIP_OK;
}

// head bitmap from an offsets array (dthip_reduce on caller-supplied groupings)
__global__ void __launch_bounds__(256) offsets_to_bitmap_kernel(const int32_t* offsets, uint32_t ngroups,
                                                                uint32_t* bitmap32) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g < ngroups) {
    const uint32_t p = (uint32_t)offsets[g];
    atomicOr(&bitmap32[p >> 5], 1u << (p & 31));
  }
}

__global__ void __launch_bounds__(256) bitmap_tile_counts_kernel(const unsigned long long* bitmap, uint32_t nwords,
                                                                 uint32_t ntiles, uint32_t* tile_counts) {
  // one wave per tile: 2048 positions = 32 words of 64 bits
  const uint32_t tile = blockIdx.x * 4 + wave_id();
  if (tile >= ntiles) return;
  const int lane = lane_id();
  const uint32_t w = tile * 32 + lane;
  uint32_t c = (lane < 32 && w < nwords) ? (uint32_t)__popcll(bitmap[w]) : 0;
  c = wave_reduce_sum_u32(c);
  if (lane == 0) tile_counts[tile] = c;
}

int launch_bitmap_from_offsets(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t n,
                               unsigned long long* bitmap, uint32_t* tile_counts, uint32_t* d_total) {
  const uint32_t nwords = (uint32_t)((n + 63) / 64);
  const uint32_t nt = ntiles_of(n);
  DTHIP_CHECK_HIP(hipMemsetAsync(bitmap, 0, (size_t)nwords * 8, ctx->stream));
  if (ngroups > 0) {
    DTHIP_LAUNCH(ctx, "offsets_to_bitmap_kernel", offsets_to_bitmap_kernel, (unsigned)((ngroups + 255) / 256), 256, 0,
                 offsets, (uint32_t)ngroups, reinterpret_cast<uint32_t*>(bitmap));
  }
  DTHIP_LAUNCH(ctx, "bitmap_tile_counts_kernel", bitmap_tile_counts_kernel, (nt + 3) / 4, 256, 0,
               bitmap, nwords, nt, tile_counts);
  DTHIP_TRY(launch_scan_tiles(ctx, tile_counts, nt, d_total));
  return DTHIP_OK;
}

// Groupby::ungroup_rowindex (groupby.cc:117-130): out[i] = index of the group that sorted position i
// belongs to; cumcount() / ngroup() (column/cumcountngroup.h:55-72) are the same walk.  The reference
// expands the offsets serially.  Here the offsets become the head bitmap (1 bit per position) and every
// position counts the heads up to itself: a tile of 2048 positions per workgroup, one bitmap byte (8
// positions) per thread, a workgroup scan of the byte popcounts on top of the heads of earlier tiles --
// streaming, instead of a binary search of the offsets per row (0.65 -> 0.25 ms per 1e8 rows).
// mode 0: group index (int32); 1: ngroup (int64); 2: cumcount (int64)
template <typename OT>
__global__ void __launch_bounds__(256) ungroup_kernel(const uint8_t* __restrict__ bitmap, const uint32_t* __restrict__ tile_first_head,
                                                      const int32_t* __restrict__ offsets, uint32_t ngroups, uint32_t n, int mode,
                                                      int rev, OT* __restrict__ out) {
  __shared__ uint32_t scratch[4];
  const int tid = threadIdx.x;
  const uint32_t p0 = blockIdx.x * GB_TILE + tid * 8;
  uint32_t hb = p0 < n ? bitmap[p0 >> 3] : 0u;
  if (p0 < n && p0 + 8 > n) hb &= (1u << (n - p0)) - 1u;
  const uint32_t before = block_excl_scan_u32<256>((uint32_t)__popc(hb), scratch, nullptr) + tile_first_head[blockIdx.x];
  if (p0 >= n) return;
  OT v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t g = before + (uint32_t)__popc(hb & ((2u << j) - 1u)) - 1u;     // heads at positions <= p0 + j, minus one
    const uint32_t p = p0 + j;
    long long r = (long long)g;
    if (p < n) {
      if (mode == 1) r = rev ? (long long)(ngroups - 1 - g) : (long long)g;
      else if (mode == 2) r = rev ? (long long)offsets[g + 1] - 1 - (long long)p : (long long)p - (long long)offsets[g];
    }
    v[j] = (OT)r;
  }
  if (p0 + 8 <= n) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* src = reinterpret_cast<const u32x4*>(v);
    u32x4* dst = reinterpret_cast<u32x4*>(out + p0);
#pragma unroll
    for (int q = 0; q < (int)(8 * sizeof(OT) / 16); q++) dst[q] = src[q];
  } else {
    for (int j = 0; j < 8 && p0 + j < n; j++) out[p0 + j] = v[j];
  }
}

// offsets -> bitmap + per-tile head counts, then the expansion.  mode / rev as in ungroup_kernel.
static int ungroup_common(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t n, int mode, int rev, void* out) {
  if (n == 0) return DTHIP_OK;
  Scratch sc(ctx);
  unsigned long long* bitmap = nullptr;
  uint32_t* tile_counts = nullptr;
  const uint32_t nt = ntiles_of(n);
  DTHIP_TRY(sc.get<unsigned long long>((size_t)((n + 63) / 64) + 1, &bitmap));
  DTHIP_TRY(sc.get<uint32_t>((size_t)nt + 4 + nt / 8192, &tile_counts));
  DTHIP_TRY(launch_bitmap_from_offsets(ctx, offsets, ngroups, n, bitmap, tile_counts, tile_counts + nt));
  if (mode == 0) {
    DTHIP_LAUNCH(ctx, "ungroup_kernel", ungroup_kernel<int32_t>, nt, 256, 0, reinterpret_cast<const uint8_t*>(bitmap), tile_counts,
                 offsets, (uint32_t)ngroups, (uint32_t)n, mode, rev, static_cast<int32_t*>(out));
  } else {
    DTHIP_LAUNCH(ctx, "ungroup_kernel", ungroup_kernel<long long>, nt, 256, 0, reinterpret_cast<const uint8_t*>(bitmap), tile_counts,
                 offsets, (uint32_t)ngroups, (uint32_t)n, mode, rev, static_cast<long long*>(out));
  }
  return DTHIP_OK;
}

int launch_ungroup(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t n, int32_t* out) {
  return ungroup_common(ctx, offsets, ngroups, n, 0, 0, out);
}

int launch_cumcount(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t n, int ngroup, int reverse, int64_t* out) {
  return ungroup_common(ctx, offsets, ngroups, n, ngroup ? 1 : 2, reverse ? 1 : 0, out);
}

// NaPosition::REMOVE (sort.cc:598-608): `skip` rows are cut off the front of the ordering.  g0 = groups
// that end inside the cut; the result's offsets are out[0] = 0, out[i] = in[g0 + i] - skip.
__global__ void __launch_bounds__(256) offsets_drop_rows_kernel(const int32_t* __restrict__ in, uint32_t ngroups, int32_t skip,
                                                                int32_t* __restrict__ out, int32_t* __restrict__ g0_out) {
  uint32_t lo = 0, hi = ngroups;                 // g0 = number of groups with in[g + 1] <= skip
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (in[mid + 1] <= skip) lo = mid + 1; else hi = mid;
  }
  const uint32_t g0 = lo;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *g0_out = (int32_t)g0;
  if (i <= ngroups - g0) out[i] = i == 0 ? 0 : in[g0 + i] - skip;
}

int launch_offsets_drop_rows(dthip_ctx* ctx, const int32_t* in, int64_t ngroups, int32_t skip, int32_t* out, int32_t* g0_out) {
  DTHIP_LAUNCH(ctx, "offsets_drop_rows_kernel", offsets_drop_rows_kernel, (unsigned)((ngroups + 256) / 256), 256, 0,
               in, (uint32_t)ngroups, skip, out, g0_out);
  return DTHIP_OK;
}

__global__ void __launch_bounds__(256) offsets_piece_kernel(const int32_t* __restrict__ in, uint32_t count, int32_t delta, int32_t* __restrict__ out) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += stride) out[i] = in[i] + delta;
}

int launch_offsets_piece(dthip_ctx* ctx, const int32_t* in, int64_t count, int32_t delta, int32_t* out) {
  if (count <= 0) return DTHIP_OK;
  long long blocks = (count + 2047) / 2048;
  if (blocks > ctx->num_cus * 8) blocks = ctx->num_cus * 8;
  DTHIP_LAUNCH(ctx, "offsets_piece_kernel", offsets_piece_kernel, (unsigned)blocks, 256, 0, in, (uint32_t)count, delta, out);
  return DTHIP_OK;
}

__global__ void __launch_bounds__(256) iota_kernel(int32_t* out, uint32_t n) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = (int32_t)i;
}

int launch_iota(dthip_ctx* ctx, int32_t* out, int64_t n) {
  if (n == 0) return DTHIP_OK;
  long long blocks = (n + 2047) / 2048;
  if (blocks > ctx->num_cus * 8) blocks = ctx->num_cus * 8;
  DTHIP_LAUNCH(ctx, "iota_kernel", iota_kernel, (unsigned)blocks, 256, 0, out, (uint32_t)n);
  return DTHIP_OK;
}

// Group-key column of a fused groupby: invert the key transform on the sorted
// packed key at the first position of each group (what the reference obtains as
// key[o[off[g]]], eval_context.cc:473-485).
struct UntransformArgs {
  const void* keys; int key64;
  const int32_t* offsets; uint32_t ngroups;
  int stype; int desc; unsigned long long edge, na_repl, inc; int shift; int bits;
  void* out;
};

__global__ void __launch_bounds__(256) untransform_kernel(UntransformArgs a) {
  typedef unsigned long long u64;
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= a.ngroups) return;
  const uint32_t p = a.offsets ? (uint32_t)a.offsets[g] : g;   // null offsets: keys[g] is the group's key
  u64 k = a.key64 ? static_cast<const u64*>(a.keys)[p] : (u64)static_cast<const uint32_t*>(a.keys)[p];
  k >>= a.shift;
  if (a.bits < 64) k &= (1ULL << a.bits) - 1ULL;
  const bool na = (k == a.na_repl);
  switch (a.stype) {
    case DTHIP_BOOL: {
      int8_t v = a.desc ? (int8_t)(2 - (int)k) : (int8_t)((int)k - 1);
      static_cast<int8_t*>(a.out)[g] = na ? INT8_MIN : v;
      break;
    }
    case DTHIP_INT8: case DTHIP_INT16: case DTHIP_INT32: case DTHIP_INT64: {
      const u64 u = a.desc ? a.edge - (k - a.inc) : (k - a.inc) + a.edge;
      const long long v = (long long)u;
      if (a.stype == DTHIP_INT8) static_cast<int8_t*>(a.out)[g] = na ? INT8_MIN : (int8_t)v;
      else if (a.stype == DTHIP_INT16) static_cast<int16_t*>(a.out)[g] = na ? INT16_MIN : (int16_t)v;
      else if (a.stype == DTHIP_INT32) static_cast<int32_t*>(a.out)[g] = na ? INT32_MIN : (int32_t)v;
      else static_cast<long long*>(a.out)[g] = na ? INT64_MIN : v;
      break;
    }
    case DTHIP_FLOAT32: {
      const uint32_t x = (uint32_t)k;
      uint32_t t;
      if (a.desc) t = (x & 0x80000000u) ? x : (x ^ 0x7FFFFFFFu);
      else t = (x & 0x80000000u) ? (x ^ 0x80000000u) : ~x;
      static_cast<uint32_t*>(a.out)[g] = na ? 0x7FC00000u : t;
      break;
    }
    default: {
      u64 t;
      if (a.desc) t = (k & 0x8000000000000000ULL) ? k : (k ^ 0x7FFFFFFFFFFFFFFFULL);
      else t = (k & 0x8000000000000000ULL) ? (k ^ 0x8000000000000000ULL) : ~k;
      static_cast<u64*>(a.out)[g] = na ? 0x7FF8000000000000ULL : t;
      break;
    }
  }
}

// The streaming case of the above (every row, no offsets): 32-bit packed keys -> an int64 column, FOUR consecutive rows per
// thread: one 16-byte load, two 16-byte stores (the one-row-per-thread form moved 6 GB per 5e8 rows at 4.0 TB/s)
typedef uint32_t gu32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) untransform_u32_i64_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift, int bits, int desc,
                                                                  unsigned long long edge, unsigned long long na_repl, unsigned long long inc,
                                                                  long long* __restrict__ out) {
  typedef unsigned long long u64;
  const uint32_t g4 = blockIdx.x * 256 + threadIdx.x;
  const uint32_t r0 = g4 * 4u;
  if (r0 >= n) return;
  const u64 mask = bits < 64 ? (1ULL << bits) - 1ULL : ~0ULL;
  if (r0 + 4u <= n) {
    const gu32x4 w = *reinterpret_cast<const gu32x4*>(keys + r0);
    const uint32_t kk[4] = {w.x, w.y, w.z, w.w};
    long long v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u64 k = ((u64)kk[j] >> shift) & mask;
      const u64 u = desc ? edge - (k - inc) : (k - inc) + edge;
      v[j] = (k == na_repl) ? (long long)INT64_MIN : (long long)u;
    }
    gu32x4 o0, o1;
    o0.x = (uint32_t)v[0]; o0.y = (uint32_t)((u64)v[0] >> 32); o0.z = (uint32_t)v[1]; o0.w = (uint32_t)((u64)v[1] >> 32);
    o1.x = (uint32_t)v[2]; o1.y = (uint32_t)((u64)v[2] >> 32); o1.z = (uint32_t)v[3]; o1.w = (uint32_t)((u64)v[3] >> 32);
    gu32x4* op = reinterpret_cast<gu32x4*>(out + r0);
    op[0] = o0; op[1] = o1;
  } else {
    for (uint32_t r = r0; r < n; r++) {
      const u64 k = ((u64)keys[r] >> shift) & mask;
      const u64 u = desc ? edge - (k - inc) : (k - inc) + edge;
      out[r] = (k == na_repl) ? (long long)INT64_MIN : (long long)u;
    }
  }
}

int launch_untransform_keys(dthip_ctx* ctx, const void* sorted_keys, int key64, const int32_t* offsets,
                            int64_t ngroups, const KeyColDev& col, int bits, void* out) {
  if (ngroups == 0) return DTHIP_OK;
  if (!offsets && !key64 && col.stype == DTHIP_INT64 && ngroups >= 4096 &&
      ((reinterpret_cast<uintptr_t>(sorted_keys) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const unsigned grid = (unsigned)((ngroups + 1023) / 1024);
    DTHIP_LAUNCH(ctx, "untransform_kernel", untransform_u32_i64_kernel, grid, 256, 0, static_cast<const uint32_t*>(sorted_keys), (uint32_t)ngroups,
                 col.shift, bits, col.desc, col.edge, col.na_repl, col.inc, static_cast<long long*>(out));
    return DTHIP_OK;
  }
  UntransformArgs a;
  a.keys = sorted_keys; a.key64 = key64; a.offsets = offsets; a.ngroups = (uint32_t)ngroups;
  a.stype = col.stype; a.desc = col.desc; a.edge = col.edge; a.na_repl = col.na_repl; a.inc = col.inc;
  a.shift = col.shift; a.bits = bits; a.out = out;
  DTHIP_LAUNCH(ctx, "untransform_kernel", untransform_kernel, (unsigned)((ngroups + 255) / 256), 256, 0, a);
  return DTHIP_OK;
}

}  // namespace dthip
