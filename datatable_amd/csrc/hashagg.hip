// hashagg.hip -- the hash combiner's kernels (split out of bucket.hip in round 6): LDS hash tables per hash bucket -> partial
// groups, and the typed columns of the partial groups that the merge reduces.  Host side: agg.hip hash_groupby_agg.
#include "agg_dev.hpp"
#include "keyxform.hpp"

namespace dthip {

// ---------------------------------------------------------------------------------------
// Hash combiner for SPARSE key ranges (wide integer ranges, float keys, > 32-bit composites): the
// slot of a key cannot be its value, so rows are partitioned by a HASH of the packed transformed key
// (same histogram / partition kernels, driven by a 24-bit pseudo key), every bucket part is
// aggregated into an LDS hash table (64-bit keys, linear probing, DS compare-and-swap), and the
// tables are written out as PARTIAL groups {key, accumulators}.  One key always hashes into one
// bucket; partial duplicates only come from big buckets that were split into parts.  The caller
// merges the partial groups (few rows) with the ordinary sort path, which also orders them by key.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 x) {       // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27; x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return x;
}

// xs[i] = packed transformed key of row i; pk[i] = top 24 bits of its hash (the pseudo key that the
// bucket histogram / partition kernels split on)
__global__ void __launch_bounds__(256) hash_xform_kernel(KeyXform kx, uint32_t n, u64* __restrict__ xs, int32_t* __restrict__ pk) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const u64 x = packed_key(kx.cols, kx.ncols, i);
    xs[i] = x;
    pk[i] = (int32_t)(mix64(x) >> 40);
  }
}

int launch_hash_xform(dthip_ctx* ctx, const KeyXform& kx, int64_t n, unsigned long long* xs, int32_t* pk) {
  if (n == 0) return DTHIP_OK;
  long long blocks = (n + 2047) / 2048;
  if (blocks > ctx->num_cus * 16) blocks = ctx->num_cus * 16;
  DTHIP_LAUNCH(ctx, "hash_xform_kernel", hash_xform_kernel, (unsigned)blocks, 256, 0, kx, (uint32_t)n, xs, pk);
  return DTHIP_OK;
}

// the same pseudo key for ONE int64 key column taken as it is: the hash tables then hold the raw key (any injective
// 64-bit image of the key works -- the merge of the partial groups orders and types them), so the 8-byte packed key
// need not be written at all: the key column itself rides through the partition as payload 0
__global__ void __launch_bounds__(256) hash_pk_raw_kernel(const u64* __restrict__ key, uint32_t n, int32_t* __restrict__ pk) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) pk[i] = (int32_t)(mix64(key[i]) >> 40);
}

int launch_hash_pk_raw(dthip_ctx* ctx, const void* key, int64_t n, int32_t* pk) {
  if (n == 0) return DTHIP_OK;
  long long blocks = (n + 2047) / 2048;
  if (blocks > ctx->num_cus * 16) blocks = ctx->num_cus * 16;
  DTHIP_LAUNCH(ctx, "hash_xform_kernel", hash_pk_raw_kernel, (unsigned)blocks, 256, 0, static_cast<const u64*>(key), (uint32_t)n, pk);
  return DTHIP_OK;
}

// Position and double-hashing step of a key inside a workgroup's hash table (C prime): 32-bit arithmetic.  Rounds 2-5 took a
// splitmix round + two 64-bit Lemire reductions -- five 64 x 64-bit multiplies = ~20 quarter-rate v_mul_lo / v_mul_hi per row,
// about a third of hash_agg_kernel's time (it is bound by its VALU work: profiles/r06_hash_tl_ab.txt).  Here: the two
// halves by two odd constants (others than hash_pk24's, which chose the bucket), one xorshift-multiply, two multiply-highs.
__device__ __forceinline__ void hash_tab_probe0(u64 x, uint32_t P, uint32_t& p, uint32_t& step) {
#ifdef DTHIP_HASH_MIX64            // (A/B flavour: `make var NAME=mix64 VFILE=hashagg FLAGS=-DDTHIP_HASH_MIX64`)
  const u64 h2 = mix64(x ^ 0x9E3779B97F4A7C15ULL);
  p = (uint32_t)(((h2 & 0xFFFFFFFFULL) * (u64)P) >> 32);
  step = 1u + (uint32_t)(((h2 >> 32) * (u64)(P - 1)) >> 32);
#else
  uint32_t h = (uint32_t)x * 0xC2B2AE3Du + (uint32_t)(x >> 32) * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
  p = __umulhi(h, P);
  step = 1u + __umulhi(h * 0x297A2D39u + 0x165667B1u, P - 1u);
#endif
}
__device__ __forceinline__ uint32_t hash_tab_pair0(u64 x, uint32_t P) {      // the first pair alone (the fast path needs no step)
  uint32_t p, step;
  hash_tab_probe0(x, P, p, step);
  return p;
}

constexpr u64 HASH_EMPTY = ~0ULL;

// occupied entries of a workgroup's hash table -> compact partial groups {key, raw accumulators} (unordered)
__device__ __forceinline__ void hash_tab_flush(const u64* hk, const LdsTab& t, int flags, uint32_t C, uint32_t* s_misc,
                                               u64* o_key, const AggTable& o_tab, uint32_t* out_n, uint32_t out_cap,
                                               uint32_t* overflow, int tid) {
  const uint32_t S = C + 1;
  uint32_t mine = 0;
  for (uint32_t s = tid; s < S; s += TA_BLOCK) mine += (s < C ? hk[s] != HASH_EMPTY : s_misc[16] != 0) ? 1u : 0u;
  uint32_t total;
  const uint32_t before = block_excl_scan_u32<TA_BLOCK>(mine, s_misc, &total);
  if (tid == 0) {
    const uint32_t base = atomicAdd(out_n, total);
    s_misc[17] = base;
    if (base + total > out_cap) { atomicOr(overflow, 2u); s_misc[17] = ~0u; }
  }
  __syncthreads();
  const uint32_t base = s_misc[17];
  if (base == ~0u) return;
  // block_excl_scan gives each thread the number of occupied entries of LOWER threads; entries of one
  // thread are strided, so positions are base + before + (rank among the thread's own entries)
  uint32_t pos = base + before;
  for (uint32_t s = tid; s < S; s += TA_BLOCK) {
    const bool occ = s < C ? hk[s] != HASH_EMPTY : s_misc[16] != 0;
    if (!occ) continue;
    o_key[pos] = s < C ? hk[s] : HASH_EMPTY;
    if (flags & ACC_CNT) o_tab.cnt[pos] = t.cnt[s];
    if (flags & ACC_VCNT) o_tab.vcnt[pos] = t.vcnt[s];
    if (flags & ACC_SUM) o_tab.sum[pos] = t.sum[s];
    if (flags & ACC_MIN) o_tab.mn[pos] = t.mn[s];
    if (flags & ACC_MAX) o_tab.mx[pos] = t.mx[s];
    if (flags & ACC_FSUM) o_tab.fsum[pos] = t.fsum[s];
    pos++;
  }
}

__device__ __forceinline__ void hash_tab_init(u64* hk, const LdsTab& t, int flags, uint32_t S, uint32_t* s_misc, int tid) {
  for (uint32_t s = tid; s < S; s += TA_BLOCK) {
    hk[s] = HASH_EMPTY;
    if (flags & ACC_SUM) t.sum[s] = 0;
    if (flags & ACC_MIN) t.mn[s] = ~0ULL;
    if (flags & ACC_MAX) t.mx[s] = 0;
    if (flags & ACC_FSUM) t.fsum[s] = 0.0;
    if (flags & ACC_CNT) t.cnt[s] = 0;
    if (flags & ACC_VCNT) t.vcnt[s] = 0;
  }
  if (tid == 0) s_misc[16] = 0;                      // special entry used?
}

struct HashAggDev {
  const WorkItem* items; const uint32_t* nitems;
  const u64* xs; const void* val;
  uint32_t C; int flags;
  // partial groups out (compact, unordered): key + raw accumulators, *out_n entries, capacity out_cap
  u64* o_key; AggTable o_tab; uint32_t* out_n; uint32_t out_cap;
  uint32_t* overflow;      // set when a table fills up or the output capacity is exceeded
};

// ---------------------------------------------------------------------------------------
// One row into a workgroup's hash table, wave by wave (round 6).  The table is C = 2 P entries read as P PAIRS (one 16-byte DS
// read shows both keys of a pair; double hashing over the pairs, P prime; a key lives in the first entry of its probe sequence
// -- pair by pair, entry 0 before entry 1 -- that was empty when it arrived).  Rows outnumber keys ~100 : 1, so nearly every row finds its
// key already in the table, and most of those at the FIRST entry of their probe sequence.  Rounds 2-5 let every lane walk a
// private queue of eight rows at its own pace inside one fat divergent loop (the whole body ran ~18 times per eight rows for
// the wave's slowest lane: the kernel was bound by that VALU work, 5.8 ms per 1e9 rows against 3 for its 16 GB).  Now:
//   step()   straight-line: hash, ONE plain DS read, key found -> accumulate.  Everything else (another key there, or an
//            empty entry = first row of a key) is a PENDING row: compacted (ballot + mbcnt) into the wave's own queue in LDS
//   drain()  when 64 rows are pending (or at the end): lane = pending row, the full probe loop with compare-and-swap claims;
//            the loop runs as long as the slowest of 64 SLOW rows needs, not of all rows
// DS instructions of a wave execute in order, so the queue needs no barrier beyond the compiler's (wave_barrier).
// ---------------------------------------------------------------------------------------
constexpr uint32_t HQ_ROWS = 64;
constexpr size_t HQ_BYTES = (size_t)(TA_BLOCK / 64) * HQ_ROWS * 16;      // every wave: 64 x {key, value bits}

template <typename VT> __device__ __forceinline__ u64 val_bits(VT v) { u64 b = 0; __builtin_memcpy(&b, &v, sizeof(VT)); return b; }
template <typename VT> __device__ __forceinline__ VT bits_val(u64 b) { VT v; __builtin_memcpy(&v, &b, sizeof(VT)); return v; }

template <typename VT>
struct HashIns {
  u64* hk; LdsTab t; uint32_t C; int flags; uint32_t* s_misc;
  u64* qx; u64* qv;          // this wave's queue
  uint32_t qn;               // pending rows in it (wave-uniform: kept in a scalar register)
  bool full;

  __device__ __forceinline__ void drain() {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t P = C >> 1;
    __builtin_amdgcn_wave_barrier();
    bool pend = lane < qn;
    u64 x = 0; VT v = VT(0);
    if (pend) {
      x = __hip_atomic_load(&qx[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      v = bits_val<VT>(__hip_atomic_load(&qv[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
    }
    if (pend && x == HASH_EMPTY) { s_misc[16] = 1; acc_row<VT, false>(t, flags, C, v); pend = false; }   // the one key that equals the empty mark
    uint32_t p = 0, step = 1, probes = 0;
    hash_tab_probe0(x, P, p, step);
    while (__ballot(pend)) {
      if (pend) {
        const bu32x4 w = *reinterpret_cast<const bu32x4*>(&hk[2u * p]);
        const u64 k0 = (u64)w.x | ((u64)w.y << 32), k1 = (u64)w.z | ((u64)w.w << 32);
        if (k0 == x || k1 == x) { acc_row<VT, false>(t, flags, 2u * p + (k0 == x ? 0u : 1u), v); pend = false; }
        else if (k0 == HASH_EMPTY || k1 == HASH_EMPTY) {
          // claim the FIRST empty entry of the pair; when another key got there first, look at the same pair again
          const uint32_t s = 2u * p + (k0 == HASH_EMPTY ? 0u : 1u);
          const u64 cur = atomicCAS(&hk[s], HASH_EMPTY, x);                  // returns what was there: EMPTY = claimed
          if (cur == HASH_EMPTY || cur == x) { acc_row<VT, false>(t, flags, s, v); pend = false; }
        } else {
          p += step; if (p >= P) p -= P;
          if (++probes >= P) { full = true; pend = false; }
        }
      }
    }
    qn = 0;
    __builtin_amdgcn_wave_barrier();
  }

  // every lane of the wave calls this together (uniform control flow); `active` lanes bring a row
  __device__ __forceinline__ void step(u64 x, VT v, bool active) {
    const uint32_t p = hash_tab_pair0(x, C >> 1);
    const bu32x4 w = *reinterpret_cast<const bu32x4*>(&hk[2u * p]);
    const u64 k0 = (u64)w.x | ((u64)w.y << 32), k1 = (u64)w.z | ((u64)w.w << 32);
    const bool hit = active && (k0 == x || k1 == x) && x != HASH_EMPTY;
    if (hit) acc_row<VT, false>(t, flags, 2u * p + (k0 == x ? 0u : 1u), v);
    const bool pend = active && !hit;
    const unsigned long long m = __ballot(pend);
    if (m) {
      const uint32_t c = (uint32_t)__popcll(m);
      if (qn + c > HQ_ROWS) drain();
      if (pend) {
        const uint32_t pos = qn + mbcnt64(m);
        __hip_atomic_store(&qx[pos], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&qv[pos], val_bits<VT>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      qn = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qn + c));
    }
  }
};

// CFLAGS >= 0: the accumulator set is a compile-time constant (the common sum / sum+count shapes: the
// unrolled inserts then carry no per-row flag tests); -1: taken from the arguments
template <typename VT, int CFLAGS>
__global__ void __launch_bounds__(TA_BLOCK) hash_agg_kernel(HashAggDev a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x >= *a.nitems) return;
  const WorkItem it = a.items[blockIdx.x];
  const int tid = threadIdx.x;
  const int flags = CFLAGS >= 0 ? CFLAGS : a.flags;
  const uint32_t C = a.C, S = C + 1;                 // entry C is reserved for the key that equals HASH_EMPTY
  u64* hk = reinterpret_cast<u64*>(smem + HQ_BYTES);            // [S] keys (behind the waves' queues)
  const LdsTab t = carve_tab(smem + HQ_BYTES + (size_t)S * 8, S, flags);
  __shared__ uint32_t s_misc[20];
  hash_tab_init(hk, t, flags, S, s_misc, tid);
  __syncthreads();
  const VT* __restrict__ val = static_cast<const VT*>(a.val);
  const bool hasval = (flags & (ACC_SUM | ACC_MIN | ACC_MAX | ACC_VCNT | ACC_FSUM)) != 0;
  HashIns<VT> hi;
  hi.hk = hk; hi.t = t; hi.C = C; hi.flags = flags; hi.s_misc = s_misc; hi.qn = 0; hi.full = false;
  hi.qx = reinterpret_cast<u64*>(smem) + (size_t)(tid >> 6) * (2 * HQ_ROWS); hi.qv = hi.qx + HQ_ROWS;
  {
    // 8 consecutive rows per thread and iteration: all their 16-byte loads are in flight before the first probe
    uint32_t a0 = (it.begin + 7u) & ~7u; if (a0 > it.end) a0 = it.end;
    uint32_t a1 = it.end & ~7u; if (a1 < a0) a1 = a0;
    const uint32_t nh = a0 - it.begin, ntl = it.end - a1;       // < 8 rows each: the ragged head and tail of the part
    {
      const bool act = (uint32_t)tid < nh || ((uint32_t)tid >= 64u && (uint32_t)tid - 64u < ntl);
      const uint32_t row = (uint32_t)tid < nh ? it.begin + tid : a1 + ((uint32_t)tid - 64u);
      u64 x = 0; VT v = VT(0);
      if (act) { x = a.xs[row]; if (hasval) v = val[row]; }
      if (tid < 128) hi.step(x, v, act);               // (waves 0 and 1, whole)
    }
    // a wave takes 512 consecutive rows at a time as four coalesced 1 KB loads of keys (two rows per lane and load) and
    // four of values, all in flight before the first probe; sums do not care about the order of the rows.  (Rounds 2-5
    // gave every lane 8 CONSECUTIVE rows: each of its load instructions then touched 64 different 64-byte lines.)
    typedef VT VT2 __attribute__((ext_vector_type(2)));
    const uint32_t lane2 = ((uint32_t)tid & 63u) * 2u;
    for (uint32_t base = a0 + (uint32_t)(tid >> 6) * 512u; base < a1; base += (TA_BLOCK / 64) * 512u) {
      bu32x4 kw[4];
      VT2 vv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t row = base + (uint32_t)j * 128u + lane2;
        kw[j].x = 0; kw[j].y = 0; kw[j].z = 0; kw[j].w = 0; vv[j].x = VT(0); vv[j].y = VT(0);
        if (row < a1) {
          kw[j] = *reinterpret_cast<const bu32x4*>(a.xs + row);
          if (hasval) vv[j] = *reinterpret_cast<const VT2*>(val + row);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool act = base + (uint32_t)j * 128u + lane2 < a1;
        hi.step((u64)kw[j].x | ((u64)kw[j].y << 32), vv[j].x, act);
        hi.step((u64)kw[j].z | ((u64)kw[j].w << 32), vv[j].y, act);
      }
    }
  }
  if (hi.qn) hi.drain();
  if (__ballot(hi.full) && (tid & 63) == 0) atomicOr(a.overflow, 1u);
  __syncthreads();
  hash_tab_flush(hk, t, flags, C, s_misc, a.o_key, a.o_tab, a.out_n, a.out_cap, a.overflow, tid);
}

size_t hash_agg_entry_bytes(int flags) { return 8 + table_agg_slot_bytes(flags); }
size_t hash_agg_queue_bytes() { return HQ_BYTES; }

template <typename VT, int CFLAGS>
static int hash_agg_t(dthip_ctx* ctx, const HashAggDev& d, uint32_t grid, size_t lds) {
  auto kfn = hash_agg_kernel<VT, CFLAGS>;
  DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 256));
  DTHIP_LAUNCH(ctx, "hash_agg_kernel", kfn, grid, TA_BLOCK, lds, d);
  return DTHIP_OK;
}

int launch_hash_agg(dthip_ctx* ctx, const HashAggArgs& a) {
  if (a.max_items == 0) return DTHIP_OK;
  HashAggDev d;
  d.items = a.items; d.nitems = a.nitems; d.xs = a.xs; d.val = a.val; d.C = a.C; d.flags = a.flags;
  d.o_key = a.o_key; d.o_tab = a.o_tab; d.out_n = a.out_n; d.out_cap = a.out_cap; d.overflow = a.overflow;
  const size_t lds = hash_agg_queue_bytes() + (size_t)(a.C + 1) * hash_agg_entry_bytes(a.flags) + 32;
  if (lds > 160 * 1024 - 512) { set_error("hash_agg: table of %zu bytes exceeds LDS", lds); return DTHIP_EINVAL; }
  if ((a.C & 1u) || a.C < 6) { set_error("hash_agg: the table is read as pairs of entries (C = %u)", a.C); return DTHIP_EINVAL; }
  const int st = a.val ? a.vstype : DTHIP_INT32;
  if (st == DTHIP_FLOAT64 && a.flags == ACC_SUM) return hash_agg_t<double, ACC_SUM>(ctx, d, a.max_items, lds);
  if (st == DTHIP_FLOAT64 && a.flags == (ACC_SUM | ACC_CNT)) return hash_agg_t<double, ACC_SUM | ACC_CNT>(ctx, d, a.max_items, lds);
  switch (st) {
    case DTHIP_INT32: return hash_agg_t<int32_t, -1>(ctx, d, a.max_items, lds);
    case DTHIP_INT64: return hash_agg_t<long long, -1>(ctx, d, a.max_items, lds);
    case DTHIP_FLOAT32: return hash_agg_t<float, -1>(ctx, d, a.max_items, lds);
    case DTHIP_FLOAT64: return hash_agg_t<double, -1>(ctx, d, a.max_items, lds);
    default: set_error("hash_agg: unsupported value stype %d", a.vstype); return DTHIP_ENOTIMPL;
  }
}

// ---------------------------------------------------------------------------------------
// Tile-local hash partition into 16-BYTE RECORDS {key, value} (round 6): ONE aligned int64 key column, one 8-byte value
// column.  1024 threads x 16 rows; every row is ranked inside its bucket (hash_pk24 >> r) by one DS atomic, the tile's
// rows go back over the tile's own records in bucket order (two rounds of 8192 records through 128 KB of LDS, dwordx4 both
// ways), dir[tile][b] = first record of bucket b inside the tile (b <= F).  Against the column layout (bucket_partition_kernel
// with the key as payload 0): the key column is read once, and bucket b's ~8 rows of a tile are 128 contiguous bytes for
// hash_agg_seg_kernel instead of 64 + 64 at two places (2.75 instead of 3.75 sectors of 64 bytes per segment).
// ---------------------------------------------------------------------------------------
struct HashPartRecArgs {
  const u64* key; const u64* val; uint32_t n; int r; uint32_t F;
  bu32x4* rec; uint16_t* dir;
};

__global__ void __launch_bounds__(1024) hash_partition_rec_kernel(HashPartRecArgs a) {
  constexpr uint32_t BLOCK = 1024, ITEMS = 16, TILE = BLOCK * ITEMS, HALF = TILE / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t F = a.F, Fp = (F + 4u) & ~3u;
  uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);          // [Fp] bucket counts, then tile-local exclusive starts
  uint32_t* misc = cnt + Fp;                                  // [32]
  bu32x4* stage = reinterpret_cast<bu32x4*>(misc + 32);       // [HALF] records
  const uint32_t tid = threadIdx.x;
  const uint32_t nt = gridDim.x, bi = blockIdx.x;
  const uint32_t xq = nt / 8, xr = nt % 8, xc = bi % 8, q0 = bi / 8;
  const uint32_t tile = xc * xq + (xc < xr ? xc : xr) + q0;
  const uint32_t tile_base = tile * TILE;
  const uint32_t nvalid = (a.n - tile_base < TILE) ? (a.n - tile_base) : TILE;
  const bool full = nvalid == TILE;
  for (uint32_t b = tid; b < F + 1u; b += BLOCK) cnt[b] = 0;
  // item j of a thread = row ((j / 2) * BLOCK + tid) * 2 + j % 2 of the tile: 16-byte loads of two consecutive rows
  bu32x4 kw[ITEMS / 2], vw[ITEMS / 2];
  const bu32x4* ksrc = reinterpret_cast<const bu32x4*>(a.key + tile_base);
  const bu32x4* vsrc = reinterpret_cast<const bu32x4*>(a.val + tile_base);
  if (full) {
#pragma unroll
    for (int q = 0; q < (int)ITEMS / 2; q++) kw[q] = ksrc[(uint32_t)q * BLOCK + tid];
#pragma unroll
    for (int q = 0; q < (int)ITEMS / 2; q++) vw[q] = vsrc[(uint32_t)q * BLOCK + tid];
  } else {
#pragma unroll
    for (int q = 0; q < (int)ITEMS / 2; q++) {
      const uint32_t r0 = ((uint32_t)q * BLOCK + tid) * 2u;
      kw[q].x = 0; kw[q].y = 0; kw[q].z = 0; kw[q].w = 0; vw[q] = kw[q];
      if (r0 < nvalid) { const u64 k = a.key[tile_base + r0], v = a.val[tile_base + r0]; kw[q].x = (uint32_t)k; kw[q].y = (uint32_t)(k >> 32); vw[q].x = (uint32_t)v; vw[q].y = (uint32_t)(v >> 32); }
      if (r0 + 1u < nvalid) { const u64 k = a.key[tile_base + r0 + 1u], v = a.val[tile_base + r0 + 1u]; kw[q].z = (uint32_t)k; kw[q].w = (uint32_t)(k >> 32); vw[q].z = (uint32_t)v; vw[q].w = (uint32_t)(v >> 32); }
    }
  }
  __syncthreads();
  // (bucket << 16) | arrival rank inside the bucket; rows past the end of a ragged last tile go to the extra bin F behind
  // the rows that count and are never written
  uint32_t pk[ITEMS];
#pragma unroll
  for (int j = 0; j < (int)ITEMS; j++) {
    const bu32x4 w = kw[j >> 1];
    const u64 k = (j & 1) ? ((u64)w.z | ((u64)w.w << 32)) : ((u64)w.x | ((u64)w.y << 32));
    uint32_t b = hash_pk24(k) >> a.r;
    if (!full && (((uint32_t)(j >> 1) * BLOCK + tid) * 2u + (uint32_t)(j & 1)) >= nvalid) b = F;
    pk[j] = (b << 16) | atomicAdd(&cnt[b], 1u);
  }
  __syncthreads();
  {
    const uint32_t K = (F + BLOCK - 1) / BLOCK;            // consecutive bins per thread (F <= 2048: K <= 2)
    uint32_t c[2], sm = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t b = tid * K + (uint32_t)k;
      c[k] = ((uint32_t)k < K && b < F) ? cnt[b] : 0u;
      sm += c[k];
    }
    uint32_t e = block_excl_scan_u32<BLOCK>(sm, misc, nullptr);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t b = tid * K + (uint32_t)k;
      if ((uint32_t)k < K && b < F) { cnt[b] = e; a.dir[(size_t)tile * (F + 1) + b] = (uint16_t)e; e += c[k]; }
    }
    if (tid == 0) { const uint32_t ngood = TILE - cnt[F]; a.dir[(size_t)tile * (F + 1) + F] = (uint16_t)ngood; cnt[F] = ngood; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < (int)ITEMS; j++) pk[j] = cnt[pk[j] >> 16] + (pk[j] & 0xFFFFu);      // place of the row among the tile's records
  bu32x4* out = a.rec + (size_t)tile_base;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    if (h) __syncthreads();
#pragma unroll
    for (int j = 0; j < (int)ITEMS; j++) {
      const uint32_t pl = pk[j] - (uint32_t)h * HALF;
      if (pl < HALF) {
        const bu32x4 wk = kw[j >> 1], wv = vw[j >> 1];
        bu32x4 rcd;
        if (j & 1) { rcd.x = wk.z; rcd.y = wk.w; rcd.z = wv.z; rcd.w = wv.w; }
        else { rcd.x = wk.x; rcd.y = wk.y; rcd.z = wv.x; rcd.w = wv.y; }
        stage[pl] = rcd;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (int)(HALF / BLOCK); i++) {
      const uint32_t sl = (uint32_t)h * HALF + (uint32_t)i * BLOCK + tid;
      if (sl < nvalid) out[sl] = stage[sl - (uint32_t)h * HALF];
    }
  }
}

int launch_hash_partition_rec(dthip_ctx* ctx, const void* key, const void* val, int64_t n, int r, uint32_t F, uint32_t ntiles,
                              void* rec, uint16_t* dir) {
  if (n == 0) return DTHIP_OK;
  if (F > 2048 || F < 1) { set_error("hash partition: F=%u", F); return DTHIP_EINVAL; }
  HashPartRecArgs a;
  a.key = static_cast<const u64*>(key); a.val = static_cast<const u64*>(val); a.n = (uint32_t)n; a.r = r; a.F = F;
  a.rec = static_cast<bu32x4*>(rec); a.dir = dir;
  const size_t lds = (size_t)(((F + 4u) & ~3u) + 32) * 4 + (size_t)8192 * 16;
  auto kfn = hash_partition_rec_kernel;
  DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 256));
  DTHIP_LAUNCH(ctx, "hash_partition_rec_kernel", kfn, ntiles, 1024, lds, a);
  return DTHIP_OK;
}

// ---------------------------------------------------------------------------------------
// The same tables fed from a TILE-LOCAL partition (round 6): 16384-row tiles written sequentially, bucket b's rows are one
// segment of ~8 rows per tile, found through the transposed directory dirT[b][tile] (bucket.hip dir_transpose_kernel).  No
// histogram pass and no scattered 48-byte runs: the partition takes 5.9 ms instead of 8.5 + 1.6 (1e9 rows).
// A wave takes 56 tiles at a time (lane = tile: coalesced directory reads), lays their segments end to end as ONE sequence of
// T ~ 448 rows (prefix sum of the lengths across the lanes) and deals it out 64 rows per step: a lane finds the segment of
// its row by binary search in the prefix sums (six ds_bpermute).  8 steps cover T <= 512 (mean + 3 sigma); all their loads are
// in flight before the first probe.  (A first version of this kernel -- 8 lanes per segment, every lane walking a private
// queue of rows -- took 14.2 ms: half the lanes idle and the walk loop's VALU work; profiles/r06_hash_tl_ab.txt.)
// ---------------------------------------------------------------------------------------
struct HashAggSegDev {
  const WorkItem* items; const uint32_t* nitems;
  const u64* xs; const void* val;          // REC: xs = the 16-byte records {key, value bits}, val unused
  const uint16_t* dirT; uint32_t dstride; uint32_t tile_rows;
  uint32_t C; int flags;
  uint32_t map;        // work item of workgroup i: 0 = XCD i % 8 takes a contiguous range of items, 1 = item i, m > 1 = item (i * m) % nitems
  u64* o_key; AggTable o_tab; uint32_t* out_n; uint32_t out_cap;
  uint32_t* overflow;
};

constexpr uint32_t HSEG_TILES = 56;

template <typename VT, int CFLAGS, bool REC>
__global__ void __launch_bounds__(TA_BLOCK) hash_agg_seg_kernel(HashAggSegDev a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // items are dealt to XCDs in contiguous bucket ranges (as in table_agg_seg_kernel): the sectors two neighbouring buckets'
  // segments share are fetched from HBM once
  const uint32_t nit = *a.nitems;
  const uint32_t bi = blockIdx.x;
  uint32_t ii;
  if (a.map == 0) {
    const uint32_t xq = nit / 8, xr = nit % 8, xc = bi % 8, q0 = bi / 8;
    if (q0 >= xq + (xc < xr ? 1u : 0u)) return;
    ii = xc * xq + (xc < xr ? xc : xr) + q0;
  } else if (a.map == 1) {
    if (bi >= nit) return;
    ii = bi;
  } else {
    if (bi >= nit) return;
    ii = (uint32_t)(((unsigned long long)bi * (unsigned long long)a.map) % nit);       // a.map coprime to nit (host)
  }
  const WorkItem it = a.items[ii];
  const int tid = threadIdx.x;
  const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
  const int flags = CFLAGS >= 0 ? CFLAGS : a.flags;
  const uint32_t C = a.C, S = C + 1;
  u64* hk = reinterpret_cast<u64*>(smem + HQ_BYTES);
  const LdsTab t = carve_tab(smem + HQ_BYTES + (size_t)S * 8, S, flags);
  __shared__ uint32_t s_misc[20];
  hash_tab_init(hk, t, flags, S, s_misc, tid);
  __syncthreads();
  const VT* __restrict__ val = static_cast<const VT*>(a.val);
  const u64* __restrict__ xs = a.xs;
  const bool hasval = (flags & (ACC_SUM | ACC_MIN | ACC_MAX | ACC_VCNT | ACC_FSUM)) != 0;
  HashIns<VT> hi;
  hi.hk = hk; hi.t = t; hi.C = C; hi.flags = flags; hi.s_misc = s_misc; hi.qn = 0; hi.full = false;
  hi.qx = reinterpret_cast<u64*>(smem) + (size_t)wave * (2 * HQ_ROWS); hi.qv = hi.qx + HQ_ROWS;
  const uint16_t* __restrict__ ds = a.dirT + (size_t)it.bucket * a.dstride;
  const uint16_t* __restrict__ de = ds + a.dstride;
  const uint32_t t1 = it.end, tr = a.tile_rows;
  constexpr uint32_t WSTEP = (TA_BLOCK / 64) * HSEG_TILES;
  // directory entries of the wave's first chunk; the next chunk's are loaded while the current one is processed
  uint32_t c = it.begin + wave * HSEG_TILES;
  uint32_t st_n = 0, ln_n = 0;
  if (lane < HSEG_TILES && c + lane < t1) { st_n = ds[c + lane]; ln_n = (uint32_t)de[c + lane] - st_n; }
  for (; c < t1; c += WSTEP) {
    const uint32_t st = st_n, ln = ln_n;
    {
      const uint32_t tn = c + WSTEP + lane;
      st_n = 0; ln_n = 0;
      if (lane < HSEG_TILES && tn < t1) { st_n = ds[tn]; ln_n = (uint32_t)de[tn] - st_n; }
    }
    uint32_t inc = ln;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, o, 64); if (lane >= (uint32_t)o) inc += u; }
    const uint32_t excl = inc - ln;
    const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    for (uint32_t k0 = 0; k0 < T; k0 += 512u) {
      u64 kx[REC ? 1 : 8]; VT vv[REC ? 1 : 8];
      bu32x4 rw[REC ? 8 : 1];
      // the eight binary searches run side by side: one round = eight independent ds_bpermute, one wait
      uint32_t lo[8];
#pragma unroll
      for (int j = 0; j < 8; j++) lo[j] = 0;
#pragma unroll
      for (int bit = 32; bit > 0; bit >>= 1) {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = (uint32_t)__shfl((int)excl, (int)(lo[j] + (uint32_t)bit), 64);
#pragma unroll
        for (int j = 0; j < 8; j++) if (e[j] <= k0 + 64u * (uint32_t)j + lane) lo[j] += (uint32_t)bit;
      }
      uint32_t fe[8], fs[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { fe[j] = (uint32_t)__shfl((int)excl, (int)lo[j], 64); fs[j] = (uint32_t)__shfl((int)st, (int)lo[j], 64); }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t r = k0 + 64u * (uint32_t)j + lane;
        const uint32_t row = (c + lo[j]) * tr + fs[j] + (r - fe[j]);
        // lanes past the end of the sequence load the first row of the chunk's first tile (always there) and ignore it: no
        // branch around the loads.  (With `if (r < T) load` the compiler merged the eight conditionally written registers
        // as one array and followed every load by s_waitcnt vmcnt(0): eight dependent round trips per 448 rows.)
        const uint32_t rs = r < T ? row : c * tr;
        if (REC) rw[j] = reinterpret_cast<const bu32x4*>(xs)[rs];      // one 16-byte record per row (hash_partition_rec_kernel)
        else { kx[j] = xs[rs]; vv[j] = hasval ? val[rs] : VT(0); }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool act = k0 + 64u * (uint32_t)j + lane < T;
        if (REC) hi.step((u64)rw[j].x | ((u64)rw[j].y << 32), bits_val<VT>((u64)rw[j].z | ((u64)rw[j].w << 32)), act);
        else hi.step(kx[j], vv[j], act);
      }
    }
  }
  if (hi.qn) hi.drain();
  if (__ballot(hi.full) && lane == 0) atomicOr(a.overflow, 1u);
  __syncthreads();
  hash_tab_flush(hk, t, flags, C, s_misc, a.o_key, a.o_tab, a.out_n, a.out_cap, a.overflow, tid);
}

template <typename VT, int CFLAGS, bool REC = false>
static int hash_agg_seg_t(dthip_ctx* ctx, const HashAggSegDev& d, uint32_t grid, size_t lds) {
  auto kfn = hash_agg_seg_kernel<VT, CFLAGS, REC>;
  DTHIP_TRY(ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kfn), 160 * 1024 - 256));
  DTHIP_LAUNCH(ctx, "hash_agg_seg_kernel", kfn, grid, TA_BLOCK, lds, d);
  return DTHIP_OK;
}

int launch_hash_agg_seg(dthip_ctx* ctx, const HashAggArgs& a, const uint16_t* dirT, uint32_t dstride, uint32_t tile_rows, bool rec) {
  if (a.max_items == 0) return DTHIP_OK;
  HashAggSegDev d;
  d.items = a.items; d.nitems = a.nitems; d.xs = a.xs; d.val = a.val; d.C = a.C; d.flags = a.flags;
  d.dirT = dirT; d.dstride = dstride; d.tile_rows = tile_rows;
  static const int map_env = getenv("DTHIP_HSEG_MAP") ? atoi(getenv("DTHIP_HSEG_MAP")) : 0;
  d.map = (uint32_t)map_env;
  d.o_key = a.o_key; d.o_tab = a.o_tab; d.out_n = a.out_n; d.out_cap = a.out_cap; d.overflow = a.overflow;
  const size_t lds = hash_agg_queue_bytes() + (size_t)(a.C + 1) * hash_agg_entry_bytes(a.flags) + 32;
  if (lds > 160 * 1024 - 512) { set_error("hash_agg_seg: table of %zu bytes exceeds LDS", lds); return DTHIP_EINVAL; }
  if ((a.C & 1u) || a.C < 6) { set_error("hash_agg_seg: the table is read as pairs of entries (C = %u)", a.C); return DTHIP_EINVAL; }
  const uint32_t grid = (a.max_items + 7u) & ~7u;
  const int st = (a.val || rec) ? a.vstype : DTHIP_INT32;
  if (rec) {        // records: one 8-byte value column
    if (st == DTHIP_FLOAT64 && a.flags == ACC_SUM) return hash_agg_seg_t<double, ACC_SUM, true>(ctx, d, grid, lds);
    if (st == DTHIP_FLOAT64 && a.flags == (ACC_SUM | ACC_CNT)) return hash_agg_seg_t<double, ACC_SUM | ACC_CNT, true>(ctx, d, grid, lds);
    if (st == DTHIP_FLOAT64) return hash_agg_seg_t<double, -1, true>(ctx, d, grid, lds);
    if (st == DTHIP_INT64) return hash_agg_seg_t<long long, -1, true>(ctx, d, grid, lds);
    set_error("hash_agg_seg: records carry an 8-byte value column"); return DTHIP_EINVAL;
  }
  if (st == DTHIP_FLOAT64 && a.flags == ACC_SUM) return hash_agg_seg_t<double, ACC_SUM>(ctx, d, grid, lds);
  if (st == DTHIP_FLOAT64 && a.flags == (ACC_SUM | ACC_CNT)) return hash_agg_seg_t<double, ACC_SUM | ACC_CNT>(ctx, d, grid, lds);
  switch (st) {
    case DTHIP_INT32: return hash_agg_seg_t<int32_t, -1>(ctx, d, grid, lds);
    case DTHIP_INT64: return hash_agg_seg_t<long long, -1>(ctx, d, grid, lds);
    case DTHIP_FLOAT32: return hash_agg_seg_t<float, -1>(ctx, d, grid, lds);
    case DTHIP_FLOAT64: return hash_agg_seg_t<double, -1>(ctx, d, grid, lds);
    default: set_error("hash_agg_seg: unsupported value stype %d", a.vstype); return DTHIP_ENOTIMPL;
  }
}

// raw accumulators of the partial groups -> typed columns the merge can reduce:
//   psum  float64 (float values) / int64 (integer values)      pfsum float64 (integer values, for mean)
//   pmin / pmax in the value's own stype (NA when no valid row)  pvcnt / pcnt int64
__global__ void __launch_bounds__(256) partial_columns_kernel(PartialColsArgs a) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= a.n) return;
  const int st = a.vstype;
  const bool isf = st == DTHIP_FLOAT32 || st == DTHIP_FLOAT64;
  const uint32_t vc = a.tab.vcnt ? a.tab.vcnt[g] : 0u;
  if (a.o_cnt) a.o_cnt[g] = (int64_t)a.tab.cnt[g];
  if (a.o_vcnt) a.o_vcnt[g] = (int64_t)vc;
  if (a.o_sum) a.o_sum[g] = a.tab.sum[g];                       // same 8 bytes: float64 bits or int64
  if (a.o_fsum) a.o_fsum[g] = a.tab.fsum[g];
  for (int which = 0; which < 2; which++) {
    void* o = which ? a.o_max : a.o_min;
    if (!o) continue;
    const u64 k = which ? a.tab.mx[g] : a.tab.mn[g];
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
}

int launch_partial_columns(dthip_ctx* ctx, const PartialColsArgs& a) {
  if (a.n == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "partial_columns_kernel", partial_columns_kernel, (a.n + 255) / 256, 256, 0, a);
  return DTHIP_OK;
}

// mean = merged sum / merged valid count (NA when the count is 0); float32 input -> float32 output
__global__ void __launch_bounds__(256) mean_div_kernel(const double* sum, const long long* cnt, uint32_t n, void* out, int out_f32) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  const long long c = cnt[g];
  const double m = c > 0 ? sum[g] / (double)c : __builtin_nan("");
  if (out_f32) static_cast<float*>(out)[g] = c > 0 ? (float)m : __builtin_nanf("");
  else static_cast<double*>(out)[g] = m;
}
__global__ void __launch_bounds__(256) cast_f64_f32_kernel(const double* in, uint32_t n, float* out) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g < n) out[g] = (float)in[g];
}
int launch_mean_div(dthip_ctx* ctx, const double* sum, const long long* cnt, int64_t n, void* out, int out_f32) {
  if (n == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "mean_div_kernel", mean_div_kernel, (unsigned)((n + 255) / 256), 256, 0, sum, cnt, (uint32_t)n, out, out_f32);
  return DTHIP_OK;
}
int launch_cast_f64_f32(dthip_ctx* ctx, const double* in, int64_t n, float* out) {
  if (n == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "cast_f64_f32_kernel", cast_f64_f32_kernel, (unsigned)((n + 255) / 256), 256, 0, in, (uint32_t)n, out);
  return DTHIP_OK;
}

__global__ void __launch_bounds__(256) narrow_i64_u32_kernel(const long long* in, uint32_t n, uint32_t* out) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g < n) out[g] = (uint32_t)in[g];
}
int launch_narrow_i64_u32(dthip_ctx* ctx, const long long* in, int64_t n, uint32_t* out) {
  if (n == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "narrow_i64_u32_kernel", narrow_i64_u32_kernel, (unsigned)((n + 255) / 256), 256, 0, in, (uint32_t)n, out);
  return DTHIP_OK;
}

// strided row sample: out[i] = first row of the i-th of m equal pieces (rows for a distinct-count estimate)
__global__ void __launch_bounds__(256) sample_rows_kernel(int32_t* out, uint32_t m, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) out[i] = (int32_t)(((unsigned long long)i * n) / m);
}
int launch_sample_rows(dthip_ctx* ctx, int32_t* out, int64_t m, int64_t n) {
  if (m == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "sample_rows_kernel", sample_rows_kernel, (unsigned)((m + 255) / 256), 256, 0, out, (uint32_t)m, (uint32_t)n);
  return DTHIP_OK;
}

}  // namespace dthip
