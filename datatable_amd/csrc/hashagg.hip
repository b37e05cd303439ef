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
__device__ __forceinline__ void hash_tab_probe0(u64 x, uint32_t C, uint32_t& p, uint32_t& step) {
#ifdef DTHIP_HASH_MIX64            // (A/B flavour: `make var NAME=mix64 VFILE=bucket FLAGS=-DDTHIP_HASH_MIX64`)
  const u64 h2 = mix64(x ^ 0x9E3779B97F4A7C15ULL);
  p = (uint32_t)(((h2 & 0xFFFFFFFFULL) * (u64)C) >> 32);
  step = 1u + (uint32_t)(((h2 >> 32) * (u64)(C - 1)) >> 32);
#else
  uint32_t h = (uint32_t)x * 0xC2B2AE3Du + (uint32_t)(x >> 32) * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
  p = __umulhi(h, C);
  step = 1u + __umulhi(h * 0x297A2D39u + 0x165667B1u, C - 1u);
#endif
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

// CFLAGS >= 0: the accumulator set is a compile-time constant (the common sum / sum+count shapes: the
// eight unrolled inserts then carry no per-row flag tests); -1: taken from the arguments
template <typename VT, int CFLAGS>
__global__ void __launch_bounds__(TA_BLOCK) hash_agg_kernel(HashAggDev a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x >= *a.nitems) return;
  const WorkItem it = a.items[blockIdx.x];
  const int tid = threadIdx.x;
  const int flags = CFLAGS >= 0 ? CFLAGS : a.flags;
  const uint32_t C = a.C, S = C + 1;                 // entry C is reserved for the key that equals HASH_EMPTY
  u64* hk = reinterpret_cast<u64*>(smem);            // [S] keys
  const LdsTab t = carve_tab(smem + (size_t)S * 8, S, flags);
  __shared__ uint32_t s_misc[20];
  hash_tab_init(hk, t, flags, S, s_misc, tid);
  __syncthreads();
  const VT* __restrict__ val = static_cast<const VT*>(a.val);
  const bool hasval = (flags & (ACC_SUM | ACC_MIN | ACC_MAX | ACC_VCNT | ACC_FSUM)) != 0;
  bool full = false;
  auto insert = [&](u64 x, VT v) {
    uint32_t p;
    if (x == HASH_EMPTY) {
      p = C;
      s_misc[16] = 1;
    } else {
      // second hash (independent of the bits that chose the bucket): start by Lemire reduction to [0, C),
      // DOUBLE hashing step in [1, C) -- C is prime, so every step visits all entries.  Linear probing
      // clusters: at load 0.6 the slowest of a wave's 64 lanes needed ~20 probes, and a wave waits for it.
      uint32_t step;
      hash_tab_probe0(x, C, p, step);
      uint32_t probes = 0;
      // most rows find their key already in the table (rows >> keys): look with a plain DS read first,
      // compare-and-swap only into an empty entry
      while (true) {
        u64 cur = __hip_atomic_load(&hk[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == HASH_EMPTY) cur = atomicCAS(&hk[p], HASH_EMPTY, x);      // returns what was there: EMPTY = claimed
        if (cur == HASH_EMPTY || cur == x) break;
        p += step; if (p >= C) p -= C;
        if (++probes >= C) { full = true; return; }
      }
    }
    acc_row<VT, false>(t, flags, p, v);
  };
  {
    // 8 consecutive rows per thread and iteration: all their 16-byte loads are in flight before the first
    // probe (a probe chain is a dependent sequence of DS operations; one row at a time is latency bound)
    uint32_t a0 = (it.begin + 7u) & ~7u; if (a0 > it.end) a0 = it.end;
    uint32_t a1 = it.end & ~7u; if (a1 < a0) a1 = a0;
    const uint32_t nh = a0 - it.begin, ntl = it.end - a1;
    if ((uint32_t)tid < nh) { const uint32_t row = it.begin + tid; insert(a.xs[row], hasval ? val[row] : VT(0)); }
    else if ((uint32_t)tid >= 64u && (uint32_t)tid - 64u < ntl) { const uint32_t row = a1 + ((uint32_t)tid - 64u); insert(a.xs[row], hasval ? val[row] : VT(0)); }
    const uint32_t ngr = (a1 - a0) >> 3;
    for (uint32_t g = tid; g < ngr; g += TA_BLOCK) {
      const uint32_t row = a0 + g * 8u;
      bu32x4 kw[4];
      const bu32x4* kp = reinterpret_cast<const bu32x4*>(a.xs + row);
#pragma unroll
      for (int j = 0; j < 4; j++) kw[j] = kp[j];
      VT v[8];
      if (hasval) {
        constexpr int NV = (int)sizeof(VT) / 2;
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
      // Every lane walks through ITS eight rows at its own pace: one probe step per loop iteration on the
      // lane's current row, and a lane that resolved its row accumulates and moves on to its next one.
      // Row-by-row in lock-step, each row costs the wave the probe count of its slowest lane (~9 at load
      // 0.6); this way a wave runs about max-over-lanes of the SUM of a lane's probes (~18 for 8 rows, not 72).
      u64 qx[8]; VT qv[8];
      {
        const u64* kx8 = reinterpret_cast<const u64*>(kw);
#pragma unroll
        for (int j = 0; j < 8; j++) { qx[j] = kx8[j]; qv[j] = v[j]; }
      }
      int left = 8;
      uint32_t p = 0, step = 0, probes = 0;
      bool fresh = true;
      while (__any(left > 0)) {
        if (left > 0) {
          const u64 x = qx[0];
          bool resolved = false;
          if (fresh) {
            fresh = false; probes = 0;
            if (x == HASH_EMPTY) { p = C; s_misc[16] = 1; resolved = true; }
            else {
              hash_tab_probe0(x, C, p, step);
            }
          }
          if (!resolved) {
            u64 cur = __hip_atomic_load(&hk[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == HASH_EMPTY) cur = atomicCAS(&hk[p], HASH_EMPTY, x);
            if (cur == HASH_EMPTY || cur == x) resolved = true;
            else {
              p += step; if (p >= C) p -= C;
              if (++probes >= C) { full = true; left = 0; }
            }
          }
          if (resolved) {
            // (acc_row's wave-uniform fast path is off here: lanes are on different rows)
            acc_row<VT, false>(t, flags, p, qv[0]);
#pragma unroll
            for (int j = 0; j < 7; j++) { qx[j] = qx[j + 1]; qv[j] = qv[j + 1]; }
            left--;
            fresh = true;
          }
        }
      }
    }
  }
  if (__ballot(full) && (tid & 63) == 0) atomicOr(a.overflow, 1u);
  __syncthreads();
  hash_tab_flush(hk, t, flags, C, s_misc, a.o_key, a.o_tab, a.out_n, a.out_cap, a.overflow, tid);
}

size_t hash_agg_entry_bytes(int flags) { return 8 + table_agg_slot_bytes(flags); }

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
  const size_t lds = (size_t)(a.C + 1) * hash_agg_entry_bytes(a.flags) + 32;
  if (lds > 160 * 1024 - 512) { set_error("hash_agg: table of %zu bytes exceeds LDS", lds); return DTHIP_EINVAL; }
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

// (round 6, measured and dropped: the same tables fed from a TILE-LOCAL partition -- 16384-row tiles, 8-row segments per bucket
// found through the transposed directory, 8 lanes per segment, every lane walking up to eight rows of eight segments.  The
// partition took 5.9 ms instead of 8.6 + 1.6 for histogram + exact positions, but this kernel 13.5 instead of 5.8: it is
// bound by the VALU work of its walk loop, and segments of 8 +- 3 rows fill the lanes' queues unevenly -- three rounds per
// 64 tiles for what eight full slots do in one.  22.1 ms against 18.5: profiles/r06_hash_tl_ab.txt, r06_hash_tabhash_ab.txt)

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
