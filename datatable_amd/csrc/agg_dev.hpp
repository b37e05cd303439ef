// agg_dev.hpp -- device pieces shared by the LDS aggregation tables of bucket.hip (slot tables) and hashagg.hip (hash tables):
// value traits, order-preserving images for min / max, the table carved out of dynamic LDS, one row into its accumulators.
#pragma once
#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

typedef uint32_t bu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t bu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

constexpr int TA_BLOCK = 1024;

template <typename VT> struct ValTraits;
template <> struct ValTraits<double> {
  static constexpr bool is_float = true;
  static __device__ __forceinline__ bool isna(double v) { return v != v; }
};
template <> struct ValTraits<float> {
  static constexpr bool is_float = true;
  static __device__ __forceinline__ bool isna(float v) { return v != v; }
};
template <> struct ValTraits<int32_t> {
  static constexpr bool is_float = false;
  static __device__ __forceinline__ bool isna(int32_t v) { return v == INT32_MIN; }
};
template <> struct ValTraits<long long> {
  static constexpr bool is_float = false;
  static __device__ __forceinline__ bool isna(long long v) { return v == INT64_MIN; }
};

// order-preserving unsigned images (the reference's float key transform, sort.cc:808-845)
__device__ __forceinline__ u64 sortable_f64(double d) {
  const u64 t = (u64)__double_as_longlong(d);
  return t ^ (0x8000000000000000ULL | (0ULL - (t >> 63)));
}
__device__ __forceinline__ double unsortable_f64(u64 k) {
  const u64 t = (k & 0x8000000000000000ULL) ? (k ^ 0x8000000000000000ULL) : ~k;
  return __longlong_as_double((long long)t);
}
__device__ __forceinline__ u64 sortable_i64(long long v) { return (u64)v ^ 0x8000000000000000ULL; }

struct LdsTab {
  u64* sum; u64* mn; u64* mx; double* fsum; uint32_t* cnt; uint32_t* vcnt; uint32_t* pres;
  uint32_t* g_na;        // ACC_CHKNA: this bucket's slice of AggTable::nacnt (GLOBAL memory): NA rows of a column guessed NA-free
};

// accumulators (or checks) that look at the VALUE of a row; without any of them the value column is not even read
constexpr int ACC_VALUE_MASK = ACC_SUM | ACC_MIN | ACC_MAX | ACC_VCNT | ACC_FSUM | ACC_CHKNA;

__device__ __forceinline__ LdsTab carve_tab(unsigned char* smem, uint32_t S, int flags) {
  LdsTab t;
  unsigned char* p = smem;
  t.sum = reinterpret_cast<u64*>(p); if (flags & ACC_SUM) p += (size_t)S * 8;
  t.mn = reinterpret_cast<u64*>(p); if (flags & ACC_MIN) p += (size_t)S * 8;
  t.mx = reinterpret_cast<u64*>(p); if (flags & ACC_MAX) p += (size_t)S * 8;
  t.fsum = reinterpret_cast<double*>(p); if (flags & ACC_FSUM) p += (size_t)S * 8;
  t.cnt = reinterpret_cast<uint32_t*>(p); if (flags & ACC_CNT) p += (size_t)S * 4;
  t.vcnt = reinterpret_cast<uint32_t*>(p); if (flags & ACC_VCNT) p += (size_t)S * 4;
  t.pres = reinterpret_cast<uint32_t*>(p); if (flags & ACC_PRES) p += (size_t)((S + 31) / 32) * 4;
  t.g_na = nullptr;
  return t;
}

__device__ __forceinline__ void lds_fadd(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ u64 wave_reduce_u64(u64 v, int op) {     // op 0 add, 1 min, 2 max; all 64 lanes active
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 w = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
    v = op == 0 ? v + w : op == 1 ? (w < v ? w : v) : (w > v ? w : v);
  }
  return v;
}
__device__ __forceinline__ double wave_reduce_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 b = (u64)__double_as_longlong(v);
    const u64 w = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)b, o, 64);
    v += __longlong_as_double((long long)w);
  }
  return v;
}

// Many lanes of a (fully active) wave address ONE slot -- sorted / constant keys (all 64), a hot key inside its bucket (most
// of them): the lanes of `in` are reduced in registers and one lane updates the table (64 same-address DS atomics would
// serialise); the other lanes contribute the identity and go the ordinary way afterwards.  All 64 lanes call.
template <typename VT>
__device__ __forceinline__ void acc_wave_masked(const LdsTab& t, int flags, uint32_t slot, VT v, bool in) {
  const bool lead = (threadIdx.x & 63) == 0;
  const uint32_t nin = (uint32_t)__popcll(__ballot(in));
  if (lead && (flags & ACC_CNT)) atomicAdd(&t.cnt[slot], nin);
  if (lead && (flags & ACC_PRES)) atomicOr(&t.pres[slot >> 5], 1u << (slot & 31));
  if (!(flags & (ACC_VALUE_MASK))) return;
  const bool valid = (flags & ACC_NONA) || !ValTraits<VT>::isna(v);
  const bool ok = in && valid;
  const uint32_t nok = (uint32_t)__popcll(__ballot(ok));
  if ((flags & ACC_CHKNA) && in && !valid) atomicAdd(&t.g_na[slot], 1u);      // (rare: the column was guessed NA-free)
  if (nok == 0) return;
  if (lead && (flags & ACC_VCNT)) atomicAdd(&t.vcnt[slot], nok);
  if (ValTraits<VT>::is_float) {
    const double d = (double)v;
    if (flags & ACC_SUM) { const double r = wave_reduce_f64(ok ? d : 0.0); if (lead) lds_fadd(reinterpret_cast<double*>(&t.sum[slot]), r); }
    if (flags & ACC_MIN) { const u64 r = wave_reduce_u64(ok ? sortable_f64(d) : ~0ULL, 1); if (lead) atomicMin(&t.mn[slot], r); }
    if (flags & ACC_MAX) { const u64 r = wave_reduce_u64(ok ? sortable_f64(d) : 0ULL, 2); if (lead) atomicMax(&t.mx[slot], r); }
  } else {
    const long long iv = (long long)v;
    if (flags & ACC_SUM) { const u64 r = wave_reduce_u64(ok ? (u64)iv : 0ULL, 0); if (lead) atomicAdd(&t.sum[slot], r); }
    if (flags & ACC_FSUM) { const double r = wave_reduce_f64(ok ? (double)iv : 0.0); if (lead) lds_fadd(&t.fsum[slot], r); }
    if (flags & ACC_MIN) { const u64 r = wave_reduce_u64(ok ? sortable_i64(iv) : ~0ULL, 1); if (lead) atomicMin(&t.mn[slot], r); }
    if (flags & ACC_MAX) { const u64 r = wave_reduce_u64(ok ? sortable_i64(iv) : 0ULL, 2); if (lead) atomicMax(&t.mx[slot], r); }
  }
}

// UNI (clustered keys: the launch's instance) or `hot` (this work item's bucket holds a hot key -- wave-uniform, decided per
// item): when at least 16 lanes of the fully active wave share the first lane's slot they are combined in registers
// (acc_wave_masked); a -0.0 sum identity aside, the table ends up as if every lane had issued its own atomics
template <typename VT, bool UNI>
__device__ __forceinline__ void acc_row(const LdsTab& t, int flags, uint32_t slot, VT v, bool hot = false) {
  if ((UNI || hot) && __ballot(1) == ~0ULL) {
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane(slot);
    const bool in = slot == s0;
    if (__popcll(__ballot(in)) >= 16) {
      acc_wave_masked<VT>(t, flags, s0, v, in);
      if (in) return;
    }
  }
  if (flags & ACC_CNT) atomicAdd(&t.cnt[slot], 1u);
  if (flags & ACC_PRES) atomicOr(&t.pres[slot >> 5], 1u << (slot & 31));
  if (flags & (ACC_VALUE_MASK)) {
    if ((flags & ACC_NONA) || !ValTraits<VT>::isna(v)) {
      if (flags & ACC_VCNT) atomicAdd(&t.vcnt[slot], 1u);
      if (ValTraits<VT>::is_float) {
        const double d = (double)v;
        if (flags & ACC_SUM) lds_fadd(reinterpret_cast<double*>(&t.sum[slot]), d);
        if (flags & (ACC_MIN | ACC_MAX)) {
          const u64 k = sortable_f64(d);
          if (flags & ACC_MIN) atomicMin(&t.mn[slot], k);
          if (flags & ACC_MAX) atomicMax(&t.mx[slot], k);
        }
      } else {
        const long long iv = (long long)v;
        if (flags & ACC_SUM) atomicAdd(&t.sum[slot], (u64)iv);
        if (flags & ACC_FSUM) lds_fadd(&t.fsum[slot], (double)iv);
        if (flags & (ACC_MIN | ACC_MAX)) {
          const u64 k = sortable_i64(iv);
          if (flags & ACC_MIN) atomicMin(&t.mn[slot], k);
          if (flags & ACC_MAX) atomicMax(&t.mx[slot], k);
        }
      }
    } else if (flags & ACC_CHKNA) {
      // the guess "this column holds no NA" was wrong for this row: it is skipped like any NA and counted apart -- no valid
      // counter per slot in LDS, and (round 6) no second aggregation either: valid rows = rows - these
      atomicAdd(&t.g_na[slot], 1u);
    }
  }
}


}  // namespace dthip
