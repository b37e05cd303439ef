// device_utils.hpp -- wave64 / workgroup primitives for gfx950 (CDNA4).
// Wavefront width is 64 everywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dthip {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// number of set bits of `m` strictly below the calling lane (v_mbcnt_lo/hi)
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// wave64 inclusive prefix sum with DPP row shifts + row broadcasts (gfx9 encodings): no lane-index
// registers and no LDS traffic, unlike a ds_bpermute ladder
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
  return (uint32_t)x;
}

__device__ __forceinline__ uint32_t wave_reduce_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Workgroup exclusive scan of one uint32 per thread.  `scratch` needs
// BLOCK/64 words of LDS.  Contains two __syncthreads(); all threads must call.
template <int BLOCK>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* scratch, uint32_t* total) {
  constexpr int WAVES = BLOCK / 64;
  const int lane = lane_id(), wave = wave_id();
  const uint32_t incl = wave_incl_scan_u32(v);
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < WAVES; w++) {
    const uint32_t s = scratch[w];
    if (w < wave) off += s;
    tot += s;
  }
  __syncthreads();
  if (total) *total = tot;
  return off + incl - v;
}

// 64-bit shuffle helpers
__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int o) {
  uint32_t lo = __shfl_up((uint32_t)v, o, 64), hi = __shfl_up((uint32_t)(v >> 32), o, 64);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double shfl_up_f64(double v, int o) {
  return __longlong_as_double((long long)shfl_up_u64((unsigned long long)__double_as_longlong(v), o));
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
  return ((unsigned long long)hi << 32) | lo;
}

// LDS counter increment that returns the arrival rank; a wave whose lanes all address one bin (sorted /
// constant keys: 64 same-address DS atomics would serialise) is counted by one lane and ranked with v_mbcnt
__device__ __forceinline__ uint32_t lds_count_rank(uint32_t* cnt, uint32_t d) {
  const uint64_t act = __ballot(1);
  const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
  if (__ballot(d == d0) == act) {
    const uint32_t r = mbcnt64(act);
    uint32_t base = 0;
    if (r == 0) base = atomicAdd(&cnt[d0], (uint32_t)__popcll(act));
    return __builtin_amdgcn_readfirstlane(base) + r;
  }
  return atomicAdd(&cnt[d], 1u);
}

// LDS counter increment for keys that arrive in runs (sorted / clustered / constant columns), where
// plain DS atomics would pile 64 lanes onto one address and serialise.  The wave's lanes are peeled
// cluster by cluster: all lanes that share the first remaining lane's bin are counted by ONE ds_add of
// the cluster's population and ranked with a masked v_mbcnt.  A first cluster smaller than 8 lanes means
// the bins are scattered (random keys): everybody falls back to an individual atomic at once, so the
// test costs one ballot.  Returns the arrival rank inside the bin (unordered between clusters).
__device__ __forceinline__ uint32_t lds_count_peel_rank(uint32_t* cnt, uint32_t d) {
  const int lane = lane_id();
  const uint64_t lt = (1ULL << lane) - 1ULL;
  uint64_t rem = __ballot(1);
  uint32_t rank = 0;
  bool done = false;
#pragma unroll 1
  for (int it = 0; it < 4 && rem; it++) {
    const int leader = __ffsll((long long)rem) - 1;
    const uint32_t d0 = (uint32_t)__shfl((int)d, leader, 64);
    const uint64_t m = __ballot(!done && d == d0) & rem;
    const uint32_t pop = (uint32_t)__popcll(m);
    if (it == 0 && pop < 8) break;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&cnt[d0], pop);
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (!done && d == d0) { rank = base + (uint32_t)__popcll(m & lt); done = true; }
    rem &= ~m;
  }
  if (!done) rank = atomicAdd(&cnt[d], 1u);
  return rank;
}

}  // namespace dthip
