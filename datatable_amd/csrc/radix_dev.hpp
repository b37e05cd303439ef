// radix_dev.hpp -- device pieces of the stable radix pass shared by radix.hip and tlsort.hip: tile geometry, vector
// typedefs, streaming-hint macros, the run-wise store, and one stable ranking round over a workgroup's rows.
#pragma once
#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

#ifndef DTHIP_RP_BLOCK
#define DTHIP_RP_BLOCK 512
#endif
#ifndef DTHIP_RP_ITEMS
#define DTHIP_RP_ITEMS 16
#endif
#ifndef DTHIP_RP_WAVES
#define DTHIP_RP_WAVES 4
#endif
constexpr int RP_BLOCK = DTHIP_RP_BLOCK, RP_ITEMS = DTHIP_RP_ITEMS, RP_TILE = RP_BLOCK * RP_ITEMS;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-B access

// build-time experiments (A/B through DTHIP_LIB): streaming hints on the pass' loads / stores
#ifdef DTHIP_RP_NT
#define RP_LD(p) __builtin_nontemporal_load(p)
#else
#define RP_LD(p) (*(p))
#endif
#ifdef DTHIP_RP_NTS
#define RP_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define RP_ST(p, v) (*(p) = (v))
#endif

// store 4 values of tile-sorted slots s0..s0+3 to their global positions: one 16-B
// (or two, for 8-byte elements) store when the four land on consecutive addresses
template <typename T>
__device__ __forceinline__ void store_group4(T* __restrict__ out, const uint32_t (&gp)[4], const T (&v)[4], uint32_t nv) {
  const bool run = nv == 4 && gp[1] == gp[0] + 1 && gp[2] == gp[0] + 2 && gp[3] == gp[0] + 3;
  if (run) {
    if (sizeof(T) == 4) {
      u32x4 w;
      w.x = (uint32_t)v[0]; w.y = (uint32_t)v[1]; w.z = (uint32_t)v[2]; w.w = (uint32_t)v[3];
      RP_ST(reinterpret_cast<u32x4_u*>(out + gp[0]), w);
    } else {
      u32x4 w0, w1;
      w0.x = (uint32_t)v[0]; w0.y = (uint32_t)((unsigned long long)v[0] >> 32);
      w0.z = (uint32_t)v[1]; w0.w = (uint32_t)((unsigned long long)v[1] >> 32);
      w1.x = (uint32_t)v[2]; w1.y = (uint32_t)((unsigned long long)v[2] >> 32);
      w1.z = (uint32_t)v[3]; w1.w = (uint32_t)((unsigned long long)v[3] >> 32);
      u32x4_u* o = reinterpret_cast<u32x4_u*>(out + gp[0]);
      RP_ST(&o[0], w0); RP_ST(&o[1], w1);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) if ((uint32_t)j < nv) RP_ST(&out[gp[j]], v[j]);
  }
}

// One STABLE ranking round over the workgroup's rows (the lane-mask ranking of the pass kernel as a function): on entry
// pos is don't-care, on exit pos[i] = number of rows of the tile that precede row i in the order (digit, current row
// order).  wh / bin_excl / misc / exch as in the pass kernel; contains barriers, all threads must call.
// nbal > 0: the lanes of an item that share a digit are found with nbal ballot rounds instead of the LDS lane masks --
// for a digit of FEW values (the bucket number inside a window: ~3) dozens of lanes would pile their ds_or onto one
// address and serialise (measured: the windowed final level 5.8 ms instead of 3.8).
// (the digit of item i is dig(i): recomputed where it is needed instead of held in ITEMS more registers -- the two-round
// final level was register-starved: 6.4 ms for its first round alone against 3.85 for the one-round kernel)
template <int BLOCK, int ITEMS, int RBMAX, typename DigF>
__device__ __forceinline__ void rank_round(DigF dig, uint32_t vmask, int bins, uint16_t* wh, uint32_t* bin_excl,
                                           uint32_t* misc, unsigned char* exch, uint32_t slice_bytes, uint32_t (&pos)[ITEMS], int nbal = 0,
                                           uint32_t* total = nullptr /* rows ranked by the whole workgroup */) {
  constexpr int WAVES = BLOCK / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < WAVES * bins / 2; i += BLOCK) reinterpret_cast<uint32_t*>(wh)[i] = 0;
  unsigned long long* mk = reinterpret_cast<unsigned long long*>(exch + (size_t)wave * slice_bytes);
  for (int b = lane; b < bins; b += 64) mk[b] = 0ULL;
  __syncthreads();
  uint16_t* mywh = wh + wave * bins;
  const unsigned long long mybit = 1ULL << lane;
  if (nbal > 0) {
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const bool valid = (vmask >> i) & 1u;
      const uint32_t d = dig(i);
      unsigned long long m = __ballot(valid);
      for (int b = 0; b < nbal; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      const uint32_t below = mbcnt64(m);
      uint32_t prev = 0;
      __builtin_amdgcn_wave_barrier();
      if (valid) prev = __hip_atomic_load(&mywh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      pos[i] = prev + below;
      if (valid && below == 0) __hip_atomic_store(&mywh[d], (uint16_t)(prev + (uint32_t)__popcll(m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  } else
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    pos[i] = 0;
    __builtin_amdgcn_wave_barrier();
    if ((vmask >> i) & 1u) {
      const uint32_t d = dig(i);
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
  __syncthreads();
  constexpr int KB = ((1 << RBMAX) + BLOCK - 1) / BLOCK;
  uint32_t tc[KB], tsum = 0;
#pragma unroll
  for (int k = 0; k < KB; k++) {
    const int b = tid * KB + k;
    tc[k] = 0;
    if (b < bins) {
      uint32_t sum = 0;
#pragma unroll
      for (int w = 0; w < WAVES; w++) {
        const uint32_t c = wh[w * bins + b];
        wh[w * bins + b] = (uint16_t)sum;
        sum += c;
      }
      tc[k] = sum;
    }
    tsum += tc[k];
  }
  uint32_t excl = block_excl_scan_u32<BLOCK>(tsum, misc, total);
#pragma unroll
  for (int k = 0; k < KB; k++) {
    const int b = tid * KB + k;
    if (b < bins) { bin_excl[b] = excl; excl += tc[k]; }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; i++)
    if ((vmask >> i) & 1u) { const uint32_t d = dig(i); pos[i] += bin_excl[d] + wh[wave * bins + d]; }
}

// ---- tile-local layout of a first sort level (tlsort.hip) and the loader that reads it back ----------------------------
// Level 1 of the fused filter -> group-rows path writes every tile's rows, ordered by their top digit b, into the tile's OWN
// row range [t * T1, t * T1 + count) -- sequential writes, no histogram pass, no run positions -- together with a directory
// of 16-bit positions: dirT[b][t] = first row of digit b inside tile t (transposed: one row per digit; row `bins` holds
// the tile's row count).  Bucket b of the next level is then the concatenation, over the tiles, of the segments
// [dirT[b][t], dirT[b + 1][t]); cc[b][tb] = rows of bucket b in the tiles before tile block tb (64 tiles per block).
// A workgroup that takes the rows [v0, v0 + nrows) of that virtual sequence builds src[v] = global row of the v-th of
// them: every wave takes tile blocks round robin (lane = tile: two coalesced directory reads, one wave scan of the
// segment lengths), short segments are written by their lane, long ones (clustered keys) by the whole wave.
template <int BLOCK>
__device__ __forceinline__ void tl_build_src(uint32_t* src, const uint16_t* __restrict__ dirT, uint32_t dstride,
                                             const uint32_t* __restrict__ cc, uint32_t ntb, uint32_t ntiles1, uint32_t T1,
                                             uint32_t b, uint32_t v0, uint32_t nrows) {
  constexpr int WAVES = BLOCK / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t* ccb = cc + (size_t)b * ntb;
  uint32_t lo = 0, hi = ntb;                       // first block whose prefix exceeds v0; the block before it holds row v0
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (ccb[mid] <= v0) lo = mid + 1; else hi = mid; }
  const uint32_t tb0 = lo ? lo - 1 : 0;
  const uint16_t* r0 = dirT + (size_t)b * dstride;
  const uint16_t* r1 = r0 + dstride;
  const uint32_t vend = v0 + nrows;
  for (uint32_t tb = tb0 + (uint32_t)wave; tb < ntb; tb += WAVES) {
    const uint32_t cb = ccb[tb];
    if (cb >= vend) break;                         // (uniform in the wave)
    const uint32_t t = tb * 64u + (uint32_t)lane;
    uint32_t st = 0, len = 0;
    if (t < ntiles1) { st = r0[t]; len = (uint32_t)r1[t] - st; }
    const uint32_t vs = cb + wave_incl_scan_u32(len) - len;        // place of the segment's first row inside the bucket
    const uint32_t g0 = t * T1 + st;
    uint32_t jlo = vs < v0 ? v0 - vs : 0u;
    uint32_t jhi = vs >= vend ? 0u : (vend - vs < len ? vend - vs : len);
    if (jlo > jhi) jlo = jhi;
    const bool big = jhi - jlo > 32u;
    if (!big) for (uint32_t j = jlo; j < jhi; j++) src[vs + j - v0] = g0 + j;
    unsigned long long lm = __ballot(big);
    while (lm) {
      const int l = __ffsll((long long)lm) - 1;
      lm &= lm - 1ULL;
      const uint32_t G0 = (uint32_t)__shfl((int)g0, l, 64), VS = (uint32_t)__shfl((int)vs, l, 64);
      const uint32_t JLO = (uint32_t)__shfl((int)jlo, l, 64), JHI = (uint32_t)__shfl((int)jhi, l, 64);
      for (uint32_t j = JLO + (uint32_t)lane; j < JHI; j += 64u) src[VS + j - v0] = G0 + j;
    }
  }
}

// The same for the FINAL level over a second tile-local level (RadixPass::g2_dirT): the rows [row0, row0 + nrows) of the
// final order are whole final buckets c0, c0 + 1, ... (a window); bucket c = (parent b1, digit d2) is the concatenation over
// the parent's tiles p of the segments [dirT2[d2][p], dirT2[d2 + 1][p]) at base tdesc[4 p].  A parent has ~n / 2^s1 / tile
// tiles (two 64-tile blocks for config 5): every wave walks all of a bucket's blocks to keep the running offset and writes
// the rows of the blocks that are its share.
template <int BLOCK>
__device__ __forceinline__ void tl_build_src_final(uint32_t* src, const uint16_t* __restrict__ dirT2, uint32_t dstride2,
                                                   const uint32_t* __restrict__ tdesc, const uint32_t* __restrict__ pfirst,
                                                   const uint32_t* __restrict__ fstart, int s2bits, uint32_t nbk, uint32_t c0,
                                                   uint32_t row0, uint32_t nrows) {
  constexpr int WAVES = BLOCK / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t vend = row0 + nrows, dmask2 = (1u << s2bits) - 1u;
  for (uint32_t c = c0; c < nbk; c++) {
    const uint32_t fs = fstart[c];
    if (fs >= vend) break;
    const uint32_t fe = fstart[c + 1];
    if (fe <= row0 || fe == fs) continue;            // before the window / empty
    const uint32_t b1 = c >> s2bits, d2 = c & dmask2;
    const uint32_t p0 = pfirst[b1], p1 = pfirst[b1 + 1];
    const uint16_t* r0 = dirT2 + (size_t)d2 * dstride2;
    const uint16_t* r1 = r0 + dstride2;
    uint32_t run = fs;
    uint32_t blk = 0;
    for (uint32_t pb = p0; pb < p1; pb += 64u, blk++) {
      const uint32_t p = pb + (uint32_t)lane;
      uint32_t st = 0, len = 0, base = 0;
      if (p < p1) { st = r0[p]; len = (uint32_t)r1[p] - st; base = tdesc[4 * p]; }
      const uint32_t incl = wave_incl_scan_u32(len);
      const uint32_t tot = (uint32_t)__shfl((int)incl, 63, 64);
      if ((blk % (uint32_t)WAVES) == (uint32_t)wave) {
        const uint32_t vs = run + incl - len;
        const uint32_t g0 = base + st;
        uint32_t jlo = vs < row0 ? row0 - vs : 0u;
        uint32_t jhi = vs >= vend ? 0u : (vend - vs < len ? vend - vs : len);
        if (jlo > jhi) jlo = jhi;
        const bool big = jhi - jlo > 32u;
        if (!big) for (uint32_t j = jlo; j < jhi; j++) src[vs + j - row0] = g0 + j;
        unsigned long long lm = __ballot(big);
        while (lm) {
          const int l = __ffsll((long long)lm) - 1;
          lm &= lm - 1ULL;
          const uint32_t G0 = (uint32_t)__shfl((int)g0, l, 64), VS = (uint32_t)__shfl((int)vs, l, 64);
          const uint32_t JLO = (uint32_t)__shfl((int)jlo, l, 64), JHI = (uint32_t)__shfl((int)jhi, l, 64);
          for (uint32_t j = JLO + (uint32_t)lane; j < JHI; j += 64u) src[VS + j - row0] = G0 + j;
        }
      }
      run += tot;
    }
  }
}

}  // namespace dthip
