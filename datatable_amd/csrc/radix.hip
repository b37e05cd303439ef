// radix.hip -- key transform + digit histograms, and the stable one-sweep
// LSD radix pass (decoupled look-back) for gfx950.
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
//                 stable; tiles chain their per-digit prefixes through
//                 64-bit {flag,count} words (agent-scope relaxed atomics);
//                 keys and payloads are re-ordered through LDS so that global
//                 stores are runs of consecutive addresses per digit.
#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

// ---------------------------------------------------------------------------
// key transform: column value -> unsigned key (sort.cc:689-720 _initB,
// :728-776 _initI, :808-845 _initF), evaluated on the fly from the raw column
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long xform_key(const KeyColDev& c, uint32_t row) {
  typedef unsigned long long u64;
  switch (c.stype) {
    case DTHIP_BOOL: {
      const uint8_t t = static_cast<const uint8_t*>(c.data)[row];
      if (t == 128) return c.na_repl;
      return c.desc ? (u64)(uint8_t)((uint8_t)(128 - t) >> 6) : (u64)(uint8_t)(t + 1);
    }
    case DTHIP_INT8: {
      const int8_t v = static_cast<const int8_t*>(c.data)[row];
      if (v == INT8_MIN) return c.na_repl;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT16: {
      const int16_t v = static_cast<const int16_t*>(c.data)[row];
      if (v == INT16_MIN) return c.na_repl;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT32: {
      const int32_t v = static_cast<const int32_t*>(c.data)[row];
      if (v == INT32_MIN) return c.na_repl;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT64: {
      const long long v = static_cast<const long long*>(c.data)[row];
      if (v == INT64_MIN) return c.na_repl;
      const u64 u = (u64)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_FLOAT32: {
      const uint32_t t = static_cast<const uint32_t*>(c.data)[row];
      if ((t & 0x7F800000u) == 0x7F800000u && (t & 0x007FFFFFu) != 0) return c.na_repl;
      return c.desc ? (u64)(uint32_t)(t ^ (0x7FFFFFFFu & ((t >> 31) - 1u)))
                    : (u64)(uint32_t)(t ^ (0x80000000u | (0u - (t >> 31))));
    }
    default: {  // FLOAT64
      const u64 t = static_cast<const u64*>(c.data)[row];
      if ((t & 0x7FF0000000000000ULL) == 0x7FF0000000000000ULL && (t & 0x000FFFFFFFFFFFFFULL) != 0)
        return c.na_repl;
      return c.desc ? t ^ (0x7FFFFFFFFFFFFFFFULL & ((t >> 63) - 1ULL))
                    : t ^ (0x8000000000000000ULL | (0ULL - (t >> 63)));
    }
  }
}

constexpr int XH_BLOCK = 256;

// Packed transformed key of every row + the digit histogram of every radix
// pass, in one streaming read of the key column(s).
__global__ void __launch_bounds__(XH_BLOCK) xform_hist_kernel(XformArgs a) {
  __shared__ uint32_t lhist[MAX_PASSES * HIST_STRIDE];
  for (int i = threadIdx.x; i < a.npass * HIST_STRIDE; i += XH_BLOCK) lhist[i] = 0;
  __syncthreads();
  const uint32_t stride = gridDim.x * XH_BLOCK;
  for (uint32_t row = blockIdx.x * XH_BLOCK + threadIdx.x; row < a.n; row += stride) {
    const uint32_t src = a.order ? (uint32_t)a.order[row] : row;
    unsigned long long k = 0;
    for (int j = 0; j < a.ncols; j++) k |= xform_key(a.cols[j], src) << a.cols[j].shift;
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
  __syncthreads();
  for (int i = threadIdx.x; i < a.npass * HIST_STRIDE; i += XH_BLOCK) {
    const uint32_t c = lhist[i];
    if (c) atomicAdd(&a.hist[i], c);
  }
}

int launch_xform_hist(dthip_ctx* ctx, const XformArgs& a) {
  if (a.n == 0) return DTHIP_OK;
  long long blocks = ((long long)a.n + XH_BLOCK * 8 - 1) / (XH_BLOCK * 8);
  const long long maxb = (long long)ctx->num_cus * 8;
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
constexpr int RP_BLOCK = 512;
constexpr int RP_ITEMS = 16;
constexpr unsigned long long ST_AGG = 1ULL << 62;   // tile's own count is published
constexpr unsigned long long ST_INCL = 2ULL << 62;  // inclusive prefix up to this tile is published
constexpr unsigned long long ST_VAL = (1ULL << 62) - 1;
constexpr uint32_t SPIN_LIMIT = 1u << 24;

template <typename KeyT>
struct PassArgsT {
  const KeyT* kin; KeyT* kout;
  uint32_t n; int shift; int bits;
  const uint32_t* base;
  unsigned long long* state;
  uint32_t* ticket;
  int* err;
  int iota;
  PayCols pay;
};

// RB   = number of ballot rounds (>= bits of every pass run with this instance)
// P0W  = byte width of payload column 0 when it is prefetched with the keys (0: none / iota)
template <typename KeyT, int RB, int P0W>
__global__ void __launch_bounds__(RP_BLOCK) radix_pass_kernel(PassArgsT<KeyT> a) {
  constexpr int BLOCK = RP_BLOCK, ITEMS = RP_ITEMS, WAVES = BLOCK / 64, TILE = BLOCK * ITEMS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bins = 1 << a.bits;
  const uint32_t dmask = (uint32_t)bins - 1u;
  uint32_t* wh = reinterpret_cast<uint32_t*>(smem);   // [WAVES][bins] per-wave digit counts
  uint32_t* bin_excl = wh + WAVES * bins;              // [bins] tile-local exclusive digit start
  uint32_t* bin_delta = bin_excl + bins;               // [bins] global start - local start
  uint32_t* misc = bin_delta + bins;                   // [16]
  unsigned char* exch = reinterpret_cast<unsigned char*>(misc + 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  if (tid == 0) misc[15] = atomicAdd(a.ticket, 1u);   // tiles start in ticket order => look-back cannot deadlock
  for (int i = tid; i < WAVES * bins; i += BLOCK) wh[i] = 0;
  __syncthreads();
  const uint32_t tile = misc[15];
  const uint32_t tile_base = tile * (uint32_t)TILE;
  const uint32_t nvalid = (a.n - tile_base < (uint32_t)TILE) ? (a.n - tile_base) : (uint32_t)TILE;
  const uint32_t wbase = (uint32_t)wave * 64u * ITEMS + (uint32_t)lane;   // wave-striped: item i at wbase + 64*i

  // ---- load keys (and payload column 0) -----------------------------------
  KeyT key[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t loc = wbase + 64u * i;
    key[i] = (loc < nvalid) ? a.kin[tile_base + loc] : KeyT(0);
  }
  typedef typename std::conditional<P0W == 8, unsigned long long, uint32_t>::type P0T;
  P0T pay0[P0W ? ITEMS : 1];
  if (P0W) {
    const P0T* pin = static_cast<const P0T*>(a.pay.in[0]);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const uint32_t loc = wbase + 64u * i;
      pay0[i] = (loc < nvalid) ? pin[tile_base + loc] : P0T(0);
    }
  }

  // ---- stable rank of every key among equal digits of its wave --------------
  volatile uint32_t* mywh = wh + wave * bins;
  uint32_t pos[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const bool valid = (wbase + 64u * i) < nvalid;
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
    if (valid) prev = mywh[d];
    pos[i] = prev + below;
    if (valid && below == 0) mywh[d] = prev + cnt;
  }
  __syncthreads();

  // ---- per-digit: wave offsets, tile count, look-back -----------------------
  uint32_t tcount = 0;
  if (tid < bins) {
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
      const uint32_t c = wh[w * bins + tid];
      wh[w * bins + tid] = s;
      s += c;
    }
    tcount = s;
  }
  const uint32_t excl = block_excl_scan_u32<BLOCK>(tcount, misc, nullptr);
  if (tid < bins) {
    bin_excl[tid] = excl;
    unsigned long long* st = a.state + (size_t)tile * bins + tid;
    uint32_t prefix = 0;
    if (tile == 0) {
      st_agent_u64(st, ST_INCL | tcount);
    } else {
      st_agent_u64(st, ST_AGG | tcount);
      long long t = (long long)tile - 1;
      uint32_t spins = 0;
      while (true) {
        const unsigned long long v = ld_agent_u64(a.state + (size_t)t * bins + tid);
        const unsigned long long flag = v >> 62;
        if (flag == 0) {
          if (++spins > SPIN_LIMIT) { *a.err = 1; break; }
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        prefix += (uint32_t)(v & ST_VAL);
        if (flag == 2 || t == 0) break;
        t--;
      }
      st_agent_u64(st, ST_INCL | (unsigned long long)(prefix + tcount));
    }
    bin_delta[tid] = a.base[tid] + prefix - excl;
  }
  __syncthreads();

  // ---- keys: registers -> LDS in tile-sorted order -> global ---------------
  KeyT* ek = reinterpret_cast<KeyT*>(exch);
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    if ((wbase + 64u * i) < nvalid) {
      const uint32_t d = (uint32_t)(key[i] >> a.shift) & dmask;
      pos[i] += bin_excl[d] + wh[wave * bins + d];
      ek[pos[i]] = key[i];
    }
  }
  __syncthreads();
  uint32_t gpos[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t slot = (uint32_t)k * BLOCK + tid;
    if (slot < nvalid) {
      const KeyT kk = ek[slot];
      const uint32_t d = (uint32_t)(kk >> a.shift) & dmask;
      gpos[k] = bin_delta[d] + slot;
      a.kout[gpos[k]] = kk;
    }
  }

  // ---- payload columns follow the same permutation --------------------------
  for (int c = 0; c < a.pay.n; c++) {
    __syncthreads();
    if (a.pay.width[c] == 4) {
      uint32_t* e4 = reinterpret_cast<uint32_t*>(exch);
      const uint32_t* pin = static_cast<const uint32_t*>(a.pay.in[c]);
      uint32_t* pout = static_cast<uint32_t*>(a.pay.out[c]);
#pragma unroll
      for (int i = 0; i < ITEMS; i++) {
        const uint32_t loc = wbase + 64u * i;
        if (loc < nvalid) {
          uint32_t v;
          if (c == 0 && a.iota) v = tile_base + loc;
          else if (c == 0 && P0W == 4) v = (uint32_t)pay0[i];
          else v = pin[tile_base + loc];
          e4[pos[i]] = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        const uint32_t slot = (uint32_t)k * BLOCK + tid;
        if (slot < nvalid) pout[gpos[k]] = e4[slot];
      }
    } else {
      unsigned long long* e8 = reinterpret_cast<unsigned long long*>(exch);
      const unsigned long long* pin = static_cast<const unsigned long long*>(a.pay.in[c]);
      unsigned long long* pout = static_cast<unsigned long long*>(a.pay.out[c]);
#pragma unroll
      for (int i = 0; i < ITEMS; i++) {
        const uint32_t loc = wbase + 64u * i;
        if (loc < nvalid) {
          unsigned long long v;
          if (c == 0 && P0W == 8) v = (unsigned long long)pay0[i];
          else v = pin[tile_base + loc];
          e8[pos[i]] = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        const uint32_t slot = (uint32_t)k * BLOCK + tid;
        if (slot < nvalid) pout[gpos[k]] = e8[slot];
      }
    }
  }
}

uint32_t radix_tile_items(int, int) { return RP_BLOCK * RP_ITEMS; }

static size_t pass_lds_bytes(int bits, int key64, int maxw) {
  const int bins = 1 << bits;
  const int w = (key64 ? 8 : 4) > maxw ? (key64 ? 8 : 4) : maxw;
  return (size_t)((RP_BLOCK / 64) * bins + 2 * bins + 16) * 4 + (size_t)RP_BLOCK * RP_ITEMS * w;
}

template <typename KeyT, int RB, int P0W>
static int launch_pass_t(dthip_ctx* ctx, const RadixPass& p, size_t lds) {
  PassArgsT<KeyT> a;
  a.kin = static_cast<const KeyT*>(p.kin); a.kout = static_cast<KeyT*>(p.kout);
  a.n = p.n; a.shift = p.shift; a.bits = p.bits; a.base = p.base; a.state = p.state;
  a.ticket = p.ticket; a.err = p.err; a.iota = p.iota; a.pay = p.pay;
  auto kfn = radix_pass_kernel<KeyT, RB, P0W>;
  static bool attr_set = false;
  if (!attr_set) {
    DTHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    attr_set = true;
  }
  const uint32_t tile = RP_BLOCK * RP_ITEMS;
  const uint32_t ntiles = (p.n + tile - 1) / tile;
  DTHIP_LAUNCH(ctx, "radix_pass_kernel", kfn, ntiles, RP_BLOCK, lds, a);
  return DTHIP_OK;
}

int launch_radix_pass(dthip_ctx* ctx, const RadixPass& p) {
  if (p.n == 0) return DTHIP_OK;
  if (p.bits < 1 || p.bits > 9) { set_error("radix pass: bad digit width %d", p.bits); return DTHIP_EINVAL; }
  int maxw = 0;
  for (int c = 0; c < p.pay.n; c++) maxw = p.pay.width[c] > maxw ? p.pay.width[c] : maxw;
  const size_t lds = pass_lds_bytes(p.bits, p.key64, maxw);
  const int p0w = (p.pay.n > 0 && !p.iota) ? p.pay.width[0] : 0;
  const bool rb9 = p.bits > 8;
#define DISPATCH(KT)                                                            \
  do {                                                                          \
    if (rb9) {                                                                  \
      if (p0w == 8) return launch_pass_t<KT, 9, 8>(ctx, p, lds);                \
      if (p0w == 4) return launch_pass_t<KT, 9, 4>(ctx, p, lds);                \
      return launch_pass_t<KT, 9, 0>(ctx, p, lds);                              \
    } else {                                                                    \
      if (p0w == 8) return launch_pass_t<KT, 8, 8>(ctx, p, lds);                \
      if (p0w == 4) return launch_pass_t<KT, 8, 4>(ctx, p, lds);                \
      return launch_pass_t<KT, 8, 0>(ctx, p, lds);                              \
    }                                                                           \
  } while (0)
  if (p.key64) DISPATCH(unsigned long long);
  else DISPATCH(uint32_t);
#undef DISPATCH
  return DTHIP_OK;
}

}  // namespace dthip
