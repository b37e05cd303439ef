// groupwise.hip -- operators that share the Groupby of the hot path:
//   sd, cov, corr            expr/head_reduce_unary.cc:194-216, head_reduce_binary.cc:113-198
//   median, nunique          expr/head_reduce_unary.cc:377-387,424-470
//   cumsum/cumprod/cummin/cummax (+reverse)   column/cumsumprod.h:52-92, column/cumminmax.h:48-98
//   cumcount/ngroup          column/cumcountngroup.h:55-72
//   fillna(reverse)          expr/fexpr_fillna.cc:85-117 (the same scan: the state is the last valid value)
//
// The reference runs one sequential loop per group (parallel over groups, so one huge group is
// serial).  Here every operator that is an associative fold is one *segmented scan* over the
// grouped row order, independent of the group-size distribution:
//
//   gw_summary_kernel   tile of 2048 positions -> state of the tile's trailing open segment
//   gw_carry_kernel     segmented exclusive scan over the tile states (single workgroup)
//   gw_apply_kernel     re-scan the tile with its carry; SCAN writes every position's running
//                       value, REDUCE emits the state at each group's last row
//
// and a policy P supplies the state and the fold:
//   MomP<1>/MomP<2>  (count, mean, M2[, mean2, M2', C12]) merged with Chan's pairwise update --
//                    the parallel form of the reference's Welford loops; inputs pre-cast to f64
//   CumP<A,OP>       running sum / product / min / max in A = int64 or double
// Segment heads come from the same 1-bit-per-position bitmap the reducers use; `reverse` mirrors
// the position order.  Float results are re-associated (tests: 1e-6 relative; float32 5e-5),
// integer results are exact (sums/products wrap like the reference's int64 arithmetic).
//
// median / nunique need the group's distinct values in order with their multiplicities: the caller
// runs the library's own fused groupby-aggregate on (group id, value) with count() -- sort-free when
// that composite key is dense or has few distinct values -- and two small kernels read the pairs.
#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

constexpr int GW_BLOCK = 256;
constexpr int GW_ITEMS = 8;
constexpr int GW_TILE = GW_BLOCK * GW_ITEMS;   // 2048 == SEG_TILE (tile_first_head granularity)
static_assert(GW_TILE == SEG_TILE, "tile size shared with group.hip");

template <typename T> __device__ __forceinline__ T shfl_up_any(const T& v, int o) {
  static_assert(sizeof(T) % 4 == 0, "state must be a multiple of 4 bytes");
  constexpr int W = sizeof(T) / 4;
  uint32_t w[W];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (int i = 0; i < W; i++) w[i] = (uint32_t)__shfl_up((int)w[i], o, 64);
  T r;
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}

__device__ __forceinline__ bool head_bit(const uint32_t* __restrict__ bm, uint32_t p) { return (bm[p >> 5] >> (p & 31)) & 1u; }
// logical position q of the scan order -> physical grouped position
__device__ __forceinline__ uint32_t phys_pos(uint32_t q, uint32_t n, int rev) { return rev ? n - 1 - q : q; }
// does a segment start at logical position q?
__device__ __forceinline__ bool seg_head(const uint32_t* __restrict__ bm, uint32_t q, uint32_t n, int rev) {
  const uint32_t p = phys_pos(q, n, rev);
  if (!rev) return head_bit(bm, p);
  return p == n - 1 || head_bit(bm, p + 1);
}

template <class St> struct TileSum { St s; uint32_t has_head; uint32_t pad; };

// The 8 rows of one thread: head bits + single-row states.  Interior forward blocks take the head
// byte in one read and the policy's 16-byte loads (8 scalar loads per thread walk the same cache
// lines 8 times: the 2-column moments summary ran at 1.7 ms per 1e8 rows instead of 0.35).
template <class P>
__device__ __forceinline__ uint32_t load_rows(const typename P::Args& a, const uint32_t* __restrict__ bm, uint32_t q0, uint32_t n,
                                              int rev, typename P::St* x) {
  uint32_t hb = 0;
  if (!rev && q0 + GW_ITEMS <= n) {
    hb = (bm[q0 >> 5] >> (q0 & 31)) & 0xFFu;      // q0 is a multiple of 8: the byte never straddles a word
    P::load_block(a, q0, x);
  } else {
#pragma unroll
    for (int j = 0; j < GW_ITEMS; j++) {
      const uint32_t q = q0 + j;
      x[j] = P::ident();
      if (q < n) {
        if (seg_head(bm, q, n, rev)) hb |= 1u << j;
        x[j] = P::load(a, phys_pos(q, n, rev));
      }
    }
  }
  return hb;
}

__device__ __forceinline__ void load8x8(const void* p, unsigned long long* v) {        // 8 x 8 bytes, p 16-byte aligned
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2* s = static_cast<const u64x2*>(p);
  const u64x2 a = s[0], b = s[1], c = s[2], d = s[3];
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = d.x; v[7] = d.y;
}
__device__ __forceinline__ void load8x4(const void* p, uint32_t* v) {                  // 8 x 4 bytes, p 16-byte aligned
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* s = static_cast<const u32x4*>(p);
  const u32x4 a = s[0], b = s[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8x8(void* p, const unsigned long long* v) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2* d = static_cast<u64x2*>(p);
  d[0] = u64x2{v[0], v[1]}; d[1] = u64x2{v[2], v[3]}; d[2] = u64x2{v[4], v[5]}; d[3] = u64x2{v[6], v[7]};
}
__device__ __forceinline__ void store8x4(void* p, const uint32_t* v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4* d = static_cast<u32x4*>(p);
  d[0] = u32x4{v[0], v[1], v[2], v[3]}; d[1] = u32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- kernel 1: per-tile trailing-segment state --------------------------------------------------
template <class P>
__global__ void __launch_bounds__(GW_BLOCK) gw_summary_kernel(typename P::Args a, const uint32_t* __restrict__ bm, uint32_t n,
                                                              int rev, TileSum<typename P::St>* __restrict__ sums) {
  typedef typename P::St St;
  __shared__ St w_val[GW_BLOCK / 64];
  __shared__ uint32_t w_flag[GW_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t q0 = blockIdx.x * GW_TILE + tid * GW_ITEMS;
  St x[GW_ITEMS];
  const uint32_t hb = load_rows<P>(a, bm, q0, n, rev, x);
  St cur = P::ident();
#pragma unroll
  for (int j = 0; j < GW_ITEMS; j++) {
    if ((hb >> j) & 1u) cur = P::ident();
    cur = P::comb(cur, x[j]);                     // rows past n are identities
  }
  const uint32_t flag = hb ? 1u : 0u;
  St sv = cur; uint32_t sf = flag;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const St pv = shfl_up_any(sv, o);
    const uint32_t pf = (uint32_t)__shfl_up((int)sf, o, 64);
    if (lane >= o) { if (!sf) sv = P::comb(pv, sv); sf |= pf; }
  }
  if (lane == 63) { w_val[wave] = sv; w_flag[wave] = sf; }
  __syncthreads();
  if (tid == 0) {
    St acc = w_val[0]; uint32_t f = w_flag[0];
    for (int w = 1; w < GW_BLOCK / 64; w++) {
      if (w_flag[w]) { acc = w_val[w]; f = 1; } else { acc = P::comb(acc, w_val[w]); }
    }
    sums[blockIdx.x].s = acc;
    sums[blockIdx.x].has_head = f;
  }
}

// ---- kernel 2: exclusive segmented scan over the tile states (one workgroup) -------------------
constexpr int GC_BLOCK = 1024;
constexpr int GC_ITEMS = 4;
template <class P>
__global__ void __launch_bounds__(GC_BLOCK) gw_carry_kernel(const TileSum<typename P::St>* __restrict__ sums, uint32_t nt,
                                                            typename P::St* __restrict__ carry) {
  typedef typename P::St St;
  __shared__ St w_val[GC_BLOCK / 64];
  __shared__ uint32_t w_flag[GC_BLOCK / 64];
  __shared__ St run_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) run_s = P::ident();
  __syncthreads();
  for (uint32_t base = 0; base < nt; base += GC_BLOCK * GC_ITEMS) {
    const uint32_t t0 = base + tid * GC_ITEMS;
    St s[GC_ITEMS]; uint32_t f[GC_ITEMS];
    St cur = P::ident(); uint32_t flag = 0;
#pragma unroll
    for (int j = 0; j < GC_ITEMS; j++) {
      s[j] = P::ident(); f[j] = 0;
      if (t0 + j < nt) { s[j] = sums[t0 + j].s; f[j] = sums[t0 + j].has_head; }
      if (f[j]) { cur = s[j]; flag = 1; } else { cur = P::comb(cur, s[j]); }
    }
    St sv = cur; uint32_t sf = flag;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const St pv = shfl_up_any(sv, o);
      const uint32_t pf = (uint32_t)__shfl_up((int)sf, o, 64);
      if (lane >= o) { if (!sf) sv = P::comb(pv, sv); sf |= pf; }
    }
    if (lane == 63) { w_val[wave] = sv; w_flag[wave] = sf; }
    __syncthreads();
    St acc = run_s;
    for (int w = 0; w < wave; w++) { if (w_flag[w]) acc = w_val[w]; else acc = P::comb(acc, w_val[w]); }
    {
      St ev = shfl_up_any(sv, 1);
      uint32_t ef = (uint32_t)__shfl_up((int)sf, 1, 64);
      if (lane == 0) { ev = P::ident(); ef = 0; }
      if (ef) acc = ev; else acc = P::comb(acc, ev);
    }
#pragma unroll
    for (int j = 0; j < GC_ITEMS; j++) {
      if (t0 + j < nt) carry[t0 + j] = acc;
      if (f[j]) acc = s[j]; else acc = P::comb(acc, s[j]);
    }
    __syncthreads();
    if (tid == GC_BLOCK - 1) run_s = acc;
    __syncthreads();
  }
}

// ---- kernel 3: tile scan with carry; SCAN stores every position, REDUCE emits at group ends -----
template <class P, int REDUCE>
__global__ void __launch_bounds__(GW_BLOCK) gw_apply_kernel(typename P::Args a, const uint32_t* __restrict__ bm,
                                                            const uint32_t* __restrict__ tile_first_head, uint32_t n, int rev,
                                                            const typename P::St* __restrict__ carry, typename P::Out out) {
  typedef typename P::St St;
  __shared__ St w_val[GW_BLOCK / 64];
  __shared__ uint32_t w_flag[GW_BLOCK / 64];
  __shared__ uint32_t w_nh[GW_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tile = blockIdx.x;
  const uint32_t q0 = tile * GW_TILE + tid * GW_ITEMS;
  St x[GW_ITEMS];
  const uint32_t hb = load_rows<P>(a, bm, q0, n, rev, x);
  St cur = P::ident();
#pragma unroll
  for (int j = 0; j < GW_ITEMS; j++) {
    if ((hb >> j) & 1u) cur = P::ident();
    cur = P::comb(cur, x[j]);
  }
  const uint32_t nh = (uint32_t)__popc(hb);
  St sv = cur; uint32_t sf = nh ? 1u : 0u; uint32_t sn = nh;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const St pv = shfl_up_any(sv, o);
    const uint32_t pf = (uint32_t)__shfl_up((int)sf, o, 64);
    const uint32_t pn = (uint32_t)__shfl_up((int)sn, o, 64);
    if (lane >= o) { if (!sf) sv = P::comb(pv, sv); sf |= pf; sn += pn; }
  }
  if (lane == 63) { w_val[wave] = sv; w_flag[wave] = sf; w_nh[wave] = sn; }
  __syncthreads();
  St acc = carry[tile];
  uint32_t k = 0;
  for (int w = 0; w < wave; w++) {
    if (w_flag[w]) acc = w_val[w]; else acc = P::comb(acc, w_val[w]);
    k += w_nh[w];
  }
  {
    St ev = shfl_up_any(sv, 1);
    uint32_t ef = (uint32_t)__shfl_up((int)sf, 1, 64), en = (uint32_t)__shfl_up((int)sn, 1, 64);
    if (lane == 0) { ev = P::ident(); ef = 0; en = 0; }
    if (ef) acc = ev; else acc = P::comb(acc, ev);
    k += en;
  }
  const uint32_t G = REDUCE ? tile_first_head[tile] : 0u;    // heads in earlier tiles
  const bool fast = !rev && q0 + GW_ITEMS <= n;
#pragma unroll
  for (int j = 0; j < GW_ITEMS; j++) {
    if ((hb >> j) & 1u) { acc = P::ident(); k++; }
    acc = P::comb(acc, x[j]);
    x[j] = acc;                                   // the running value of row j
    if (REDUCE) {
      const uint32_t q = q0 + j;
      if (q < n) {
        const bool last = (q == n - 1) || (j < GW_ITEMS - 1 && fast ? ((hb >> (j + 1)) & 1u) != 0 : seg_head(bm, q + 1, n, rev));
        if (last) P::emit(out, G + k - 1, acc);
      }
    }
  }
  if (!REDUCE) {
    if (fast) {
      P::store_block(out, q0, x);
    } else {
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) if (q0 + j < n) P::store(out, phys_pos(q0 + j, n, rev), x[j]);
    }
  }
}

template <class P, int REDUCE>
static int run_segscan(dthip_ctx* ctx, const typename P::Args& a, const uint32_t* bm, const uint32_t* tile_first_head,
                       int64_t n, int rev, const typename P::Out& out, const char* name) {
  typedef typename P::St St;
  const uint32_t nt = (uint32_t)((n + GW_TILE - 1) / GW_TILE);
  if (nt == 0) return DTHIP_OK;
  Scratch sc(ctx);
  TileSum<St>* sums = nullptr; St* carry = nullptr;
  DTHIP_TRY(sc.get<TileSum<St>>(nt, &sums));
  DTHIP_TRY(sc.get<St>(nt, &carry));
  DTHIP_LAUNCH(ctx, "gw_summary_kernel", gw_summary_kernel<P>, nt, GW_BLOCK, 0, a, bm, (uint32_t)n, rev, sums);
  DTHIP_LAUNCH(ctx, "gw_carry_kernel", gw_carry_kernel<P>, 1, GC_BLOCK, 0, sums, nt, carry);
  DTHIP_LAUNCH(ctx, name, (gw_apply_kernel<P, REDUCE>), nt, GW_BLOCK, 0, a, bm, tile_first_head, (uint32_t)n, rev, carry, out);
  return DTHIP_OK;
}

// ---- policy: running moments (sd / cov / corr) ---------------------------------------------------
struct MomArgs { const double* x; const double* y; };
struct MomOut { void* out; int op; int f32; uint8_t* nonfinite; };   // op: 0 sd, 1 cov, 2 corr; nonfinite[g]: see cov_seq_kernel
template <int NC> struct MomSt;
template <> struct MomSt<1> { double n, mx, cxx; };
template <> struct MomSt<2> { double n, mx, my, cxx, cyy, cxy; };

template <int NC> struct MomP;
template <> struct MomP<1> {
  typedef MomSt<1> St; typedef MomArgs Args; typedef MomOut Out;
  static __device__ __forceinline__ St ident() { return St{0.0, 0.0, 0.0}; }
  static __device__ __forceinline__ St comb(const St& a, const St& b) {
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    St r;
    r.n = a.n + b.n;
    const double d = b.mx - a.mx, f = b.n / r.n;
    r.mx = a.mx + d * f;
    r.cxx = a.cxx + b.cxx + d * d * a.n * f;
    return r;
  }
  static __device__ __forceinline__ St load(const Args& a, uint32_t p) {
    return make(a.x[p]);                  // M2 of one row = x - x: 0, or NaN for +-inf (the reference's m2 turns NaN too)
  }
  static __device__ __forceinline__ St make(double x) {
    if (x != x) return ident();
    return St{1.0, x, x - x};
  }
  static __device__ __forceinline__ void load_block(const Args& a, uint32_t q0, St* x) {
    if (aligned16(a.x)) {
      unsigned long long v[GW_ITEMS];
      load8x8(a.x + q0, v);
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) x[j] = make(__longlong_as_double((long long)v[j]));
    } else {
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) x[j] = make(a.x[q0 + j]);
    }
  }
  static __device__ __forceinline__ void store_block(const Out&, uint32_t, const St*) {}
  static __device__ __forceinline__ void emit(const Out& o, uint32_t g, const St& s) {
    double r = __builtin_nan("");
    if (s.n > 1.0 && !(s.cxx != s.cxx)) r = s.cxx >= 0.0 ? sqrt(s.cxx / (s.n - 1.0)) : 0.0;
    if (o.f32) static_cast<float*>(o.out)[g] = (float)r; else static_cast<double*>(o.out)[g] = r;
  }
  static __device__ __forceinline__ void store(const Out&, uint32_t, const St&) {}
};
template <> struct MomP<2> {
  typedef MomSt<2> St; typedef MomArgs Args; typedef MomOut Out;
  static __device__ __forceinline__ St ident() { return St{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; }
  static __device__ __forceinline__ St comb(const St& a, const St& b) {
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    St r;
    r.n = a.n + b.n;
    const double dx = b.mx - a.mx, dy = b.my - a.my, f = b.n / r.n, w = a.n * f;
    r.mx = a.mx + dx * f;
    r.my = a.my + dy * f;
    r.cxx = a.cxx + b.cxx + dx * dx * w;
    r.cyy = a.cyy + b.cyy + dy * dy * w;
    r.cxy = a.cxy + b.cxy + dx * dy * w;
    return r;
  }
  static __device__ __forceinline__ St load(const Args& a, uint32_t p) {
    return make(a.x[p], a.y[p]);                    // a row counts only when both values are valid
  }
  static __device__ __forceinline__ St make(double x, double y) {
    if (x != x || y != y) return ident();
    return St{1.0, x, y, x - x, y - y, (x - x) * (y - y)};
  }
  static __device__ __forceinline__ void load_block(const Args& a, uint32_t q0, St* x) {
    if (aligned16(a.x) && aligned16(a.y)) {
      unsigned long long vx[GW_ITEMS], vy[GW_ITEMS];
      load8x8(a.x + q0, vx);
      load8x8(a.y + q0, vy);
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) x[j] = make(__longlong_as_double((long long)vx[j]), __longlong_as_double((long long)vy[j]));
    } else {
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) x[j] = make(a.x[q0 + j], a.y[q0 + j]);
    }
  }
  static __device__ __forceinline__ void store_block(const Out&, uint32_t, const St*) {}
  static __device__ __forceinline__ void emit(const Out& o, uint32_t g, const St& s) {
    double r = __builtin_nan("");
    if (s.cxx != s.cxx || s.cyy != s.cyy) { o.nonfinite[g] = 1; return; }     // +-inf among the pairs: redone sequentially
    if (o.op == 1) { if (s.n > 1.0) r = s.cxy / (s.n - 1.0); }
    else { const double vv = s.cxx * s.cyy; if (s.n > 1.0 && vv > 0.0) r = s.cxy / sqrt(vv); }
    if (o.f32) static_cast<float*>(o.out)[g] = (float)r; else static_cast<double*>(o.out)[g] = r;
  }
  static __device__ __forceinline__ void store(const Out&, uint32_t, const St&) {}
};

// value of a column element as double, NA -> NaN (SentinelFw get_element + cast_inplace(FLOAT64))
__device__ __forceinline__ double load_as_f64(const void* data, int stype, int64_t j) {
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: { const int8_t v = static_cast<const int8_t*>(data)[j]; return v == INT8_MIN ? __builtin_nan("") : (double)v; }
    case DTHIP_INT16: { const int16_t v = static_cast<const int16_t*>(data)[j]; return v == INT16_MIN ? __builtin_nan("") : (double)v; }
    case DTHIP_INT32: { const int32_t v = static_cast<const int32_t*>(data)[j]; return v == INT32_MIN ? __builtin_nan("") : (double)v; }
    case DTHIP_INT64: { const long long v = static_cast<const long long*>(data)[j]; return v == INT64_MIN ? __builtin_nan("") : (double)v; }
    case DTHIP_FLOAT32: return (double)static_cast<const float*>(data)[j];
    default: return static_cast<const double*>(data)[j];
  }
}

__global__ void __launch_bounds__(256) gather_f64_kernel(const void* __restrict__ data, int stype, const int32_t* __restrict__ ri,
                                                         uint32_t n, double* __restrict__ out) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < n; p += stride) {
    const int32_t j = ri ? ri[p] : (int32_t)p;
    out[p] = j < 0 ? __builtin_nan("") : load_as_f64(data, stype, j);
  }
}

int launch_gather_f64(dthip_ctx* ctx, const void* data, int stype, const int32_t* ri, int64_t n, double* out) {
  if (n == 0) return DTHIP_OK;
  long long blocks = (n + 1023) / 1024;
  if (blocks > (long long)ctx->num_cus * 16) blocks = (long long)ctx->num_cus * 16;
  DTHIP_LAUNCH(ctx, "gather_f64_kernel", gather_f64_kernel, (unsigned)blocks, 256, 0, data, stype, ri, (uint32_t)n, out);
  return DTHIP_OK;
}

// cov / corr of a group that holds +-inf among its valid pairs: what the reference's sequential update
// yields then (+-inf or NaN) depends on the row order -- e.g. an infinity in the last row leaves cov at
// +-inf, one earlier usually turns it into NaN -- so those (rare) groups are re-evaluated exactly as
// cov_reducer / corr_reducer do (head_reduce_binary.cc:113-135,167-198), one thread per group, in T.
template <typename T>
__global__ void __launch_bounds__(256) cov_seq_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                      const int32_t* __restrict__ offsets, uint32_t ngroups,
                                                      const uint8_t* __restrict__ nonfinite, int op, T* __restrict__ out) {
#pragma clang fp contract(off)
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups || !nonfinite[g]) return;
  T mean1 = 0, mean2 = 0, var1 = 0, var2 = 0, cov = 0;
  long long n = 0;
  for (int32_t p = offsets[g]; p < offsets[g + 1]; p++) {
    const double xd = x[p], yd = y[p];
    if (xd != xd || yd != yd) continue;
    const T v1 = (T)xd, v2 = (T)yd;
    n++;
    const T d1 = v1 - mean1, d2 = v2 - mean2;
    mean1 += d1 / (T)n;
    mean2 += d2 / (T)n;
    const T t1 = v1 - mean1, t2 = v2 - mean2;
    cov += t1 * d2;
    var1 += t1 * d1;
    var2 += t2 * d2;
  }
  T r = (T)__builtin_nan("");
  if (op == 1) { if (n > 1) r = cov / (T)(n - 1); }
  else { const T vv = var1 * var2; if (n > 1 && vv > 0) r = cov / (T)sqrt((double)vv); }
  out[g] = r;
}

int launch_moments(dthip_ctx* ctx, const double* x, const double* y, const uint8_t* bitmap, const uint32_t* tile_first_head,
                   int64_t n, int op, void* out, int out_f32, const int32_t* offsets, int64_t ngroups) {
  MomArgs a{x, y};
  MomOut o{out, op, out_f32, nullptr};
  const uint32_t* bm = reinterpret_cast<const uint32_t*>(bitmap);
  if (op == 0) return run_segscan<MomP<1>, 1>(ctx, a, bm, tile_first_head, n, 0, o, "gw_apply_kernel<sd>");
  Scratch sc(ctx);
  DTHIP_TRY(sc.get<uint8_t>((size_t)ngroups, &o.nonfinite));
  DTHIP_CHECK_HIP(hipMemsetAsync(o.nonfinite, 0, (size_t)ngroups, ctx->stream));
  DTHIP_TRY((run_segscan<MomP<2>, 1>(ctx, a, bm, tile_first_head, n, 0, o, "gw_apply_kernel<cov>")));
  const unsigned grid = (unsigned)((ngroups + 255) / 256);
  if (out_f32) DTHIP_LAUNCH(ctx, "cov_seq_kernel", cov_seq_kernel<float>, grid, 256, 0, x, y, offsets, (uint32_t)ngroups, o.nonfinite, op,
                            static_cast<float*>(out));
  else DTHIP_LAUNCH(ctx, "cov_seq_kernel", cov_seq_kernel<double>, grid, 256, 0, x, y, offsets, (uint32_t)ngroups, o.nonfinite, op,
                    static_cast<double*>(out));
  return DTHIP_OK;
}

// ---- policy: cumulative sum / product / min / max -------------------------------------------------
struct CumArgs { const void* data; int stype; const int32_t* ri; };
struct CumOut { void* out; int ostype; };
template <typename A> struct CumSt { A v; uint32_t has; uint32_t pad; };

template <typename A> __device__ __forceinline__ bool load_as(const void* data, int stype, int64_t j, A* out);
template <> __device__ __forceinline__ bool load_as<long long>(const void* data, int stype, int64_t j, long long* out) {
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: { const int8_t v = static_cast<const int8_t*>(data)[j]; *out = v; return v != INT8_MIN; }
    case DTHIP_INT16: { const int16_t v = static_cast<const int16_t*>(data)[j]; *out = v; return v != INT16_MIN; }
    case DTHIP_INT32: { const int32_t v = static_cast<const int32_t*>(data)[j]; *out = v; return v != INT32_MIN; }
    default: { const long long v = static_cast<const long long*>(data)[j]; *out = v; return v != INT64_MIN; }
  }
}
template <> __device__ __forceinline__ bool load_as<double>(const void* data, int stype, int64_t j, double* out) {
  const double v = stype == DTHIP_FLOAT32 ? (double)static_cast<const float*>(data)[j] : static_cast<const double*>(data)[j];
  *out = v;
  return !(v != v);
}

enum { CUM_SUM = 0, CUM_PROD = 1, CUM_MIN = 2, CUM_MAX = 3, CUM_FILL = 6 };     // (= enum dthip_cumop; 4 / 5: cumcount / ngroup)

template <typename A, int OP> struct CumP {
  typedef CumSt<A> St; typedef CumArgs Args; typedef CumOut Out;
  static __device__ __forceinline__ St ident() { return St{OP == CUM_PROD ? A(1) : A(0), 0u, 0u}; }
  static __device__ __forceinline__ A add(A a, A b) { return a + b; }
  static __device__ __forceinline__ A mul(A a, A b) { return a * b; }
  static __device__ __forceinline__ St comb(const St& a, const St& b) {   // a = earlier rows, b = later rows
    St r; r.pad = 0;
    r.has = a.has | b.has;
    if (OP == CUM_SUM) r.v = add(a.v, b.v);
    else if (OP == CUM_PROD) r.v = mul(a.v, b.v);
    else if (!b.has) r.v = a.v;
    else if (!a.has || OP == CUM_FILL) r.v = b.v;                         // fillna: the last valid value (fexpr_fillna.cc:101-113)
    else if (OP == CUM_MIN) r.v = (a.v < b.v) ? a.v : b.v;               // ties keep the later row's value
    else r.v = (a.v > b.v) ? a.v : b.v;                                  // (cumminmax.h:83-87)
    return r;
  }
  static __device__ __forceinline__ St load(const Args& a, uint32_t p) {
    const int32_t j = a.ri ? a.ri[p] : (int32_t)p;
    A v;
    if (j < 0 || !load_as<A>(a.data, a.stype, j, &v)) return ident();     // NA: 0 / 1 / "nothing yet"
    return St{v, 1u, 0u};
  }
  static __device__ __forceinline__ void store(const Out& o, uint32_t p, const St& s) {
    const bool valid = (OP == CUM_SUM || OP == CUM_PROD) ? true : (s.has != 0);
    switch (o.ostype) {
      case DTHIP_BOOL: case DTHIP_INT8: static_cast<int8_t*>(o.out)[p] = valid ? (int8_t)s.v : INT8_MIN; break;
      case DTHIP_INT16: static_cast<int16_t*>(o.out)[p] = valid ? (int16_t)s.v : INT16_MIN; break;
      case DTHIP_INT32: static_cast<int32_t*>(o.out)[p] = valid ? (int32_t)s.v : INT32_MIN; break;
      case DTHIP_INT64: static_cast<long long*>(o.out)[p] = valid ? (long long)s.v : INT64_MIN; break;
      case DTHIP_FLOAT32: static_cast<float*>(o.out)[p] = valid ? (float)s.v : __builtin_nanf(""); break;
      default: static_cast<double*>(o.out)[p] = valid ? (double)s.v : __builtin_nan(""); break;
    }
  }
  static __device__ __forceinline__ void emit(const Out&, uint32_t, const St&) {}

  // 8 consecutive rows of a column in grouped order (no RowIndex): 16-byte loads for 8- and 4-byte stypes
  static __device__ __forceinline__ void load_block(const Args& a, uint32_t q0, St* x) {
    const int sz = (a.stype == DTHIP_INT64 || a.stype == DTHIP_FLOAT64) ? 8 : (a.stype == DTHIP_INT32 || a.stype == DTHIP_FLOAT32) ? 4 : 0;
    if (!a.ri && sz == 8 && aligned16(a.data)) {
      unsigned long long v[GW_ITEMS];
      load8x8(static_cast<const unsigned long long*>(a.data) + q0, v);
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) {
        A t;
        x[j] = load_as<A>(v + j, a.stype, 0, &t) ? St{t, 1u, 0u} : ident();
      }
    } else if (!a.ri && sz == 4 && aligned16(a.data)) {
      uint32_t v[GW_ITEMS];
      load8x4(static_cast<const uint32_t*>(a.data) + q0, v);
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) {
        A t;
        x[j] = load_as<A>(v + j, a.stype, 0, &t) ? St{t, 1u, 0u} : ident();
      }
    } else {
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) x[j] = load(a, q0 + j);
    }
  }
  static __device__ __forceinline__ void store_block(const Out& o, uint32_t q0, const St* x) {
    const int sz = (o.ostype == DTHIP_INT64 || o.ostype == DTHIP_FLOAT64) ? 8 : (o.ostype == DTHIP_INT32 || o.ostype == DTHIP_FLOAT32) ? 4 : 0;
    if (sz == 8 && aligned16(o.out)) {
      unsigned long long v[GW_ITEMS];
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) { CumOut t{v, o.ostype}; store(t, (uint32_t)j, x[j]); }
      store8x8(static_cast<unsigned long long*>(o.out) + q0, v);
    } else if (sz == 4 && aligned16(o.out)) {
      uint32_t v[GW_ITEMS];
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) { CumOut t{v, o.ostype}; store(t, (uint32_t)j, x[j]); }
      store8x4(static_cast<uint32_t*>(o.out) + q0, v);
    } else {
#pragma unroll
      for (int j = 0; j < GW_ITEMS; j++) store(o, q0 + j, x[j]);
    }
  }
};
// int64 sums and products wrap (two's complement), as the reference's do in practice
template <> __device__ __forceinline__ long long CumP<long long, CUM_SUM>::add(long long a, long long b) {
  return (long long)((unsigned long long)a + (unsigned long long)b);
}
template <> __device__ __forceinline__ long long CumP<long long, CUM_PROD>::mul(long long a, long long b) {
  return (long long)((unsigned long long)a * (unsigned long long)b);
}

int launch_cumulate(dthip_ctx* ctx, const void* data, int stype, const int32_t* ri, const uint8_t* bitmap, int64_t n,
                    int op, int reverse, void* out, int ostype) {
  // Both passes read the column; through a RowIndex that is a random gather, so it is done once into
  // grouped order (2.0 ms per 1e8 float64 rows) and the two scan passes stream (0.2 + 0.4 ms).
  Scratch sc(ctx);
  if (ri) {
    const int sz = stype_size(stype);
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>((size_t)n * sz, &t));
    DTHIP_TRY(launch_gather(ctx, data, stype, ri, n, t));
    data = t;
    ri = nullptr;
  }
  CumArgs a{data, stype, ri};
  CumOut o{out, ostype};
  const uint32_t* bm = reinterpret_cast<const uint32_t*>(bitmap);
  const bool isf = stype == DTHIP_FLOAT32 || stype == DTHIP_FLOAT64;
  const int rev = reverse ? 1 : 0;
#define DTHIP_CUM_CASE(OPC)                                                                                         \
  case OPC:                                                                                                         \
    return isf ? run_segscan<CumP<double, OPC>, 0>(ctx, a, bm, nullptr, n, rev, o, "gw_apply_kernel<cum>")          \
               : run_segscan<CumP<long long, OPC>, 0>(ctx, a, bm, nullptr, n, rev, o, "gw_apply_kernel<cum>");
  switch (op) {
    DTHIP_CUM_CASE(CUM_SUM)
    DTHIP_CUM_CASE(CUM_PROD)
    DTHIP_CUM_CASE(CUM_MIN)
    DTHIP_CUM_CASE(CUM_MAX)
    DTHIP_CUM_CASE(CUM_FILL)
    default: set_error("bad cumulative op %d", op); return DTHIP_EINVAL;
  }
#undef DTHIP_CUM_CASE
}

// cumcount() / ngroup(): launch_cumcount in group.hip (the head-bitmap expansion shared with ungroup)

// ---- median / nunique over rows sorted by (group, value), NA first ---------------------------------
template <typename T> struct NaOf;
template <> struct NaOf<int8_t> { static __device__ __forceinline__ bool isna(int8_t v) { return v == INT8_MIN; } };
template <> struct NaOf<int16_t> { static __device__ __forceinline__ bool isna(int16_t v) { return v == INT16_MIN; } };
template <> struct NaOf<int32_t> { static __device__ __forceinline__ bool isna(int32_t v) { return v == INT32_MIN; } };
template <> struct NaOf<long long> { static __device__ __forceinline__ bool isna(long long v) { return v == INT64_MIN; } };
template <> struct NaOf<float> { static __device__ __forceinline__ bool isna(float v) { return v != v; } };
template <> struct NaOf<double> { static __device__ __forceinline__ bool isna(double v) { return v != v; } };

// median / nunique read the DISTINCT (group, value) pairs of the grouped column, ordered by (group,
// value) with NA first, and the number of rows before each pair (`pair_off`): what the fused
// groupby-aggregate returns for keys (group id, value) and count().  Rows of group g occupy sorted
// positions [offsets[g], offsets[g+1]); the value at sorted position p belongs to the last pair whose
// pair_off <= p.

__device__ __forceinline__ uint32_t pair_of(const int32_t* __restrict__ pair_off, uint32_t npairs, uint32_t p) {
  uint32_t lo = 0, hi = npairs;                  // largest s with pair_off[s] <= p
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((uint32_t)pair_off[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// Median_ColumnImpl::get_element (head_reduce_unary.cc:446-466): skip the leading NAs of the sorted
// group, then the middle element, or the mean of the two middle ones computed in U
template <typename T, typename U>
__global__ void __launch_bounds__(256) median_kernel(const T* __restrict__ pair_val, const int32_t* __restrict__ pair_off,
                                                     uint32_t npairs, const int32_t* __restrict__ offsets, uint32_t ngroups,
                                                     U* __restrict__ out) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const uint32_t i0 = (uint32_t)offsets[g], i1 = (uint32_t)offsets[g + 1];
  const uint32_t s0 = pair_of(pair_off, npairs, i0);
  uint32_t a = i0;                               // first valid position: the NA pair, if any, leads the group
  if (NaOf<T>::isna(pair_val[s0])) a = (uint32_t)pair_off[s0 + 1];
  if (a >= i1) { out[g] = (U)__builtin_nan(""); return; }
  const uint32_t j = (a + i1) >> 1;
  const T v1 = pair_val[pair_of(pair_off, npairs, j)];
  if ((i1 - a) & 1u) out[g] = (U)v1;
  else out[g] = ((U)v1 + (U)pair_val[pair_of(pair_off, npairs, j - 1)]) / (U)2;
}

// op_nunique (head_reduce_unary.cc:377-387): distinct valid values per group.  One thread per pair; a
// pair counts when its value is valid and differs -- as a VALUE, so -0.0 == 0.0 like std::set's
// ordering -- from the previous pair of the same group.  Lanes of a wave in the same group add once.
template <typename T>
__global__ void __launch_bounds__(256) nunique_kernel(const T* __restrict__ pair_val, const int32_t* __restrict__ pair_gid,
                                                      uint32_t npairs, unsigned long long* __restrict__ out) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int32_t g = -1;
  bool distinct = false;
  if (s < npairs) {
    g = pair_gid[s];
    const T v = pair_val[s];
    distinct = !NaOf<T>::isna(v);
    if (distinct && s > 0 && pair_gid[s - 1] == g && pair_val[s - 1] == v) distinct = false;
  }
  const int32_t gp = __shfl_up(g, 1, 64);
  const bool leader = lane == 0 || g != gp;
  const unsigned long long leaders = __ballot(leader);
  const unsigned long long dmask = __ballot(distinct);
  if (leader && g >= 0) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long higher = leaders & ~(below | (1ull << lane));
    const int end = higher ? (__ffsll((long long)higher) - 1) : 64;
    const unsigned long long upto = end == 64 ? ~0ull : ((1ull << end) - 1ull);
    const int c = __popcll(dmask & upto & ~below);
    if (c) atomicAdd(&out[g], (unsigned long long)c);
  }
}

int launch_median(dthip_ctx* ctx, const void* pair_val, int stype, const int32_t* pair_off, int64_t npairs, const int32_t* offsets,
                  int64_t ngroups, void* out) {
  if (ngroups == 0) return DTHIP_OK;
  const unsigned grid = (unsigned)((ngroups + 255) / 256);
#define DTHIP_MED(T, U) DTHIP_LAUNCH(ctx, "median_kernel", (median_kernel<T, U>), grid, 256, 0, static_cast<const T*>(pair_val), pair_off, \
                                     (uint32_t)npairs, offsets, (uint32_t)ngroups, static_cast<U*>(out)); break
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: DTHIP_MED(int8_t, double);
    case DTHIP_INT16: DTHIP_MED(int16_t, double);
    case DTHIP_INT32: DTHIP_MED(int32_t, double);
    case DTHIP_INT64: DTHIP_MED(long long, double);
    case DTHIP_FLOAT32: DTHIP_MED(float, float);
    case DTHIP_FLOAT64: DTHIP_MED(double, double);
    default: set_error("median: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
#undef DTHIP_MED
  return DTHIP_OK;
}

int launch_nunique(dthip_ctx* ctx, const void* pair_val, int stype, const int32_t* pair_gid, int64_t npairs, int64_t ngroups,
                   int64_t* out) {
  DTHIP_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(int64_t) * (size_t)ngroups, ctx->stream));
  if (npairs == 0) return DTHIP_OK;
  const unsigned grid = (unsigned)((npairs + 255) / 256);
#define DTHIP_NU(T) DTHIP_LAUNCH(ctx, "nunique_kernel", nunique_kernel<T>, grid, 256, 0, static_cast<const T*>(pair_val), pair_gid, \
                                 (uint32_t)npairs, reinterpret_cast<unsigned long long*>(out)); break
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: DTHIP_NU(int8_t);
    case DTHIP_INT16: DTHIP_NU(int16_t);
    case DTHIP_INT32: DTHIP_NU(int32_t);
    case DTHIP_INT64: DTHIP_NU(long long);
    case DTHIP_FLOAT32: DTHIP_NU(float);
    case DTHIP_FLOAT64: DTHIP_NU(double);
    default: set_error("nunique: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
#undef DTHIP_NU
  return DTHIP_OK;
}

// ---- the same two reducers over ROWS sorted by (group, value) ------------------------------------
// Used for float columns, whose values are mostly distinct: the pairs would be as many as the rows and
// materialising them costs more (2 ms per 1e8 rows) than reading the sorted rows through their order.
// Median_ColumnImpl::get_element (head_reduce_unary.cc:446-466): skip the leading NAs of the sorted
// group, then the middle element, or the mean of the two middle ones computed in U
template <typename T, typename U>
__global__ void __launch_bounds__(256) median_sorted_kernel(const T* __restrict__ vg, const int32_t* __restrict__ order,
                                                     const int32_t* __restrict__ offsets, uint32_t ngroups, U* __restrict__ out) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const uint32_t i0 = (uint32_t)offsets[g], i1 = (uint32_t)offsets[g + 1];
  uint32_t lo = i0, hi = i1;                     // first valid position
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (NaOf<T>::isna(vg[order[mid]])) lo = mid + 1; else hi = mid;
  }
  if (lo == i1) { out[g] = (U)__builtin_nan(""); return; }
  const uint32_t j = (lo + i1) >> 1;
  const T v1 = vg[order[j]];
  if ((i1 - lo) & 1u) out[g] = (U)v1;
  else out[g] = ((U)v1 + (U)vg[order[j - 1]]) / (U)2;
}

// op_nunique (head_reduce_unary.cc:377-387): distinct valid values per group.  One thread per
// (group, value) run of the sorted order; a run counts when its value is valid and differs -- as a
// VALUE, so -0.0 == 0.0 like std::set's ordering -- from the previous run of the same group.
// Lanes of a wave that fall into the same group add once.
template <typename T>
__global__ void __launch_bounds__(256) nunique_sorted_kernel(const T* __restrict__ vg, const int32_t* __restrict__ gid,
                                                      const int32_t* __restrict__ order, const int32_t* __restrict__ run_offsets,
                                                      uint32_t nruns, unsigned long long* __restrict__ out) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int32_t g = -1;
  bool distinct = false;
  if (s < nruns) {
    const int32_t r = order[run_offsets[s]];
    g = gid[r];
    const T v = vg[r];
    distinct = !NaOf<T>::isna(v);
    if (distinct && s > 0) {
      const int32_t rp = order[run_offsets[s - 1]];
      if (gid[rp] == g && vg[rp] == v) distinct = false;
    }
  }
  const int32_t gp = __shfl_up(g, 1, 64);
  const bool leader = lane == 0 || g != gp;
  const unsigned long long leaders = __ballot(leader);
  const unsigned long long dmask = __ballot(distinct);
  if (leader && g >= 0) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long higher = leaders & ~(below | (1ull << lane));
    const int end = higher ? (__ffsll((long long)higher) - 1) : 64;
    const unsigned long long upto = end == 64 ? ~0ull : ((1ull << end) - 1ull);
    const int c = __popcll(dmask & upto & ~below);
    if (c) atomicAdd(&out[g], (unsigned long long)c);
  }
}

int launch_median_sorted(dthip_ctx* ctx, const void* vg, int stype, const int32_t* order, const int32_t* offsets, int64_t ngroups,
                  void* out) {
  if (ngroups == 0) return DTHIP_OK;
  const unsigned grid = (unsigned)((ngroups + 255) / 256);
#define DTHIP_MEDS(T, U) DTHIP_LAUNCH(ctx, "median_sorted_kernel", (median_sorted_kernel<T, U>), grid, 256, 0, static_cast<const T*>(vg), order, offsets, \
                                     (uint32_t)ngroups, static_cast<U*>(out)); break
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: DTHIP_MEDS(int8_t, double);
    case DTHIP_INT16: DTHIP_MEDS(int16_t, double);
    case DTHIP_INT32: DTHIP_MEDS(int32_t, double);
    case DTHIP_INT64: DTHIP_MEDS(long long, double);
    case DTHIP_FLOAT32: DTHIP_MEDS(float, float);
    case DTHIP_FLOAT64: DTHIP_MEDS(double, double);
    default: set_error("median: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
#undef DTHIP_MEDSS
  return DTHIP_OK;
}

int launch_nunique_sorted(dthip_ctx* ctx, const void* vg, int stype, const int32_t* gid, const int32_t* order, const int32_t* run_offsets,
                   int64_t nruns, int64_t ngroups, int64_t* out) {
  DTHIP_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(int64_t) * (size_t)ngroups, ctx->stream));
  if (nruns == 0) return DTHIP_OK;
  const unsigned grid = (unsigned)((nruns + 255) / 256);
#define DTHIP_NUS(T) DTHIP_LAUNCH(ctx, "nunique_sorted_kernel", nunique_sorted_kernel<T>, grid, 256, 0, static_cast<const T*>(vg), gid, order, run_offsets, \
                                 (uint32_t)nruns, reinterpret_cast<unsigned long long*>(out)); break
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: DTHIP_NUS(int8_t);
    case DTHIP_INT16: DTHIP_NUS(int16_t);
    case DTHIP_INT32: DTHIP_NUS(int32_t);
    case DTHIP_INT64: DTHIP_NUS(long long);
    case DTHIP_FLOAT32: DTHIP_NUS(float);
    case DTHIP_FLOAT64: DTHIP_NUS(double);
    default: set_error("nunique: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
#undef DTHIP_NUS
  return DTHIP_OK;
}



}  // namespace dthip
