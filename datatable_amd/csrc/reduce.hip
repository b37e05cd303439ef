// reduce.hip -- per-group reducers sum / mean / min / max / count as one
// segmented reduction over the grouped row order.
//
// Reference semantics (src/core/column/sumprod.h:34-59, mean.h:33-52,
// minmax.h:33-62, count.h:35-58): for each group g the rows
// [offsets[g], offsets[g+1]) are visited in order, NA values are skipped,
//   sum   = T-typed sum (int64 for all integer stypes, wraps), 0 for an all-NA group
//   mean  = double sum / valid count, NA for an all-NA group (float32 in -> float32 out)
//   min/max = first valid then strict compare (ties keep the earlier row), NA if none
//   count = number of valid rows
// The reference runs one sequential loop per group and per reducer, each
// re-gathering the column through the RowIndex (column/view.cc:140-145).  Here
// every position is visited exactly once per value column and all five
// statistics are produced together:
//   - workgroup = 2048 consecutive grouped positions, 8 per thread, values
//     optionally gathered through the RowIndex
//   - group heads come from the head bitmap written by group.hip (1 bit per
//     position); the index of the first head of each tile from its scan
//   - thread-serial combine, then a wave64 shuffle segmented scan, then across
//     the 4 waves through LDS
//   - groups entirely inside a tile are finalised and stored directly; the two
//     open ends of a tile go to a side buffer that `seg_fixup_kernel` stitches
//     (one wave per tile, scanning forward until the next tile with a head)
// Combine order is position order, so integer results and min/max (including
// which of equal values wins) are identical to the sequential loops; float
// sums are re-associated (<= 1e-6 relative, stated in the tests).
#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

constexpr int SR_BLOCK = 256;
constexpr int SR_ITEMS = 8;
constexpr int SR_TILE = SR_BLOCK * SR_ITEMS;   // 2048, must equal group.hip's GB_TILE
typedef uint32_t ru32x4 __attribute__((ext_vector_type(4)));

// ---- accumulator states ---------------------------------------------------
struct StF {   // float32 / float64 values
  double sum, mn, mx;
  long long cnt;
};
struct StI {   // bool / int8 / int16 / int32 / int64 values
  unsigned long long isum;
  double dsum;
  long long mn, mx, cnt;
};

__device__ __forceinline__ StF ident(StF*) { return StF{0.0, __builtin_huge_val(), -__builtin_huge_val(), 0}; }
__device__ __forceinline__ StI ident(StI*) { return StI{0ULL, 0.0, INT64_MAX, INT64_MIN, 0}; }

// a = earlier positions, b = later positions
__device__ __forceinline__ StF comb(const StF& a, const StF& b) {
  StF r;
  r.sum = a.sum + b.sum;
  r.mn = (b.mn < a.mn) ? b.mn : a.mn;
  r.mx = (b.mx > a.mx) ? b.mx : a.mx;
  r.cnt = a.cnt + b.cnt;
  return r;
}
__device__ __forceinline__ StI comb(const StI& a, const StI& b) {
  StI r;
  r.isum = a.isum + b.isum;
  r.dsum = a.dsum + b.dsum;
  r.mn = (b.mn < a.mn) ? b.mn : a.mn;
  r.mx = (b.mx > a.mx) ? b.mx : a.mx;
  r.cnt = a.cnt + b.cnt;
  return r;
}

__device__ __forceinline__ StF shfl_up_st(const StF& s, int o) {
  StF r;
  r.sum = shfl_up_f64(s.sum, o); r.mn = shfl_up_f64(s.mn, o); r.mx = shfl_up_f64(s.mx, o);
  r.cnt = (long long)shfl_up_u64((unsigned long long)s.cnt, o);
  return r;
}
__device__ __forceinline__ StI shfl_up_st(const StI& s, int o) {
  StI r;
  r.isum = shfl_up_u64(s.isum, o); r.dsum = shfl_up_f64(s.dsum, o);
  r.mn = (long long)shfl_up_u64((unsigned long long)s.mn, o);
  r.mx = (long long)shfl_up_u64((unsigned long long)s.mx, o);
  r.cnt = (long long)shfl_up_u64((unsigned long long)s.cnt, o);
  return r;
}
__device__ __forceinline__ StF shfl_st(const StF& s, int src) {
  StF r;
  r.sum = __longlong_as_double((long long)shfl_u64((unsigned long long)__double_as_longlong(s.sum), src));
  r.mn = __longlong_as_double((long long)shfl_u64((unsigned long long)__double_as_longlong(s.mn), src));
  r.mx = __longlong_as_double((long long)shfl_u64((unsigned long long)__double_as_longlong(s.mx), src));
  r.cnt = (long long)shfl_u64((unsigned long long)s.cnt, src);
  return r;
}
__device__ __forceinline__ StI shfl_st(const StI& s, int src) {
  StI r;
  r.isum = shfl_u64(s.isum, src);
  r.dsum = __longlong_as_double((long long)shfl_u64((unsigned long long)__double_as_longlong(s.dsum), src));
  r.mn = (long long)shfl_u64((unsigned long long)s.mn, src);
  r.mx = (long long)shfl_u64((unsigned long long)s.mx, src);
  r.cnt = (long long)shfl_u64((unsigned long long)s.cnt, src);
  return r;
}

// ---- per-stype value traits -------------------------------------------------
template <typename T> struct VT;
template <> struct VT<double> {
  typedef StF St; typedef double SumT; typedef double MeanT;
  static __device__ __forceinline__ bool isna(double v) { return v != v; }
  static __device__ __forceinline__ double na() { return __builtin_nan(""); }
};
template <> struct VT<float> {
  typedef StF St; typedef float SumT; typedef float MeanT;
  static __device__ __forceinline__ bool isna(float v) { return v != v; }
  static __device__ __forceinline__ float na() { return __builtin_nanf(""); }
};
#define DTHIP_INT_VT(T, NA)                                                     \
  template <> struct VT<T> {                                                    \
    typedef StI St; typedef long long SumT; typedef double MeanT;               \
    static __device__ __forceinline__ bool isna(T v) { return v == (NA); }      \
    static __device__ __forceinline__ T na() { return (NA); }                   \
  };
DTHIP_INT_VT(int8_t, INT8_MIN)
DTHIP_INT_VT(int16_t, INT16_MIN)
DTHIP_INT_VT(int32_t, INT32_MIN)
DTHIP_INT_VT(long long, INT64_MIN)

template <typename T> __device__ __forceinline__ void accum(StF& s, T v) {
  const double d = (double)v;
  s.sum += d;
  if (d < s.mn) s.mn = d;
  if (d > s.mx) s.mx = d;
  s.cnt += 1;
}
template <typename T> __device__ __forceinline__ void accum(StI& s, T v) {
  const long long x = (long long)v;
  s.isum += (unsigned long long)x;
  s.dsum += (double)x;
  if (x < s.mn) s.mn = x;
  if (x > s.mx) s.mx = x;
  s.cnt += 1;
}

// PROD: the `sum` slot of an INTEGER state carries the wrapping product instead (prod(), column/sumprod.h:34-59 with SUM = false:
// identity 1, NA skipped; multiplication mod 2^64 is associative, so the segmented scan is exact).  Float products are not
// re-associated at all (prod_seq_kernel below).
template <bool PROD> __device__ __forceinline__ StF identP(StF* p) { return ident(p); }
template <bool PROD> __device__ __forceinline__ StI identP(StI* p) { StI r = ident(p); if (PROD) r.isum = 1ULL; return r; }
template <bool PROD> __device__ __forceinline__ StF combP(const StF& a, const StF& b) { return comb(a, b); }
template <bool PROD> __device__ __forceinline__ StI combP(const StI& a, const StI& b) {
  StI r = comb(a, b);
  if (PROD) r.isum = a.isum * b.isum;
  return r;
}
template <typename T, bool PROD> __device__ __forceinline__ void accumP(StF& s, T v) { accum<T>(s, v); }
template <typename T, bool PROD> __device__ __forceinline__ void accumP(StI& s, T v) {
  if (PROD) { const unsigned long long keep = s.isum; accum<T>(s, v); s.isum = keep * (unsigned long long)(long long)v; }
  else accum<T>(s, v);
}

template <typename T>
struct OutsT {
  typename VT<T>::SumT* sum;
  typename VT<T>::MeanT* mean;
  T* mn;
  T* mx;
  long long* count;
};

template <typename T> __device__ __forceinline__ void emit(const OutsT<T>& o, uint32_t g, const StF& s) {
  if (o.sum) o.sum[g] = (typename VT<T>::SumT)s.sum;
  if (o.mean) o.mean[g] = s.cnt ? (typename VT<T>::MeanT)(s.sum / (double)s.cnt) : (typename VT<T>::MeanT)VT<T>::na();
  if (o.mn) o.mn[g] = s.cnt ? (T)s.mn : VT<T>::na();
  if (o.mx) o.mx[g] = s.cnt ? (T)s.mx : VT<T>::na();
  if (o.count) o.count[g] = s.cnt;
}
template <typename T> __device__ __forceinline__ void emit(const OutsT<T>& o, uint32_t g, const StI& s) {
  if (o.sum) o.sum[g] = (long long)s.isum;
  if (o.mean) o.mean[g] = s.cnt ? s.dsum / (double)s.cnt : __builtin_nan("");
  if (o.mn) o.mn[g] = s.cnt ? (T)s.mn : VT<T>::na();
  if (o.mx) o.mx[g] = s.cnt ? (T)s.mx : VT<T>::na();
  if (o.count) o.count[g] = s.cnt;
}

// per-tile open ends
template <typename St>
struct TileSide {
  St first;      // positions before the first head of the tile (whole tile if it has none)
  St last;       // positions from the last head to the end of the tile
  uint32_t has_head;
  uint32_t last_gid;   // group of `last`
};

template <typename T, bool PROD>
__global__ void __launch_bounds__(SR_BLOCK) seg_reduce_kernel(const T* __restrict__ vals, const int32_t* __restrict__ ri,
                                                              const uint8_t* __restrict__ bitmap,
                                                              const uint32_t* __restrict__ tile_first_head,
                                                              uint32_t n, OutsT<T> outs,
                                                              TileSide<typename VT<T>::St>* side, int nona) {
  typedef typename VT<T>::St St;
  __shared__ St w_val[SR_BLOCK / 64];
  __shared__ uint32_t w_flag[SR_BLOCK / 64];
  __shared__ uint32_t w_nh[SR_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tile = blockIdx.x;
  const uint32_t p0 = tile * SR_TILE + tid * SR_ITEMS;    // this thread's first grouped position
  const uint32_t G = tile_first_head[tile];               // group index of the first head in this tile

  // head bits of this thread's 8 positions: exactly one bitmap byte
  uint32_t hb = 0;
  if (p0 < n) hb = bitmap[p0 >> 3];

  // values (optionally gathered through the RowIndex), NA -> not accumulated.
  // Interior threads read their 8 consecutive elements with 16-byte loads.
  T x[SR_ITEMS];
  bool ok[SR_ITEMS];
  const bool whole = p0 + SR_ITEMS <= n;
  if (whole && !ri && sizeof(T) >= 2 && ((reinterpret_cast<uintptr_t>(vals) & 15) == 0)) {
    constexpr int NV = SR_ITEMS * (int)sizeof(T) / 16 > 0 ? SR_ITEMS * (int)sizeof(T) / 16 : 1;
    ru32x4 v[NV];
    const ru32x4* src = reinterpret_cast<const ru32x4*>(vals + p0);
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = src[j];
    const T* vt = reinterpret_cast<const T*>(v);
#pragma unroll
    for (int j = 0; j < SR_ITEMS; j++) { x[j] = vt[j]; ok[j] = nona || !VT<T>::isna(x[j]); }
  } else if (whole && ri && ((reinterpret_cast<uintptr_t>(ri) & 15) == 0)) {
    ru32x4 rv[2];
    const ru32x4* rsrc = reinterpret_cast<const ru32x4*>(ri + p0);
    rv[0] = rsrc[0]; rv[1] = rsrc[1];
    const int32_t* rr = reinterpret_cast<const int32_t*>(rv);
#pragma unroll
    for (int j = 0; j < SR_ITEMS; j++) {
      const int32_t r = rr[j];
      ok[j] = r >= 0;
      x[j] = ok[j] ? vals[r] : T(0);
      if (ok[j] && !nona && VT<T>::isna(x[j])) ok[j] = false;
    }
  } else {
#pragma unroll
    for (int j = 0; j < SR_ITEMS; j++) {
      const uint32_t p = p0 + j;
      ok[j] = false;
      x[j] = T(0);
      if (p < n) {
        if (ri) {
          const int32_t r = ri[p];
          if (r >= 0) { x[j] = vals[r]; ok[j] = true; }
        } else {
          x[j] = vals[p]; ok[j] = true;
        }
        if (ok[j] && !nona && VT<T>::isna(x[j])) ok[j] = false;
      } else {
        hb &= ~(1u << j);
      }
    }
  }

  // thread summary: value of the open segment at the end of the thread's range
  St cur = identP<PROD>((St*)nullptr);
#pragma unroll
  for (int j = 0; j < SR_ITEMS; j++) {
    if ((hb >> j) & 1u) cur = identP<PROD>((St*)nullptr);
    if (ok[j]) accumP<T, PROD>(cur, x[j]);
  }
  const uint32_t nh = (uint32_t)__popc(hb);
  uint32_t flag = nh ? 1u : 0u;

  // inclusive segmented scan over the wave: (flag, value, nheads)
  St sv = cur; uint32_t sf = flag; uint32_t sn = nh;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const St pv = shfl_up_st(sv, o);
    const uint32_t pf = __shfl_up(sf, o, 64);
    const uint32_t pn = __shfl_up(sn, o, 64);
    if (lane >= o) {
      if (!sf) sv = combP<PROD>(pv, sv);
      sf |= pf;
      sn += pn;
    }
  }
  if (lane == 63) { w_val[wave] = sv; w_flag[wave] = sf; w_nh[wave] = sn; }
  __syncthreads();
  // exclusive carry for this thread = (carry from earlier waves) (+) (exclusive within wave)
  St carry = identP<PROD>((St*)nullptr); uint32_t cflag = 0, hc = 0;
  for (int w = 0; w < wave; w++) {
    if (w_flag[w]) { carry = w_val[w]; cflag = 1; } else { carry = combP<PROD>(carry, w_val[w]); }
    hc += w_nh[w];
  }
  {
    St ev = shfl_up_st(sv, 1);
    uint32_t ef = __shfl_up(sf, 1, 64), en = __shfl_up(sn, 1, 64);
    if (lane == 0) { ev = identP<PROD>((St*)nullptr); ef = 0; en = 0; }
    if (ef) { carry = ev; cflag = 1; } else { carry = combP<PROD>(carry, ev); }
    hc += en;
  }

  // emit every segment that ends inside this thread's range
  St acc = carry;
  uint32_t k = hc;
#pragma unroll
  for (int j = 0; j < SR_ITEMS; j++) {
    if ((hb >> j) & 1u) {
      if (k == 0) side[tile].first = acc;           // started in an earlier tile
      else emit<T>(outs, G + k - 1, acc);           // complete group
      acc = identP<PROD>((St*)nullptr);
      k++;
    }
    if (ok[j]) accumP<T, PROD>(acc, x[j]);
  }
  if (tid == SR_BLOCK - 1) {
    const bool any = cflag || flag;
    if (any) {
      side[tile].last = acc;
    } else {
      side[tile].first = acc;                       // no head in this tile at all
      side[tile].last = identP<PROD>((St*)nullptr);
    }
    side[tile].has_head = any ? 1u : 0u;
    side[tile].last_gid = G + k - 1;                // meaningful only if any
  }
}

// One wave per tile that contains a head: its trailing open segment is
// completed with the `first` parts of the following tiles up to and including
// the next tile that has a head, then finalised.
template <typename T, bool PROD>
__global__ void __launch_bounds__(256) seg_fixup_kernel(const TileSide<typename VT<T>::St>* side, uint32_t ntiles,
                                                        OutsT<T> outs) {
  typedef typename VT<T>::St St;
  const uint32_t t = blockIdx.x * 4 + wave_id();
  if (t >= ntiles) return;
  if (!side[t].has_head) return;
  const int lane = lane_id();
  St acc = side[t].last;
  uint32_t u = t + 1;
  bool done = (u >= ntiles);
  // common case first: the next tile has a head
  if (!done) {
    const uint32_t hh = side[u].has_head;
    acc = combP<PROD>(acc, side[u].first);
    u++;
    done = hh || (u >= ntiles);
  }
  while (!done) {
    const uint32_t v = u + lane;
    St s = identP<PROD>((St*)nullptr);
    uint32_t hh = 0;
    if (v < ntiles) { s = side[v].first; hh = side[v].has_head; }
    const unsigned long long bal = __ballot(hh != 0);
    const int stop = bal ? (__ffsll((long long)bal) - 1) : 63;   // last lane that contributes
    if (lane > stop) s = identP<PROD>((St*)nullptr);
    // ordered inclusive scan, take the value at lane `stop`
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const St pv = shfl_up_st(s, o);
      if (lane >= o) s = combP<PROD>(pv, s);
    }
    const St tot = shfl_st(s, stop);
    acc = combP<PROD>(acc, tot);
    u += 64;
    done = (bal != 0) || (u >= ntiles);
  }
  if (lane == 0) emit<T>(outs, side[t].last_gid, acc);
}

template <typename T, bool PROD = false>
static int reduce_t(dthip_ctx* ctx, const void* values, const int32_t* ri, const uint8_t* bitmap,
                    const uint32_t* tile_first_head, int64_t nrows, const ReduceOuts& o, int nona) {
  typedef typename VT<T>::St St;
  const uint32_t nt = (uint32_t)((nrows + SR_TILE - 1) / SR_TILE);
  if (nt == 0) return DTHIP_OK;
  Scratch sc(ctx);
  TileSide<St>* side = nullptr;
  DTHIP_TRY(sc.get<TileSide<St>>(nt, &side));
  OutsT<T> outs;
  outs.sum = static_cast<typename VT<T>::SumT*>(o.sum);
  outs.mean = static_cast<typename VT<T>::MeanT*>(o.mean);
  outs.mn = static_cast<T*>(o.mn);
  outs.mx = static_cast<T*>(o.mx);
  outs.count = reinterpret_cast<long long*>(o.count);
  DTHIP_LAUNCH(ctx, "seg_reduce_kernel", (seg_reduce_kernel<T, PROD>), nt, SR_BLOCK, 0,
               static_cast<const T*>(values), ri, bitmap, tile_first_head, (uint32_t)nrows, outs, side, nona);
  DTHIP_LAUNCH(ctx, "seg_fixup_kernel", (seg_fixup_kernel<T, PROD>), (nt + 3) / 4, 256, 0, side, nt, outs);
  return DTHIP_OK;
}

int launch_reduce(dthip_ctx* ctx, const void* values, int stype, const int32_t* rowindex,
                  const uint8_t* bitmap, const uint32_t* tile_first_head, int64_t nrows,
                  const ReduceOuts& outs, int nona) {
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: return reduce_t<int8_t>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, nona);
    case DTHIP_INT16: return reduce_t<int16_t>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, nona);
    case DTHIP_INT32: return reduce_t<int32_t>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, nona);
    case DTHIP_INT64: return reduce_t<long long>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, nona);
    case DTHIP_FLOAT32: return reduce_t<float>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, nona);
    case DTHIP_FLOAT64: return reduce_t<double>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, nona);
    default: set_error("reduce: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
}

// prod() of an integer column (bool / int8..int64 -> int64, wrapping): outs.sum receives the products
int launch_reduce_prod_int(dthip_ctx* ctx, const void* values, int stype, const int32_t* rowindex,
                           const uint8_t* bitmap, const uint32_t* tile_first_head, int64_t nrows, void* out) {
  ReduceOuts outs;
  outs.sum = out;
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: return reduce_t<int8_t, true>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, 0);
    case DTHIP_INT16: return reduce_t<int16_t, true>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, 0);
    case DTHIP_INT32: return reduce_t<int32_t, true>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, 0);
    case DTHIP_INT64: return reduce_t<long long, true>(ctx, values, rowindex, bitmap, tile_first_head, nrows, outs, 0);
    default: set_error("prod: stype %d is not an integer type", stype); return DTHIP_ENOTIMPL;
  }
}

// countna(col) = rows of the group - valid rows (count.h:35-58 with COUNTNA = true): `out` holds count(col) on entry
__global__ void __launch_bounds__(256) countna_kernel(const int32_t* offsets, uint32_t ngroups, long long* out) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g < ngroups) out[g] = (long long)offsets[g + 1] - (long long)offsets[g] - out[g];
}

int launch_countna_from_count(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t* out) {
  if (ngroups == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "countna_kernel", countna_kernel, (unsigned)((ngroups + 255) / 256), 256, 0,
               offsets, (uint32_t)ngroups, reinterpret_cast<long long*>(out));
  return DTHIP_OK;
}

// count(): rows per group = offsets[g+1] - offsets[g]   (count.h:77-88)
__global__ void __launch_bounds__(256) count0_kernel(const int32_t* offsets, uint32_t ngroups, long long* out) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g < ngroups) out[g] = (long long)offsets[g + 1] - (long long)offsets[g];
}

int launch_count0(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t* out) {
  if (ngroups == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "count0_kernel", count0_kernel, (unsigned)((ngroups + 255) / 256), 256, 0,
               offsets, (uint32_t)ngroups, reinterpret_cast<long long*>(out));
  return DTHIP_OK;
}

// sum(float32 column) EXACTLY as the reference computes it (option "f32_sum" = 1): one float accumulator per group,
// the valid rows added one by one in grouped row order (SumProd_ColumnImpl<float>::get_element, column/sumprod.h:48-55).
// One thread per group: sequential by definition, so a single huge group is slow -- it is a reproduction switch, the
// default accumulates in float64 (include/dthip.h, DTHIP_SUM).
__global__ void __launch_bounds__(256) sum_f32_seq_kernel(const float* __restrict__ v, const int32_t* __restrict__ ri,
                                                          const int32_t* __restrict__ offsets, uint32_t ngroups, float* __restrict__ out) {
#pragma clang fp contract(off)
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  float s = 0.0f;
  for (int32_t p = offsets[g]; p < offsets[g + 1]; p++) {
    const int32_t j = ri ? ri[p] : p;
    if (j < 0) continue;
    const float x = v[j];
    if (x == x) s += x;
  }
  out[g] = s;
}

// prod(float32 / float64 column) in the reference's own order: one T accumulator per group starting at 1, the valid rows
// multiplied in one by one (SumProd_ColumnImpl<T, false>::get_element, column/sumprod.h:48-55).  A product that overflows
// or underflows on the way depends on that order, so it is not re-associated: bit-exact, one thread per group.
template <typename T>
__global__ void __launch_bounds__(256) prod_seq_kernel(const T* __restrict__ v, const int32_t* __restrict__ ri,
                                                       const int32_t* __restrict__ offsets, uint32_t ngroups, T* __restrict__ out) {
#pragma clang fp contract(off)
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  T s = T(1);
  for (int32_t p = offsets[g]; p < offsets[g + 1]; p++) {
    const int32_t j = ri ? ri[p] : p;
    if (j < 0) continue;
    const T x = v[j];
    if (x == x) s *= x;
  }
  out[g] = s;
}

int launch_prod_float_seq(dthip_ctx* ctx, const void* values, int stype, const int32_t* ri, const int32_t* offsets, int64_t ngroups, void* out) {
  if (ngroups == 0) return DTHIP_OK;
  const unsigned nb = (unsigned)((ngroups + 255) / 256);
  if (stype == DTHIP_FLOAT32)
    DTHIP_LAUNCH(ctx, "prod_seq_kernel", prod_seq_kernel<float>, nb, 256, 0, static_cast<const float*>(values), ri, offsets, (uint32_t)ngroups, static_cast<float*>(out));
  else if (stype == DTHIP_FLOAT64)
    DTHIP_LAUNCH(ctx, "prod_seq_kernel", prod_seq_kernel<double>, nb, 256, 0, static_cast<const double*>(values), ri, offsets, (uint32_t)ngroups, static_cast<double*>(out));
  else { set_error("prod: stype %d is not a float type", stype); return DTHIP_ENOTIMPL; }
  return DTHIP_OK;
}

int launch_sum_f32_seq(dthip_ctx* ctx, const void* values, const int32_t* ri, const int32_t* offsets, int64_t ngroups, void* out) {
  if (ngroups == 0) return DTHIP_OK;
  DTHIP_LAUNCH(ctx, "sum_f32_seq_kernel", sum_f32_seq_kernel, (unsigned)((ngroups + 255) / 256), 256, 0,
               static_cast<const float*>(values), ri, offsets, (uint32_t)ngroups, static_cast<float*>(out));
  return DTHIP_OK;
}

}  // namespace dthip
