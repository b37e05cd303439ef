// stats.hip -- min / max / valid count of an integer key column.
//
// Reference: NumericStats<T>::compute_minmax (src/core/stats.cc:601-640), called
// from SortContext::_initI (src/core/sort.cc:731-732) to find the key range that
// fixes the number of significant radix bits.  One streaming pass over the
// column; per-workgroup partials are merged with 64-bit integer atomics.
#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

typedef uint32_t su32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void mm_acc(T v, T na, long long& mn, long long& mx, long long& cn) {
  if (v != na) {
    const long long x = (long long)v;
    mn = x < mn ? x : mn;
    mx = x > mx ? x : mx;
    cn++;
  }
}

// VEC: the column base is 16-byte aligned -> 16-byte loads, two in flight per thread
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) minmax_kernel(const T* __restrict__ data, uint32_t n, T na, MinMax* out) {
  __shared__ long long smn[4], smx[4], scn[4];
  long long mn = INT64_MAX, mx = INT64_MIN, cn = 0;
  const uint32_t stride = gridDim.x * 256;
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  if (VEC) {
    constexpr uint32_t E = 16 / sizeof(T);
    const uint32_t nvec = n / E;
    const su32x4* src = reinterpret_cast<const su32x4*>(data);
    uint32_t i = gid;
    for (; i + stride < nvec; i += 2 * stride) {
      const su32x4 a = src[i], b = src[i + stride];
      const T* ta = reinterpret_cast<const T*>(&a);
      const T* tb = reinterpret_cast<const T*>(&b);
#pragma unroll
      for (uint32_t j = 0; j < E; j++) { mm_acc<T>(ta[j], na, mn, mx, cn); mm_acc<T>(tb[j], na, mn, mx, cn); }
    }
    if (i < nvec) {
      const su32x4 a = src[i];
      const T* ta = reinterpret_cast<const T*>(&a);
#pragma unroll
      for (uint32_t j = 0; j < E; j++) mm_acc<T>(ta[j], na, mn, mx, cn);
    }
    for (uint32_t k = nvec * E + gid; k < n; k += stride) mm_acc<T>(data[k], na, mn, mx, cn);
  } else {
    for (uint32_t i = gid; i < n; i += stride) mm_acc<T>(data[i], na, mn, mx, cn);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long omn = (long long)shfl_u64((unsigned long long)mn, lane_id() ^ o);
    const long long omx = (long long)shfl_u64((unsigned long long)mx, lane_id() ^ o);
    const long long ocn = (long long)shfl_u64((unsigned long long)cn, lane_id() ^ o);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
    cn += ocn;
  }
  if (lane_id() == 0) { smn[wave_id()] = mn; smx[wave_id()] = mx; scn[wave_id()] = cn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      mn = smn[w] < mn ? smn[w] : mn;
      mx = smx[w] > mx ? smx[w] : mx;
      cn += scn[w];
    }
    if (cn) {
      atomicMin(&out->mn, mn);
      atomicMax(&out->mx, mx);
      atomicAdd(reinterpret_cast<unsigned long long*>(&out->nvalid), (unsigned long long)cn);
    }
  }
}

// Range GUESS from a sample: nsamp evenly spaced pieces of 16 bytes plus the first and the last
// piece (sorted columns keep their extremes there).  The bucketed aggregation widens the guess by
// a margin and its histogram pass verifies every row against it (bucket.hip), so a wrong guess
// costs a second attempt with the exact range, never a wrong result.
template <typename T>
__global__ void __launch_bounds__(256) minmax_sample_kernel(const T* __restrict__ data, uint32_t n, uint32_t nsamp, T na,
                                                            MinMax* out) {
  __shared__ long long smn[4], smx[4], scn[4];
  constexpr uint32_t E = 16 / sizeof(T);
  long long mn = INT64_MAX, mx = INT64_MIN, cn = 0;
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  if (gid < nsamp) {
    uint32_t r0 = (uint32_t)(((unsigned long long)gid * n) / nsamp);
    if (gid + 1 == nsamp) r0 = n > E ? n - E : 0;
#pragma unroll
    for (uint32_t j = 0; j < E; j++) if (r0 + j < n) mm_acc<T>(data[r0 + j], na, mn, mx, cn);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long omn = (long long)shfl_u64((unsigned long long)mn, lane_id() ^ o);
    const long long omx = (long long)shfl_u64((unsigned long long)mx, lane_id() ^ o);
    const long long ocn = (long long)shfl_u64((unsigned long long)cn, lane_id() ^ o);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
    cn += ocn;
  }
  if (lane_id() == 0) { smn[wave_id()] = mn; smx[wave_id()] = mx; scn[wave_id()] = cn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      mn = smn[w] < mn ? smn[w] : mn;
      mx = smx[w] > mx ? smx[w] : mx;
      cn += scn[w];
    }
    if (cn) {
      atomicMin(&out->mn, mn);
      atomicMax(&out->mx, mx);
      atomicAdd(reinterpret_cast<unsigned long long*>(&out->nvalid), (unsigned long long)cn);
    }
  }
}

__global__ void minmax_init_kernel(MinMax* out) {
  out->mn = INT64_MAX; out->mx = INT64_MIN; out->nvalid = 0;
}

int launch_minmax(dthip_ctx* ctx, const void* data, int stype, int64_t n, MinMax* d_out) {
  DTHIP_LAUNCH(ctx, "minmax_init_kernel", minmax_init_kernel, 1, 1, 0, d_out);
  if (n == 0) return DTHIP_OK;
  long long blocks = (n + 256 * 32 - 1) / (256 * 32);
  if (blocks > ctx->num_cus * 8) blocks = ctx->num_cus * 8;
  const unsigned g = (unsigned)blocks;
  const uint32_t nn = (uint32_t)n;
  const bool vec = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
#define MM_LAUNCH(T, NA)                                                                                     \
  do {                                                                                                        \
    if (vec) { DTHIP_LAUNCH(ctx, "minmax_kernel", (minmax_kernel<T, true>), g, 256, 0,                        \
                            static_cast<const T*>(data), nn, (T)(NA), d_out); }                               \
    else { DTHIP_LAUNCH(ctx, "minmax_kernel", (minmax_kernel<T, false>), g, 256, 0,                           \
                        static_cast<const T*>(data), nn, (T)(NA), d_out); }                                   \
  } while (0)
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: MM_LAUNCH(int8_t, INT8_MIN); break;
    case DTHIP_INT16: MM_LAUNCH(int16_t, INT16_MIN); break;
    case DTHIP_INT32: MM_LAUNCH(int32_t, INT32_MIN); break;
    case DTHIP_INT64: MM_LAUNCH(long long, INT64_MIN); break;
    default: set_error("minmax: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
#undef MM_LAUNCH
  return DTHIP_OK;
}

int launch_minmax_sample(dthip_ctx* ctx, const void* data, int stype, int64_t n, uint32_t nsamp, MinMax* d_out) {
  DTHIP_LAUNCH(ctx, "minmax_init_kernel", minmax_init_kernel, 1, 1, 0, d_out);
  if (n == 0) return DTHIP_OK;
  const unsigned g = (nsamp + 255) / 256;
  const uint32_t nn = (uint32_t)n;
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8:
      DTHIP_LAUNCH(ctx, "minmax_sample_kernel", minmax_sample_kernel<int8_t>, g, 256, 0, static_cast<const int8_t*>(data), nn, nsamp, (int8_t)INT8_MIN, d_out);
      break;
    case DTHIP_INT16:
      DTHIP_LAUNCH(ctx, "minmax_sample_kernel", minmax_sample_kernel<int16_t>, g, 256, 0, static_cast<const int16_t*>(data), nn, nsamp, (int16_t)INT16_MIN, d_out);
      break;
    case DTHIP_INT32:
      DTHIP_LAUNCH(ctx, "minmax_sample_kernel", minmax_sample_kernel<int32_t>, g, 256, 0, static_cast<const int32_t*>(data), nn, nsamp, (int32_t)INT32_MIN, d_out);
      break;
    case DTHIP_INT64:
      DTHIP_LAUNCH(ctx, "minmax_sample_kernel", minmax_sample_kernel<long long>, g, 256, 0, static_cast<const long long*>(data), nn, nsamp, (long long)INT64_MIN, d_out);
      break;
    default: set_error("minmax: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
  return DTHIP_OK;
}

}  // namespace dthip
