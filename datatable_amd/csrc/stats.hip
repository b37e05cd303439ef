// stats.hip -- min / max / valid count of an integer key column.
//
// Reference: NumericStats<T>::compute_minmax (src/core/stats.cc:601-640), called
// from SortContext::_initI (src/core/sort.cc:731-732) to find the key range that
// fixes the number of significant radix bits.  One streaming pass over the
// column; per-workgroup partials are merged with 64-bit integer atomics.
#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

template <typename T>
__global__ void __launch_bounds__(256) minmax_kernel(const T* __restrict__ data, uint32_t n, T na, MinMax* out) {
  __shared__ long long smn[4], smx[4], scn[4];
  long long mn = INT64_MAX, mx = INT64_MIN, cn = 0;
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const T v = data[i];
    if (v != na) {
      const long long x = (long long)v;
      mn = x < mn ? x : mn;
      mx = x > mx ? x : mx;
      cn++;
    }
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
  long long blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > ctx->num_cus * 8) blocks = ctx->num_cus * 8;
  const unsigned g = (unsigned)blocks;
  const uint32_t nn = (uint32_t)n;
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8:
      DTHIP_LAUNCH(ctx, "minmax_kernel", minmax_kernel<int8_t>, g, 256, 0, static_cast<const int8_t*>(data), nn,
                   (int8_t)INT8_MIN, d_out);
      break;
    case DTHIP_INT16:
      DTHIP_LAUNCH(ctx, "minmax_kernel", minmax_kernel<int16_t>, g, 256, 0, static_cast<const int16_t*>(data), nn,
                   (int16_t)INT16_MIN, d_out);
      break;
    case DTHIP_INT32:
      DTHIP_LAUNCH(ctx, "minmax_kernel", minmax_kernel<int32_t>, g, 256, 0, static_cast<const int32_t*>(data), nn,
                   (int32_t)INT32_MIN, d_out);
      break;
    case DTHIP_INT64:
      DTHIP_LAUNCH(ctx, "minmax_kernel", minmax_kernel<long long>, g, 256, 0, static_cast<const long long*>(data), nn,
                   (long long)INT64_MIN, d_out);
      break;
    default: set_error("minmax: unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
  return DTHIP_OK;
}

}  // namespace dthip
