// arrow.hip -- Arrow-layout fixed-width columns -> the sentinel layout of every other entry point, ON THE DEVICE.
//
// The reference keeps a column that came from an Arrow table as two buffers, a validity bitmap and the values
// (ArrowFw_ColumnImpl, src/core/column/arrow_fw.cc:63-72: element i is valid when the bitmap is absent or bit
// `validity[i / 8] & (1 << (i & 7))` is set; booleans are bit-packed too, ArrowBool_ColumnImpl, column/arrow_bool.cc;
// built by Column::from_arrow, column_from_arrow.cc:55-59).  Before such a column can be grouped or reduced by anything
// that wants SentinelFw data the reference materialises it element by element on the CPU.  Here the two buffers go to
// HBM as they are (1/8 + 1 x the column's bytes over PCIe in host mode, nothing in device mode) and one streaming kernel
// writes the values with the stype's NA sentinel at the invalid rows: 16-byte loads and stores, a validity byte per 8 rows.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "device_utils.hpp"

namespace dthip {

template <typename T> __device__ __forceinline__ T arrow_na();
template <> __device__ __forceinline__ int8_t arrow_na<int8_t>() { return INT8_MIN; }
template <> __device__ __forceinline__ int16_t arrow_na<int16_t>() { return INT16_MIN; }
template <> __device__ __forceinline__ int32_t arrow_na<int32_t>() { return INT32_MIN; }
template <> __device__ __forceinline__ int64_t arrow_na<int64_t>() { return INT64_MIN; }
// floats travel as their bit patterns (a NaN payload must not be "canonicalised" by a float move): the NA is the quiet
// NaN the reference writes, GETNA<float>() / GETNA<double>() = std::numeric_limits<T>::quiet_NaN() (stype.h:186-197)
struct f32bits { uint32_t u; };
struct f64bits { unsigned long long u; };
template <> __device__ __forceinline__ f32bits arrow_na<f32bits>() { return f32bits{0x7FC00000u}; }
template <> __device__ __forceinline__ f64bits arrow_na<f64bits>() { return f64bits{0x7FF8000000000000ull}; }

// V = 16 / sizeof(T) rows per thread: one 16-byte load, the V validity bits out of one or two bitmap bytes, one 16-byte
// store.  V divides 8 or is 16, and a thread's first row is a multiple of V, so its bits never straddle more than the
// bytes read here.  The last (partial) vector of the column goes row by row.
template <typename T>
__global__ void __launch_bounds__(256) arrow_fw_kernel(const T* __restrict__ values, const uint8_t* __restrict__ validity,
                                                       uint64_t n, T* __restrict__ out) {
  constexpr int V = 16 / (int)sizeof(T);
  const uint64_t v0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (v0 >= n) return;
  uint32_t bits;
  if (V <= 8) bits = ((uint32_t)validity[v0 >> 3] >> (v0 & 7)) & ((1u << V) - 1u);
  else bits = (uint32_t)validity[v0 >> 3] | ((v0 + 8 < n ? (uint32_t)validity[(v0 >> 3) + 1] : 0u) << 8);
  if (v0 + V <= n) {
    union { uint4 q; T e[V]; } u;
    u.q = *reinterpret_cast<const uint4*>(values + v0);
#pragma unroll
    for (int j = 0; j < V; j++) if (!((bits >> j) & 1u)) u.e[j] = arrow_na<T>();
    *reinterpret_cast<uint4*>(out + v0) = u.q;
  } else {
    for (int j = 0; v0 + j < n; j++) out[v0 + j] = ((bits >> j) & 1u) ? values[v0 + j] : arrow_na<T>();
  }
}

// Arrow booleans: value bit i and validity bit i -> bool8 (0 / 1 / NA = -128).  16 rows per thread: two bytes of each
// bitmap in, one 16-byte store out.
__global__ void __launch_bounds__(256) arrow_bool_kernel(const uint8_t* __restrict__ valbits, const uint8_t* __restrict__ validity,
                                                         uint64_t n, int8_t* __restrict__ out) {
  const uint64_t v0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (v0 >= n) return;
  const bool two = v0 + 8 < n;
  const uint32_t d = (uint32_t)valbits[v0 >> 3] | ((two ? (uint32_t)valbits[(v0 >> 3) + 1] : 0u) << 8);
  const uint32_t m = validity ? ((uint32_t)validity[v0 >> 3] | ((two ? (uint32_t)validity[(v0 >> 3) + 1] : 0u) << 8)) : 0xFFFFu;
  if (v0 + 16 <= n) {
    union { uint4 q; int8_t e[16]; } u;
#pragma unroll
    for (int j = 0; j < 16; j++) u.e[j] = ((m >> j) & 1u) ? (int8_t)((d >> j) & 1u) : (int8_t)INT8_MIN;
    *reinterpret_cast<uint4*>(out + v0) = u.q;
  } else {
    for (int j = 0; v0 + j < n; j++) out[v0 + j] = ((m >> j) & 1u) ? (int8_t)((d >> j) & 1u) : (int8_t)INT8_MIN;
  }
}

template <typename T>
static int launch_fw(dthip_ctx* ctx, const void* values, const uint8_t* validity, int64_t n, void* dst) {
  constexpr int V = 16 / (int)sizeof(T);
  const uint64_t vecs = ((uint64_t)n + V - 1) / V;
  DTHIP_LAUNCH(ctx, "arrow_fw_kernel", arrow_fw_kernel<T>, (unsigned)((vecs + 255) / 256), 256, 0, static_cast<const T*>(values), validity,
               (uint64_t)n, static_cast<T*>(dst));
  return DTHIP_OK;
}

}  // namespace dthip

using namespace dthip;

extern "C" int dthip_from_arrow(dthip_ctx* ctx, const void* values, const uint8_t* validity, int64_t nrows, int stype, int mem,
                                void* dst) {
  if (!ctx || !dst || (!values && nrows > 0)) { set_error("null argument"); return DTHIP_EINVAL; }
  if (nrows < 0 || nrows > INT32_MAX) { set_error("nrows %lld outside [0, 2^31-1]", (long long)nrows); return DTHIP_EINVAL; }
  size_t esz = 0;
  switch (stype) {
    case DTHIP_BOOL: esz = 0; break;
    case DTHIP_INT8: esz = 1; break;
    case DTHIP_INT16: esz = 2; break;
    case DTHIP_INT32: case DTHIP_FLOAT32: esz = 4; break;
    case DTHIP_INT64: case DTHIP_FLOAT64: esz = 8; break;
    default: set_error("unsupported stype %d", stype); return DTHIP_ENOTIMPL;
  }
  if (nrows == 0) return DTHIP_OK;
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  const size_t bm_bytes = ((size_t)nrows + 7) / 8;
  const size_t val_bytes = stype == DTHIP_BOOL ? bm_bytes : (size_t)nrows * esz;
  if ((reinterpret_cast<uintptr_t>(dst) & 15u) != 0) { set_error("dst must be 16-byte aligned"); return DTHIP_EINVAL; }
  const void* dval = values;
  const uint8_t* dbm = validity;
  void* stage_v = nullptr; void* stage_b = nullptr;
  int rc = DTHIP_OK;
  auto cleanup = [&]() { if (stage_v) dev_release(ctx, stage_v); if (stage_b) dev_release(ctx, stage_b); };
  if (mem == DTHIP_HOST) {
    // the column's two buffers cross PCIe as they are; a column WITHOUT nulls lands in dst directly (no kernel at all)
    if (!validity && stype != DTHIP_BOOL) {
      DTHIP_CHECK_HIP(hipMemcpyAsync(dst, values, val_bytes, hipMemcpyHostToDevice, ctx->stream));
      DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      return DTHIP_OK;
    }
    rc = dev_alloc(ctx, val_bytes, &stage_v);
    if (rc == DTHIP_OK && validity) rc = dev_alloc(ctx, bm_bytes, &stage_b);
    if (rc != DTHIP_OK) { cleanup(); return rc; }
    if (hipMemcpyAsync(stage_v, values, val_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        (validity && hipMemcpyAsync(stage_b, validity, bm_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)) {
      set_error("host -> device copy of an Arrow buffer failed: %s", hipGetErrorString(hipGetLastError()));
      cleanup();
      return DTHIP_EDEVICE;
    }
    dval = stage_v; dbm = static_cast<const uint8_t*>(stage_b);
  } else if (mem != DTHIP_DEVICE) {
    set_error("mem must be DTHIP_HOST or DTHIP_DEVICE"); return DTHIP_EINVAL;
  } else {
    if (!validity && stype != DTHIP_BOOL) {
      if (dst != values) DTHIP_CHECK_HIP(hipMemcpyAsync(dst, values, val_bytes, hipMemcpyDeviceToDevice, ctx->stream));
      return DTHIP_OK;
    }
    if (stype != DTHIP_BOOL && (reinterpret_cast<uintptr_t>(values) & 15u) != 0) {
      set_error("device values buffer must be 16-byte aligned"); return DTHIP_EINVAL;
    }
  }
  auto run = [&]() -> int {
    switch (stype) {
      case DTHIP_BOOL: {
        const uint64_t vecs = ((uint64_t)nrows + 15) / 16;
        DTHIP_LAUNCH(ctx, "arrow_bool_kernel", arrow_bool_kernel, (unsigned)((vecs + 255) / 256), 256, 0,
                     static_cast<const uint8_t*>(dval), dbm, (uint64_t)nrows, static_cast<int8_t*>(dst));
        return DTHIP_OK;
      }
      case DTHIP_INT8: return launch_fw<int8_t>(ctx, dval, dbm, nrows, dst);
      case DTHIP_INT16: return launch_fw<int16_t>(ctx, dval, dbm, nrows, dst);
      case DTHIP_INT32: return launch_fw<int32_t>(ctx, dval, dbm, nrows, dst);
      case DTHIP_FLOAT32: return launch_fw<f32bits>(ctx, dval, dbm, nrows, dst);
      case DTHIP_INT64: return launch_fw<int64_t>(ctx, dval, dbm, nrows, dst);
      default: return launch_fw<f64bits>(ctx, dval, dbm, nrows, dst);
    }
  };
  rc = run();
  if (mem == DTHIP_HOST) {
    // staged buffers go back to the cache only when the kernel is done with them (the cache is stream-ordered, and this
    // call is synchronous for host data like every other host-mode entry point)
    if (rc == DTHIP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
      set_error("arrow conversion failed: %s", hipGetErrorString(hipGetLastError())); rc = DTHIP_EDEVICE;
    }
    cleanup();
  }
  return rc;
}
