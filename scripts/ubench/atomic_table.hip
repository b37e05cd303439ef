// microbenchmark: direct global-atomic aggregation  table[key] += value  for 1e9 random keys over a
// 1e7-slot float64 table (80 MB: fits the 256 MB Infinity Cache).  Answers whether a single-pass
// atomic design could beat partition + LDS tables.   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void gen(long long* k, double* v, size_t n, uint32_t range) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t x = i * 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    k[i] = (long long)(x % range); v[i] = (double)(x & 1023) * 0.001;
  }
}
template <int MODE>   // 0: device-scope f64 add, 1: + u32 count, 2: workgroup-scope (L2-local, WRONG across XCDs: speed only)
__global__ void __launch_bounds__(256) agg(const long long* __restrict__ k, const double* __restrict__ v, size_t n, double* t, unsigned* c) {
  size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2, stride = (size_t)gridDim.x * blockDim.x * 2;
  for (; i + 1 < n; i += stride) {
    const longlong2 kk = *reinterpret_cast<const longlong2*>(k + i);
    const double2 vv = *reinterpret_cast<const double2*>(v + i);
    if (MODE == 2) {
      __hip_atomic_fetch_add(&t[kk.x], vv.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&t[kk.y], vv.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      __hip_atomic_fetch_add(&t[kk.x], vv.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&t[kk.y], vv.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 1) { atomicAdd(&c[kk.x], 1u); atomicAdd(&c[kk.y], 1u); }
    }
  }
}
int main(int argc, char** argv) {
  size_t n = argc > 1 ? (size_t)atof(argv[1]) : 1000000000ULL;
  uint32_t range = argc > 2 ? (uint32_t)atof(argv[2]) : 10000000u;
  long long* k; double* v; double* t; unsigned* c;
  hipMalloc(&k, n * 8); hipMalloc(&v, n * 8); hipMalloc(&t, (size_t)range * 8); hipMalloc(&c, (size_t)range * 4);
  gen<<<4096, 256>>>(k, v, n, range);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(t, 0, (size_t)range * 8); hipMemset(c, 0, (size_t)range * 4);
      hipEventRecord(a);
      if (mode == 0) agg<0><<<256 * 8, 256>>>(k, v, n, t, c);
      if (mode == 1) agg<1><<<256 * 8, 256>>>(k, v, n, t, c);
      if (mode == 2) agg<2><<<256 * 8, 256>>>(k, v, n, t, c);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("mode %d rep %d: %.3f ms  (%.3g rows/s)\n", mode, rep, ms, n / (ms * 1e-3));
    }
  }
  std::vector<double> h(16); hipMemcpy(h.data(), t, 128, hipMemcpyDeviceToHost);
  printf("t[0..3] = %g %g %g %g\n", h[0], h[1], h[2], h[3]);
  return 0;
}
