// microbenchmark: what this MI355X sustains for a pure 16-byte-per-lane streaming read, and for a copy
// (read + write), over 8 GB -- the practical ceilings the HBM-bound kernels of DESIGN.md are compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) rd(const u32x4* __restrict__ p, size_t nv, uint32_t* out) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  uint32_t acc = 0;
  for (; i + 3 * stride < nv; i += 4 * stride) {
    const u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < nv; i += stride) { const u32x4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) cp(const u32x4* __restrict__ p, u32x4* __restrict__ q, size_t nv) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (; i < nv; i += stride) q[i] = p[i];
}
int main() {
  const size_t bytes = 8ULL << 30, nv = bytes / 16;
  u32x4 *p, *q; uint32_t* o;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&q, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
  if (hipMemset(p, 1, bytes) != hipSuccess) return 1;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int grid : {2048, 4096, 8192, 16384}) {
    float best_r = 1e9f, best_c = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      float ms;
      (void)hipEventRecord(a); rd<<<grid, 256>>>(p, nv, o); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      (void)hipEventElapsedTime(&ms, a, b); if (ms < best_r) best_r = ms;
      (void)hipEventRecord(a); cp<<<grid, 256>>>(p, q, nv); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      (void)hipEventElapsedTime(&ms, a, b); if (ms < best_c) best_c = ms;
    }
    printf("grid %5d: read %.3f ms = %.2f TB/s   copy %.3f ms = %.2f TB/s (read+write)\n", grid, best_r, bytes / best_r / 1e9,
           best_c, 2.0 * bytes / best_c / 1e9);
  }
  return 0;
}
