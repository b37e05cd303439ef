// microbenchmark: can a workgroup re-read its own block of keys cheaply (L2 / Infinity Cache) shortly after
// reading it?  Models a "supertile" partition: phase 1 reads the keys of T tiles (12288 rows each, 8 B keys),
// phase 2 re-reads the keys tile by tile together with the 8 B values and writes 10 B/row (8 B value + 2 B key)
// fully coalesced.  Compared with the one-pass stream (keys + values read once, same writes) and with a
// keys-only read: if twice<T> ~= once + (keys-only from HBM), the re-read is free; if it is ~= once + 2 x
// keys-only, the re-read goes to HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int BLOCK = 1024, ITEMS = 12, TILE = BLOCK * ITEMS;

__device__ __forceinline__ uint32_t mix(u32x4 a) { return a.x ^ a.y ^ a.z ^ a.w; }

__device__ __forceinline__ void tile_pass(const u32x4* __restrict__ k, const u32x4* __restrict__ v, u32x4* __restrict__ ov,
                                          uint32_t* __restrict__ ok, size_t tile, int tid, uint32_t& acc) {
  // 12288 rows: keys 6144 x 16 B, values 6144 x 16 B; out values 6144 x 16 B, out keys 12288 x 2 B = 6144 x 4 B
  const size_t base = tile * (TILE / 2);
  u32x4 kk[ITEMS / 2], vv[ITEMS / 2];
#pragma unroll
  for (int q = 0; q < ITEMS / 2; q++) kk[q] = k[base + q * BLOCK + tid];
#pragma unroll
  for (int q = 0; q < ITEMS / 2; q++) vv[q] = v[base + q * BLOCK + tid];
#pragma unroll
  for (int q = 0; q < ITEMS / 2; q++) {
    ov[base + q * BLOCK + tid] = vv[q];
    ok[base + q * BLOCK + tid] = (kk[q].x & 0xFFFFu) | (kk[q].z << 16);
    acc ^= kk[q].y;
  }
}

__global__ void __launch_bounds__(BLOCK) once(const u32x4* k, const u32x4* v, u32x4* ov, uint32_t* ok, uint32_t* sink) {
  uint32_t acc = 0;
  tile_pass(k, v, ov, ok, blockIdx.x, threadIdx.x, acc);
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void __launch_bounds__(BLOCK) keys_only(const u32x4* k, uint32_t* sink) {
  const size_t base = (size_t)blockIdx.x * (TILE / 2);
  uint32_t acc = 0;
#pragma unroll
  for (int q = 0; q < ITEMS / 2; q++) acc ^= mix(k[base + q * BLOCK + threadIdx.x]);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int T>
__global__ void __launch_bounds__(BLOCK) twice(const u32x4* k, const u32x4* v, u32x4* ov, uint32_t* ok, uint32_t* sink, int xcd_contig) {
  __shared__ uint32_t cnt[1024];
  uint32_t bi = blockIdx.x;
  if (xcd_contig) {   // supertiles dealt to XCDs in contiguous ranges
    const uint32_t nt = gridDim.x, xq = nt / 8, xr = nt % 8, xc = bi % 8, q = bi / 8;
    bi = xc * xq + (xc < xr ? xc : xr) + q;
  }
  cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t acc = 0;
  for (int t = 0; t < T; t++) {
    const size_t base = ((size_t)bi * T + t) * (TILE / 2);
    u32x4 kk[ITEMS / 2];
#pragma unroll
    for (int q = 0; q < ITEMS / 2; q++) kk[q] = k[base + q * BLOCK + threadIdx.x];
#pragma unroll
    for (int q = 0; q < ITEMS / 2; q++) { atomicAdd(&cnt[kk[q].x & 1023u], 1u); atomicAdd(&cnt[kk[q].z & 1023u], 1u); }
  }
  __syncthreads();
  acc ^= cnt[threadIdx.x];
  for (int t = 0; t < T; t++) tile_pass(k, v, ov, ok, (size_t)bi * T + t, threadIdx.x, acc);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int T>
static float run_twice(const u32x4* k, const u32x4* v, u32x4* ov, uint32_t* ok, uint32_t* sink, int ntiles, int xc, hipEvent_t a, hipEvent_t b) {
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    float ms;
    (void)hipEventRecord(a);
    twice<T><<<ntiles / T, BLOCK>>>(k, v, ov, ok, sink, xc);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int ntiles = 81920;                       // 1.0066e9 rows
  const size_t rows = (size_t)ntiles * TILE;
  u32x4 *k, *v, *ov; uint32_t *ok, *sink;
  if (hipMalloc(&k, rows * 8) != hipSuccess || hipMalloc(&v, rows * 8) != hipSuccess || hipMalloc(&ov, rows * 8) != hipSuccess ||
      hipMalloc(&ok, rows * 2) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  (void)hipMemset(k, 3, rows * 8); (void)hipMemset(v, 1, rows * 8);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float t_once = 1e9f, t_keys = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    float ms;
    (void)hipEventRecord(a); once<<<ntiles, BLOCK>>>(k, v, ov, ok, sink); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b); if (ms < t_once) t_once = ms;
    (void)hipEventRecord(a); keys_only<<<ntiles, BLOCK>>>(k, sink); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b); if (ms < t_keys) t_keys = ms;
  }
  printf("rows %zu\nonce (16 R + 10 W per row): %.3f ms = %.2f TB/s\nkeys only (8 R): %.3f ms = %.2f TB/s\n", rows, t_once,
         rows * 26.0 / t_once / 1e9, t_keys, rows * 8.0 / t_keys / 1e9);
  for (int xc = 0; xc < 2; xc++) {
    printf("twice, supertiles %s:\n", xc ? "in XCD-contiguous ranges" : "round-robin over XCDs");
    printf("  T=1  (96 KB of keys per WG)  %.3f ms\n", run_twice<1>(k, v, ov, ok, sink, ntiles, xc, a, b));
    printf("  T=2  (192 KB) %.3f ms\n", run_twice<2>(k, v, ov, ok, sink, ntiles, xc, a, b));
    printf("  T=4  (384 KB) %.3f ms\n", run_twice<4>(k, v, ov, ok, sink, ntiles, xc, a, b));
    printf("  T=8  (768 KB) %.3f ms\n", run_twice<8>(k, v, ov, ok, sink, ntiles, xc, a, b));
    printf("  T=16 (1.5 MB) %.3f ms\n", run_twice<16>(k, v, ov, ok, sink, ntiles, xc, a, b));
    printf("  T=64 (6 MB)   %.3f ms\n", run_twice<64>(k, v, ov, ok, sink, ntiles, xc, a, b));
  }
  return 0;
}
