// kbench_offset.hip -- does the relative placement of the output buffer matter for the 64 MiB + 64 MiB stream?
// y = ybase + delta bytes; ring of 5 (x, y) pairs, cold.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/kbench_offset.bin tools/kbench_offset.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k(const float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ scales) {
  const uint32_t row = blockIdx.x;
  const f4* x = reinterpret_cast<const f4*>(xs) + (int64_t)row * 1024 + threadIdx.x;
  f4* y = reinterpret_cast<f4*>(ys) + (int64_t)row * 1024 + threadIdx.x;
  f4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(x + u * 256);
  const float s = scales[row], inv = 1.0f / s;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f4 r;
    r.x = fminf(fmaxf(__builtin_rintf(v[u].x * inv), -128.f), 127.f) * s; r.y = fminf(fmaxf(__builtin_rintf(v[u].y * inv), -128.f), 127.f) * s;
    r.z = fminf(fmaxf(__builtin_rintf(v[u].z * inv), -128.f), 127.f) * s; r.w = fminf(fmaxf(__builtin_rintf(v[u].w * inv), -128.f), 127.f) * s;
    __builtin_nontemporal_store(r, y + u * 256);
  }
}
int main() {
  const size_t bytes = 64u << 20, slack = 8u << 20;
  const int RING = 5;
  char* xb[RING]; char* yb[RING]; float* scales;
  std::vector<float> h(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f * 4.f - 2.f;
  std::vector<float> hs(4096, 1.f / 64);
  for (int r = 0; r < RING; ++r) { CK(hipMalloc(&xb[r], bytes)); CK(hipMalloc(&yb[r], bytes + slack)); CK(hipMemcpy(xb[r], h.data(), bytes, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&scales, 4096 * 4)); CK(hipMemcpy(scales, hs.data(), 4096 * 4, hipMemcpyHostToDevice));
  for (int r = 0; r < RING; ++r) printf("pair %d x=%p y=%p  (y-x) mod 2MiB = %ld\n", r, xb[r], yb[r], (long)(((uintptr_t)yb[r] - (uintptr_t)xb[r]) & ((2u << 20) - 1)));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<size_t> deltas = {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1u << 20, 2u << 20, 4u << 20,
                                768, 1536, 3072, 6144, 12288, 24576, 49152, 98304, 196608, 393216, 786432, 3u << 19};
  std::sort(deltas.begin(), deltas.end());
  for (int round = 0; round < 2; ++round)
  for (size_t d : deltas) {
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, st, (const float*)xb[i % RING], (float*)(yb[i % RING] + d), scales);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    const int iters = 300;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, st, (const float*)xb[i % RING], (float*)(yb[i % RING] + d), scales);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("round %d delta %8zu B : %6.2f us  %5.0f GB/s\n", round, d, ms * 1000 / iters, 2.0 * bytes / (ms * 1000 / iters) / 1e3);
  }
  // per-pair timing at delta 0 (which pairs are fast?)
  for (int r = 0; r < RING; ++r) {
    // cold: touch other pairs in between is not possible per-pair; report warm-ish per pair for relative comparison only
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, st, (const float*)xb[r], (float*)yb[r], scales);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("pair %d alone (warm) %6.2f us\n", r, ms * 1000 / 50);
  }
  return 0;
}
