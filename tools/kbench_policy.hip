// kbench_policy.hip -- cache-policy bits and block->tile permutations for the cfg2 stream (64 MiB in, 64 MiB out, cold ring)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float fq(float x, float s, float inv) { return fminf(fmaxf(__builtin_rintf(x * inv), -128.f), 127.f) * s; }

// LA/SA: aux (cache policy) immediates of the raw buffer load/store
template <int LA, int SA>
__global__ __launch_bounds__(256) void k_buf(const float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ scales, uint32_t mul, uint32_t ntiles) {
  const uint32_t t = (uint32_t)(((uint64_t)blockIdx.x * mul) % ntiles);      // block -> row permutation (mul = 1: identity)
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, 0x7fffffff, 0x27000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)ys, 0, 0x7fffffff, 0x27000);
  const uint32_t base = t * 16384u + threadIdx.x * 16u;     // bytes
  i4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, base + u * 4096u, 0, LA);
  const float s = scales[t], inv = 1.0f / s;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f4 f = __builtin_bit_cast(f4, v[u]);
    f4 r; r.x = fq(f.x, s, inv); r.y = fq(f.y, s, inv); r.z = fq(f.z, s, inv); r.w = fq(f.w, s, inv);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, r), ry, base + u * 4096u, 0, SA);
  }
}
struct V { const char* name; void (*l)(const float*, float*, const float*, uint32_t, hipStream_t); uint32_t mul; };
template <int LA, int SA> void L(const float* x, float* y, const float* s, uint32_t mul, hipStream_t st) {
  hipLaunchKernelGGL((k_buf<LA, SA>), dim3(4096), dim3(256), 0, st, x, y, s, mul, 4096u);
}
int main() {
  const size_t bytes = 64u << 20; const int RING = 5;
  float *x[RING], *y[RING], *scales;
  std::vector<float> h(bytes / 4); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f * 4.f - 2.f;
  std::vector<float> hs(4096, 1.f / 64);
  for (int r = 0; r < RING; ++r) { CK(hipMalloc(&x[r], bytes)); CK(hipMalloc(&y[r], bytes)); CK(hipMemcpy(x[r], h.data(), bytes, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&scales, 4096 * 4)); CK(hipMemcpy(scales, hs.data(), 4096 * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // aux bits (gfx940+): 1 = sc0, 2 = nt, 16 = sc1
  std::vector<V> vs = {
    {"ld 0      st 0     ", L<0, 0>, 1}, {"ld nt     st nt    ", L<2, 2>, 1}, {"ld nt     st 0     ", L<2, 0>, 1}, {"ld 0      st nt    ", L<0, 2>, 1},
    {"ld sc1    st sc1   ", L<16, 16>, 1}, {"ld nt sc1 st nt sc1", L<18, 18>, 1}, {"ld nt     st sc0sc1", L<2, 17>, 1}, {"ld nt     st nt sc1", L<2, 18>, 1},
    {"ld nt sc1 st nt    ", L<18, 2>, 1}, {"ld sc0    st nt    ", L<1, 2>, 1}, {"ld nt sc0 st nt    ", L<3, 2>, 1}, {"ld nt     st nt sc0", L<2, 3>, 1},
    {"ld nt st nt  perm x3   ", L<2, 2>, 3}, {"ld nt st nt  perm x7   ", L<2, 2>, 7}, {"ld nt st nt  perm x33  ", L<2, 2>, 33}, {"ld nt st nt  perm x257 ", L<2, 2>, 257},
    {"ld nt st nt  perm x1021", L<2, 2>, 1021}, {"ld nt st nt  perm x2049", L<2, 2>, 2049},
  };
  std::vector<std::vector<float>> res(vs.size());
  for (int round = 0; round < 3; ++round)
    for (size_t vi = 0; vi < vs.size(); ++vi) {
      for (int i = 0; i < 10; ++i) vs[vi].l(x[i % RING], y[i % RING], scales, vs[vi].mul, st);
      CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st));
      const int iters = 300;
      for (int i = 0; i < iters; ++i) vs[vi].l(x[i % RING], y[i % RING], scales, vs[vi].mul, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); res[vi].push_back(ms * 1000 / iters);
    }
  for (size_t vi = 0; vi < vs.size(); ++vi) { std::sort(res[vi].begin(), res[vi].end()); printf("%-26s med %6.2f us  %5.0f GB/s\n", vs[vi].name, res[vi][1], 2.0 * bytes / res[vi][1] / 1e3); }
  return 0;
}
