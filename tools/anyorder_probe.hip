// anyorder_probe.hip -- does hipExtAnyOrderLaunch do anything on gfx950?  (not part of the product)
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; nobody had tried.  If honoured, a launch would not wait for
// the previous packets of its queue: the 1.4 us bubble between back-to-back launches of the headline kernel (and the
// overlap of one launch's drain with the next one's ramp) would be the caller's to give away when launches are independent.
// The probe streams a config-2-sized quantize (4096 x 4096 float32, 16 B per lane, 4 per lane and tile) over a cold ring of 5
// buffer pairs, 400 launches, through hipExtLaunchKernelGGL with flags 0 and with hipExtAnyOrderLaunch, and reports the
// launch period; a dependent chain (launch k+1 reads what launch k wrote) shows whether order is actually relaxed.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/anyorder_probe tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void quant(const float* __restrict__ x, float* __restrict__ y, float s, float inv) {
  const int64_t base = ((int64_t)blockIdx.x * 1024 + threadIdx.x) * 4;
  f4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(x + base + u * 1024));
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_fmed3f(__builtin_rintf(v[u][i] * inv), -128.f, 127.f) * s;
    __builtin_nontemporal_store(r, reinterpret_cast<f4*>(y + base + u * 1024));
  }
}

// y = x + 1 over n floats: a chain of these is order-sensitive
__global__ __launch_bounds__(256) void inc(const float* __restrict__ x, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  y[i] = x[i] + 1.0f;
}

int main() {
  const int64_t n = 4096ll * 4096;
  const int ring = 5, launches = 400;
  std::vector<float*> xs(ring), ys(ring);
  for (int i = 0; i < ring; ++i) { CK(hipMalloc(&xs[i], n * 4)); CK(hipMalloc(&ys[i], n * 4)); CK(hipMemset(xs[i], 0x3c, n * 4)); }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned blocks = (unsigned)(n / 4096);
  for (int pass = 0; pass < 2; ++pass) {                      // the ordinary launch macro, for reference
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(quant, dim3(blocks), dim3(256), 0, st, (const float*)xs[i % ring], ys[i % ring], 0.02f, 50.f);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(quant, dim3(blocks), dim3(256), 0, st, (const float*)xs[i % ring], ys[i % ring], 0.02f, 50.f);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipLaunchKernelGGL (what the library uses by default): %.2f us per launch (%.0f GB/s)\n", ms * 1e3 / launches, n * 8.0 / (ms * 1e-3 / launches) / 1e9);
  }
  {                                                           // the same through 4x larger launches (two buffers of 256 MiB each way)
    float *bx, *by; CK(hipMalloc(&bx, 4 * n * 4)); CK(hipMalloc(&by, 4 * n * 4)); CK(hipMemset(bx, 0x3c, 4 * n * 4));
    float *bx2, *by2; CK(hipMalloc(&bx2, 4 * n * 4)); CK(hipMalloc(&by2, 4 * n * 4)); CK(hipMemset(bx2, 0x3c, 4 * n * 4));
    for (int variant = 0; variant < 3; ++variant) {
      auto go = [&](int i) {
        const float* x = i & 1 ? bx2 : bx; float* y = i & 1 ? by2 : by;
        if (variant == 0) hipLaunchKernelGGL(quant, dim3(4 * blocks), dim3(256), 0, st, x, y, 0.02f, 50.f);
        else hipExtLaunchKernelGGL(quant, dim3(4 * blocks), dim3(256), 0, st, nullptr, nullptr, variant == 2 ? (uint32_t)hipExtAnyOrderLaunch : 0u, x, y, 0.02f, 50.f);
      };
      for (int i = 0; i < 300; ++i) go(i);
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 100; ++i) go(i);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("512 MiB launches, %s: %.2f us per launch (%.0f GB/s)\n", variant == 0 ? "hipLaunchKernelGGL" : variant == 1 ? "ext, flags 0" : "ext, any order", ms * 1e3 / 100, 4 * n * 8.0 / (ms * 1e-3 / 100) / 1e9);
    }
    CK(hipFree(bx)); CK(hipFree(by)); CK(hipFree(bx2)); CK(hipFree(by2));
  }
  for (int pass = 0; pass < 3; ++pass)
    for (uint32_t flags : {0u, (uint32_t)hipExtAnyOrderLaunch}) {
      for (int i = 0; i < 2000; ++i) hipExtLaunchKernelGGL(quant, dim3(blocks), dim3(256), 0, st, nullptr, nullptr, flags, (const float*)xs[i % ring], ys[i % ring], 0.02f, 50.f);
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < launches; ++i) hipExtLaunchKernelGGL(quant, dim3(blocks), dim3(256), 0, st, nullptr, nullptr, flags, (const float*)xs[i % ring], ys[i % ring], 0.02f, 50.f);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("flags %s: %.2f us per launch (%.0f GB/s)\n", flags ? "hipExtAnyOrderLaunch" : "0 (in order)      ", ms * 1e3 / launches, n * 8.0 / (ms * 1e-3 / launches) / 1e9);
    }
  // order check: 64 dependent increments of a small buffer; in order the result is 64 everywhere
  const int64_t m = 1 << 20;
  float *a, *b; CK(hipMalloc(&a, m * 4)); CK(hipMalloc(&b, m * 4));
  for (uint32_t flags : {0u, (uint32_t)hipExtAnyOrderLaunch}) {
    int wrong_runs = 0;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipMemsetAsync(a, 0, m * 4, st));
      for (int k = 0; k < 64; ++k) hipExtLaunchKernelGGL(inc, dim3((unsigned)(m / 256)), dim3(256), 0, st, nullptr, nullptr, flags, (const float*)(k % 2 ? b : a), k % 2 ? a : b);
      CK(hipStreamSynchronize(st));
      std::vector<float> h(m);
      CK(hipMemcpy(h.data(), a, m * 4, hipMemcpyDeviceToHost));
      int64_t bad = 0;
      for (int64_t i = 0; i < m; ++i) bad += h[i] != 64.0f;
      wrong_runs += bad != 0;
    }
    printf("dependent chain of 64 launches, flags %s: %d of 20 runs with a wrong result%s\n", flags ? "hipExtAnyOrderLaunch" : "0", wrong_runs,
           flags ? (wrong_runs ? "  -> order IS relaxed" : "  -> order kept (flag ignored, or the runtime still serialises)") : "");
  }
  return 0;
}
