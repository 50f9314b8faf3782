// launch_probe.hip -- host cost per kernel launch on this machine, by launch API (not part of the product).
// The per-tensor activation path is launch-bound (BASELINE config 3, N <= 8): what remains of a call after the
// compiled binding is the HIP launch itself.  Which entry point of the runtime is cheapest for a kernel with the
// flat kernel's argument list (two small structs, two pointers, a count)?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/launch_probe.bin tools/launch_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Op { const float* a; const int* b; float lo, hi; };
struct Param { float s, inv, z; };

__global__ __launch_bounds__(256) void k(Op op, Param p, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = fminf(fmaxf(__builtin_rintf(x[i] * p.inv) + p.z, op.lo), op.hi) * p.s;
}

template <class F>
double per_call_us(F f, int n, hipStream_t st) {
  for (int i = 0; i < 2000; ++i) f();
  CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  auto t1 = std::chrono::steady_clock::now();
  CK(hipStreamSynchronize(st));
  auto t2 = std::chrono::steady_clock::now();
  printf("   host %.2f us/call, incl. drain %.2f us/call\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
         std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
  return 0;
}

int main() {
  const int64_t n = 150528;   // config 3, N = 1
  float *x, *y;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMemset(x, 0, n * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  Op op{nullptr, nullptr, 0.f, 255.f};
  Param p{0.02f, 50.f, 114.f};
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const int N = 200000;
  printf("hipLaunchKernelGGL (what libmctq_hip.so uses)\n");
  per_call_us([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, st, op, p, (const float*)x, y, n); }, N, st);
  printf("hipLaunchKernelGGL + hipGetLastError\n");
  per_call_us([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, st, op, p, (const float*)x, y, n); (void)hipGetLastError(); }, N, st);
  hipFunction_t fn;
  CK(hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(&k)));
  const float* xc = x;
  int64_t nn = n;
  void* params[] = {&op, &p, &xc, &y, &nn};
  printf("hipModuleLaunchKernel, kernelParams\n");
  per_call_us([&] { (void)hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, st, params, nullptr); }, N, st);
  struct __attribute__((packed, aligned(8))) Buf { Op op; Param p; uint32_t pad; const float* x; float* y; int64_t n; } buf{op, p, 0, x, y, n};
  size_t bsz = sizeof(buf);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &buf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bsz, HIP_LAUNCH_PARAM_END};
  printf("hipModuleLaunchKernel, one argument buffer (extra), sizeof %zu\n", bsz);
  per_call_us([&] { (void)hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, st, nullptr, extra); }, N, st);
  printf("hipExtModuleLaunchKernel, kernelParams\n");
  per_call_us([&] { (void)hipExtModuleLaunchKernel(fn, blocks * 256, 1, 1, 256, 1, 1, 0, st, params, nullptr, nullptr, nullptr, 0); }, N, st);
  printf("hipLaunchKernel (C API), args array\n");
  per_call_us([&] { (void)hipLaunchKernel(reinterpret_cast<const void*>(&k), dim3(blocks), dim3(256), params, 0, st); }, N, st);
  // graph replay of 1 and of 16 kernel nodes
  for (int nodes : {1, 16}) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, st, op, p, (const float*)x, y, n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("hipGraphLaunch of %d captured launches (per launch)\n", nodes);
    for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 20000 / nodes;
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    printf("   host %.2f us/launch, incl. drain %.2f us/launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps / nodes,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / reps / nodes);
  }
  // GPU-side period of back-to-back tiny kernels
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, st, op, p, (const float*)x, y, n);
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("GPU period of 20000 back-to-back launches of this kernel: %.2f us each\n", ms * 1000 / 20000);
  return 0;
}
