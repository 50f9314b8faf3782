// kbench.hip -- standalone variant explorer for the cfg2 stream (4096x4096 fp32 per-channel fake-quant).
// Not part of the product: used to choose the launch shape of mctq_kernels.hip and to measure the
// copy ceiling on the same machine (guide rule: ceilings come from a known-good reference measured
// on the same hardware).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/kbench tools/kbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include <dlfcn.h>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ f4 ld4(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st4(f4* p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__device__ __forceinline__ float fq(float x, float s, float inv, float z, float lo, float hi) {
  float q = __builtin_rintf(x * inv) + z;
  q = fminf(fmaxf(q, lo), hi);
  return (q - z) * s;
}
template <bool FQ>
__device__ __forceinline__ f4 op4(f4 v, float s, float inv, float z, float lo, float hi) {
  if (!FQ) return v;
  f4 r; r.x = fq(v.x, s, inv, z, lo, hi); r.y = fq(v.y, s, inv, z, lo, hi); r.z = fq(v.z, s, inv, z, lo, hi); r.w = fq(v.w, s, inv, z, lo, hi);
  return r;
}

// A: one tile per block (row tile), T threads, U float4 per lane.
template <int T, int U, bool NTL, bool NTS, bool FQ>
__global__ __launch_bounds__(T) void k_tile(const float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ scales,
                                            uint32_t tiles_per_row, uint32_t inner4) {
  const uint32_t row = blockIdx.x / tiles_per_row;
  const uint32_t tile = blockIdx.x - row * tiles_per_row;
  const float s = scales[row];
  const float inv = 1.0f / s;
  const f4* x = reinterpret_cast<const f4*>(xs) + (int64_t)row * inner4;
  f4* y = reinterpret_cast<f4*>(ys) + (int64_t)row * inner4;
  const uint32_t col = tile * (T * U) + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = ld4<NTL>(x + col + u * T);
#pragma unroll
  for (int u = 0; u < U; ++u) st4<NTS>(y + col + u * T, op4<FQ>(v[u], s, inv, 0.f, -128.f, 127.f));
}

// B: persistent grid-stride over tiles, software prefetch of the next tile.
template <int T, int U, bool NTL, bool NTS, bool FQ, bool XCD>
__global__ __launch_bounds__(T) void k_persist(const float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ scales,
                                               uint32_t tiles_per_row, uint32_t inner4, uint32_t ntiles) {
  uint32_t b = blockIdx.x, nb = gridDim.x;
  uint32_t t0, t1, step;
  if (XCD) {
    // blocks b, b+8, b+16.. share an XCD: give each XCD one contiguous eighth of the tiles
    const uint32_t xcd = b & 7, idx = b >> 3, per_xcd_blocks = nb >> 3;
    const uint32_t chunk = (ntiles + 7) / 8;
    t0 = xcd * chunk + idx; t1 = min(ntiles, (xcd + 1) * chunk); step = per_xcd_blocks;
  } else { t0 = b; t1 = ntiles; step = nb; }
  if (t0 >= t1) return;
  f4 v[U];
  {
    const uint32_t row = t0 / tiles_per_row, tile = t0 - row * tiles_per_row;
    const f4* x = reinterpret_cast<const f4*>(xs) + (int64_t)row * inner4 + tile * (T * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4<NTL>(x + u * T);
  }
  for (uint32_t t = t0; t < t1; t += step) {
    const uint32_t row = t / tiles_per_row, tile = t - row * tiles_per_row;
    const float s = scales[row];
    const float inv = 1.0f / s;
    f4 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = v[u];
    const uint32_t tn = t + step;
    if (tn < t1) {
      const uint32_t rown = tn / tiles_per_row, tilen = tn - rown * tiles_per_row;
      const f4* x = reinterpret_cast<const f4*>(xs) + (int64_t)rown * inner4 + tilen * (T * U) + threadIdx.x;
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld4<NTL>(x + u * T);
    }
    f4* y = reinterpret_cast<f4*>(ys) + (int64_t)row * inner4 + tile * (T * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) st4<NTS>(y + u * T, op4<FQ>(w[u], s, inv, 0.f, -128.f, 127.f));
  }
}


// read-only stream: loads, folds, never stores (store guarded by an impossible runtime condition)
template <int T, int U, bool NT>
__global__ __launch_bounds__(T) void k_read(const float* __restrict__ xs, float* __restrict__ ys, uint32_t inner4, float never) {
  const f4* x = reinterpret_cast<const f4*>(xs) + (int64_t)blockIdx.x * (T * U) + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = ld4<NT>(x + u * T);
  f4 a = v[0];
#pragma unroll
  for (int u = 1; u < U; ++u) a += v[u];
  if (a.x + a.y + a.z + a.w == never) ys[threadIdx.x] = a.x;
}
template <int T, int U, bool NT>
__global__ __launch_bounds__(T) void k_write(const float* __restrict__ xs, float* __restrict__ ys, uint32_t inner4, float val) {
  f4* y = reinterpret_cast<f4*>(ys) + (int64_t)blockIdx.x * (T * U) + threadIdx.x;
  f4 v; v.x = val; v.y = val + 1; v.z = val + 2; v.w = val + 3;
#pragma unroll
  for (int u = 0; u < U; ++u) st4<NT>(y + u * T, v);
}
// tile kernel that waits for ALL of the wave's loads before its first store (+ optional sleep)
template <int T, int U, int SLEEP>
__global__ __launch_bounds__(T) void k_phase(const float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ scales,
                                             uint32_t tiles_per_row, uint32_t inner4) {
  const uint32_t row = blockIdx.x / tiles_per_row;
  const uint32_t tile = blockIdx.x - row * tiles_per_row;
  const float s = scales[row];
  const float inv = 1.0f / s;
  const f4* x = reinterpret_cast<const f4*>(xs) + (int64_t)row * inner4;
  f4* y = reinterpret_cast<f4*>(ys) + (int64_t)row * inner4;
  const uint32_t col = tile * (T * U) + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = ld4<true>(x + col + u * T);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
#pragma unroll
  for (int u = 0; u < U; ++u) st4<true>(y + col + u * T, op4<true>(v[u], s, inv, 0.f, -128.f, 127.f));
}

struct Variant { std::string name; void (*launch)(const float*, float*, const float*, hipStream_t); double traffic = 2.0; /* x 64 MiB: 2 = read + write, 1 = one direction only */ };

static constexpr uint32_t ROWS = 4096, INNER4 = 1024;
static int g_cus = 256;

template <int T, int U, bool NTL, bool NTS, bool FQ>
void l_tile(const float* x, float* y, const float* s, hipStream_t st) {
  const uint32_t tpr = INNER4 / (T * U);
  hipLaunchKernelGGL((k_tile<T, U, NTL, NTS, FQ>), dim3(ROWS * tpr), dim3(T), 0, st, x, y, s, tpr, INNER4);
}
template <int T, int U, bool NTL, bool NTS, bool FQ, bool XCD, int BPC>
void l_persist(const float* x, float* y, const float* s, hipStream_t st) {
  const uint32_t tpr = INNER4 / (T * U);
  hipLaunchKernelGGL((k_persist<T, U, NTL, NTS, FQ, XCD>), dim3(g_cus * BPC), dim3(T), 0, st, x, y, s, tpr, INNER4, ROWS * tpr);
}

template <int T, int U, bool NT>
void l_read(const float* x, float* y, const float* s, hipStream_t st) {
  hipLaunchKernelGGL((k_read<T, U, NT>), dim3(ROWS * INNER4 / (T * U)), dim3(T), 0, st, x, y, INNER4, -12345.678f);
}
template <int T, int U, bool NT>
void l_write(const float* x, float* y, const float* s, hipStream_t st) {
  hipLaunchKernelGGL((k_write<T, U, NT>), dim3(ROWS * INNER4 / (T * U)), dim3(T), 0, st, x, y, INNER4, 1.0f);
}
template <int T, int U, int SLEEP>
void l_phase(const float* x, float* y, const float* s, hipStream_t st) {
  static_assert(INNER4 % (T * U) == 0 && INNER4 / (T * U) >= 1, "tile does not fit the row");
  const uint32_t tpr = INNER4 / (T * U);
  hipLaunchKernelGGL((k_phase<T, U, SLEEP>), dim3(ROWS * tpr), dim3(T), 0, st, x, y, s, tpr, INNER4);
}
typedef int (*fqpc_t)(const float*, float*, int64_t, int64_t, int64_t, const float*, const int32_t*, int32_t, int32_t, void*);
static fqpc_t g_fqpc = nullptr;
static int32_t* g_zps = nullptr;
void l_lib(const float* x, float* y, const float* s, hipStream_t st) {
  g_fqpc(x, y, 1, ROWS, INNER4 * 4, s, g_zps, -128, 127, (void*)st);
}
void l_memcpy(const float* x, float* y, const float*, hipStream_t st) {
  CK(hipMemcpyAsync(y, x, (size_t)ROWS * INNER4 * 16, hipMemcpyDeviceToDevice, st));
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 200;
  const char* filter = argc > 2 ? argv[2] : "";
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  printf("device %s CUs %d\n", prop.name, g_cus);
  const size_t n = (size_t)ROWS * INNER4 * 4, bytes = n * 4;
  const int RING = 5;
  float *x[RING], *y[RING], *scales;
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f * 4.f - 2.f;
  std::vector<float> hs(ROWS);
  for (uint32_t i = 0; i < ROWS; ++i) hs[i] = (0.5f + (i % 97) / 97.f) / 64.f;
  for (int r = 0; r < RING; ++r) { CK(hipMalloc(&x[r], bytes)); CK(hipMalloc(&y[r], bytes)); CK(hipMemcpy(x[r], h.data(), bytes, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&scales, ROWS * 4)); CK(hipMemcpy(scales, hs.data(), ROWS * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  {
    void* h = dlopen("./mct_quantizers_amd/lib/libmctq_hip.so", RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    g_fqpc = (fqpc_t)dlsym(h, "mctq_fq_per_channel_f32");
    CK(hipMalloc(&g_zps, ROWS * 4)); CK(hipMemset(g_zps, 0, ROWS * 4));
  }
  std::vector<Variant> vs = {
    {"lib mctq_fq_per_channel_f32", l_lib},
    {"memcpyD2D", l_memcpy},
#define TV(T, U, NL, NS, FQ) {std::string("tile T" #T " U" #U " ntl" #NL " nts" #NS) + (FQ ? " fq" : " copy"), l_tile<T, U, NL, NS, FQ>}
    TV(256, 4, false, false, false), TV(256, 4, true, true, false), TV(256, 4, true, false, false), TV(256, 4, false, true, false),
    TV(256, 4, false, false, true), TV(256, 4, true, true, true), TV(256, 4, true, false, true), TV(256, 4, false, true, true),
    TV(256, 2, true, true, true), TV(256, 1, true, true, true),
    TV(512, 2, true, true, true), TV(512, 1, true, true, true), TV(1024, 1, true, true, true), TV(128, 4, true, true, true), TV(128, 8, true, true, true), TV(64, 8, true, true, true), TV(64, 16, true, true, true),
#define PV(T, U, X, B) {"persist T" #T " U" #U " xcd" #X " bpc" #B " nt fq", l_persist<T, U, true, true, true, X, B>}
    PV(256, 4, false, 1), PV(256, 4, false, 2), PV(256, 4, false, 4), PV(256, 4, false, 8),
    PV(256, 4, true, 1), PV(256, 4, true, 2), PV(256, 4, true, 4), PV(256, 4, true, 8),
    PV(256, 2, false, 4), PV(256, 2, false, 8), PV(256, 2, true, 8), PV(256, 1, false, 8), PV(256, 1, true, 8),
    PV(512, 2, false, 2), PV(512, 2, false, 4), PV(512, 2, true, 4), PV(1024, 1, false, 2), PV(1024, 1, true, 2),
    {"read-only T256 U4 nt (64MiB)", l_read<256, 4, true>, 1.0}, {"read-only T256 U4 (64MiB)", l_read<256, 4, false>, 1.0},
    {"read-only T256 U8 nt (64MiB)", l_read<256, 8, true>, 1.0},
    {"write-only T256 U4 nt (64MiB)", l_write<256, 4, true>, 1.0}, {"write-only T256 U4 (64MiB)", l_write<256, 4, false>, 1.0},
    // (a row of 1024 float4 holds ONE tile of 256 x 4: U8 tiles do not exist for this shape -- round 1 listed
    //  "phase T256 U8" variants whose grid was 0 blocks; they never ran and are gone)
    {"phase T256 U4 sleep0", l_phase<256, 4, 0>}, {"phase T256 U2 sleep0", l_phase<256, 2, 0>}, {"phase T256 U4 sleep20", l_phase<256, 4, 20>},
    {"phase T256 U4 sleep60", l_phase<256, 4, 60>},
    {"persist copy T256 U4 bpc4", l_persist<256, 4, true, true, false, false, 4>},
    {"persist copy T256 U4 bpc8", l_persist<256, 4, true, true, false, false, 8>},
  };
  // interleaved rounds: every variant runs `iters` launches per round, 3 rounds; report median & min round
  const int ROUNDS = 3;
  std::vector<std::vector<float>> res(vs.size());
  std::vector<bool> failed(vs.size(), false);
  for (int round = 0; round < ROUNDS; ++round) {
    for (size_t vi = 0; vi < vs.size(); ++vi) {
      if (*filter && !strstr(vs[vi].name.c_str(), filter)) continue;
      if (failed[vi]) continue;
      for (int i = 0; i < 10; ++i) vs[vi].launch(x[i % RING], y[i % RING], scales, st);
      { hipError_t le = hipGetLastError(); if (le != hipSuccess) { printf("%-44s FAILED to launch: %s\n", vs[vi].name.c_str(), hipGetErrorString(le)); failed[vi] = true; continue; } }
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) vs[vi].launch(x[i % RING], y[i % RING], scales, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      res[vi].push_back(ms * 1000.f / iters);
    }
  }
  for (size_t vi = 0; vi < vs.size(); ++vi) {
    if (res[vi].empty() || failed[vi]) continue;
    std::sort(res[vi].begin(), res[vi].end());
    float med = res[vi][res[vi].size() / 2], mn = res[vi][0];
    const double moved = vs[vi].traffic * bytes;
    printf("%-44s med %7.2f us  %6.0f GB/s | min %7.2f us %6.0f GB/s\n", vs[vi].name.c_str(), med, moved / med / 1e3, mn, moved / mn / 1e3);
  }
  return 0;
}
