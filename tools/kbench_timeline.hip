// kbench_timeline.hip -- where do the ~2 us of fixed cost per launch of the cfg2 stream go?
// (VERDICT r01 "What's weak" #1/#6: the 128 MiB launch sits at 0.74 of spec while the 512 MiB one reaches 0.82.)
// Not part of the product.  Every launch is checked with hipGetLastError; a variant that fails to launch is
// reported as FAILED, never timed.
//
//   A. timeline: the library's launch shape (T256, U4, nt loads + nt stores, one tile per block) with every
//      block stamping the 100 MHz constant clock (s_memrealtime) at entry, when its loads have landed, and after
//      its stores have been issued -> dispatch ramp, drain, in-kernel span vs the event-measured period,
//      and the bubble between consecutive launches on one stream.
//   B. steady state: the same kernel on 2x / 4x / 8x the rows in ONE launch (fixed cost amortised).
//   C. chip-wide phase separation for real: 512 resident blocks load the WHOLE 64 MiB into registers
//      (32 x 16 B per lane), grid barrier, then all store.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/kbench_timeline.bin tools/kbench_timeline.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float fq(float x, float s, float inv) {
  float q = __builtin_rintf(x * inv);
  q = fminf(fmaxf(q, -128.f), 127.f);
  return q * s;
}
__device__ __forceinline__ f4 fq4(f4 v, float s, float inv) {
  f4 r; r.x = fq(v.x, s, inv); r.y = fq(v.y, s, inv); r.z = fq(v.z, s, inv); r.w = fq(v.w, s, inv);
  return r;
}
__device__ __forceinline__ uint64_t now() { return __builtin_readsteadycounter(); }   // s_memrealtime, 100 MHz

// one tile per block, optional stamps: stamps[3*b + {0,1,2}] = entry, loads landed, stores issued
template <int T, int U, bool STAMP>
__global__ __launch_bounds__(T) void k_tile(const float* __restrict__ xs, float* __restrict__ ys,
                                            const float* __restrict__ scales, uint32_t tiles_per_row, uint32_t inner4,
                                            uint64_t* __restrict__ stamps) {
  uint64_t t0 = 0, t1 = 0;
  if (STAMP) t0 = now();
  const uint32_t row = blockIdx.x / tiles_per_row;
  const uint32_t tile = blockIdx.x - row * tiles_per_row;
  const f4* x = reinterpret_cast<const f4*>(xs) + (int64_t)row * inner4;
  f4* y = reinterpret_cast<f4*>(ys) + (int64_t)row * inner4;
  const uint32_t col = tile * (T * U) + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(x + col + u * T);
  const float s = scales[row];
  const float inv = 1.0f / s;
  if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t1 = now(); }
#pragma unroll
  for (int u = 0; u < U; ++u) __builtin_nontemporal_store(fq4(v[u], s, inv), y + col + u * T);
  if (STAMP && threadIdx.x == 0) {
    uint64_t* p = stamps + 3ull * blockIdx.x;
    p[0] = t0; p[1] = t1; p[2] = now();
  }
}

// E: the same tile kernel, but the blocks of the FIRST residency round (blockIdx < first) delay their loads by
// (blockIdx / per_slot) * delay sleeps of 64 clocks: the launch otherwise starts 2048 blocks in lockstep (all read,
// then all write), and the timeline shows reads starving for ~2.5 us behind the first write burst.
template <int T, int U>
__global__ __launch_bounds__(T) void k_tile_stagger(const float* __restrict__ xs, float* __restrict__ ys,
                                                    const float* __restrict__ scales, uint32_t tiles_per_row, uint32_t inner4,
                                                    uint32_t first, uint32_t per_slot, uint32_t delay, uint32_t mod) {
  if (blockIdx.x < first) {
    uint32_t k = blockIdx.x / per_slot;
    if (mod) k %= mod;
    for (uint32_t i = 0; i < k * delay; ++i) __builtin_amdgcn_s_sleep(1);
  }
  const uint32_t row = blockIdx.x / tiles_per_row;
  const uint32_t tile = blockIdx.x - row * tiles_per_row;
  const f4* x = reinterpret_cast<const f4*>(xs) + (int64_t)row * inner4;
  f4* y = reinterpret_cast<f4*>(ys) + (int64_t)row * inner4;
  const uint32_t col = tile * (T * U) + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(x + col + u * T);
  const float s = scales[row];
  const float inv = 1.0f / s;
#pragma unroll
  for (int u = 0; u < U; ++u) __builtin_nontemporal_store(fq4(v[u], s, inv), y + col + u * T);
}

// C: phase separated.  grid = blocks resident at once; lane holds NV float4.  barrier counter is monotonic:
// launch number `epoch` waits for epoch * gridDim.x arrivals.
template <int T, int NV>
__global__ __launch_bounds__(T) void k_coop(const float* __restrict__ xs, float* __restrict__ ys,
                                            const float* __restrict__ scales, uint32_t inner4, uint32_t total4,
                                            unsigned int* counter, unsigned int target, int do_barrier) {
  // block b owns lane-vectors [b * T * NV, (b+1) * T * NV): T*NV divides inner4 * k, rows found per vector
  const uint32_t base = blockIdx.x * (T * NV) + threadIdx.x;
  const f4* x = reinterpret_cast<const f4*>(xs);
  f4* y = reinterpret_cast<f4*>(ys);
  f4 v[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) v[u] = __builtin_nontemporal_load(x + base + u * T);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (do_barrier) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const uint32_t i = base + u * T;
    const uint32_t row = i / inner4;                  // wave-uniform (T*NV and inner4 are multiples of 64 vectors)
    const float s = scales[row];
    const float inv = 1.0f / s;
    __builtin_nontemporal_store(fq4(v[u], s, inv), y + i);
  }
}

static const uint32_t INNER4 = 1024;   // 4096 floats per row

static bool launched_ok(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("%-52s FAILED to launch: %s\n", what, hipGetErrorString(e)); return false; }
  return true;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const uint32_t MAXROWS = 4096 * 8;
  const size_t maxbytes = (size_t)MAXROWS * INNER4 * 16;
  const int RING = 5;
  float *x[RING], *y[RING], *scales, *xbig, *ybig;
  const uint32_t ROWS = 4096;
  const size_t n = (size_t)ROWS * INNER4 * 4, bytes = n * 4;
  std::vector<float> h((size_t)MAXROWS * INNER4 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f * 4.f - 2.f;
  std::vector<float> hs(MAXROWS);
  for (uint32_t i = 0; i < MAXROWS; ++i) hs[i] = (0.5f + (i % 97) / 97.f) / 64.f;
  for (int r = 0; r < RING; ++r) { CK(hipMalloc(&x[r], bytes)); CK(hipMalloc(&y[r], bytes)); CK(hipMemcpy(x[r], h.data(), bytes, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&xbig, maxbytes)); CK(hipMalloc(&ybig, maxbytes)); CK(hipMemcpy(xbig, h.data(), maxbytes, hipMemcpyHostToDevice));
  CK(hipMalloc(&scales, MAXROWS * 4)); CK(hipMemcpy(scales, hs.data(), MAXROWS * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // fixed-duration pre-warm so every section below sees the same clocks
  {
    for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL((k_tile<256, 4, false>), dim3(ROWS), dim3(256), 0, st, x[i % RING], y[i % RING], scales, 1u, INNER4, (uint64_t*)nullptr);
    CK(hipStreamSynchronize(st));
  }

  // ---- A. timeline -------------------------------------------------------------------------------
  const int K = 24;
  const uint32_t blocks = ROWS;                       // tiles_per_row = INNER4 / (256*4) = 1
  uint64_t* d_stamps; CK(hipMalloc(&d_stamps, (size_t)K * blocks * 3 * 8));
  auto time_events = [&](auto&& launch, int reps) -> float {
    for (int i = 0; i < 10; ++i) launch(i);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
  };
  for (int ring = RING; ring >= 1; ring -= RING - 1) {
    float plain = time_events([&](int i) { hipLaunchKernelGGL((k_tile<256, 4, false>), dim3(blocks), dim3(256), 0, st, x[i % ring], y[i % ring], scales, 1u, INNER4, (uint64_t*)nullptr); }, iters);
    if (!launched_ok("tile plain")) return 1;
    float stamped = time_events([&](int i) { hipLaunchKernelGGL((k_tile<256, 4, true>), dim3(blocks), dim3(256), 0, st, x[i % ring], y[i % ring], scales, 1u, INNER4, d_stamps + (size_t)(i % K) * blocks * 3); }, iters);
    if (!launched_ok("tile stamped")) return 1;
    printf("\n[A] %s  (ring %d)  period by events: plain %.2f us (%.0f GB/s), stamped %.2f us\n", ring > 1 ? "COLD" : "WARM", ring, plain, 2.0 * bytes / plain / 1e3, stamped);
    // K stamped launches back to back, then read the stamps
    CK(hipMemset(d_stamps, 0, (size_t)K * blocks * 3 * 8));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL((k_tile<256, 4, true>), dim3(blocks), dim3(256), 0, st, x[i % ring], y[i % ring], scales, 1u, INNER4, d_stamps + (size_t)i * blocks * 3);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> hst((size_t)K * blocks * 3);
    CK(hipMemcpy(hst.data(), d_stamps, hst.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> span, bubble, first_done, ramp90, drain10;
    uint64_t prev_end = 0;
    for (int k = 0; k < K; ++k) {
      const uint64_t* s = hst.data() + (size_t)k * blocks * 3;
      uint64_t mn = ~0ull, mx = 0, mn_done = ~0ull;
      std::vector<uint64_t> starts(blocks), ends(blocks), landed(blocks);
      for (uint32_t b = 0; b < blocks; ++b) {
        mn = std::min(mn, s[3 * b]); mx = std::max(mx, s[3 * b + 2]); mn_done = std::min(mn_done, s[3 * b + 1]);
        starts[b] = s[3 * b]; ends[b] = s[3 * b + 2]; landed[b] = s[3 * b + 1];
      }
      std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end()); std::sort(landed.begin(), landed.end());
      span.push_back((mx - mn) * 0.01);
      first_done.push_back((mn_done - mn) * 0.01);
      if (k) bubble.push_back(((double)mn - (double)prev_end) * 0.01);
      prev_end = mx;
      if (k == K / 2) {
        printf("    launch %d timeline (us from first block entry): blocks entered / finished, cumulative\n", k);
        for (double t = 0.5; t < (mx - mn) * 0.01 + 0.5; t += 0.5) {
          const uint64_t lim = mn + (uint64_t)(t * 100.0);
          size_t a = std::upper_bound(starts.begin(), starts.end(), lim) - starts.begin();
          size_t c = std::upper_bound(ends.begin(), ends.end(), lim) - ends.begin();
          size_t l = std::upper_bound(landed.begin(), landed.end(), lim) - landed.begin();
          const uint64_t lim0 = mn + (uint64_t)((t - 0.5) * 100.0);
          size_t l0 = std::upper_bound(landed.begin(), landed.end(), lim0) - landed.begin();
          size_t c0 = std::upper_bound(ends.begin(), ends.end(), lim0) - ends.begin();
          // 16 KiB read per block landed, 16 KiB written per block finished, per 0.5 us bin -> GB/s
          printf("      t=%5.1f  entered %5zu  loads landed %5zu  finished %5zu  resident %5zu | this bin: read %5.0f GB/s  write(issued) %5.0f GB/s\n",
                 t, a, l, c, a - c, (l - l0) * 16384.0 / 0.5e-6 / 1e9, (c - c0) * 16384.0 / 0.5e-6 / 1e9);
        }
        // block lifetime distribution
        std::vector<double> life(blocks);
        for (uint32_t b = 0; b < blocks; ++b) life[b] = (s[3 * b + 2] - s[3 * b]) * 0.01;
        std::sort(life.begin(), life.end());
        printf("    block lifetime us: p10 %.2f p50 %.2f p90 %.2f max %.2f\n", life[blocks / 10], life[blocks / 2], life[blocks * 9 / 10], life[blocks - 1]);
      }
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("    K=%d stamped launches: period by events %.2f us | in-kernel span (first entry -> last store issued) median %.2f us"
           " | bubble (last store of k -> first entry of k+1) median %.2f us | first loads landed after %.2f us\n",
           K, ms * 1000.f / K, med(span), med(bubble), med(first_done));
  }

  // ---- B. steady state: same kernel, more rows per launch ----------------------------------------
  printf("\n[B] one launch over R rows (cold: buffers of 8x config 2, each launch touches fresh lines beyond the 256 MiB cache for R >= 8192)\n");
  for (uint32_t rows : {4096u, 8192u, 16384u, 32768u}) {
    const size_t b2 = (size_t)rows * INNER4 * 16;
    // rotate the start offset inside the big buffer so that small R is cold too
    const uint32_t slots = MAXROWS / rows;
    float us = time_events([&](int i) {
      const size_t off = (size_t)(i % slots) * rows * INNER4 * 4;
      hipLaunchKernelGGL((k_tile<256, 4, false>), dim3(rows), dim3(256), 0, st, xbig + off, ybig + off, scales, 1u, INNER4, (uint64_t*)nullptr);
    }, std::max(20, iters * 4096 / (int)rows));
    if (!launched_ok("tile rows")) return 1;
    printf("    R=%6u  %8.2f us  %6.0f GB/s\n", rows, us, 2.0 * b2 / us / 1e3);
  }

  // ---- D. residency caps: fewer blocks per CU (dynamic LDS as the limiter) and fatter / thinner tiles ----
  printf("\n[D] one tile per block, residency capped by an LDS allocation (160 KiB per CU), cold ring\n");
  {
    auto run = [&](auto kern, const char* name, uint32_t threads, uint32_t tiles_per_row, size_t lds) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      float us = time_events([&](int i) {
        hipLaunchKernelGGL(kern, dim3(ROWS * tiles_per_row), dim3(threads), lds, st, x[i % RING], y[i % RING], scales, tiles_per_row, INNER4, (uint64_t*)nullptr);
      }, iters);
      if (!launched_ok(name)) return;
      printf("    %-34s lds %6zu B (<= %2zu blocks/CU)  %8.2f us  %6.0f GB/s\n", name, lds, lds ? std::min<size_t>(8, 160 * 1024 / lds) : 8, us, 2.0 * bytes / us / 1e3);
    };
    for (size_t lds : {(size_t)0, (size_t)27 * 1024, (size_t)32 * 1024, (size_t)40 * 1024, (size_t)53 * 1024, (size_t)80 * 1024, (size_t)160 * 1024}) {
      run(k_tile<256, 4, false>, "T256 U4 (16 KiB in per block)", 256, 1, lds);
      run(k_tile<256, 2, false>, "T256 U2 ( 8 KiB in per block)", 256, 2, lds);
      run(k_tile<256, 1, false>, "T256 U1 ( 4 KiB in per block)", 256, 4, lds);
    }
    run(k_tile<512, 2, false>, "T512 U2 (16 KiB in per block)", 512, 1, 0);
    run(k_tile<1024, 1, false>, "T1024 U1 (16 KiB in per block)", 1024, 1, 0);
  }

  // ---- E. staggered first residency round ----
  printf("\n[E] first-round stagger: block b < first sleeps ((b / per_slot) %% mod) * delay * 64 clocks before its loads (cold ring)\n");
  {
    for (uint32_t first : {2048u, 1024u}) {
      for (uint32_t per_slot : {256u, 32u, 8u}) {
        for (uint32_t delay : {0u, 2u, 4u, 8u, 12u, 16u, 24u}) {
          const uint32_t mod = per_slot == 256 ? 0 : 8;
          float us = time_events([&](int i) {
            hipLaunchKernelGGL((k_tile_stagger<256, 4>), dim3(ROWS), dim3(256), 0, st, x[i % RING], y[i % RING], scales, 1u, INNER4, first, per_slot, delay, mod);
          }, iters);
          if (!launched_ok("stagger")) return 1;
          printf("    first %4u per_slot %3u delay %2u x64clk (max %.2f us)  %8.2f us  %6.0f GB/s\n", first, per_slot, delay,
                 (mod ? 7 : (first / per_slot - 1)) * delay * 64 / 2100.0, us, 2.0 * bytes / us / 1e3);
        }
      }
    }
  }

  // ---- C. phase-separated (all load -> grid barrier -> all store) ---------------------------------
  printf("\n[C] chip-wide phase separation: 512 blocks x 256 lanes x 32 float4 per lane = 64 MiB in registers\n");
  {
    unsigned int* counter; CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
    const uint32_t total4 = ROWS * INNER4;
    constexpr int NV = 32;
    const uint32_t grid = total4 / (256 * NV);        // 512
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_coop<256, NV>, 256, 0));
    printf("    grid %u blocks, occupancy API says %d blocks/CU -> %d resident slots\n", grid, occ, occ * prop.multiProcessorCount);
    if ((uint32_t)(occ * prop.multiProcessorCount) < grid) printf("    NOT all resident: barrier variant skipped\n");
    unsigned int epoch = 0;
    for (int barrier = 0; barrier <= ((uint32_t)(occ * prop.multiProcessorCount) >= grid ? 1 : 0); ++barrier) {
      float us = time_events([&](int i) {
        ++epoch;
        hipLaunchKernelGGL((k_coop<256, NV>), dim3(grid), dim3(256), 0, st, x[i % RING], y[i % RING], scales, INNER4, total4, counter,
                           barrier ? epoch * grid : 0u, barrier);
      }, iters);
      if (!launched_ok("coop")) return 1;
      if (!barrier) { CK(hipMemset(counter, 0, 4)); epoch = 0; }
      printf("    %-44s %8.2f us  %6.0f GB/s\n", barrier ? "load all -> grid barrier -> store all" : "load all (32 in flight/lane) -> store, no barrier", us, 2.0 * bytes / us / 1e3);
    }
    // correctness of the phase kernel vs the tile kernel
    hipLaunchKernelGGL((k_tile<256, 4, false>), dim3(blocks), dim3(256), 0, st, x[0], y[0], scales, 1u, INNER4, (uint64_t*)nullptr);
    ++epoch;
    hipLaunchKernelGGL((k_coop<256, NV>), dim3(grid), dim3(256), 0, st, x[0], y[1], scales, INNER4, total4, counter, 0u, 0);
    CK(hipStreamSynchronize(st));
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), y[0], bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y[1], bytes, hipMemcpyDeviceToHost));
    printf("    outputs equal: %s\n", memcmp(a.data(), b.data(), bytes) == 0 ? "yes" : "NO");
  }
  return 0;
}
