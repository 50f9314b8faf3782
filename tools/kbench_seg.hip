// kbench_seg: how does the contiguous run per row per load instruction affect streaming a [N][K] byte matrix?
// Each block owns 16 rows, its 8 waves split K in 1 KiB-per-row steps; one wave instruction (64 lanes x 16 B)
// covers (1024 / SEG) rows x SEG contiguous bytes.  hipcc --offload-arch=gfx950 -O3 tools/kbench_seg.hip -o kbench_seg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int SEG, bool NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* __restrict__ w, int* __restrict__ out, int N, long K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;          // lanes per row
  constexpr int RPI = 64 / LPR;          // rows per instruction
  const long n0 = (long)blockIdx.x * 16;
  i32x4 acc = {0, 0, 0, 0};
  // the wave reads, per step, 16 rows x 256 B (4 KiB) as 4 instructions; layout decided by SEG
  for (long kb = wave * 256; kb < K; kb += WAVES * 256) {
    i32x4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // instruction p covers rows [p*RPI .. ) x SEG bytes when SEG<=256; for SEG > 256 use a larger K step
      long row, off;
      if constexpr (SEG <= 256) {
        constexpr int CH = 256 / SEG;                   // chunks per row in this step
        const int q = p * RPI + lane / LPR;             // (row, chunk) index 0..(16*CH-1)
        row = q / CH; off = (q % CH) * SEG + (lane % LPR) * 16;
      } else { row = 0; off = 0; }
      const char* ptr = w + (n0 + row) * K + kb + off;
      v[p] = NT ? __builtin_nontemporal_load((const i32x4*)ptr) : *(const i32x4*)ptr;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) acc += v[p];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) out[0] = 1;
}

template <int SEG, bool NT, int WAVES>
float run(const std::vector<char*>& ws, int* out, int N, long K, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<SEG, NT, WAVES>), dim3(N / 16), dim3(WAVES * 64), 0, 0, ws[i % ws.size()], out, N, K);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<SEG, NT, WAVES>), dim3(N / 16), dim3(WAVES * 64), 0, 0, ws[i % ws.size()], out, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000 / iters;
}

int main() {
  const int N = 4096; const long K = 4096;
  std::vector<char*> ws(24);
  for (auto& p : ws) { hipMalloc(&p, N * K); hipMemset(p, 1, N * K); }
  int* out; hipMalloc(&out, 4);
#define R(SEG, NT, WV) printf("SEG=%4d NT=%d waves=%d : %.2f us  %.0f GB/s\n", SEG, NT, WV, run<SEG, NT, WV>(ws, out, N, K, 60), N * K / run<SEG, NT, WV>(ws, out, N, K, 60) / 1e3);
  R(64, 1, 8) R(64, 0, 8) R(128, 1, 8) R(128, 0, 8) R(256, 1, 8) R(256, 0, 8)
  R(64, 1, 4) R(256, 1, 4) R(64, 0, 4) R(256, 0, 4)
  R(64, 1, 16) R(256, 1, 16) R(64, 0, 16) R(256, 0, 16)
  return 0;
}
