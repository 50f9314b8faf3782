// kbench_l2: every block reads the SAME [64][K] byte matrix (the activation codes of mctq_qlinear_i8) from L2.
// One wave instruction covers (1024 / SEG) rows x SEG contiguous bytes.  Reports aggregate L2->CU bandwidth.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int SEG, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* __restrict__ a, int* __restrict__ out, long K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16, RPI = 64 / LPR;
  i32x4 acc = {0, 0, 0, 0};
  const long rot = (blockIdx.x * 256L) % K;
  for (long kb0 = wave * 1024L; kb0 < K; kb0 += WAVES * 1024L) {      // 64 rows x 1 KiB per step = 64 instructions
    long kb = kb0 + rot; if (kb >= K) kb -= K;
#pragma unroll 16
    for (int p = 0; p < 64; ++p) {
      constexpr int CH = 1024 / SEG;                      // chunks per row in a step
      const long row = (p / CH) * RPI + lane / LPR;       // one instruction: RPI rows x SEG contiguous bytes
      const long off = (p % CH) * SEG + (lane % LPR) * 16;
      acc += *(const i32x4*)(a + row * K + kb + off);
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) out[0] = 1;
}

template <int SEG, int WAVES>
void run(const char* a, int* out, long K) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<SEG, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, a, out, K);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<SEG, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, a, out, K);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const float us = ms * 1000 / 50;
  printf("SEG=%4d waves=%2d : %.2f us  %.1f TB/s L2->CU\n", SEG, WAVES, us, 256.0 * 64 * K / us / 1e6);
}

int main() {
  const long K = 4096;
  char* a; (void)hipMalloc(&a, 64 * K); (void)hipMemset(a, 1, 64 * K);
  int* out; (void)hipMalloc(&out, 4);
  run<64, 4>(a, out, K); run<128, 4>(a, out, K); run<256, 4>(a, out, K); run<1024, 4>(a, out, K);
  run<64, 8>(a, out, K); run<128, 8>(a, out, K); run<256, 8>(a, out, K); run<1024, 8>(a, out, K);
  run<64, 16>(a, out, K); run<256, 16>(a, out, K); run<1024, 16>(a, out, K);
  return 0;
}
