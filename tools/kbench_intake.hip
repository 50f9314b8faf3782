// kbench_intake: what one CU can take in from L2 -- plain 16-byte loads into registers against direct-to-LDS copies,
// with 4 / 8 / 16 waves per block (one block per CU), from a region every block shares (L2 hits) and from per-block
// regions of a large matrix (first touch).  Prints GB/s per CU.
//   hipcc --offload-arch=gfx950 -O3 -o kbench_intake.bin tools/kbench_intake.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// A block walks REPS times over its region of `bytes` bytes (a multiple of WAVES * 8 KiB); a wave step = 8 instructions
// of 1 KiB (64 lanes x 16 B, contiguous).
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* __restrict__ base, long stride, long bytes, int reps, int* out) {
  __shared__ i32x4 lds[WAVES * 8 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* a = base + (long)blockIdx.x * stride;
  i32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (long o = (long)wave * 8192; o < bytes; o += (long)WAVES * 8192) {
      if constexpr (MODE == 0) {
        i32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const i32x4*)(a + o + j * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          __builtin_amdgcn_global_load_lds((glb_void*)(a + o + j * 1024 + lane * 16), (lds_void*)&lds[(wave * 8 + j) * 64], 16, 0, 0);
        if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // 8 KiB per wave in flight, then drain
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                             // keep 8 older copies flying (they overwrite: timing only)
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE != 0) acc = lds[threadIdx.x];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) out[0] = 1;
}

template <int MODE, int WAVES>
void run(const char* name, const char* a, long stride, long bytes, int reps, int* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, a, stride, bytes, reps, out);
  (void)hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, a, stride, bytes, reps, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / n;
  printf("%-34s %-22s waves=%2d : %8.2f us  %6.1f GB/s per CU  (%5.2f TB/s chip)\n", name,
         MODE == 0 ? "loads into registers" : MODE == 1 ? "direct-to-LDS, drain" : "direct-to-LDS, counted", WAVES, us,
         (double)bytes * reps / us / 1e3, 256.0 * bytes * reps / us / 1e6);
}

int main() {
  const long region = 512 * 1024;                    // operand bytes of a 64 x 64 tile over K = 4096
  char* a; (void)hipMalloc(&a, 256 * region); (void)hipMemset(a, 1, 256 * region);
  int* out; (void)hipMalloc(&out, 4);
#define ALL(NAME, STRIDE, REPS)                                                         \
  run<0, 4>(NAME, a, STRIDE, region, REPS, out); run<0, 8>(NAME, a, STRIDE, region, REPS, out); run<0, 16>(NAME, a, STRIDE, region, REPS, out); \
  run<1, 4>(NAME, a, STRIDE, region, REPS, out); run<1, 8>(NAME, a, STRIDE, region, REPS, out); run<1, 16>(NAME, a, STRIDE, region, REPS, out); \
  run<2, 4>(NAME, a, STRIDE, region, REPS, out); run<2, 8>(NAME, a, STRIDE, region, REPS, out); run<2, 16>(NAME, a, STRIDE, region, REPS, out);
  ALL("one 512 KiB region, all blocks (L2)", 0L, 8)
  ALL("a region per block (128 MiB, L3)", region, 1)
  return 0;
}
