// kbench_ql_stamps: DIAGNOSTIC build of the tiled consumer kernel (-DMCTQ_QL_STAMP): per wave of one block, the cycles of a
// K step spent (a) in the wait + barrier that certifies a tile, (b) issuing the copies of a later tile, (c) reading
// fragments and multiplying.  Shares only: the stamps' fences forbid overlaps the real kernel has.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMCTQ_QL_STAMP -Iinclude -Imct_quantizers_amd/csrc \
//     -mllvm -amdgpu-kernarg-preload-count=16 tools/kbench_ql_stamps.hip mct_quantizers_amd/csrc/mctq_qlinear.hip mct_quantizers_amd/csrc/mctq_misc.hip -o tools/kbench_ql_stamps.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "mctq_hip.h"
extern "C" int mctq_debug_ql_stamps(unsigned long long* out16x8);

int main(int argc, char** argv) {
  const long N = 4096, K = 4096;
  std::vector<int8_t*> ws(24);
  for (auto& p : ws) { (void)hipMalloc(&p, N * K); (void)hipMemset(p, 1, N * K); }
  float *sc, *bias, *y; int32_t* rs; uint8_t* a;
  (void)hipMalloc(&sc, N * 4); (void)hipMalloc(&bias, N * 4); (void)hipMalloc(&rs, N * 4); (void)hipMalloc(&y, 2048 * N * 4);
  (void)hipMalloc(&a, 2048 * K); (void)hipMemset(a, 3, 2048 * K); (void)hipMemset(sc, 0, N * 4); (void)hipMemset(bias, 0, N * 4); (void)hipMemset(rs, 0, N * 4);
  struct Case { long M; int variant; const char* what; };
  const Case cases[] = {{256, 662, "64x64x256, 4 waves, 2 buffers"}, {256, 6623, "64x64x256, 4 waves, ring 3"}, {256, 6624, "64x64x256, 4 waves, ring 4"},
                        {256, 86623, "64x64x256, 8 waves, ring 3"}, {256, 166623, "64x64x256, 16 waves, ring 3"},
                        {1024, 12123, "128x128x128, 4 waves, ring 3"}, {1024, 812123, "128x128x128, 8 waves, ring 3"}, {512, 1612623, "128x64x256, 16 waves, ring 3"}};
  for (const Case& c : cases) {
    if (mctq_set_tuning("ql_variant", c.variant) != 0) { printf("variant %d refused\n", c.variant); continue; }
    for (int i = 0; i < 30; ++i)
      if (mctq_qlinear_i8(a, MCTQ_CODE_U8, 114, 0.02f, ws[i % ws.size()], sc, rs, bias, y, c.M, N, K, 0) != 0) { printf("launch failed: %s\n", mctq_last_error()); return 1; }
    (void)hipDeviceSynchronize();
    unsigned long long h[16 * 8];
    if (mctq_debug_ql_stamps(h) != 0) { printf("stamp read failed\n"); return 1; }
    printf("M=%ld variant %d (%s): K steps %llu\n", c.M, c.variant, c.what, h[4]);
    const int waves = c.variant >= 1000000 ? 16 : c.variant >= 80000 ? 8 : 4;
    for (int w = 0; w < waves; w += (waves > 4 ? waves / 4 : 1)) {
      const unsigned long long* o = h + w * 8;
      const double tot = (double)o[3], n = (double)o[4];
      printf("   wave %2d: loop %8.0f ticks = %6.0f per step: wait+barrier %5.1f %%  copy issue %5.1f %%  fragments+products %5.1f %%  (per step %5.0f / %5.0f / %5.0f)\n",
             w, tot, tot / n, 100.0 * o[0] / tot, 100.0 * o[1] / tot, 100.0 * o[2] / tot, o[0] / n, o[1] / n, o[2] / n);
    }
  }
  return 0;
}
