// kbench_ql: timing of mctq_qlinear_i8 and of ablated builds (-DMCTQ_QL_ABLATE_A, -DMCTQ_QL_ABLATE_MFMA).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imct_quantizers_amd/csrc [-D...]
//       tools/kbench_ql.hip mct_quantizers_amd/csrc/mctq_qlinear.hip mct_quantizers_amd/csrc/mctq_misc.hip -o kbench_ql
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "mctq_hip.h"

int main() {
  const long N = 4096, K = 4096;
  std::vector<int8_t*> ws(24);
  for (auto& p : ws) { (void)hipMalloc(&p, N * K); (void)hipMemset(p, 1, N * K); }
  float *sc, *bias, *y; int32_t* rs; uint8_t* a;
  (void)hipMalloc(&sc, N * 4); (void)hipMalloc(&bias, N * 4); (void)hipMalloc(&rs, N * 4); (void)hipMalloc(&y, 1024 * N * 4);
  (void)hipMalloc(&a, 1024 * K); (void)hipMemset(a, 3, 1024 * K); (void)hipMemset(sc, 0, N * 4); (void)hipMemset(bias, 0, N * 4); (void)hipMemset(rs, 0, N * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int variant : {41, 81, 44, 84}) {
    mctq_set_tuning("ql_variant", variant);
    for (long M : {1L, 16L, 64L, 256L, 1024L}) {
      for (int i = 0; i < 3; ++i) mctq_qlinear_i8(a, MCTQ_CODE_U8, 114, 0.02f, ws[i], sc, rs, bias, y, M, N, K, 0);
      (void)hipEventRecord(e0);
      const int iters = 40;
      for (int i = 0; i < iters; ++i) mctq_qlinear_i8(a, MCTQ_CODE_U8, 114, 0.02f, ws[i % ws.size()], sc, rs, bias, y, M, N, K, 0);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("variant %d M=%4ld : %.2f us\n", variant, M, ms * 1000 / iters);
    }
  }
  return 0;
}
