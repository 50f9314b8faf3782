// Host check of the compact decision table builder (round 4 experiment), formerly part of tests/native/table_builder_check.cpp.
//   g++ -fsanitize=address,undefined -I mct_quantizers_amd/csrc -I include -I tools/experiments/lut_compact \
//       tools/experiments/lut_compact/compact_builder_check.cpp -o /tmp/ccheck && /tmp/ccheck
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "compact_table_builder.h"

// compact table: cell byte -> step entry -> one comparison equals the literal scan (and therefore the full table) on a dense
// sweep around every half-unit point, around every step's threshold and on a grid; `expect_fail`: codebooks with more than
// 255 steps must be refused
static int check_compact(const std::vector<float>& lut, float mult, float cmin, float cmax, bool expect_fail = false) {
  const int K = mctq_tb::table_entries(cmin, cmax);
  if (K < 0) return 1;
  std::vector<uint32_t> blob(mctq_tb::compact_words_for(K, (int)lut.size()) + 8, 0xdeadbeefu);
  int nw = 0;
  const char* err = mctq_tb::build_compact(lut.data(), (int)lut.size(), mult, cmin, cmax, blob.data(), &nw);
  if (expect_fail) { if (!err) printf("compact: accepted a codebook it must refuse\n"); return err ? 0 : 1; }
  if (err) { printf("build_compact: %s\n", err); return 1; }
  if (nw > mctq_tb::compact_words_for(K, (int)lut.size()) || blob[nw] != 0xdeadbeefu) { printf("compact: size\n"); return 1; }
  const int CW = mctq_tb::compact_cell_words(K);
  const uint8_t* cell = reinterpret_cast<const uint8_t*>(blob.data());
  const uint32_t* step = blob.data() + CW;
  float pf; memcpy(&pf, &blob[nw - 1], 4);
  const int P = (int)pf;
  if (nw != CW + 2 * (P + 1) + 2) { printf("compact: n_words\n"); return 1; }
  auto model = [&](float t) {
    const float kf = t * 2.0f + (0.5f - 2.0f * cmin);       // (the kernel's fma: exact here, |t| small)
    int k = (int)(kf < 0.0f ? 0.0f : (kf > (float)(K - 1) ? (float)(K - 1) : kf));
    const int j = cell[k];
    float T; memcpy(&T, &step[2 * j], 4);
    const uint32_t pair = step[2 * j + 1];
    return mctq_tb::f16_to_f32((uint16_t)((t >= T) ? (pair >> 16) : (pair & 0xffffu)));
  };
  long bad = 0;
  auto probe = [&](float t) { if (t >= cmin && t <= cmax && model(t) != mctq_tb::literal(t, lut.data(), (int)lut.size()) / mult) ++bad; };
  for (int k = 0; k < K; ++k) {
    const float Pk = cmin + 0.5f * (float)k;
    for (int d = -60; d <= 60; ++d) probe(mctq_tb::ord2f(mctq_tb::f2ord(Pk) + (uint32_t)d));
    for (int d = -3; d <= 3; ++d) probe(mctq_tb::ord2f(mctq_tb::f2ord(Pk + 0.25f) + (uint32_t)d));      // cell borders
  }
  for (int j = 0; j < P; ++j) {
    float T; memcpy(&T, &step[2 * j], 4);
    for (int d = -200; d <= 200; ++d) probe(mctq_tb::ord2f(mctq_tb::f2ord(T) + (uint32_t)d));
  }
  const double span = (double)cmax - (double)cmin;
  for (int i = 0; i <= 400000; ++i) probe((float)((double)cmin + span * i / 400000.0));
  if (bad) printf("compact: %ld mismatches\n", bad);
  return bad != 0;
}


int main() {
  int bad = 0;
  const std::vector<float> l16 = {-128, -96, -64, -40, -24, -12, -5, 0, 5, 12, 24, 40, 64, 96, 120, 127};
  bad += check_compact(l16, 128.0f, -128.0f, 127.0f);
  bad += check_compact({-3, -1, 0, 2}, 4.0f, -4.0f, 3.0f);
  printf(bad ? "FAILED\n" : "ok\n");
  return bad;
}
