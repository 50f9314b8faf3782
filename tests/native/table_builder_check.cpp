// Sanitizer-checked run of the host table builder (g++ -fsanitize=address,undefined; see tests/test_abi_exports.py).
// Builds decision tables for several codebooks and compares every table decision with the literal scan on a dense
// sweep around each half-integer point; also round-trips the binary16 conversion over all 65536 bit patterns.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "mctq_table_builder.h"

static int check(const std::vector<float>& lut, float mult, float cmin, float cmax) {
  const int K = mctq_tb::table_entries(cmin, cmax);
  if (K < 0) return 1;
  std::vector<float> tab(2 * (K + 1));
  if (const char* err = mctq_tb::build(lut.data(), (int)lut.size(), mult, cmin, cmax, tab.data())) { printf("build: %s\n", err); return 1; }
  long bad = 0;
  for (int k = 0; k < K; ++k) {
    const float P = cmin + 0.5f * (float)k;
    for (int d = -60; d <= 60; ++d) {
      uint32_t o = mctq_tb::f2ord(P) + (uint32_t)d;
      float t = mctq_tb::ord2f(o);
      if (!(t >= cmin && t <= cmax)) continue;
      const int kk = (int)(t * 2.0f + (0.5f - 2.0f * cmin));
      const int ki = kk < 0 ? 0 : (kk > K - 1 ? K - 1 : kk);
      uint32_t pair; memcpy(&pair, &tab[2 * ki + 1], 4);
      const float q = mctq_tb::f16_to_f32((uint16_t)((t >= tab[2 * ki]) ? (pair >> 16) : (pair & 0xffffu)));
      if (q != mctq_tb::literal(t, lut.data(), (int)lut.size()) / mult) ++bad;
    }
  }
  if (bad) printf("%ld mismatches\n", bad);
  return bad != 0;
}

int main() {
  int rc = 0;
  for (uint32_t h = 0; h < 65536; ++h) {                    // binary16 round trip
    const float f = mctq_tb::f16_to_f32((uint16_t)h);
    if (f != f) continue;
    if (mctq_tb::f32_to_f16(f) != (uint16_t)h) { printf("f16 round trip failed at %u\n", h); rc = 1; break; }
  }
  rc |= check({-5, 5}, 128, -128, 127);
  rc |= check({3, 3, -8}, 128, -128, 127);
  rc |= check({22, -53, 62, 0, -66, -21, 44, -40}, 128, -128, 127);
  rc |= check({-128, -96, -64, -40, -24, -12, -5, 0, 5, 12, 24, 40, 64, 96, 120, 127}, 128, -128, 127);
  rc |= check({0, 13, 50, 90, 128, 200, 255, 256}, 256, 0, 255);
  std::vector<float> wide; for (int v = -512; v < 512; v += 64) wide.push_back((float)v);
  rc |= check(wide, 512, -512, 511);
  std::vector<float> all; for (int v = 127; v >= -128; --v) all.push_back((float)((v * 37) % 256 - 128 + ((v * 37) % 256 < 0 ? 256 : 0)));
  rc |= check(all, 128, -128, 127);
  float bad_lut[2] = {0.5f, 1.0f}; float tmp[8];
  if (!mctq_tb::build(bad_lut, 2, 128, -128, 127, tmp)) { printf("non-integer codebook accepted\n"); rc = 1; }
  printf(rc ? "FAILED\n" : "table builder ok\n");
  return rc;
}
