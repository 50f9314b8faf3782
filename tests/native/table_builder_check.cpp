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

// threshold list: the binary search over T equals the literal scan around every threshold and on a coarse sweep
static int check_steps(const std::vector<float>& lut, float mult, float cmin, float cmax) {
  std::vector<float> steps(mctq_tb::steps_words_for((int)lut.size()));
  int P = 0;
  if (const char* err = mctq_tb::build_steps(lut.data(), (int)lut.size(), mult, cmin, cmax, steps.data(), &P)) { printf("build_steps: %s\n", err); return 1; }
  if (2 * P + 2 > (int)steps.size() || steps[2 * P + 1] != (float)P) { printf("bad P\n"); return 1; }
  const float* T = steps.data();
  const float* Q = T + P;
  auto model = [&](float t) { int idx = 0; for (int s = P >> 1; s > 0; s >>= 1) idx += (t >= T[idx + s]) ? s : 0; return Q[idx]; };
  long bad = 0;
  for (int k = 1; k < P; ++k) {
    if (!(T[k] > -INFINITY && T[k] < INFINITY)) continue;
    for (int d = -200; d <= 200; ++d) {
      const float t = mctq_tb::ord2f(mctq_tb::f2ord(T[k]) + (uint32_t)d);
      if (!(t >= cmin && t <= cmax)) continue;
      if (model(t) != mctq_tb::literal(t, lut.data(), (int)lut.size()) / mult) ++bad;
    }
  }
  const double span = (double)cmax - (double)cmin;
  for (int i = 0; i <= 200000; ++i) {
    const float t = (float)((double)cmin + span * i / 200000.0);
    if (model(t) != mctq_tb::literal(t, lut.data(), (int)lut.size()) / mult) ++bad;
  }
  // cell index of a long list: the count through the cells equals the binary search
  int distinct = 0;
  { std::vector<float> seen; for (float v : lut) { bool dup = false; for (float u : seen) dup = dup || u == v; if (!dup) seen.push_back(v); } distinct = (int)seen.size(); }
  const int extra = mctq_tb::build_step_cells(steps.data(), P, distinct, cmin, cmax);
  if (extra) {
    const float* h = steps.data() + 2 * P + 2;
    const int G = (int)h[0], maxc = (int)h[1];
    const float gscale = h[2];
    auto via_cells = [&](float t) {
      const int c = mctq_tb::steps_cell(t, cmin, gscale, G);
      uint32_t e; memcpy(&e, &h[4 + c], 4);
      const int first = (int)(e & 0xffffu), n = (int)(e >> 16);
      int idx = first;
      for (int j = 0; j < maxc; ++j) { const int k = 1 + first + j < P - 1 ? 1 + first + j : P - 1; idx += (j < n && t >= T[k]) ? 1 : 0; }
      return Q[idx];
    };
    if (G != mctq_tb::steps_cells_for(P)) { printf("bad G\n"); return 1; }
    for (int i = 0; i <= 400000; ++i) {
      const float t = (float)((double)cmin + span * i / 400000.0);
      if (via_cells(t) != model(t)) ++bad;
    }
    for (int k = 1; k < P; ++k) {
      if (!(T[k] > -INFINITY && T[k] < INFINITY)) continue;
      for (int d = -3; d <= 3; ++d) {
        const float t = mctq_tb::ord2f(mctq_tb::f2ord(T[k]) + (uint32_t)d);
        if (t >= cmin && t <= cmax && via_cells(t) != model(t)) ++bad;
      }
    }
  } else if (P >= mctq_tb::kCellMinP) { printf("note: no cell index for P=%d\n", P); }
  if (bad) printf("steps: %ld mismatches\n", bad);
  return bad != 0;
}

// the float64 threshold list against the literal double scan on a grid, at every threshold +-3 ulps and at midpoints
static int check_steps64(const std::vector<float>& lut, float mult, float cmin, float cmax) {
  std::vector<unsigned char> blob(mctq_tb::steps64_bytes(mctq_tb::steps_pow2((int)lut.size())));
  int P = 0;
  if (const char* err = mctq_tb::build_steps64(lut.data(), (int)lut.size(), mult, cmin, cmax, blob.data(), &P)) {
    printf("build_steps64 failed: %s\n", err);
    return 1;
  }
  std::vector<double> T(P); std::vector<float> Q(P);
  memcpy(T.data(), blob.data(), (size_t)P * 8); memcpy(Q.data(), blob.data() + (size_t)P * 8, (size_t)P * 4);
  auto model = [&](double t) { int idx = 0; for (int s = P >> 1; s > 0; s >>= 1) idx += (t >= T[idx + s]) ? s : 0; return (double)(Q[idx] * mult); };
  long bad = 0;
  for (int i = 0; i <= 400000; ++i) {
    const double t = (double)cmin + ((double)cmax - (double)cmin) * i / 400000.0;
    if (mctq_tb::literal64(t, lut.data(), (int)lut.size()) != model(t)) ++bad;
  }
  for (int k = 1; k < P; ++k) {
    if (!(T[k] > -INFINITY && T[k] < INFINITY)) continue;
    for (int d = -3; d <= 3; ++d) {
      const double t = mctq_tb::ord2d(mctq_tb::d2ord(T[k]) + (uint64_t)(int64_t)d);
      if (t >= cmin && t <= cmax && mctq_tb::literal64(t, lut.data(), (int)lut.size()) != model(t)) ++bad;
    }
  }
  if (bad) printf("steps64: %ld mismatches\n", bad);
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
  rc |= check_steps({-5, 5}, 2048, -2048, 2047);
  rc |= check_steps({3, 3, -8}, 2048, -2048, 2047);
  rc |= check_steps({7}, 2048, -2048, 2047);
  rc |= check_steps({2047, -2048, 0, 1, -1, 1000, -1000, 512, 3}, 2048, -2048, 2047);
  rc |= check_steps({0, 1, 2, 4095, 4096, 77, 900}, 4096, 0, 4095);                       // centre above the clip range
  std::vector<float> w16; for (int v = -32768; v < 32768; v += 257) w16.push_back((float)(((v * 31) % 65536 + 65536) % 65536 - 32768));
  rc |= check_steps(w16, 32768, -32768, 32767);
  rc |= check_steps(all, 128, -128, 127);
  {
    std::vector<float> st(mctq_tb::steps_words_for(2)); int P = 0;
    if (!mctq_tb::build_steps(bad_lut, 2, 2048, -2048, 2047, st.data(), &P)) { printf("non-integer codebook accepted (steps)\n"); rc = 1; }
  }
  rc |= check_steps64({-5, 5}, 128, -128, 127);
  rc |= check_steps64({3, 3, -8}, 2048, -2048, 2047);
  rc |= check_steps64({22, -53, 62, 0, -66, -21, 44, -40}, 128, -128, 127);
  rc |= check_steps64(w16, 32768, -32768, 32767);
  rc |= check_steps64(all, 128, -128, 127);
  printf(rc ? "FAILED\n" : "table builder ok\n");
  return rc;
}
