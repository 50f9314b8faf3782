// mctq_table_builder.h -- host-only construction of the LUT decision table (no HIP dependency).
//
// Pure C++ so that it can be compiled on its own (tests/native/table_builder_check.cpp builds it with
// g++ -fsanitize=address,undefined); mctq_lut_table.hip wraps it behind the C ABI.  The algorithm and its
// correctness argument are described at LutTableOp in mctq_kernels.hpp and in DESIGN.md.
#pragma once
#include <math.h>

#include <vector>
#include <stdint.h>
#include <string.h>

namespace mctq_tb {

// first-minimum argmin over fl32(|t - lut[j]|) in list order: the reference's literal scan
inline float literal(float t, const float* lut, int n) {
  float best_c = lut[0];
  float best_d = fabsf(t - lut[0]);
  for (int j = 1; j < n; ++j) {
    const float d = fabsf(t - lut[j]);
    if (d < best_d) { best_d = d; best_c = lut[j]; }
  }
  return best_c;
}

// order-preserving map float <-> uint32 (finite values)
inline uint32_t f2ord(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
inline float ord2f(uint32_t o) { uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; memcpy(&f, &u, 4); return f; }

// IEEE binary16 <-> binary32 (round to nearest even; enough for the finite values used here)
inline uint16_t f32_to_f16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const uint32_t abs = u & 0x7fffffffu;
  if (abs >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((abs > 0x7f800000u) ? 0x200u : 0u));   // inf / NaN
  if (abs >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                          // overflow -> inf
  if (abs < 0x33000001u) return (uint16_t)sign;                                                        // underflow -> 0
  int e = (int)(abs >> 23) - 127;
  uint32_t m = (abs & 0x7fffffu) | 0x800000u;
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;
  uint32_t half = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
  if (rem > mid || (rem == mid && (half & 1u))) ++half;
  uint32_t he = (e < -14) ? 0u : (uint32_t)(e + 15);
  // `half` holds the 11-bit significand (implicit bit included for normals); adding lets a carry bump the exponent
  const uint32_t out = (e < -14) ? half : ((he << 10) + (half - 0x400u));
  return (uint16_t)(sign | out);
}
inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
  if (e == 0) {
    if (m == 0) { u = sign; }
    else { int s = 0; while (!(m & 0x400u)) { m <<= 1; ++s; } u = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13); }
  } else if (e == 31) { u = sign | 0x7f800000u | (m << 13); }
  else { u = sign | ((e + 112u) << 23) | (m << 13); }
  float f; memcpy(&f, &u, 4); return f;
}

// number of table points for a clip range, or -1 if unsupported
inline int table_entries(float cmin, float cmax) {
  if (!(cmin < cmax) || cmin != floorf(cmin) || cmax != floorf(cmax)) return -1;
  const double k = 2.0 * ((double)cmax - (double)cmin) + 1.0;
  if (k > 2048.0) return -1;                 // the table must stay well inside LDS
  return (int)k;
}

// Fills table[2 * (K + 1)] words; returns NULL on success or a static message.
inline const char* build(const float* lut, int n_lut, float mult, float clip_min, float clip_max, float* table) {
  if (!lut || !table) return "NULL pointer";
  if (n_lut < 1 || n_lut > 4096) return "n_lut must be in [1, 4096]";
  int e = 0;
  if (!(mult > 0.0f) || frexpf(mult, &e) != 0.5f) return "mult must be a positive power of two";
  const int K = table_entries(clip_min, clip_max);
  if (K < 0) return "decision table unsupported for this clip range";
  for (int j = 0; j < n_lut; ++j)
    if (!(lut[j] == floorf(lut[j])) || fabsf(lut[j]) > 16777216.0f) return "decision table needs an integer codebook";
  uint32_t rng = 0x9E3779B9u;
  for (int k = 0; k < K; ++k) {
    const float P = clip_min + 0.5f * (float)k;
    const float lo = fmaxf(clip_min, P - 0.25f), hi = fminf(clip_max, P + 0.25f);
    const float cb = literal(lo, lut, n_lut), ca = literal(hi, lut, n_lut);
    float T = -INFINITY;
    if (cb != ca) {
      uint32_t a = f2ord(lo), b = f2ord(hi);          // F(a) == cb, F(b) == ca; find the smallest b with F == ca
      while (b - a > 1) {
        const uint32_t m = a + (b - a) / 2;
        if (literal(ord2f(m), lut, n_lut) == ca) b = m; else a = m;
      }
      T = ord2f(b);
      if (literal(ord2f(b - 1), lut, n_lut) != cb || literal(T, lut, n_lut) != ca) return "codebook decision is not a single step";
    }
    // spot-check the single-step model on pseudo-random points of the cell
    const uint32_t span = f2ord(hi) - f2ord(lo);
    for (int r = 0; r < 32; ++r) {
      rng = rng * 1664525u + 1013904223u;
      const float t = ord2f(f2ord(lo) + (span ? rng % (span + 1u) : 0u));
      if (literal(t, lut, n_lut) != ((t >= T) ? ca : cb)) return "codebook decision is not a single step";
    }
    const float qb = cb / mult, qa = ca / mult;
    const uint16_t hb = f32_to_f16(qb), ha = f32_to_f16(qa);
    if (f16_to_f32(hb) != qb || f16_to_f32(ha) != qa) return "codebook centre not exact in fp16";
    const uint32_t pair = (uint32_t)hb | ((uint32_t)ha << 16);
    table[2 * k + 0] = T;
    memcpy(&table[2 * k + 1], &pair, 4);
  }
  table[2 * K + 0] = lut[0] / mult;                  // NaN input: every distance is NaN, argmin = index 0
  table[2 * K + 1] = (float)K;
  return nullptr;
}

// ---- threshold list ("steps") for integer codebooks of any clip range (LutStepsOp in mctq_kernels.hpp) ----------
// P = smallest power of two >= number of DISTINCT centres; the array holds 2 * P + 2 floats.
inline int steps_pow2(int n_distinct) { int p = 1; while (p < n_distinct) p <<= 1; return p; }
// Long lists (P >= kCellMinP) may carry a cell index behind the list: kCellHeader words {G, max thresholds per cell,
// gscale, clip_min} and G words {first threshold of the cell | thresholds in it << 16} (see LutStepsOp).
constexpr int kCellHeader = 4, kCellMinP = 128, kCellMaxPer = 4;
inline int steps_cells_for(int P) { return P < kCellMinP ? 0 : (P <= 256 ? 1024 : (P <= 1024 ? 4096 : 8192)); }
inline int steps_words_for(int n_lut) {                                         // upper bound from the list length
  const int p = steps_pow2(n_lut);
  return 2 * p + 2 + (p >= kCellMinP ? kCellHeader + steps_cells_for(p) : 0);
}
// the cell of a clipped value: the SAME float operations as the kernel (no contraction), monotone in t
inline int steps_cell(float t, float clip_min, float gscale, int G) {
  volatile float d = t - clip_min;
  volatile float v = d * gscale;
  const float vv = v;
  if (!(vv > 0.0f)) return 0;                                                    // also -inf / NaN
  if (vv >= (float)(G - 1)) return G - 1;
  return (int)vv;
}

// Fills steps[2 * P + 2] (P returned through *p_out); returns NULL on success or a static message.
inline const char* build_steps(const float* lut, int n_lut, float mult, float clip_min, float clip_max, float* steps,
                               int* p_out) {
  if (!lut || !steps || !p_out) return "NULL pointer";
  if (n_lut < 1 || n_lut > 4096) return "n_lut must be in [1, 4096]";
  int e = 0;
  if (!(mult > 0.0f) || frexpf(mult, &e) != 0.5f) return "mult must be a positive power of two";
  if (!(clip_min < clip_max) || fabsf(clip_min) > 1048576.0f || fabsf(clip_max) > 1048576.0f) return "clip range unsupported";
  for (int j = 0; j < n_lut; ++j)
    if (!(lut[j] == floorf(lut[j])) || fabsf(lut[j]) > 1048576.0f) return "the threshold list needs an integer codebook within 2^20";
  // distinct centres, ascending (insertion sort: n <= 4096, construction time only)
  std::vector<float> vs_store(4096);                      // heap: the builders keep nothing large on the stack
  float* vs = vs_store.data();
  int D = 0;
  for (int j = 0; j < n_lut; ++j) {
    int pos = 0;
    while (pos < D && vs[pos] < lut[j]) ++pos;
    if (pos < D && vs[pos] == lut[j]) continue;
    for (int k = D; k > pos; --k) vs[k] = vs[k - 1];
    vs[pos] = lut[j];
    ++D;
  }
  const int P = steps_pow2(D);
  float* T = steps;
  float* Qv = steps + P;
  for (int k = 0; k < P; ++k) { T[k] = INFINITY; Qv[k] = vs[D - 1] / mult; }
  T[0] = -INFINITY;                                       // unused by the search
  const uint32_t olo = f2ord(clip_min), ohi = f2ord(clip_max);
  for (int k = 0; k < D; ++k) Qv[k] = vs[k] / mult;
  for (int k = 1; k < D; ++k) {
    // T_k = the smallest t in [clip_min, clip_max] whose literal result is >= vs[k] (monotone predicate)
    if (literal(clip_min, lut, n_lut) >= vs[k]) { T[k] = -INFINITY; continue; }
    if (literal(clip_max, lut, n_lut) < vs[k]) { T[k] = INFINITY; continue; }
    uint32_t a = olo, b = ohi;                            // literal(a) < vs[k] <= literal(b)
    while (b - a > 1) {
      const uint32_t m = a + (b - a) / 2;
      if (literal(ord2f(m), lut, n_lut) >= vs[k]) b = m; else a = m;
    }
    T[k] = ord2f(b);
  }
  for (int k = 2; k < D; ++k)
    if (T[k] < T[k - 1]) return "codebook decision is not a monotone staircase";
  // check the staircase model against the literal scan: at every threshold and its predecessor, and on pseudo-random
  // points of the clip range
  auto model = [&](float t) {
    int idx = 0;
    for (int sft = P >> 1; sft > 0; sft >>= 1) idx += (t >= T[idx + sft]) ? sft : 0;
    return Qv[idx] * mult;
  };
  uint32_t rng = 0x2545F491u;
  for (int k = 1; k < D; ++k) {
    if (!(T[k] > -INFINITY && T[k] < INFINITY)) continue;
    const uint32_t o = f2ord(T[k]);
    for (int d = -2; d <= 2; ++d) {
      const uint32_t oo = o + (uint32_t)d;
      if (oo < olo || oo > ohi) continue;
      const float t = ord2f(oo);
      if (literal(t, lut, n_lut) != model(t)) return "codebook decision is not a monotone staircase";
    }
  }
  const uint32_t span = ohi - olo;
  for (int r = 0; r < 4096; ++r) {
    rng = rng * 1664525u + 1013904223u;
    const float t = ord2f(olo + (span ? rng % (span + 1u) : 0u));
    if (literal(t, lut, n_lut) != model(t)) return "codebook decision is not a monotone staircase";
  }
  steps[2 * P + 0] = lut[0] / mult;                       // NaN input: argmin over all-NaN distances = index 0
  steps[2 * P + 1] = (float)P;
  *p_out = P;
  return nullptr;
}

// Cell index for a long threshold list (appended behind the 2 * P + 2 words build_steps wrote; `steps` must have room
// for steps_words_for()).  Returns the number of words appended: 0 when the list is short or some cell would hold more
// than kCellMaxPer thresholds (the kernel then keeps the binary search).  Exactness does not depend on where the cell
// edges fall: cell() is monotone, so a threshold in an earlier cell than t's is <= t, one in a later cell is > t, and
// the thresholds of t's own cell are compared one by one.
inline int build_step_cells(float* steps, int P, int n_distinct, float clip_min, float clip_max) {
  const int G = steps_cells_for(P);
  if (!G) return 0;
  // the list AND the index are staged in LDS by every block: past the 64 KiB a launch gets without opting in, keep the
  // binary search over the list alone (P = 4096: 32 KiB)
  if ((size_t)(2 * P + 2 + kCellHeader + G) * sizeof(float) > 64u * 1024u) return 0;
  const float gscale = (float)G / (clip_max - clip_min);
  const float* T = steps;
  std::vector<uint32_t> first_store(G), count_store(G);
  uint32_t* first = first_store.data();
  uint32_t* count = count_store.data();
  for (int c = 0; c < G; ++c) first[c] = count[c] = 0;
  std::vector<int> cells_store(n_distinct > 1 ? n_distinct : 1);
  int* cells_of = cells_store.data();
  for (int k = 1; k < n_distinct; ++k) { cells_of[k] = steps_cell(T[k], clip_min, gscale, G); ++count[cells_of[k]]; }
  for (int k = 2; k < n_distinct; ++k) if (cells_of[k] < cells_of[k - 1]) return 0;    // (cannot happen: T is sorted)
  uint32_t run = 0, maxc = 0;
  for (int c = 0; c < G; ++c) { first[c] = run; run += count[c]; if (count[c] > maxc) maxc = count[c]; }
  if (maxc > (uint32_t)kCellMaxPer) return 0;
  float* out = steps + 2 * P + 2;
  out[0] = (float)G; out[1] = (float)maxc; out[2] = gscale; out[3] = clip_min;
  for (int c = 0; c < G; ++c) {
    const uint32_t word = first[c] | (count[c] << 16);
    memcpy(&out[kCellHeader + c], &word, 4);
  }
  return kCellHeader + G;
}

// ---- the same threshold list for float64 tensors: the reference's chain evaluates quotient, clip and distances in
// double there (type promotion, quantizer_utils.py:126-134), so the staircase's steps sit at DOUBLE thresholds.
// Same argument as above (integer centres: distances to non-neighbours differ by >= 1, fl(t - a) is monotone in t),
// same construction with the literal scan and the bisection carried out in double.
// Blob layout: double T[P] (T[0] unused), float Q[P], float q_nan, float P  ->  12 P + 8 bytes (P <= 4096: 48 KiB).
inline double literal64(double t, const float* lut, int n) {
  float best_c = lut[0];
  double best_d = fabs(t - (double)lut[0]);
  for (int j = 1; j < n; ++j) {
    const double d = fabs(t - (double)lut[j]);
    if (d < best_d) { best_d = d; best_c = lut[j]; }
  }
  return (double)best_c;
}
inline uint64_t d2ord(double f) { uint64_t u; memcpy(&u, &f, 8); return (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
inline double ord2d(uint64_t o) { uint64_t u = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o; double f; memcpy(&f, &u, 8); return f; }
inline size_t steps64_bytes(int P) { return (size_t)P * 12u + 8u; }

inline const char* build_steps64(const float* lut, int n_lut, float mult, float clip_min, float clip_max, void* blob,
                                 int* p_out) {
  if (!lut || !blob || !p_out) return "NULL pointer";
  if (n_lut < 1 || n_lut > 4096) return "n_lut must be in [1, 4096]";
  int e = 0;
  if (!(mult > 0.0f) || frexpf(mult, &e) != 0.5f) return "mult must be a positive power of two";
  if (!(clip_min < clip_max) || fabsf(clip_min) > 1048576.0f || fabsf(clip_max) > 1048576.0f) return "clip range unsupported";
  for (int j = 0; j < n_lut; ++j)
    if (!(lut[j] == floorf(lut[j])) || fabsf(lut[j]) > 1048576.0f) return "the threshold list needs an integer codebook within 2^20";
  std::vector<float> vs(n_lut);
  int D = 0;
  for (int j = 0; j < n_lut; ++j) {
    int pos = 0;
    while (pos < D && vs[pos] < lut[j]) ++pos;
    if (pos < D && vs[pos] == lut[j]) continue;
    for (int k = D; k > pos; --k) vs[k] = vs[k - 1];
    vs[pos] = lut[j];
    ++D;
  }
  const int P = steps_pow2(D);
  std::vector<double> T(P, INFINITY);
  std::vector<float> Qv(P, vs[D - 1] / mult);
  T[0] = -INFINITY;
  for (int k = 0; k < D; ++k) Qv[k] = vs[k] / mult;
  const double cmin = clip_min, cmax = clip_max;
  const uint64_t olo = d2ord(cmin), ohi = d2ord(cmax);
  for (int k = 1; k < D; ++k) {
    if (literal64(cmin, lut, n_lut) >= (double)vs[k]) { T[k] = -INFINITY; continue; }
    if (literal64(cmax, lut, n_lut) < (double)vs[k]) { T[k] = INFINITY; continue; }
    uint64_t a = olo, b = ohi;
    while (b - a > 1) {
      const uint64_t m = a + (b - a) / 2;
      if (literal64(ord2d(m), lut, n_lut) >= (double)vs[k]) b = m; else a = m;
    }
    T[k] = ord2d(b);
  }
  for (int k = 2; k < D; ++k)
    if (T[k] < T[k - 1]) return "codebook decision is not a monotone staircase";
  auto model = [&](double t) {
    int idx = 0;
    for (int sft = P >> 1; sft > 0; sft >>= 1) idx += (t >= T[idx + sft]) ? sft : 0;
    return (double)(Qv[idx] * mult);
  };
  for (int k = 1; k < D; ++k) {
    if (!(T[k] > -INFINITY && T[k] < INFINITY)) continue;
    const uint64_t o = d2ord(T[k]);
    for (int d = -2; d <= 2; ++d) {
      const uint64_t oo = o + (uint64_t)(int64_t)d;
      if (oo < olo || oo > ohi) continue;
      const double t = ord2d(oo);
      if (literal64(t, lut, n_lut) != model(t)) return "codebook decision is not a monotone staircase";
    }
  }
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  for (int r = 0; r < 4096; ++r) {
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    const double t = cmin + (cmax - cmin) * ((double)(rng >> 11) * (1.0 / 9007199254740992.0));
    if (literal64(t, lut, n_lut) != model(t)) return "codebook decision is not a monotone staircase";
  }
  uint8_t* out = static_cast<uint8_t*>(blob);
  memcpy(out, T.data(), (size_t)P * 8u);
  memcpy(out + (size_t)P * 8u, Qv.data(), (size_t)P * 4u);
  const float tail[2] = {lut[0] / mult, (float)P};
  memcpy(out + (size_t)P * 12u, tail, 8);
  *p_out = P;
  return nullptr;
}

}  // namespace mctq_tb
