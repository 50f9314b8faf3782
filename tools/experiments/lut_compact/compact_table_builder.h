// Host builder of the compact decision table (round 4 experiment; see lut_compact_op.hpp).  Include after
// mct_quantizers_amd/csrc/mctq_table_builder.h.
#pragma once
#include "mctq_table_builder.h"

namespace mctq_tb {

// ---- compact form of the decision table (LutCompactOp in mctq_kernels.hpp) ---------------------------------------
// The decision table above spends 8 bytes on every half-unit cell although a codebook of L centres has only L - 1 steps:
// 4 KB for an 8-bit clip range, staged into LDS by EVERY block -- measured, that staging (not its barrier) is what keeps
// the table kernel behind the affine one on the same tensor (profiles/r04/lut_staging_ablation.log).  The compact form
// keeps the very same cells and thresholds:
//   cell[k]  (1 byte, k < K)   j = number of steps in cells below k = index of the first step at or above cell k
//   step[j]  (8 bytes, j <= P) {T_j, half2(q below T_j, q above T_j)}; step[P] = {+inf, (q_top, q_top)}
//   trailer  {q for NaN input (float32), P (float32)}
// and the kernel answers  q = (v >= T_j) ? above_j : below_j  with j = cell[k(v)]: k() is monotone, so the steps of lower
// cells are <= v and those of higher cells > v -- only the step of v's own cell (if any) needs the comparison, and when the
// cell has none, step[j] lies above v and `below_j` is the cell's constant.  511 + 8 * 17 + 8 bytes for 16 centres.
// Words: CW = ceil(K / 4) of cell bytes, 2 * (P + 1) of steps, 2 of trailer.
constexpr int kCompactMaxSteps = 255;
inline int compact_cell_words(int K) { return (K + 3) / 4; }
inline int compact_words_for(int K, int n_lut) {                 // upper bound: at most n_lut - 1 steps
  int p = n_lut - 1; if (p > kCompactMaxSteps) p = kCompactMaxSteps; if (p < 0) p = 0;
  return compact_cell_words(K) + 2 * (p + 1) + 2;
}

// Fills blob[*n_words] 32-bit words; returns NULL on success or a static message (then use the decision table).
inline const char* build_compact(const float* lut, int n_lut, float mult, float clip_min, float clip_max, uint32_t* blob,
                                 int* n_words) {
  if (!blob || !n_words) return "NULL pointer";
  const int K = table_entries(clip_min, clip_max);
  if (K < 0) return "decision table unsupported for this clip range";
  std::vector<float> table((size_t)2 * (K + 1));
  if (const char* err = build(lut, n_lut, mult, clip_min, clip_max, table.data())) return err;
  auto pair_of = [&](int k) { uint32_t u; memcpy(&u, &table[2 * k + 1], 4); return u; };
  const int CW = compact_cell_words(K);
  std::vector<uint8_t> cells((size_t)CW * 4, 0);
  std::vector<uint32_t> steps;
  int P = 0;
  for (int k = 0; k < K; ++k) {
    const uint32_t pr = pair_of(k);
    cells[k] = (uint8_t)P;
    if ((pr & 0xffffu) != (pr >> 16)) {
      if (P == kCompactMaxSteps) return "too many steps for the compact table";
      uint32_t t; memcpy(&t, &table[2 * k], 4);
      steps.push_back(t); steps.push_back(pr);
      ++P;
    }
  }
  for (int k = K; k < CW * 4; ++k) cells[k] = (uint8_t)P;
  const uint32_t top = P ? (steps[2 * P - 1] >> 16) : (pair_of(0) & 0xffffu);
  const float inf = INFINITY;
  uint32_t inf_bits; memcpy(&inf_bits, &inf, 4);
  steps.push_back(inf_bits); steps.push_back(top | (top << 16));
  // the compact form must reproduce every cell of the table it was derived from
  for (int k = 0; k < K; ++k) {
    const uint32_t pr = pair_of(k), st = steps[2 * cells[k] + 1];
    const bool has = (pr & 0xffffu) != (pr >> 16);
    if ((st & 0xffffu) != (pr & 0xffffu)) return "compact table: staircase not consistent";
    uint32_t tk; memcpy(&tk, &table[2 * k], 4);
    if (has && (st != pr || steps[2 * cells[k]] != tk)) return "compact table: step mismatch";
    if (has && k + 1 < K && cells[k + 1] != cells[k] + 1) return "compact table: cell index mismatch";
  }
  memcpy(blob, cells.data(), (size_t)CW * 4);
  memcpy(blob + CW, steps.data(), steps.size() * 4);
  const float pf = (float)P;
  memcpy(blob + CW + 2 * (P + 1), &table[2 * K], 4);            // q for NaN input
  memcpy(blob + CW + 2 * (P + 1) + 1, &pf, 4);
  *n_words = CW + 2 * (P + 1) + 2;
  return nullptr;
}


}  // namespace mctq_tb
