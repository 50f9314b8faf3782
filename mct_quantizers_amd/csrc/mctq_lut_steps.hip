// mctq_lut_steps.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h); kernels: mctq_kernels.hpp
// LUT quantizers through the sorted threshold list (LutStepsOp): integer codebooks whose clip range is too large for
// the decision table (lut_values_bitwidth > 10).
#include "mctq_kernels.hpp"
#include "mctq_table_builder.h"

#include <vector>

using namespace mctq;

// 2 P + 2 words, P a power of two, optionally followed by the cell index (header + cells): the sizes cannot collide
static int steps_layout(int32_t n_words, int* P, int* cells) {
  const auto pow2 = [](int v) { return v >= 1 && v <= 4096 && (v & (v - 1)) == 0; };
  *P = 0; *cells = 0;
  if (n_words >= 4 && !(n_words & 1) && pow2((n_words - 2) / 2)) { *P = (n_words - 2) / 2; return 0; }
  for (int p = mctq_tb::kCellMinP; p <= 4096; p <<= 1)
    if (n_words == 2 * p + 2 + mctq_tb::kCellHeader + mctq_tb::steps_cells_for(p)) { *P = p; *cells = 1; return 0; }
  return fail_arg("bad steps size");
}

template <class Op>
static int make_steps_op(Op& op, const float* thr, float eps, const float* steps, int32_t n_words, int P, float mult,
                         float cmin, float cmax, int step_round) {
  if (!steps) return fail_arg("steps is NULL");
  if (int rc = check_pow2(mult)) return rc;
  fill_lut_common(op, thr, eps, mult, cmin, cmax, step_round);
  op.steps = steps; op.P = P;
  if constexpr (std::is_same<Op, LutCellsOp>::value) op.n_words = n_words;
  return 0;
}

extern "C" {

int32_t mctq_lut_steps_words(int32_t n_lut) {
  if (n_lut < 1 || n_lut > 4096) return fail_arg("n_lut must be in [1, 4096]");
  return mctq_tb::steps_words_for(n_lut);
}

int mctq_lut_build_steps(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                         float* steps_host, int32_t* n_words) {
  int P = 0;
  if (const char* err = mctq_tb::build_steps(lut_host, n_lut, mult, clip_min, clip_max, steps_host, &P)) return fail_arg(err);
  int distinct = 1;                                     // distinct centres = finite-or-infinite thresholds + 1: recount
  {
    std::vector<float> seen(n_lut); int d = 0;
    for (int j = 0; j < n_lut; ++j) { bool dup = false; for (int k = 0; k < d; ++k) dup = dup || seen[k] == lut_host[j]; if (!dup) seen[d++] = lut_host[j]; }
    distinct = d;
  }
  const int extra = mctq_tb::build_step_cells(steps_host, P, distinct, clip_min, clip_max);
  if (n_words) *n_words = 2 * P + 2 + extra;
  return 0;
}

int mctq_luts_per_tensor(const void* x, float* y, int64_t n, int32_t dtype, int32_t step_round, float thr_div,
                         float thr_mul, const float* steps, int32_t n_words, float mult, float clip_min,
                         float clip_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (step_round != 0 && step_round != MCTQ_DT_F16 && step_round != MCTQ_DT_BF16) return fail_arg("bad step_round");
  int P, cells;
  if (int rc = steps_layout(n_words, &P, &cells)) return rc;
  const LutCommon::Param p = LutCommon::make(thr_div, thr_mul, mult);
  const size_t lds = (size_t)((n_words + 3) & ~3) * sizeof(float);
  const auto run = [&](auto& op) {
    if (int rc = make_steps_op(op, nullptr, 0.f, steps, n_words, P, mult, clip_min, clip_max, step_round)) return rc;
    return with_lut_types(dtype, [&](auto ti, auto to) {
      return launch_flat<decltype(ti), decltype(to)>(op, p, x, y, n, lds, (hipStream_t)stream);
    });
  };
  if (cells) { LutCellsOp op; return run(op); }
  LutStepsOp op;
  return run(op);
}

int mctq_luts_per_channel(const void* x, float* y, int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                          const float* thresholds, float eps, const float* steps, int32_t n_words, float mult,
                          float clip_min, float clip_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !thresholds)) return fail_arg("NULL pointer");
  int P, cells;
  if (int rc = steps_layout(n_words, &P, &cells)) return rc;
  const size_t lds = (size_t)((n_words + 3) & ~3) * sizeof(float);
  const auto run = [&](auto& op) {
    if (int rc = make_steps_op(op, thresholds, eps, steps, n_words, P, mult, clip_min, clip_max, 0)) return rc;
    return with_lut_types(dtype, [&](auto ti, auto to) {
      return launch_channels<decltype(ti), decltype(to)>(op, x, y, outer, channels, inner, lds, (hipStream_t)stream);
    });
  };
  if (cells) { LutCellsOp op; return run(op); }
  LutStepsOp op;
  return run(op);
}

}  // extern "C"
