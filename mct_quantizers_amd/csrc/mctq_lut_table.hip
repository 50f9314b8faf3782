// mctq_lut_table.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h); kernels: mctq_kernels.hpp
#include "mctq_kernels.hpp"
#include "mctq_table_builder.h"

using namespace mctq;

extern "C" {

// ---- LUT, decision table --------------------------------------------------------------------------

int32_t mctq_lut_table_entries(float clip_min, float clip_max) {
  const int k = mctq_tb::table_entries(clip_min, clip_max);
  if (k < 0) return fail_arg("decision table unsupported for this clip range");
  return k;
}

int mctq_lut_build_table(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                         float* table_host) {
  if (const char* err = mctq_tb::build(lut_host, n_lut, mult, clip_min, clip_max, table_host)) return fail_arg(err);
  return 0;
}

int mctq_lutt_per_tensor(const void* x, float* y, int64_t n, int32_t dtype, int32_t step_round, float thr_div,
                         float thr_mul, const float* table, int32_t entries, float mult, float clip_min,
                         float clip_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (step_round != 0 && step_round != MCTQ_DT_F16 && step_round != MCTQ_DT_BF16) return fail_arg("bad step_round");
  LutTableOp op;
  if (int rc = make_table_op(op, nullptr, 0.f, table, entries, mult, clip_min, clip_max, step_round)) return rc;
  const LutCommon::Param p = LutCommon::make(thr_div, thr_mul, mult);
  return with_lut_types(dtype, [&](auto ti, auto to) {
    return launch_flat<decltype(ti), decltype(to)>(op, p, x, y, n, table_bytes(entries), (hipStream_t)stream);
  });
}

int mctq_lutt_per_channel(const void* x, float* y, int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                          const float* thresholds, float eps, const float* table, int32_t entries, float mult,
                          float clip_min, float clip_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !thresholds)) return fail_arg("NULL pointer");
  LutTableOp op;
  if (int rc = make_table_op(op, thresholds, eps, table, entries, mult, clip_min, clip_max, 0)) return rc;
  return with_lut_types(dtype, [&](auto ti, auto to) {
    return launch_channels<decltype(ti), decltype(to)>(op, x, y, outer, channels, inner, table_bytes(entries),
                                                       (hipStream_t)stream);
  });
}

int mctq_lutt_per_tensor_f32(const float* x, float* y, int64_t n, float thr_div, float thr_mul, const float* table,
                             int32_t entries, float mult, float clip_min, float clip_max, void* stream) {
  return mctq_lutt_per_tensor(x, y, n, MCTQ_DT_F32, 0, thr_div, thr_mul, table, entries, mult, clip_min, clip_max, stream);
}

int mctq_lutt_per_channel_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                              const float* thresholds, float eps, const float* table, int32_t entries, float mult,
                              float clip_min, float clip_max, void* stream) {
  return mctq_lutt_per_channel(x, y, outer, channels, inner, MCTQ_DT_F32, thresholds, eps, table, entries, mult,
                               clip_min, clip_max, stream);
}

}  // extern "C"
