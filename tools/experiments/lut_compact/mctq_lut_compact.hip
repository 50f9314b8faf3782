// mctq_lut_compact.hip -- the compact decision table of round 4 as an EXPERIMENT translation unit (not part of the shipped
// libmctq_hip.so since ABI v8): python tools/build_variant.py lut_compact links it with the regular objects into
// tools/ablate/libmctq_hip_lut_compact.so, which exports the regular ABI plus the four entry points below.
#include "lut_compact_op.hpp"
#include "compact_table_builder.h"

using namespace mctq;

extern "C" {

// ---- LUT, compact decision table ---------------------------------------------------------------------

int32_t mctq_lut_compact_words(float clip_min, float clip_max, int32_t n_lut) {
  const int k = mctq_tb::table_entries(clip_min, clip_max);
  if (k < 0) return fail_arg("decision table unsupported for this clip range");
  if (n_lut < 1) return fail_arg("n_lut must be >= 1");
  return mctq_tb::compact_words_for(k, n_lut);
}

int mctq_lut_build_compact(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                           void* blob_host, int32_t* n_words) {
  int nw = 0;
  if (const char* err = mctq_tb::build_compact(lut_host, n_lut, mult, clip_min, clip_max, static_cast<uint32_t*>(blob_host), &nw))
    return fail_arg(err);
  *n_words = nw;
  return 0;
}

int mctq_lutc_per_tensor(const void* x, float* y, int64_t n, int32_t dtype, int32_t step_round, float thr_div,
                         float thr_mul, const void* blob, int32_t n_words, float mult, float clip_min,
                         float clip_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (step_round != 0 && step_round != MCTQ_DT_F16 && step_round != MCTQ_DT_BF16) return fail_arg("bad step_round");
  LutCompactOp op;
  if (int rc = make_compact_op(op, nullptr, 0.f, blob, n_words, mult, clip_min, clip_max, step_round)) return rc;
  const LutCommon::Param p = LutCommon::make(thr_div, thr_mul, mult);
  return with_lut_types(dtype, [&](auto ti, auto to) {
    return launch_flat<decltype(ti), decltype(to)>(op, p, x, y, n, compact_bytes(n_words), (hipStream_t)stream);
  });
}

int mctq_lutc_per_channel(const void* x, float* y, int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                          const float* thresholds, float eps, const void* blob, int32_t n_words, float mult,
                          float clip_min, float clip_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !thresholds)) return fail_arg("NULL pointer");
  LutCompactOp op;
  if (int rc = make_compact_op(op, thresholds, eps, blob, n_words, mult, clip_min, clip_max, 0)) return rc;
  return with_lut_types(dtype, [&](auto ti, auto to) {
    return launch_channels<decltype(ti), decltype(to)>(op, x, y, outer, channels, inner, compact_bytes(n_words),
                                                       (hipStream_t)stream);
  });
}

}  // extern "C"
