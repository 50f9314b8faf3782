// mctq_lut_scan.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h); kernels: mctq_kernels.hpp
#include "mctq_kernels.hpp"

using namespace mctq;

namespace mctq {        // mctq_f64.hip
int lut64_per_channel(const void* x, float* y, int64_t outer, int64_t channels, int64_t inner, const float* thr, float eps,
                      const float* lut, int32_t n_lut, float mult, float cmin, float cmax, hipStream_t st);
}

extern "C" {

// ---- LUT, literal scan ----------------------------------------------------------------------------

int mctq_lut_per_tensor(const void* x, float* y, int64_t n, int32_t dtype, int32_t step_round, float thr_div,
                        float thr_mul, const float* lut, int32_t n_lut, float mult, float clip_min, float clip_max,
                        void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (int rc = check_lut_args(lut, n_lut, mult)) return rc;
  if (step_round != 0 && step_round != MCTQ_DT_F16 && step_round != MCTQ_DT_BF16) return fail_arg("bad step_round");
  hipStream_t st = (hipStream_t)stream;
  const LutCommon::Param p = LutCommon::make(thr_div, thr_mul, mult);
  if (dtype == MCTQ_DT_F64)        // float64 tensor, float32 divisor widened (weights quantizers, per_channel=False)
    return mctq_lut_per_tensor_f64(static_cast<const double*>(x), y, n, (double)thr_div, thr_mul, lut, n_lut, mult,
                                   clip_min, clip_max, stream);
  return with_lut_types(dtype, [&](auto ti, auto to) {
    typedef decltype(ti) TI;
    typedef decltype(to) TO;
    switch (lut_class(n_lut)) {
      case 16: return launch_flat<TI, TO>(make_lut_op<16>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max, step_round),
                                          p, x, y, n, 0, st);
      default: return launch_flat<TI, TO>(make_lut_op<0>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max, step_round),
                                          p, x, y, n, (size_t)((n_lut + 3) & ~3) * sizeof(float), st);
    }
  });
}

int mctq_lut_per_channel(const void* x, float* y, int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                         const float* thresholds, float eps, const float* lut, int32_t n_lut, float mult,
                         float clip_min, float clip_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !thresholds)) return fail_arg("NULL pointer");
  if (int rc = check_lut_args(lut, n_lut, mult)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MCTQ_DT_F64)
    return lut64_per_channel(x, y, outer, channels, inner, thresholds, eps, lut, n_lut, mult, clip_min, clip_max, st);
  return with_lut_types(dtype, [&](auto ti, auto to) {
    typedef decltype(ti) TI;
    typedef decltype(to) TO;
    switch (lut_class(n_lut)) {
      case 16: return launch_channels<TI, TO>(make_lut_op<16>(thresholds, eps, lut, n_lut, mult, clip_min, clip_max, 0),
                                              x, y, outer, channels, inner, 0, st);
      default: return launch_channels<TI, TO>(make_lut_op<0>(thresholds, eps, lut, n_lut, mult, clip_min, clip_max, 0),
                                              x, y, outer, channels, inner, (size_t)((n_lut + 3) & ~3) * sizeof(float), st);
    }
  });
}

int mctq_lut_per_tensor_f32(const float* x, float* y, int64_t n, float thr_div, float thr_mul, const float* lut,
                            int32_t n_lut, float mult, float clip_min, float clip_max, void* stream) {
  return mctq_lut_per_tensor(x, y, n, MCTQ_DT_F32, 0, thr_div, thr_mul, lut, n_lut, mult, clip_min, clip_max, stream);
}

int mctq_lut_per_channel_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                             const float* thresholds, float eps, const float* lut, int32_t n_lut, float mult,
                             float clip_min, float clip_max, void* stream) {
  return mctq_lut_per_channel(x, y, outer, channels, inner, MCTQ_DT_F32, thresholds, eps, lut, n_lut, mult, clip_min,
                              clip_max, stream);
}

}  // extern "C"
