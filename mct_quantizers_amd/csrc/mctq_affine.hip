// mctq_affine.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h); kernels: mctq_kernels.hpp
#include "mctq_kernels.hpp"

using namespace mctq;

namespace mctq {        // mctq_f64.hip
int fq64_per_tensor(const void* x, void* y, int64_t n, float scale, int32_t zp, int32_t qmin, int32_t qmax, hipStream_t st);
int fq64_per_channel(const void* x, void* y, int64_t outer, int64_t channels, int64_t inner, const float* scales,
                     const int32_t* zps, int32_t qmin, int32_t qmax, bool wide, hipStream_t st);
// mctq_batched.hip: one float32 tensor through the batched grid's tile code (per-lane-vector parameters)
int fq_gather_one_f32(const void* x, void* y, int64_t outer, int64_t channels, int64_t inner, const float* scales,
                      const int32_t* zps, int32_t qmin, int32_t qmax, hipStream_t st);
extern int g_shortrows;
extern int g_paced;
}

extern "C" {

// ---- affine ---------------------------------------------------------------------------------------

int mctq_fq_per_tensor(const void* x, void* y, int64_t n, int32_t dtype, float scale, int32_t zero_point,
                       int32_t quant_min, int32_t quant_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (quant_min > quant_max) return fail_arg("quant_min > quant_max");
  if (dtype == MCTQ_DT_F64) return fq64_per_tensor(x, y, n, scale, zero_point, quant_min, quant_max, (hipStream_t)stream);
  AffineOp op;
  op.scales = nullptr; op.zps = nullptr;
  op.lo = (float)quant_min; op.hi = (float)quant_max;
  const AffineOp::Param p = AffineOp::make(scale, zero_point);   // host IEEE division == ATen's 1.0f / scale
  return with_affine_types(dtype, [&](auto ti, auto to) {
    return launch_flat<decltype(ti), decltype(to)>(op, p, x, y, n, 0, (hipStream_t)stream);
  });
}

int mctq_fq_per_channel(const void* x, void* y, int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                        const float* scales, const int32_t* zero_points, int32_t quant_min, int32_t quant_max,
                        void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  if (quant_min > quant_max) return fail_arg("quant_min > quant_max");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !scales)) return fail_arg("NULL pointer");
  if (dtype == MCTQ_DT_F64)
    return fq64_per_channel(x, y, outer, channels, inner, scales, zero_points, quant_min, quant_max, true, (hipStream_t)stream);
  // float32 rows that are neither long and vector-divisible (rows_kernel) nor the fastest axis (lastaxis_kernel), of at
  // least 32 elements: the batched grid's per-lane-vector parameter path (no LDS window, no block barrier) as a one-tensor
  // launch (descriptor in preloaded scalar arguments) is 6-9 % faster than window_kernel there (inner 64 ... 1020, 4099-wide rows: 24.3 -> 22.4 us per 128 MiB,
  // profiles/r03/short_rows_probe.log); 16-bit storage and inner < 32 stay with the window kernel (equal or better).
  // ... and long vector-divisible rows whose last row tile would be mostly idle lanes (4100-wide: 1025 lane-vectors = four
  // full 256-lane tiles + one lane: 23.1 -> 22.5 us); rows that nearly fill their tiles (11008-wide: 2 % idle) stay.
  // ... and rows of 256-511 lane-vectors, which rows_kernel could only cut into 4 KiB tiles (16384 x 1024: 22.9 -> 22.3 us).
  bool long_rows = inner % 4 == 0 && inner / 4 >= 2 * kThreads;
  if (long_rows) {
    const int64_t innerv = inner / 4, tiles = (innerv + kThreads - 1) / kThreads;
    if ((tiles * kThreads - innerv) * 12 > tiles * kThreads) long_rows = false;      // > 1/12 of the row's lanes idle
  }
  // (symmetric float32 launches of 3/4 ... 1 round with whole-vector rows: shortrows_kernel with paced stores, launch_channels)
  const bool paced_window = g_paced == 1 && g_shortrows != 0 && !zero_points && paced_rows_window<float>(n, inner, channels);
  if (dtype == MCTQ_DT_F32 && inner >= 32 && channels > 1 && !long_rows && g_shortrows != 2 && !paced_window &&
      (((uintptr_t)x | (uintptr_t)y) & 15u) == 0 && n < (1ll << 31) - 4096 && channels <= 0x7fffffffLL) {
    return fq_gather_one_f32(x, y, outer, channels, inner, scales, zero_points, quant_min, quant_max, (hipStream_t)stream);
  }
  AffineOp op;
  op.scales = scales; op.zps = zero_points;
  op.lo = (float)quant_min; op.hi = (float)quant_max;
  return with_affine_types(dtype, [&](auto ti, auto to) {
    return launch_channels<decltype(ti), decltype(to)>(op, x, y, outer, channels, inner, 0, (hipStream_t)stream);
  });
}

// Per-tensor fake-quant with the scale and zero point READ ON THE DEVICE (1-element tensors): the
// tensor-qparams overload of torch.fake_quantize_per_tensor_affine, which is what the per-tensor weights
// quantizers call (weights_symmetric_inferable_quantizer.py:147-151) and what an fx trace of a wrapper records.
// One row of one channel: the parameters arrive through scalar loads, no device->host read anywhere.
int mctq_fq_per_tensor_tqp(const void* x, void* y, int64_t n, int32_t dtype, const float* scale,
                           const int32_t* zero_point, int32_t quant_min, int32_t quant_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!scale || !zero_point)) return fail_arg("scale or zero_point is NULL");
  if (dtype == MCTQ_DT_F64) {          // per-tensor flavour of the float64 arithmetic (float32 product, widened)
    if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
    if (quant_min > quant_max) return fail_arg("quant_min > quant_max");
    return fq64_per_channel(x, y, n > 0 ? 1 : 0, 1, n, scale, zero_point, quant_min, quant_max, false, (hipStream_t)stream);
  }
  return mctq_fq_per_channel(x, y, n > 0 ? 1 : 0, 1, n, dtype, scale, zero_point, quant_min, quant_max, stream);
}

int mctq_fq_per_tensor_f32(const float* x, float* y, int64_t n, float scale, int32_t zero_point, int32_t quant_min,
                           int32_t quant_max, void* stream) {
  return mctq_fq_per_tensor(x, y, n, MCTQ_DT_F32, scale, zero_point, quant_min, quant_max, stream);
}

int mctq_fq_per_channel_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                            const float* scales, const int32_t* zero_points, int32_t quant_min, int32_t quant_max,
                            void* stream) {
  return mctq_fq_per_channel(x, y, outer, channels, inner, MCTQ_DT_F32, scales, zero_points, quant_min, quant_max,
                             stream);
}

}  // extern "C"
