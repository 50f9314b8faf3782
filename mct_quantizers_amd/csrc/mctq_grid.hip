// mctq_grid.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h); kernels: mctq_kernels.hpp
#include "mctq_kernels.hpp"

using namespace mctq;

extern "C" {

// ---- export-time grid arithmetic (clip -> true division -> round -> scale) ---------------------------

int mctq_grid_per_tensor_f32(const float* x, float* y, int64_t n, float lo, float hi, float step, int32_t shifted,
                             void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (shifted != 0 && shifted != 1) return fail_arg("shifted must be 0 or 1");
  GridOp op;
  op.los = nullptr; op.his = nullptr; op.steps = nullptr; op.shifted = shifted;
  const GridOp::Param p = GridOp::make(lo, hi, step);
  return launch_flat<float, float>(op, p, x, y, n, 0, (hipStream_t)stream);
}

int mctq_grid_per_channel_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                              const float* los, const float* his, const float* steps, int32_t shifted, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  if (shifted != 0 && shifted != 1) return fail_arg("shifted must be 0 or 1");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !los || !his || !steps)) return fail_arg("NULL pointer");
  GridOp op;
  op.los = los; op.his = his; op.steps = steps; op.shifted = shifted;
  return launch_channels<float, float>(op, x, y, outer, channels, inner, 0, (hipStream_t)stream);
}

}  // extern "C"
