// mctq_codes.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h); kernels: mctq_kernels.hpp
// Integer-code outputs of the affine quantizers.
#include "mctq_kernels.hpp"

using namespace mctq;

namespace mctq {        // mctq_codes4.hip
int codes4_per_tensor(const void* x, void* codes, int64_t n, int dtype, float scale, int zero_point, int qmin, int qmax,
                      hipStream_t st);
int codes4_per_channel(const void* x, void* codes, int64_t outer, int64_t channels, int64_t inner, int dtype,
                       const float* scales, const int32_t* zps, int qmin, int qmax, hipStream_t st);
}

static int check_code_range(int32_t code_dtype, int32_t qmin, int32_t qmax) {
  if (qmin > qmax) return fail_arg("quant_min > quant_max");
  if (code_dtype == MCTQ_CODE_I4 && (qmin < -8 || qmax > 7)) return fail_arg("clamp domain does not fit int4");
  if (code_dtype == MCTQ_CODE_U4 && (qmin < 0 || qmax > 15)) return fail_arg("clamp domain does not fit uint4");
  if (code_dtype == MCTQ_CODE_I8 && (qmin < -128 || qmax > 127)) return fail_arg("clamp domain does not fit int8");
  if (code_dtype == MCTQ_CODE_U8 && (qmin < 0 || qmax > 255)) return fail_arg("clamp domain does not fit uint8");
  return 0;
}

extern "C" {

int mctq_fq_codes_per_tensor(const void* x, void* codes, int64_t n, int32_t dtype, int32_t code_dtype, float scale,
                             int32_t zero_point, int32_t quant_min, int32_t quant_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !codes)) return fail_arg("x or codes is NULL");
  if (int rc = check_code_range(code_dtype, quant_min, quant_max)) return rc;
  if (code_dtype == MCTQ_CODE_I4 || code_dtype == MCTQ_CODE_U4)
    return codes4_per_tensor(x, codes, n, dtype, scale, zero_point, quant_min, quant_max, (hipStream_t)stream);
  AffineCodesOp op;
  op.scales = nullptr; op.zps = nullptr;
  op.lo = (float)quant_min; op.hi = (float)quant_max;
  const AffineOp::Param p = AffineOp::make(scale, zero_point);
  return with_codes_types(dtype, code_dtype, [&](auto ti, auto to) {
    return launch_flat<decltype(ti), decltype(to)>(op, p, x, codes, n, 0, (hipStream_t)stream);
  });
}

int mctq_fq_codes_per_channel(const void* x, void* codes, int64_t outer, int64_t channels, int64_t inner,
                              int32_t dtype, int32_t code_dtype, const float* scales, const int32_t* zero_points,
                              int32_t quant_min, int32_t quant_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !codes || !scales)) return fail_arg("NULL pointer");
  if (int rc = check_code_range(code_dtype, quant_min, quant_max)) return rc;
  if (code_dtype == MCTQ_CODE_I4 || code_dtype == MCTQ_CODE_U4)
    return codes4_per_channel(x, codes, outer, channels, inner, dtype, scales, zero_points, quant_min, quant_max,
                              (hipStream_t)stream);
  AffineCodesOp op;
  op.scales = scales; op.zps = zero_points;
  op.lo = (float)quant_min; op.hi = (float)quant_max;
  return with_codes_types(dtype, code_dtype, [&](auto ti, auto to) {
    return launch_channels<decltype(ti), decltype(to)>(op, x, codes, outer, channels, inner, 0, (hipStream_t)stream);
  });
}

}  // extern "C"
