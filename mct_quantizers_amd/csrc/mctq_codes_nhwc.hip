// mctq_codes_nhwc.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h).
//
// Per-tensor activation codes with the layout change a pointwise-convolution consumer wants: x is [B][C][P]
// (NCHW with P = H * W), the 8-bit codes leave as [B][P][C] (NHWC) -- one pass at 5 (float32) or 3 (16-bit)
// algorithmic bytes per element instead of a codes pass plus a byte transposition.  Same arithmetic as the codes
// kernels (AffineOp::make; q = clamp(rint(x * (1/s)) + z)).  A block moves a 128 (channels) x 32 (pixels) tile:
// 128-byte row pieces in (four consecutive pixels per lane), quantized bytes transposed through LDS, 128-byte row
// pieces out (16 consecutive channels per lane).
#include "mctq_kernels.hpp"

namespace mctq {

constexpr int kTC = 128, kTP = 32;            // tile: channels x pixels

template <class TI>
__global__ __launch_bounds__(kThreads) void codes_nchw_to_nhwc_kernel(const TI* __restrict__ x, uint8_t* __restrict__ y,
                                                                      int C, int64_t P, int c_tiles, int64_t p_tiles,
                                                                      AffineOp::Param prm, float lo, float hi, int vec_ok) {
  __shared__ uint8_t tile[kTP][kTC + 16];                 // [pixel][channel]; padded rows: 16-byte chunks stay aligned
  int64_t id = blockIdx.x;
  const int64_t pt = id % p_tiles; id /= p_tiles;
  const int ct = (int)(id % c_tiles);
  const int64_t b = id / c_tiles;
  const int c0 = ct * kTC;
  const int64_t p0 = pt * kTP;
  const TI* xb = x + (b * C) * P;
  // in: thread t -> channel row t / 8 + 32 j (j = 0..3), pixels 4 (t % 8) .. +3
  const int pr = (threadIdx.x & 7) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cr = (threadIdx.x >> 3) + 32 * j;
    const int c = c0 + cr;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
      const TI* src = xb + (int64_t)c * P + p0 + pr;
      if (vec_ok && p0 + pr + 4 <= P) {
        typedef typename VecT<TI, 4>::type V4;
        const V4 v = __builtin_nontemporal_load(reinterpret_cast<const V4*>(src));
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (p0 + pr + i < P) f[i] = (float)src[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float q = __builtin_rintf(f[i] * prm.inv) + prm.zf;
      q = fminf(fmaxf(q, lo), hi);
      tile[pr + i][cr] = (uint8_t)(int)q;                  // int8 codes: the low byte is the two's-complement value
    }
  }
  __syncthreads();
  // out: thread t -> pixel t / 8, channels 16 (t % 8) .. +15
  const int op = threadIdx.x >> 3, oc = (threadIdx.x & 7) * 16;
  if (p0 + op < P && c0 + oc < C) {
    uint8_t* dst = y + ((b * P + p0 + op) * C) + c0 + oc;
    if (c0 + oc + 16 <= C && (C & 15) == 0) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&tile[op][oc]), reinterpret_cast<u32x4*>(dst));
    } else {
      for (int i = 0; i < 16 && c0 + oc + i < C; ++i) dst[i] = tile[op][oc + i];
    }
  }
}

}  // namespace mctq

using namespace mctq;

extern "C" {

int mctq_fq_codes_nchw_to_nhwc(const void* x, void* codes, int64_t batch, int64_t channels, int64_t pixels, int32_t dtype,
                               int32_t code_dtype, float scale, int32_t zero_point, int32_t quant_min, int32_t quant_max,
                               void* stream) {
  if (batch < 0 || channels < 0 || pixels < 0) return fail_arg("negative extent");
  if (quant_min > quant_max) return fail_arg("quant_min > quant_max");
  if (code_dtype == MCTQ_CODE_I8 ? (quant_min < -128 || quant_max > 127)
                                 : code_dtype == MCTQ_CODE_U8 ? (quant_min < 0 || quant_max > 255) : true)
    return fail_arg("code_dtype must be MCTQ_CODE_I8 / MCTQ_CODE_U8 with a clamp domain that fits it");
  if (batch * channels * pixels == 0) return 0;
  if (!x || !codes) return fail_arg("x or codes is NULL");
  if (channels > 0x7fffffffLL) return fail_arg("too many channels");
  const int64_t c_tiles = (channels + kTC - 1) / kTC, p_tiles = (pixels + kTP - 1) / kTP;
  const int64_t blocks = batch * c_tiles * p_tiles;
  if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
  const AffineOp::Param p = AffineOp::make(scale, zero_point);
  const hipStream_t st = (hipStream_t)stream;
  const auto launch = [&](auto ti) {
    typedef decltype(ti) TI;
    const int vec_ok = (pixels % 4 == 0) && ((uintptr_t)x % (4 * sizeof(TI)) == 0);
    hipLaunchKernelGGL((codes_nchw_to_nhwc_kernel<TI>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                       static_cast<const TI*>(x), static_cast<uint8_t*>(codes), (int)channels, pixels, (int)c_tiles, p_tiles, p,
                       (float)quant_min, (float)quant_max, vec_ok);
    return check_launch("codes nchw->nhwc launch");
  };
  switch (dtype) {
    case MCTQ_DT_F32: return launch(float());
    case MCTQ_DT_F16: return launch(_Float16());
    case MCTQ_DT_BF16: return launch(__bf16());
    default: return fail_arg("unknown dtype");
  }
}

}  // extern "C"
