// lut_conflicts.hip -- EXPERIMENT, not part of the shipped library: the decision-table LUT kernel of config 4
// with other LDS layouts of the SAME table, to see what the bank conflicts of its 8-byte gather cost and whether a layout
// can remove them.  One rows_kernel instantiation per MODE (U = 4, NT = 1, float32):
//   0  shipped layout: entry k = {T_k, half2(q below, q above)} at 8 k (one ds_read_b64 per element)
//   1  swizzled cell index: entry k lives at k ^ ((k >> 4) & 15)          (cells 16 apart no longer share a bank pair)
//   2  two 4-byte planes: T_k at dword k, the half2 at dword 528 + k         (two ds_read_b32, planes 16 banks apart)
//   3  NO conflicts, results WRONG: every lane reads cell 255 (the upper bound of what any layout could gain)
//   4  half the gather, results WRONG: only T_k is read (4-byte random gather), the centres are constants
// Modes 0-2 are exact (checked against mode 0 by tools/experiments/lut_conflicts/run.py); 3 and 4 are timing-only.
// Build: python tools/build_variant.py lut_conflicts   ->  tools/ablate/libmctq_hip_lut_conflicts.so
#include "mctq_kernels.hpp"

namespace mctq {

template <int MODE>
struct LutTableXOp : LutTableOp {
  static constexpr const char* kName = MODE == 0 ? "LutTableX<shipped>" : MODE == 1 ? "LutTableX<swizzle>"
      : MODE == 2 ? "LutTableX<planes>" : MODE == 3 ? "LutTableX<one cell>" : "LutTableX<T only>";
  static constexpr int kFixedU = 4;
  static constexpr uint32_t kPlane = 528;                 // dwords between the two planes of mode 2 (16 banks apart)

  __device__ __forceinline__ uint32_t book_words() const { return 1064; }       // >= 2 * 512 + 2 (swizzled slots) and 528 + 512
  __device__ __forceinline__ static uint32_t swz(uint32_t k) { return MODE == 1 ? (k ^ ((k >> 4) & 15u)) : k; }

  __device__ __forceinline__ Book commit(const Prefetch& p, float* lds) const {
    f32x2* dst = reinterpret_cast<f32x2*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = (int)threadIdx.x + i * kThreads;
      if (j <= entries) {
        if (MODE == 2 || MODE == 4) { if (j < entries) { lds[j] = p.r[i].x; lds[kPlane + j] = p.r[i].y; } else lds[1060] = p.r[i].x; }
        else if (j < entries) dst[swz((uint32_t)j)] = p.r[i];
        else lds[1060] = p.r[i].x;                         // q for NaN input, outside the swizzled range
      }
    }
    __syncthreads();
    Book b; b.tab = dst; b.nan_q = lds[1060];
    return b;
  }
  __device__ __forceinline__ Book setup(float* lds) const { return commit(prefetch(), lds); }

  template <bool FAST, int NE>
  __device__ __forceinline__ void tile(const float* in, float* out, const Param& p, const Book& b) const {
    float v[NE];
    int k[NE];
    f32x2 e[NE];
    const float* lds = reinterpret_cast<const float*>(b.tab);
#pragma unroll
    for (int i = 0; i < NE; ++i) locate<FAST>(in[i], p, v[i], k[i]);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      if (MODE == 2) { e[i].x = lds[k[i]]; e[i].y = lds[kPlane + k[i]]; }
      else if (MODE == 3) { e[i] = b.tab[255]; }
      else if (MODE == 4) { e[i].x = lds[k[i]]; e[i].y = __uint_as_float(0x3c003800u); }
      else e[i] = b.tab[swz((uint32_t)k[i])];
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) out[i] = decide<FAST>(in[i], v[i], e[i], p, b);
  }
  template <bool FAST = false>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book& b) const {
    float in[1] = {x}, out[1];
    tile<FAST, 1>(in, out, p, b);
    return out[0];
  }
};

}  // namespace mctq

using namespace mctq;

extern "C" int mctq_x_lutt_per_channel_f32(int32_t mode, const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                                           const float* thresholds, float eps, const float* table, int32_t entries, float mult,
                                           float clip_min, float clip_max, void* stream) {
  const auto run = [&](auto op) {
    if (int rc = make_table_op(op, thresholds, eps, table, entries, mult, clip_min, clip_max, 0)) return rc;
    return launch_channels<float, float>(op, x, y, outer, channels, inner, (size_t)1064 * 4, (hipStream_t)stream);
  };
  switch (mode) {
    case 0: return run(LutTableXOp<0>());
    case 1: return run(LutTableXOp<1>());
    case 2: return run(LutTableXOp<2>());
    case 3: return run(LutTableXOp<3>());
    case 4: return run(LutTableXOp<4>());
    default: return fail_arg("mode");
  }
}
