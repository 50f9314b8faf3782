// mctq_codes4.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h).
//
// 4-bit code output of the affine quantizers: the clamp index of num_bits <= 4 quantizers stored two per byte
// (element 2j of the storage order in the low nibble, 2j + 1 in the high nibble; signed codes as two's-complement
// nibbles) -- 0.5 B written per element instead of 4 (4.5 instead of 8 algorithmic bytes per float32 element).
// Same arithmetic as every other affine kernel (AffineOp::make / the codes op): q = clamp(rint(x * (1/s)) + z).
// A lane vector is one 16-byte load (4 float32 or 8 half-precision elements -> 2 or 4 bytes of codes), so the
// supported layouts are the vector-aligned ones: per tensor with n % 8 == 0, per channel with
// inner % 8 == 0 (one block = one tile of one row, parameters through scalar loads) or inner == 1 with
// channels % 8 == 0 (lanes keep their 8 channels while stepping down the rows, as lastaxis_kernel).
#include "mctq_kernels.hpp"

namespace mctq {

constexpr int kC4U = 8;                       // lane vectors per lane: 8 x 16 B of input in flight

// A lane vector is one 16-byte load: 4 float32 elements -> 2 bytes of codes, or 8 half-precision elements ->
// 4 bytes; consecutive lanes read consecutive 16-byte pieces (full lines per wave instruction).
template <class TI> struct C4Vec;
template <> struct C4Vec<float> {
  static constexpr int N = 4;
  typedef uint16_t Out;
  template <int NT>
  __device__ __forceinline__ static void load(const float* p, float* f) {
    const f32x4 a = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = a[i];
  }
};
template <> struct C4Vec<_Float16> {
  static constexpr int N = 8;
  typedef uint32_t Out;
  template <int NT>
  __device__ __forceinline__ static void load(const _Float16* p, float* f) {
    const f16x8 a = NT ? __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p)) : *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)a[i];
  }
};
template <> struct C4Vec<__bf16> {
  static constexpr int N = 8;
  typedef uint32_t Out;
  template <int NT>
  __device__ __forceinline__ static void load(const __bf16* p, float* f) {
    const b16x8 a = NT ? __builtin_nontemporal_load(reinterpret_cast<const b16x8*>(p)) : *reinterpret_cast<const b16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)a[i];
  }
};

// The code's nibble without an int conversion: with M = 2^23 + 16, rint(x * inv) + (z + M) clamped to [lo + M, hi + M] is
// an integer-valued float in [2^23 + 8, 2^23 + 31] (one binade, ulp 1) whose low mantissa bits ARE the two's-complement nibble
// (M = 0 mod 16; the sums are exact: |z|, |lo|, |hi| <= 15).  NaN -> lo, as the 8-bit codes op; the caller masks.
__device__ __forceinline__ uint32_t c4_code(float x, const AffineOp::Param& p, float lo, float hi) {
  float q = __builtin_rintf(x * p.inv) + (p.zf + 8388624.0f);
  q = fminf(fmaxf(q, lo + 8388624.0f), hi + 8388624.0f);
  return __float_as_uint(q);
}
template <int N>
__device__ __forceinline__ uint32_t c4_pack(const uint32_t* c) {
  uint32_t w = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) w |= (c[i] & 0xFu) << (4 * i);    // element i -> nibble i: bytes hold (2j, 2j + 1)
  return w;
}

// per tensor: nv lane vectors
template <class TI>
__global__ __launch_bounds__(kThreads) void codes4_flat_kernel(const TI* __restrict__ x, typename C4Vec<TI>::Out* __restrict__ y,
                                                               int64_t nv, AffineOp::Param p, float lo, float hi) {
  typedef C4Vec<TI> V;
  const int64_t base = (int64_t)blockIdx.x * (kThreads * kC4U) + threadIdx.x;
  float f[kC4U][V::N];
#pragma unroll
  for (int u = 0; u < kC4U; ++u)
    if (base + u * kThreads < nv) V::template load<1>(x + (base + u * kThreads) * V::N, f[u]);
#pragma unroll
  for (int u = 0; u < kC4U; ++u) {
    const int64_t v = base + u * kThreads;
    if (v < nv) {
      uint32_t c[V::N];
#pragma unroll
      for (int i = 0; i < V::N; ++i) c[i] = c4_code(f[u][i], p, lo, hi);
      __builtin_nontemporal_store((typename V::Out)c4_pack<V::N>(c), &y[v]);
    }
  }
}

// per channel, inner % 8 == 0: block = (row, tile); the row's parameters are wave-uniform (scalar loads)
template <class TI, int U>
__global__ __launch_bounds__(kThreads) void codes4_rows_kernel(const TI* __restrict__ x, typename C4Vec<TI>::Out* __restrict__ y,
                                                               uint32_t tiles, uint32_t innerv, uint32_t channels,
                                                               const float* __restrict__ scales,
                                                               const int32_t* __restrict__ zps, float lo, float hi) {
  typedef C4Vec<TI> V;
  const uint32_t row = blockIdx.x / tiles, tile = blockIdx.x - row * tiles;
  const uint32_t first = tile * (kThreads * U) + threadIdx.x;
  const int64_t rbase = (int64_t)row * innerv;
  float f[U][V::N];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (first + u * kThreads < innerv) V::template load<1>(x + (rbase + first + u * kThreads) * V::N, f[u]);
  const uint32_t c = row % channels;
  const AffineOp::Param p = AffineOp::make(scales[c], zps ? zps[c] : 0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t v = first + u * kThreads;
    if (v < innerv) {
      uint32_t cd[V::N];
#pragma unroll
      for (int i = 0; i < V::N; ++i) cd[i] = c4_code(f[u][i], p, lo, hi);
      __builtin_nontemporal_store((typename V::Out)c4_pack<V::N>(cd), &y[rbase + v]);
    }
  }
}

// per channel along the fastest axis (inner == 1, channels % 8 == 0): the lanes of bps neighbouring blocks cover k
// whole rows and step down k rows at a time, keeping their channels (cf. lastaxis_kernel)
template <class TI>
__global__ __launch_bounds__(kThreads) void codes4_lastaxis_kernel(const TI* __restrict__ x,
                                                                   typename C4Vec<TI>::Out* __restrict__ y, uint64_t rows,
                                                                   uint32_t vc, uint32_t k, uint32_t bps,
                                                                   const float* __restrict__ scales,
                                                                   const int32_t* __restrict__ zps, float lo, float hi) {
  typedef C4Vec<TI> V;
  const uint32_t g = (blockIdx.x % bps) * kThreads + threadIdx.x;
  const uint32_t ro = g / vc, col = g - ro * vc;
  if (ro >= k) return;
  const uint64_t row0 = (uint64_t)(blockIdx.x / bps) * ((uint64_t)kC4U * k) + ro;
  float f[kC4U][V::N];
#pragma unroll
  for (int u = 0; u < kC4U; ++u) {
    const uint64_t row = row0 + (uint64_t)u * k;
    if (row < rows) V::template load<1>(x + (row * vc + col) * V::N, f[u]);
  }
  AffineOp::Param p[V::N];
#pragma unroll
  for (int i = 0; i < V::N; ++i) p[i] = AffineOp::make(scales[col * V::N + i], zps ? zps[col * V::N + i] : 0);
#pragma unroll
  for (int u = 0; u < kC4U; ++u) {
    const uint64_t row = row0 + (uint64_t)u * k;
    if (row < rows) {
      uint32_t cd[V::N];
#pragma unroll
      for (int i = 0; i < V::N; ++i) cd[i] = c4_code(f[u][i], p[i], lo, hi);
      __builtin_nontemporal_store((typename V::Out)c4_pack<V::N>(cd), &y[row * vc + col]);
    }
  }
}

template <class F>
static int with_in_types(int dtype, F f) {
  switch (dtype) {
    case MCTQ_DT_F32: return f(float());
    case MCTQ_DT_F16: return f(_Float16());
    case MCTQ_DT_BF16: return f(__bf16());
    default: return fail_arg("unknown dtype");
  }
}

int codes4_per_tensor(const void* x, void* codes, int64_t n, int dtype, float scale, int zero_point, int qmin, int qmax,
                      hipStream_t st) {
  if (n % 8 != 0) return fail_arg("4-bit codes: n must be a multiple of 8");
  if ((((uintptr_t)x) & 15u) || (((uintptr_t)codes) & 3u)) return fail_arg("4-bit codes: x must be 16-byte and codes 4-byte aligned");
  if (n == 0) return 0;
  const AffineOp::Param p = AffineOp::make(scale, zero_point);
  return with_in_types(dtype, [&](auto ti) {
    typedef decltype(ti) TI;
    typedef C4Vec<TI> V;
    const int64_t nv = n / V::N, blocks = (nv + kThreads * kC4U - 1) / (kThreads * kC4U);
    if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
    hipLaunchKernelGGL((codes4_flat_kernel<TI>), dim3((unsigned)blocks), dim3(kThreads), 0, st, static_cast<const TI*>(x),
                       static_cast<typename V::Out*>(codes), nv, p, (float)qmin, (float)qmax);
    return check_launch("codes4 flat launch");
  });
}

int codes4_per_channel(const void* x, void* codes, int64_t outer, int64_t channels, int64_t inner, int dtype,
                       const float* scales, const int32_t* zps, int qmin, int qmax, hipStream_t st) {
  const int64_t n = outer * channels * inner;
  if (n == 0) return 0;
  if ((((uintptr_t)x) & 15u) || (((uintptr_t)codes) & 3u)) return fail_arg("4-bit codes: x must be 16-byte and codes 4-byte aligned");
  if (inner % 8 == 0) {
    return with_in_types(dtype, [&](auto ti) {
      typedef decltype(ti) TI;
      typedef C4Vec<TI> V;
      const int64_t innerv = inner / V::N, rows = outer * channels;
      // lane vectors per lane: 8, or 4 when that leaves fewer idle slots in a row's last tile
      const auto waste = [&](int64_t u) { const int64_t per = kThreads * u; return (per - innerv % per) % per; };
      const int u_sel = waste(4) < waste(8) ? 4 : 8;
      const int64_t tiles = (innerv + kThreads * u_sel - 1) / (kThreads * u_sel);
      if (rows * tiles > 0x7fffffffLL || innerv > 0x7fffffffLL || channels > 0x7fffffffLL)
        return fail_arg("tensor too large for one launch");
      if (u_sel == 4)
        hipLaunchKernelGGL((codes4_rows_kernel<TI, 4>), dim3((unsigned)(rows * tiles)), dim3(kThreads), 0, st,
                           static_cast<const TI*>(x), static_cast<typename V::Out*>(codes), (uint32_t)tiles, (uint32_t)innerv,
                           (uint32_t)channels, scales, zps, (float)qmin, (float)qmax);
      else
        hipLaunchKernelGGL((codes4_rows_kernel<TI, 8>), dim3((unsigned)(rows * tiles)), dim3(kThreads), 0, st,
                           static_cast<const TI*>(x), static_cast<typename V::Out*>(codes), (uint32_t)tiles, (uint32_t)innerv,
                           (uint32_t)channels, scales, zps, (float)qmin, (float)qmax);
      return check_launch("codes4 rows launch");
    });
  }
  if (inner == 1 && channels % 8 == 0) {
    return with_in_types(dtype, [&](auto ti) {
      typedef decltype(ti) TI;
      typedef C4Vec<TI> V;
      const int64_t vc = channels / V::N;
      int64_t k = (2048 + vc - 1) / vc, best = -1;
      for (int64_t c = k; c < k + 16; ++c) {
        const int64_t waste = (kThreads - (c * vc) % kThreads) % kThreads * 4096 / (c * vc);
        if (best < 0 || waste < best) { best = waste; k = c; }
      }
      const int64_t bps = (k * vc + kThreads - 1) / kThreads;
      const int64_t blocks = bps * ((outer + kC4U * k - 1) / (kC4U * k));
      if (blocks > 0x7fffffffLL || k * vc > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
      hipLaunchKernelGGL((codes4_lastaxis_kernel<TI>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                         static_cast<const TI*>(x), static_cast<typename V::Out*>(codes), (uint64_t)outer, (uint32_t)vc,
                         (uint32_t)k, (uint32_t)bps, scales, zps, (float)qmin, (float)qmax);
      return check_launch("codes4 lastaxis launch");
    });
  }
  return fail_arg("4-bit codes need inner % 8 == 0, or inner == 1 with channels % 8 == 0");
}

}  // namespace mctq
