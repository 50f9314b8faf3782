// mctq_f64.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h): float64 tensors.
//
// The reference hands whatever tensor it gets to ATen (weights_symmetric_inferable_quantizer.py:139-151,
// activation_uniform_inferable_quantizer.py:124, quantizer_utils.py:126-137), and ATen's float64 arithmetic is NOT
// "float32 arithmetic on wider storage" (measured against the reference in this repository's goldens,
// tests/golden/cases_f64.*):
//   affine, every overload:  q = clamp(rint(x * (double)(1.0f / s)) + z, qmin, qmax)          (double product and rounding)
//     per tensor (float or tensor qparams):  y = (double)( (float)(q - z) * s )                (float32 product, widened)
//     per channel:                           y = (double)(q - z) * (double)s                   (double product)
//   LUT:  t = clip((x / (double)d) * m, cmin, cmax) in double, first-minimum argmin of |t - (double)lut[j]| in double,
//         y = float32( (lut[j] / m) * thr )  -- the OUTPUT is float32, as for half inputs (the float32 codebook and
//         threshold tensors decide the result type).  d = fl32(thr + fl32(eps)) per channel (a float32 tensor sum);
//         for the activation quantizer the threshold is a Python float, so d = thr + eps stays a double.
// float64 tensors are rare on this path (the default dtype of the models MCT exports is float32), so these kernels
// are plain: 16-byte accesses, one 32-bit division per lane-vector to find the channel, parameters from the
// L1/L2-resident tables.  16 algorithmic bytes per element for the affine ops, 12 for LUT.
#include "mctq_kernels.hpp"
#include "mctq_table_builder.h"

using namespace mctq;

namespace mctq {

typedef double f64x2 __attribute__((ext_vector_type(2)));

struct Fq64 {
  const float* __restrict__ scales;    // device [channels], or NULL: scale0 / zp0 below
  const int32_t* __restrict__ zps;     // device [channels] or NULL (all zero)
  float scale0;
  int32_t zp0;
  double lo, hi;
  int wide;                            // 1: double product (per channel), 0: float32 product widened (per tensor)

  struct P { float s; double inv, z; };
  __device__ __forceinline__ P fetch(uint32_t c) const {
    P p;
    p.s = scales ? scales[c] : scale0;
    p.inv = (double)(1.0f / p.s);
    p.z = (double)(scales ? (zps ? zps[c] : 0) : zp0);
    return p;
  }
  __device__ __forceinline__ double apply(double x, const P& p) const {
    double q = __builtin_rint(x * p.inv) + p.z;               // v_rndne_f64: ties to even
    q = fmin(fmax(q, lo), hi);                                // NaN -> lo, +inf -> hi, -inf -> lo
    const double d = q - p.z;                                 // exact small integer
    return wide ? d * (double)p.s : (double)((float)d * p.s);
  }
};

// Block b owns elements [b*TILE, (b+1)*TILE); lane-vector = 2 doubles.
template <typename IdxT, bool VEC>
__global__ __launch_bounds__(kThreads) void fq64_kernel(Fq64 op, const double* __restrict__ x, double* __restrict__ y,
                                                        IdxT n, IdxT inner, uint32_t channels) {
  constexpr int U = 4;
  constexpr uint32_t V = VEC ? 2 : 1;
  constexpr uint32_t TILE = kThreads * U * V;
  const IdxT e0 = (IdxT)blockIdx.x * TILE;
  const IdxT left = n - e0;
  const uint32_t count = left < (IdxT)TILE ? (uint32_t)left : TILE;
  f64x2 v[U];
  if (VEC) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t off = (u * kThreads + threadIdx.x) * V;
      if (off + V <= count) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x + e0 + off));
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * V;
    if (off >= count) continue;
    const IdxT pos = e0 + off;
    uint32_t c = 0;
    IdxT rem = 0;
    if (channels > 1) {                                       // uniform
      const IdxT row = pos / inner;
      rem = pos - row * inner;
      c = (uint32_t)(row % channels);
    }
    if (VEC && off + V <= count) {
      f64x2 r;
      if (channels <= 1 || rem + V <= inner) {
        const Fq64::P p = op.fetch(c);
        r.x = op.apply(v[u].x, p); r.y = op.apply(v[u].y, p);
      } else {                                                // the pair straddles two rows
        r.x = op.apply(v[u].x, op.fetch(c));
        if (++c == channels) c = 0;
        r.y = op.apply(v[u].y, op.fetch(c));
      }
      __builtin_nontemporal_store(r, reinterpret_cast<f64x2*>(y + e0 + off));
    } else {
      for (uint32_t j = 0; j < V && off + j < count; ++j) {
        y[e0 + off + j] = op.apply(x[e0 + off + j], op.fetch(c));
        if (channels > 1 && ++rem == inner) { rem = 0; if (++c == channels) c = 0; }
      }
    }
  }
}

static int launch_fq64(const Fq64& op, const void* xv, void* yv, int64_t outer, int64_t channels, int64_t inner,
                       hipStream_t st) {
  const int64_t n = outer * channels * inner;
  if (n == 0) return 0;
  if (channels > 0x7fffffffLL) return fail_arg("channels exceed 2^31-1");
  const double* x = static_cast<const double*>(xv);
  double* y = static_cast<double*>(yv);
  const bool vec = (((uintptr_t)x | (uintptr_t)y) & 15u) == 0;
  const int64_t tile = kThreads * 4 * (vec ? 2 : 1);
  const int64_t blocks = (n + tile - 1) / tile;
  if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
  const bool idx32 = n <= 0xffffffffLL - tile && inner <= 0xffffffffLL;
#define MCTQ_FQ64(IDX_, VEC_)                                                                                       \
  hipLaunchKernelGGL((fq64_kernel<IDX_, VEC_>), dim3((unsigned)blocks), dim3(kThreads), 0, st, op, x, y, (IDX_)n, \
                     (IDX_)inner, (uint32_t)channels)
  if (idx32) { if (vec) MCTQ_FQ64(uint32_t, true); else MCTQ_FQ64(uint32_t, false); }
  else { if (vec) MCTQ_FQ64(uint64_t, true); else MCTQ_FQ64(uint64_t, false); }
#undef MCTQ_FQ64
  return check_launch("float64 affine launch");
}

// ---- LUT, float64 input, float32 output: literal first-minimum scan in double ----
struct Lut64 {
  const float* __restrict__ thr;       // device [channels] or NULL: d0 / t0 below
  const float* __restrict__ lut;       // device [n_lut]
  int n_lut;
  float eps;
  double d0;                           // per tensor: divisor (a double)
  float t0;                            // per tensor: final multiplier, float32(threshold)
  double mult, cmin, cmax;
  float inv_mult;
};

template <typename IdxT>
__global__ __launch_bounds__(kThreads) void lut64_kernel(Lut64 op, const double* __restrict__ x, float* __restrict__ y,
                                                         IdxT n, IdxT inner, uint32_t channels) {
  extern __shared__ __attribute__((aligned(16))) float book[];
  for (int j = threadIdx.x; j < op.n_lut; j += kThreads) book[j] = op.lut[j];
  __syncthreads();
  const IdxT stride = (IdxT)gridDim.x * kThreads;
  for (IdxT i = (IdxT)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    double d = op.d0;
    float tm = op.t0;
    if (op.thr) {
      const uint32_t c = channels > 1 ? (uint32_t)((i / inner) % channels) : 0u;
      tm = op.thr[c];
      d = (double)(tm + op.eps);                              // float32 tensor + scalar: a float32 sum
    }
    const double xv = x[i];
    const double v = (xv / d) * op.mult;
    double t = fmin(fmax(v, op.cmin), op.cmax);
    t = (v != v) ? v : t;                                     // torch.clip keeps NaN
    float best_c = book[0];
    double best_d = fabs(t - (double)best_c);
    for (int j = 1; j < op.n_lut; ++j) {
      const float c = book[j];
      const double dist = fabs(t - (double)c);
      const bool lt = dist < best_d;                          // strict: first minimum; NaN never wins
      best_d = lt ? dist : best_d;
      best_c = lt ? c : best_c;
    }
    y[i] = (best_c * op.inv_mult) * tm;
  }
}

static int launch_lut64(const Lut64& op, const void* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                        hipStream_t st) {
  const int64_t n = outer * channels * inner;
  if (n == 0) return 0;
  if (channels > 0x7fffffffLL) return fail_arg("channels exceed 2^31-1");
  int64_t blocks = (n + kThreads - 1) / kThreads;
  const int64_t cap = (int64_t)cu_count() * 32;
  if (blocks > cap) blocks = cap;
  const size_t lds = (size_t)((op.n_lut + 3) & ~3) * 4;
  if (n <= 0x7fffffffLL && inner <= 0x7fffffffLL)
    hipLaunchKernelGGL((lut64_kernel<uint32_t>), dim3((unsigned)blocks), dim3(kThreads), lds, st, op,
                       static_cast<const double*>(x), y, (uint32_t)n, (uint32_t)inner, (uint32_t)channels);
  else
    hipLaunchKernelGGL((lut64_kernel<uint64_t>), dim3((unsigned)blocks), dim3(kThreads), lds, st, op,
                       static_cast<const double*>(x), y, (uint64_t)n, (uint64_t)inner, (uint32_t)channels);
  return check_launch("float64 LUT launch");
}

// ---- LUT, float64 input, integer codebook: sorted DOUBLE threshold list (mctq_tb::build_steps64) in LDS, branchless binary
// search -- log2(P) 8-byte LDS reads per element instead of 4 double operations per codebook entry; one tile of
// 256 x 4 lane-vectors (2 doubles each) per block, loads first.  Same results as lut64_kernel (tested against it and
// against the reference's float64 fixtures).
struct Lut64Steps {
  const float* __restrict__ thr;       // device [channels] or NULL: d0 / t0 below
  const void* __restrict__ steps;      // device blob: double T[P], float Q[P], float q_nan, float P
  int P;
  float eps;
  double d0;
  float t0;
  double mult, cmin, cmax;
};

template <typename IdxT, bool VEC>
__global__ __launch_bounds__(kThreads) void lut64_steps_kernel(Lut64Steps op, const double* __restrict__ x,
                                                               float* __restrict__ y, IdxT n, IdxT inner, uint32_t channels) {
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  constexpr int U = 4;
  constexpr uint32_t V = VEC ? 2 : 1;
  constexpr uint32_t TILE = kThreads * U * V;
  const IdxT e0 = (IdxT)blockIdx.x * TILE;
  const IdxT left = n - e0;
  const uint32_t count = left < (IdxT)TILE ? (uint32_t)left : TILE;
  f64x2 v[U];
  if (VEC) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t off = (u * kThreads + threadIdx.x) * V;
      if (off + V <= count) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x + e0 + off));
    }
  }
  const int P = op.P;
  {                                                           // stage the list: 12 P + 8 bytes as dwords
    const uint32_t* src = static_cast<const uint32_t*>(op.steps);
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds64);
    for (int j = threadIdx.x; j < 3 * P + 2; j += kThreads) dst[j] = src[j];
  }
  __syncthreads();
  const double* T = lds64;
  const float* Q = reinterpret_cast<const float*>(lds64 + P);
  const float nan_q = Q[P];
  auto one = [&](double xv, double d, float tm) -> float {
    const double val = (xv / d) * op.mult;
    double t = fmin(fmax(val, op.cmin), op.cmax);
    int idx = 0;
    for (int s = P >> 1; s > 0; s >>= 1) idx += (t >= T[idx + s]) ? s : 0;
    const float q = (val != val) ? nan_q : Q[idx];            // torch.clip keeps NaN: every distance NaN, argmin = entry 0
    return q * tm;
  };
  auto params = [&](uint32_t cc, double& d, float& tm) {
    if (op.thr) { tm = op.thr[cc]; d = (double)(tm + op.eps); }        // float32 tensor + scalar: a float32 sum
    else { tm = op.t0; d = op.d0; }
  };
  // the whole tile inside one row (rows of at least a tile, tile-aligned -- Linear weights along axis 0), or one parameter
  // set for the tensor: the divisor and the multiplier are wave-uniform, no per-lane row search
  const IdxT row0 = (op.thr && channels > 1) ? e0 / inner : 0;
  const IdxT rem0 = e0 - row0 * inner;
  if (!(op.thr && channels > 1) || rem0 + count <= inner) {
    double d; float tm;
    params((op.thr && channels > 1) ? (uint32_t)(row0 % channels) : 0u, d, tm);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t off = (u * kThreads + threadIdx.x) * V;
      if (off >= count) continue;
      if (VEC && off + V <= count) {
        f32x2 r;
        r.x = one(v[u].x, d, tm); r.y = one(v[u].y, d, tm);
        *reinterpret_cast<f32x2*>(y + e0 + off) = r;
      } else {
        for (uint32_t j = 0; j < V && off + j < count; ++j) y[e0 + off + j] = one(x[e0 + off + j], d, tm);
      }
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * V;
    if (off >= count) continue;
    const IdxT pos = e0 + off;
    uint32_t c = 0;
    IdxT rem = 0;
    if (op.thr && channels > 1) {                             // uniform
      const IdxT row = pos / inner;
      rem = pos - row * inner;
      c = (uint32_t)(row % channels);
    }
    double d; float tm;
    params(c, d, tm);
    if (VEC && off + V <= count) {
      f32x2 r;
      r.x = one(v[u].x, d, tm);
      if (op.thr && channels > 1 && rem + 1 == inner) { if (++c == channels) c = 0; params(c, d, tm); }
      r.y = one(v[u].y, d, tm);
      *reinterpret_cast<f32x2*>(y + e0 + off) = r;
    } else {
      for (uint32_t j = 0; j < V && off + j < count; ++j) {
        y[e0 + off + j] = one(x[e0 + off + j], d, tm);
        if (op.thr && channels > 1 && ++rem == inner) { rem = 0; if (++c == channels) c = 0; params(c, d, tm); }
      }
    }
  }
}

static int launch_lut64_steps(const Lut64Steps& op, const void* xv, float* y, int64_t outer, int64_t channels, int64_t inner,
                              hipStream_t st) {
  const int64_t n = outer * channels * inner;
  if (n == 0) return 0;
  if (channels > 0x7fffffffLL) return fail_arg("channels exceed 2^31-1");
  if (!op.steps || op.P < 1 || op.P > 4096 || (op.P & (op.P - 1))) return fail_arg("bad float64 threshold list");
  const double* x = static_cast<const double*>(xv);
  const bool vec = (((uintptr_t)x & 15u) | ((uintptr_t)y & 7u)) == 0;
  const int64_t tile = kThreads * 4 * (vec ? 2 : 1);
  const int64_t blocks = (n + tile - 1) / tile;
  if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
  const size_t lds = ((size_t)op.P * 12u + 8u + 15u) & ~(size_t)15u;
  const bool idx32 = n <= 0xffffffffLL - tile && inner <= 0xffffffffLL;
#define MCTQ_L64S(IDX_, VEC_)                                                                                          \
  hipLaunchKernelGGL((lut64_steps_kernel<IDX_, VEC_>), dim3((unsigned)blocks), dim3(kThreads), lds, st, op, x, y, (IDX_)n, \
                     (IDX_)inner, (uint32_t)channels)
  if (idx32) { if (vec) MCTQ_L64S(uint32_t, true); else MCTQ_L64S(uint32_t, false); }
  else { if (vec) MCTQ_L64S(uint64_t, true); else MCTQ_L64S(uint64_t, false); }
#undef MCTQ_L64S
  g_note.shape = "lut64_steps_kernel"; g_note.op = "LutSteps64"; g_note.unroll = 4; g_note.nt = 1; g_note.in_bytes = 8; g_note.out_bytes = 4; ++g_note.count;
  if (g_launch_log) log_launch();
  return check_launch("float64 LUT threshold-list launch");
}

// entry points used by mctq_affine.hip / mctq_lut_scan.hip for dtype == MCTQ_DT_F64
int fq64_per_tensor(const void* x, void* y, int64_t n, float scale, int32_t zp, int32_t qmin, int32_t qmax, hipStream_t st) {
  Fq64 op;
  op.scales = nullptr; op.zps = nullptr; op.scale0 = scale; op.zp0 = zp; op.lo = qmin; op.hi = qmax; op.wide = 0;
  return launch_fq64(op, x, y, n > 0 ? 1 : 0, 1, n, st);
}
int fq64_per_channel(const void* x, void* y, int64_t outer, int64_t channels, int64_t inner, const float* scales,
                     const int32_t* zps, int32_t qmin, int32_t qmax, bool wide, hipStream_t st) {
  Fq64 op;
  op.scales = scales; op.zps = zps; op.scale0 = 1.f; op.zp0 = 0; op.lo = qmin; op.hi = qmax; op.wide = wide ? 1 : 0;
  return launch_fq64(op, x, y, outer, channels, inner, st);
}
int lut64_per_channel(const void* x, float* y, int64_t outer, int64_t channels, int64_t inner, const float* thr, float eps,
                      const float* lut, int32_t n_lut, float mult, float cmin, float cmax, hipStream_t st) {
  Lut64 op;
  op.thr = thr; op.lut = lut; op.n_lut = n_lut; op.eps = eps; op.d0 = 1.0; op.t0 = 1.f;
  op.mult = mult; op.cmin = cmin; op.cmax = cmax; op.inv_mult = 1.0f / mult;
  return launch_lut64(op, x, y, outer, channels, inner, st);
}

}  // namespace mctq

extern "C" {

int mctq_lut_per_tensor_f64(const double* x, float* y, int64_t n, double thr_div, float thr_mul, const float* lut,
                            int32_t n_lut, float mult, float clip_min, float clip_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (int rc = check_lut_args(lut, n_lut, mult)) return rc;
  Lut64 op;
  op.thr = nullptr; op.lut = lut; op.n_lut = n_lut; op.eps = 0.f; op.d0 = thr_div; op.t0 = thr_mul;
  op.mult = mult; op.cmin = clip_min; op.cmax = clip_max; op.inv_mult = 1.0f / mult;
  return launch_lut64(op, x, y, n > 0 ? 1 : 0, 1, n, (hipStream_t)stream);
}

int32_t mctq_lut_steps_f64_bytes(int32_t n_lut) {
  if (n_lut < 1 || n_lut > 4096) return fail_arg("n_lut must be in [1, 4096]");
  return (int32_t)mctq_tb::steps64_bytes(mctq_tb::steps_pow2(n_lut));
}

int mctq_lut_build_steps_f64(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                             void* steps_host, int32_t* p_out) {
  int P = 0;
  if (const char* err = mctq_tb::build_steps64(lut_host, n_lut, mult, clip_min, clip_max, steps_host, &P)) return fail_arg(err);
  if (p_out) *p_out = P;
  return 0;
}

int mctq_luts_per_tensor_f64(const double* x, float* y, int64_t n, double thr_div, float thr_mul, const void* steps,
                             int32_t P, float mult, float clip_min, float clip_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (int rc = check_pow2(mult)) return rc;
  Lut64Steps op;
  op.thr = nullptr; op.steps = steps; op.P = P; op.eps = 0.f; op.d0 = thr_div; op.t0 = thr_mul;
  op.mult = mult; op.cmin = clip_min; op.cmax = clip_max;
  return launch_lut64_steps(op, x, y, n > 0 ? 1 : 0, 1, n, (hipStream_t)stream);
}

int mctq_luts_per_channel_f64(const double* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                              const float* thresholds, float eps, const void* steps, int32_t P, float mult,
                              float clip_min, float clip_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  if (outer * channels * inner > 0 && (!x || !y || !thresholds)) return fail_arg("NULL pointer");
  if (int rc = check_pow2(mult)) return rc;
  Lut64Steps op;
  op.thr = thresholds; op.steps = steps; op.P = P; op.eps = eps; op.d0 = 1.0; op.t0 = 1.f;
  op.mult = mult; op.cmin = clip_min; op.cmax = clip_max;
  return launch_lut64_steps(op, x, y, outer, channels, inner, (hipStream_t)stream);
}

}  // extern "C"
