// mctq_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of libmctq_hip.so.
//
// Hot path of sony/mct_quantizers' PyTorch inferable quantizers, written for CDNA4:
// one fused load -> scale -> round-half-even -> clamp -> dequant -> store pass (8 algorithmic
// bytes per element) for the affine quantizers, and one fused divide -> clamp -> literal
// first-minimum codebook scan -> dequant pass for the LUT quantizers.  The work is elementwise
// and HBM-bound, so there is no MFMA here; what matters is 16-byte-per-lane coalesced traffic
// (1 KiB per wave instruction), enough loads in flight per lane, and keeping the per-channel
// parameters out of the vector memory pipe (SGPR broadcast or an LDS window).
//
// Three launch shapes per op:
//   flat    per-tensor parameters, passed in kernel arguments (SGPRs).
//   rows    per-channel, inner >= 1024 and inner % 4 == 0: one block = one tile of one
//           (outer, channel) row; the channel's parameters are fetched once per block with a
//           wave-uniform index (scalar loads -> SGPR broadcast).
//   window  per-channel, any inner (channel-last, conv kernels, ragged): one block = one
//           contiguous tile; the parameters of the rows that tile touches are staged in LDS
//           once per block and looked up per element without any per-element division.
//
// Arithmetic contract: include/mctq_hip.h.  Compile with -ffp-contract=off, no fast-math.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "mctq_hip.h"

namespace mctq {

constexpr int kThreads = 256;
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ f4 ld4(const f4* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st4(f4* p, f4 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ------------------------------------------------------------------------------------------
// Ops.  An Op describes: per-channel Param (kWords floats when staged in LDS), fetch(c) that
// builds it from the device tables, an optional Book (codebook) set up once per block, and
// apply(x, param, book).
// ------------------------------------------------------------------------------------------

struct NoBook {};

struct AffineOp {
  const float* __restrict__ scales;    // [C] (per-channel launches only)
  const int32_t* __restrict__ zps;     // [C]
  float lo, hi;                        // clamp domain as floats (exact: |q| < 2^24)

  struct Param { float s, inv, zf; };
  typedef NoBook Book;
  static constexpr int kWords = 3;
  static constexpr bool kHeavy = false;       // a few VALU ops per element: pure streaming

  __device__ __forceinline__ Param fetch(uint32_t c) const {
    Param p;
    p.s = scales[c];
    p.inv = 1.0f / p.s;                // correctly rounded IEEE division, as ATen's 1.0f / scale
    p.zf = (float)zps[c];
    return p;
  }
  __device__ __forceinline__ static void put(float* lds, uint32_t i, uint32_t stride, const Param& p) {
    lds[i] = p.s; lds[stride + i] = p.inv; lds[2 * stride + i] = p.zf;
  }
  __device__ __forceinline__ static Param get(const float* lds, uint32_t i, uint32_t stride) {
    Param p; p.s = lds[i]; p.inv = lds[stride + i]; p.zf = lds[2 * stride + i]; return p;
  }
  __device__ __forceinline__ uint32_t book_words() const { return 0; }
  __device__ __forceinline__ Book setup(float*) const { return Book(); }
  __device__ __forceinline__ static bool can_fast(const Param&) { return true; }

  template <bool FAST = true>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book&) const {
    float q = __builtin_rintf(x * p.inv) + p.zf;      // v_rndne_f32: ties to even
    q = fminf(fmaxf(q, lo), hi);                      // NaN -> lo, +inf -> hi, -inf -> lo
    return (q - p.zf) * p.s;
  }
};

// Register-resident codebook of LP entries (padded with +inf, which never wins a strict '<').
template <int LP>
struct RegBook { float c[LP]; };

struct LdsBook { const float* c; int n; };

// Shared by the two LUT ops: per-channel parameters and the shared-divisor division.
struct LutCommon {
  const float* __restrict__ thr;       // [C] thresholds (per-channel launches only)
  float eps;
  float mult, inv_mult, cmin, cmax;

  // divisor fl32(thr + eps), multiplier thr, and r = RN(1/d) when the fast exact division below
  // is valid for this divisor (r == 0 selects the plain IEEE division).
  struct Param { float d, t, r; };
  static constexpr int kWords = 3;
  static constexpr bool kHeavy = true;         // blocks loop over tiles (set-up paid once per block)

  __host__ __device__ __forceinline__ static Param make(float d, float t) {
    Param p; p.d = d; p.t = t;
    const float a = fabsf(d);
    p.r = (a > 0x1p-60f && a < 0x1p60f) ? 1.0f / d : 0.0f;
    return p;
  }
  __device__ __forceinline__ Param fetch(uint32_t c) const {
    const float t = thr[c];
    return make(t + eps, t);
  }
  __device__ __forceinline__ static void put(float* lds, uint32_t i, uint32_t stride, const Param& p) {
    lds[i] = p.d; lds[stride + i] = p.t; lds[2 * stride + i] = p.r;
  }
  __device__ __forceinline__ static Param get(const float* lds, uint32_t i, uint32_t stride) {
    Param p; p.d = lds[i]; p.t = lds[stride + i]; p.r = lds[2 * stride + i]; return p;
  }

  // x / d, correctly rounded, for a divisor shared by many elements.  FAST: with r = RN(1/d),
  // q0 = x*r followed by two residual corrections with exact FMA residuals -- the same recurrence
  // the compiler's IEEE expansion runs after its v_rcp/Newton steps, minus the per-element
  // reciprocal and scaling (7 VALU ops instead of ~11, and no v_div_* wait states).  Outside
  // |q| < 2^60 (and for inf/NaN) q0 is returned: there the result is clamped anyway, and below
  // |q| ~ 2^-31 the codebook decision does not depend on the last bits (every |t - c| with c != 0
  // rounds to |c|).  Verified exhaustively against '/' on the GPU
  // (tests/test_gpu_parity.py::test_fast_division_is_exact).  Kernels pick FAST per block (rows,
  // flat: the divisor is wave-uniform) when p.r != 0; FAST = false is the plain IEEE division.
  template <bool FAST>
  __device__ __forceinline__ static float divide(float x, const Param& p) {
    if constexpr (!FAST) {
      return x / p.d;
    } else {
      const float q0 = x * p.r;
      const float e0 = __builtin_fmaf(-q0, p.d, x);
      const float q1 = __builtin_fmaf(e0, p.r, q0);
      const float e1 = __builtin_fmaf(-q1, p.d, x);
      const float q2 = __builtin_fmaf(e1, p.r, q1);
      return (fabsf(q0) < 0x1p60f) ? q2 : q0;
    }
  }
  __device__ __forceinline__ static bool can_fast(const Param& p) { return p.r != 0.0f; }
};

// Literal codebook scan (any codebook, any bit width): the reference's first-minimum argmin.
template <int LP>   // LP > 0: codebook broadcast into LP scalar registers; LP == 0: codebook in LDS
struct LutOp : LutCommon {
  const float* __restrict__ lut;       // [n_lut] device codebook, caller's order
  int n_lut;

  typedef typename std::conditional<(LP > 0), RegBook<(LP > 0 ? LP : 1)>, LdsBook>::type Book;

  __device__ __forceinline__ uint32_t book_words() const { return LP > 0 ? 0u : (uint32_t)((n_lut + 3) & ~3); }

  // Called by every thread of the block before any apply().
  __device__ __forceinline__ Book setup(float* lds) const {
    if constexpr (LP > 0) {
      // One coalesced load per wave: lane j holds lut[j]; every entry is then broadcast to a
      // scalar register with v_readlane (wave-level shuffle), so the scan below reads SGPRs.
      const int lane = threadIdx.x & 63;
      float v = INFINITY;
      if (lane < n_lut) v = lut[lane];
      Book b;
#pragma unroll
      for (int j = 0; j < LP; ++j)
        b.c[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
      return b;
    } else {
      for (int j = threadIdx.x; j < n_lut; j += kThreads) lds[j] = lut[j];
      __syncthreads();
      Book b; b.c = lds; b.n = n_lut;
      return b;
    }
  }

  template <bool FAST = false>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book& b) const {
    float v = divide<FAST>(x, p) * mult;               // == (x / d) * mult (quantizer_utils.py:169)
    float t = fminf(fmaxf(v, cmin), cmax);
    t = (v != v) ? v : t;                              // torch.clip keeps NaN
    float best_c, best_d;
    if constexpr (LP > 0) {
      best_c = b.c[0];
      best_d = fabsf(t - best_c);
#pragma unroll
      for (int j = 1; j < LP; ++j) {
        const float c = b.c[j];
        const float d = fabsf(t - c);
        const bool lt = d < best_d;                    // strict: first minimum wins; NaN never wins
        best_d = lt ? d : best_d;
        best_c = lt ? c : best_c;
      }
    } else {
      best_c = b.c[0];
      best_d = fabsf(t - best_c);
      for (int j = 1; j < b.n; ++j) {
        const float c = b.c[j];                        // same address in every lane: LDS broadcast
        const float d = fabsf(t - c);
        const bool lt = d < best_d;
        best_d = lt ? d : best_d;
        best_c = lt ? c : best_c;
      }
    }
    return (best_c * inv_mult) * p.t;                  // mult is a power of two: * (1/mult) == / mult
  }
};

// Decision-table codebook quantizer.  For integer codebooks every decision boundary of the literal
// scan lies within a few ulps of a half-integer point P_k = clip_min + k/2 of the scaled value t
// (midpoints of integers), and around each P_k the literal result is a single monotone step
// (fl(t - a) is non-decreasing and fl(b - t) non-increasing in t).  mctq_lut_build_table() evaluates
// the literal scan on the host and records, per point, the exact float32 threshold T_k of that
// step and the dequantized centres c/mult below/above it (exact in fp16: |c| <= 2^10, mult = 2^j).
// The kernel then needs one 8-byte LDS read, one compare and one select per element, whatever the
// codebook size: entry k = {T_k, half2(q_below, q_above)}, k = trunc(2*(t - clip_min) + 0.5);
// entry `entries` holds {q for NaN inputs (float32), K}.
typedef float f2 __attribute__((ext_vector_type(2)));
struct LutTableBook { const f2* tab; float nan_q; };

struct LutTableOp : LutCommon {
  const float* __restrict__ table;     // device, (entries + 1) x 2 words
  int entries;
  float koff;                          // 0.5 - 2*clip_min

  typedef LutTableBook Book;
  __device__ __forceinline__ uint32_t book_words() const { return ((uint32_t)(entries + 1) * 2u + 3u) & ~3u; }

  __device__ __forceinline__ Book setup(float* lds) const {
    const f2* src = reinterpret_cast<const f2*>(table);
    f2* dst = reinterpret_cast<f2*>(lds);
    for (int j = threadIdx.x; j <= entries; j += kThreads) dst[j] = src[j];
    __syncthreads();
    Book b; b.tab = dst; b.nan_q = dst[entries].x;
    return b;
  }

  // stage 1: scaled, clamped value and its table index
  template <bool FAST>
  __device__ __forceinline__ void locate(float x, const Param& p, float& v, float& t, int& k) const {
    v = divide<FAST>(x, p) * mult;
    t = fminf(fmaxf(v, cmin), cmax);                  // NaN -> cmin here, overridden in decide()
    k = (int)__builtin_fmaf(t, 2.0f, koff);           // nearest half-integer point (any tie is fine)
  }
  // stage 3: pick the side of the step, dequantize
  __device__ __forceinline__ float decide(float v, float t, f2 e, const Param& p, const Book& b) const {
    const uint32_t pair = __float_as_uint(e.y);
    const uint32_t h = (t >= e.x) ? (pair >> 16) : (pair & 0xffffu);
    float q = __half2float(__ushort_as_half((unsigned short)h));
    q = (v != v) ? b.nan_q : q;                       // all-NaN distances: argmin is index 0
    return q * p.t;
  }

  template <bool FAST = false>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book& b) const {
    float v, t; int k;
    locate<FAST>(x, p, v, t, k);
    return decide(v, t, b.tab[k], p, b);
  }

  // A whole tile: all indices first, then all LDS reads back to back, then all selects, so one
  // s_waitcnt covers U*4 lookups instead of one per element.
  template <bool FAST, int U>
  __device__ __forceinline__ void tile(const f4 (&w)[U], f4 (&r)[U], const Param& p, const Book& b) const {
    float v[U * 4], t[U * 4];
    int k[U * 4];
    f2 e[U * 4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      locate<FAST>(w[u].x, p, v[4 * u + 0], t[4 * u + 0], k[4 * u + 0]);
      locate<FAST>(w[u].y, p, v[4 * u + 1], t[4 * u + 1], k[4 * u + 1]);
      locate<FAST>(w[u].z, p, v[4 * u + 2], t[4 * u + 2], k[4 * u + 2]);
      locate<FAST>(w[u].w, p, v[4 * u + 3], t[4 * u + 3], k[4 * u + 3]);
    }
#pragma unroll
    for (int i = 0; i < U * 4; ++i) e[i] = b.tab[k[i]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      r[u].x = decide(v[4 * u + 0], t[4 * u + 0], e[4 * u + 0], p, b);
      r[u].y = decide(v[4 * u + 1], t[4 * u + 1], e[4 * u + 1], p, b);
      r[u].z = decide(v[4 * u + 2], t[4 * u + 2], e[4 * u + 2], p, b);
      r[u].w = decide(v[4 * u + 3], t[4 * u + 3], e[4 * u + 3], p, b);
    }
  }
};

template <bool FAST, class Op>
__device__ __forceinline__ f4 apply4(const Op& op, f4 v, const typename Op::Param& p, const typename Op::Book& b) {
  f4 r;
  r.x = op.template apply<FAST>(v.x, p, b);
  r.y = op.template apply<FAST>(v.y, p, b);
  r.z = op.template apply<FAST>(v.z, p, b);
  r.w = op.template apply<FAST>(v.w, p, b);
  return r;
}

template <class Op, class = void>
struct HasTile : std::false_type {};
template <class Op>
struct HasTile<Op, std::void_t<decltype(&Op::template tile<true, 1>)>> : std::true_type {};

template <bool FAST, class Op, int U>
__device__ __forceinline__ void apply_tile(const Op& op, const f4 (&w)[U], f4 (&r)[U], const typename Op::Param& p,
                                           const typename Op::Book& b) {
  if constexpr (HasTile<Op>::value) {
    op.template tile<FAST, U>(w, r, p, b);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = apply4<FAST>(op, w[u], p, b);
  }
}

// ------------------------------------------------------------------------------------------
// flat: per-tensor parameters.  Block b owns float4s [b*256*U, (b+1)*256*U); lane accesses are
// 16 B, consecutive lanes consecutive addresses, U independent loads in flight per lane.
// ------------------------------------------------------------------------------------------
template <class Op, int U, bool NT>
__global__ __launch_bounds__(kThreads) void flat_kernel(Op op, typename Op::Param p,
                                                        const float* __restrict__ xs, float* __restrict__ ys,
                                                        int64_t n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const typename Op::Book book = op.setup(smem);
  const int64_t n4 = n >> 2;
  const f4* __restrict__ x = reinterpret_cast<const f4*>(xs);
  f4* __restrict__ y = reinterpret_cast<f4*>(ys);
  const int64_t base = (int64_t)blockIdx.x * (kThreads * U) + threadIdx.x;
  f4 v[U];
  if (((int64_t)blockIdx.x + 1) * (kThreads * U) <= n4) {   // full tile (wave-uniform test): no per-lane guards
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4<NT>(x + base + u * kThreads);
#pragma unroll
    for (int u = 0; u < U; ++u) st4<NT>(y + base + u * kThreads, apply4<false>(op, v[u], p, book));
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kThreads;
      if (i < n4) v[u] = ld4<NT>(x + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kThreads;
      if (i < n4) st4<NT>(y + i, apply4<false>(op, v[u], p, book));
    }
  }
  if (blockIdx.x == 0) {                                   // n % 4 trailing elements
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (i < n) ys[i] = op.template apply<false>(xs[i], p, book);
  }
}

// flat, one element per lane: used when x or y is not 16-byte aligned.
template <class Op>
__global__ __launch_bounds__(kThreads) void flat_scalar_kernel(Op op, typename Op::Param p,
                                                               const float* __restrict__ x, float* __restrict__ y,
                                                               int64_t n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const typename Op::Book book = op.setup(smem);
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
    y[i] = op.template apply<false>(x[i], p, book);
}

// ------------------------------------------------------------------------------------------
// rows: tensor viewed as [rows = outer*C][inner4] float4.  blockIdx -> (row, tile); the row's
// channel index is wave-uniform, so fetch() compiles to scalar loads and the parameters sit in
// SGPRs for the whole block.
// ------------------------------------------------------------------------------------------
template <class Op, int U, bool NT>
__global__ __launch_bounds__(kThreads) void rows_kernel(Op op, const float* __restrict__ xs, float* __restrict__ ys,
                                                        uint32_t tiles_per_row, uint32_t inner4, uint32_t channels) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t row = blockIdx.x, tile = 0;
  if (tiles_per_row != 1) {                                // uniform branch: skip the division for 1 tile/row
    row = blockIdx.x / tiles_per_row;
    tile = blockIdx.x - row * tiles_per_row;
  }
  const int64_t rbase = (int64_t)row * inner4;
  const f4* __restrict__ x = reinterpret_cast<const f4*>(xs) + rbase;
  f4* __restrict__ y = reinterpret_cast<f4*>(ys) + rbase;
  const uint32_t col = tile * (kThreads * U) + threadIdx.x;
  const bool full = (tile + 1) * (kThreads * U) <= inner4;     // wave-uniform
  f4 v[U];
  // Issue the data loads FIRST; the parameter fetch (two dependent scalar loads + an IEEE divide)
  // and the codebook set-up then run in the shadow of the HBM latency.
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4<NT>(x + col + u * kThreads);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = col + u * kThreads;
      if (i < inner4) v[u] = ld4<NT>(x + i);
    }
  }
  __builtin_amdgcn_sched_barrier(0);                       // keep the loads above the fetch in the schedule
  const typename Op::Book book = op.setup(smem);
  uint32_t c = row;
  if (c >= channels) c = row % channels;                   // uniform; outer == 1 needs no modulo
  const typename Op::Param p = op.fetch(c);
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) st4<NT>(y + col + u * kThreads, apply4<false>(op, v[u], p, book));
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = col + u * kThreads;
      if (i < inner4) st4<NT>(y + i, apply4<false>(op, v[u], p, book));
    }
  }
}

// Loop bodies for heavy ops.  Straight-line code over the U*4 elements of a full tile (uniform
// test), so the LDS table reads of a tile are issued together instead of one wait per element.
template <bool FAST, class Op, int U, bool NT>
__device__ __forceinline__ void loop_tiles(const Op& op, const typename Op::Param& p, const typename Op::Book& book,
                                           const f4* __restrict__ x, f4* __restrict__ y, f4 (&v)[U],
                                           int64_t t0, int64_t t1, int64_t tstep, int64_t n4) {
  constexpr int64_t TILE = (int64_t)kThreads * U;
  for (int64_t t = t0; t < t1; t += tstep) {
    f4 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = v[u];
    const int64_t tn = t + tstep;
    if (tn < t1) {                                           // prefetch the next tile (uniform branches)
      if ((tn + 1) * TILE <= n4) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld4<NT>(x + tn * TILE + u * kThreads + threadIdx.x);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t i = tn * TILE + u * kThreads + threadIdx.x;
          if (i < n4) v[u] = ld4<NT>(x + i);
        }
      }
    }
    if ((t + 1) * TILE <= n4) {
      f4 r[U];
      apply_tile<FAST, Op, U>(op, w, r, p, book);
#pragma unroll
      for (int u = 0; u < U; ++u) st4<NT>(y + t * TILE + u * kThreads + threadIdx.x, r[u]);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = t * TILE + u * kThreads + threadIdx.x;
        if (i < n4) st4<NT>(y + i, apply4<FAST>(op, w[u], p, book));
      }
    }
  }
}

// rows, heavy ops: persistent blocks.  The work is cut into tiles of 256*U float4 that never cross
// a row; block b takes tiles b, b+grid, b+2*grid, ... so every CU finishes at the same time (a
// one-block-per-row grid leaves the last round of blocks mostly empty), prefetches the next tile's
// loads before it computes the current one, and pays the table / codebook staging once.
template <class Op, int U, bool NT>
__device__ __forceinline__ void load_row_tile(f4 (&v)[U], const float* __restrict__ xs, uint32_t item,
                                              uint32_t tiles_per_row, uint32_t inner4) {
  const uint32_t row = item / tiles_per_row;
  const uint32_t tile = item - row * tiles_per_row;
  const f4* __restrict__ x = reinterpret_cast<const f4*>(xs) + (int64_t)row * inner4;
  const uint32_t col = tile * (kThreads * U) + threadIdx.x;
  if ((tile + 1) * (kThreads * U) <= inner4) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4<NT>(x + col + u * kThreads);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (col + u * kThreads < inner4) v[u] = ld4<NT>(x + col + u * kThreads);
  }
}

template <bool FAST, class Op, int U, bool NT>
__device__ __forceinline__ void finish_row_tile(const Op& op, const typename Op::Param& p, const typename Op::Book& book,
                                                const f4 (&w)[U], float* __restrict__ ys, uint32_t row, uint32_t tile,
                                                uint32_t inner4) {
  f4* __restrict__ y = reinterpret_cast<f4*>(ys) + (int64_t)row * inner4;
  const uint32_t col = tile * (kThreads * U) + threadIdx.x;
  if ((tile + 1) * (kThreads * U) <= inner4) {
    f4 r[U];
    apply_tile<FAST, Op, U>(op, w, r, p, book);
#pragma unroll
    for (int u = 0; u < U; ++u) st4<NT>(y + col + u * kThreads, r[u]);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (col + u * kThreads < inner4) st4<NT>(y + col + u * kThreads, apply4<FAST>(op, w[u], p, book));
  }
}

template <class Op, int U, bool NT>
__global__ __launch_bounds__(kThreads) void rows_persist_kernel(Op op, const float* __restrict__ xs, float* __restrict__ ys,
                                                                uint32_t tiles_per_row, uint32_t total_tiles,
                                                                uint32_t inner4, uint32_t channels) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t item = blockIdx.x;                                // grid <= total_tiles
  f4 v[U];
  load_row_tile<Op, U, NT>(v, xs, item, tiles_per_row, inner4);
  const typename Op::Book book = op.setup(smem);
  uint32_t cur_row = 0xffffffffu;
  typename Op::Param p;
  bool fast = false;
  for (; item < total_tiles; item += gridDim.x) {
    const uint32_t row = item / tiles_per_row;
    const uint32_t tile = item - row * tiles_per_row;
    if (row != cur_row) {                                    // wave-uniform
      uint32_t c = row;
      if (c >= channels) c = row % channels;
      p = op.fetch(c);
      fast = __builtin_amdgcn_readfirstlane((int)Op::can_fast(p)) != 0;
      cur_row = row;
    }
    f4 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = v[u];
    const uint32_t next = item + gridDim.x;
    if (next < total_tiles) load_row_tile<Op, U, NT>(v, xs, next, tiles_per_row, inner4);
    if (fast) finish_row_tile<true, Op, U, NT>(op, p, book, w, ys, row, tile, inner4);
    else finish_row_tile<false, Op, U, NT>(op, p, book, w, ys, row, tile, inner4);
  }
}

// flat, heavy ops: persistent blocks stride over the tiles with the same one-tile prefetch.
template <class Op, int U, bool NT>
__global__ __launch_bounds__(kThreads) void flat_loop_kernel(Op op, typename Op::Param p,
                                                             const float* __restrict__ xs, float* __restrict__ ys,
                                                             int64_t n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t n4 = n >> 2;
  const int64_t tiles = (n4 + kThreads * U - 1) / (kThreads * U);
  const f4* __restrict__ x = reinterpret_cast<const f4*>(xs);
  f4* __restrict__ y = reinterpret_cast<f4*>(ys);
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = (int64_t)blockIdx.x * (kThreads * U) + u * kThreads + threadIdx.x;
    if (i < n4) v[u] = ld4<NT>(x + i);
  }
  const typename Op::Book book = op.setup(smem);
  if (Op::can_fast(p))                                      // kernel argument: uniform
    loop_tiles<true, Op, U, NT>(op, p, book, x, y, v, blockIdx.x, tiles, gridDim.x, n4);
  else
    loop_tiles<false, Op, U, NT>(op, p, book, x, y, v, blockIdx.x, tiles, gridDim.x, n4);
  if (blockIdx.x == 0) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (i < n) ys[i] = op.template apply<false>(xs[i], p, book);
  }
}

// ------------------------------------------------------------------------------------------
// window: block b owns elements [b*TILE, (b+1)*TILE), TILE = 256*U*V (V = 4: float4 accesses,
// V = 1: one float per lane for unaligned tensors).  The tile touches rows
// row0 .. row0+nrows-1 of the [outer*C][inner] view; their parameters are staged in LDS
// (structure-of-arrays, so lanes that read different rows hit different banks) and a lane
// finds its row with ONE 32-bit division per access, then walks row boundaries incrementally.
// If the whole table is smaller than the tile's row span (channel-last layouts: inner == 1,
// C small) the whole table is staged instead and indexed modulo C.
// ------------------------------------------------------------------------------------------
template <class Op, int U, int V, bool NT, typename IdxT>
__global__ __launch_bounds__(kThreads) void window_kernel(Op op, const float* __restrict__ xs, float* __restrict__ ys,
                                                          IdxT n, uint32_t inner, uint32_t channels,
                                                          uint32_t stride /* LDS entries per param word */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr uint32_t TILE = kThreads * U * V;
  const typename Op::Book book = op.setup(smem);
  float* tab = smem + op.book_words();

  const IdxT e0 = (IdxT)blockIdx.x * TILE;
  const IdxT row0 = e0 / inner;                            // uniform, once per block
  const uint32_t rem0 = (uint32_t)(e0 - row0 * inner);
  const IdxT left = n - e0;
  const uint32_t count = left < (IdxT)TILE ? (uint32_t)left : TILE;
  const uint32_t nrows = (rem0 + count - 1) / inner + 1;
  const bool whole = channels <= nrows;
  const uint32_t c0 = (uint32_t)(row0 % channels);
  if (whole) {
    for (uint32_t i = threadIdx.x; i < channels; i += kThreads) Op::put(tab, i, stride, op.fetch(i));
  } else {
    for (uint32_t i = threadIdx.x; i < nrows; i += kThreads) {
      uint32_t c = c0 + i;                                 // nrows < channels here: at most one wrap
      if (c >= channels) c -= channels;
      Op::put(tab, i, stride, op.fetch(c));
    }
  }
  __syncthreads();

#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * V;
    if (off >= count) continue;
    const uint32_t pos = rem0 + off;
    uint32_t lrow = pos / inner;
    uint32_t lrem = pos - lrow * inner;
    uint32_t li = lrow;
    if (whole) li = (c0 + lrow) % channels;
    if (V == 4 && off + 4 <= count) {
      const f4 v = ld4<NT>(reinterpret_cast<const f4*>(xs + e0 + off));
      float in[4] = {v.x, v.y, v.z, v.w};
      float out[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        out[j] = op.template apply<false>(in[j], Op::get(tab, li, stride), book);
        if (++lrem == inner) {
          lrem = 0;
          ++li;
          if (whole && li == channels) li = 0;
        }
      }
      f4 r; r.x = out[0]; r.y = out[1]; r.z = out[2]; r.w = out[3];
      st4<NT>(reinterpret_cast<f4*>(ys + e0 + off), r);
    } else {
      for (uint32_t j = 0; j < (uint32_t)V && off + j < count; ++j) {
        ys[e0 + off + j] = op.template apply<false>(xs[e0 + off + j], Op::get(tab, li, stride), book);
        if (++lrem == inner) {
          lrem = 0;
          ++li;
          if (whole && li == channels) li = 0;
        }
      }
    }
  }
}

// Self-test: the fast division of LutOp against the compiler's IEEE '/' for EVERY float32 numerator.
__global__ __launch_bounds__(kThreads) void selftest_division_kernel(const float* __restrict__ divisors, int n_div,
                                                                     unsigned long long* __restrict__ mismatches) {
  const uint64_t stride = (uint64_t)gridDim.x * kThreads;
  for (int j = 0; j < n_div; ++j) {
    const LutCommon::Param p = LutCommon::make(divisors[j], divisors[j]);
    unsigned int bad = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * kThreads + threadIdx.x; b < (1ull << 32); b += stride) {
      const float x = __uint_as_float((uint32_t)b);
      const float slow = x / p.d;
      const float fast = LutCommon::can_fast(p) ? LutCommon::divide<true>(x, p) : LutCommon::divide<false>(x, p);
      const float a = fabsf(slow);
      bool ok;
      if (a >= 0x1p-40f && a < 0x1p59f) ok = __float_as_uint(slow) == __float_as_uint(fast);    // exact domain
      else if (slow != slow) ok = fast != fast;                                                   // NaN stays NaN
      else if (a >= 0x1p59f) ok = fabsf(fast) >= 0x1p58f && (slow < 0) == (fast < 0);             // clamps alike
      else ok = fabsf(fast) < 0x1p-39f;                                                           // stays tiny
      bad += ok ? 0u : 1u;
    }
    if (bad) atomicAdd(&mismatches[j], (unsigned long long)bad);
  }
}

// ------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
static int g_nt = 1;   // non-temporal loads/stores: +7% on the cold 4096x4096 stream (profiles/)
static int g_unroll = 4;
static int g_heavy_unroll = 0;   // 0 = automatic

static int fail_arg(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return MCTQ_E_ARG;
}
static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -(int)e;
  }
  return 0;
}
static int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else cus = 256;
  }
  return cus;
}
static bool aligned16(const void* a, const void* b) {
  return (((uintptr_t)a | (uintptr_t)b) & 15u) == 0;
}

#define MCTQ_DISPATCH_U_NT(U_, NT_, CALL)                      \
  do {                                                         \
    if (NT_) {                                                 \
      switch (U_) {                                            \
        case 1: { constexpr int U = 1; constexpr bool NT = true; CALL; } break;  \
        case 2: { constexpr int U = 2; constexpr bool NT = true; CALL; } break;  \
        case 8: { constexpr int U = 8; constexpr bool NT = true; CALL; } break;  \
        default: { constexpr int U = 4; constexpr bool NT = true; CALL; } break; \
      }                                                        \
    } else {                                                   \
      switch (U_) {                                            \
        case 1: { constexpr int U = 1; constexpr bool NT = false; CALL; } break;  \
        case 2: { constexpr int U = 2; constexpr bool NT = false; CALL; } break;  \
        case 8: { constexpr int U = 8; constexpr bool NT = false; CALL; } break;  \
        default: { constexpr int U = 4; constexpr bool NT = false; CALL; } break; \
      }                                                        \
    }                                                          \
  } while (0)

template <class Op>
static int launch_flat(const Op& op, const typename Op::Param& p, const float* x, float* y, int64_t n,
                       size_t book_bytes, hipStream_t st) {
  if (n == 0) return 0;
  if (aligned16(x, y)) {
    const int64_t n4 = n >> 2;
    if constexpr (Op::kHeavy) {
      constexpr int U = 2;
      int64_t blocks = (n4 + kThreads * U - 1) / (kThreads * U);
      const int64_t cap = (int64_t)cu_count() * 16;
      if (blocks > cap) blocks = cap;
      if (blocks == 0) blocks = 1;
      if (g_nt) hipLaunchKernelGGL((flat_loop_kernel<Op, U, true>), dim3((unsigned)blocks), dim3(kThreads), book_bytes, st, op, p, x, y, n);
      else      hipLaunchKernelGGL((flat_loop_kernel<Op, U, false>), dim3((unsigned)blocks), dim3(kThreads), book_bytes, st, op, p, x, y, n);
      return check_launch("flat loop launch");
    }
    MCTQ_DISPATCH_U_NT(g_unroll, g_nt, {
      int64_t blocks = (n4 + kThreads * U - 1) / (kThreads * U);
      if (blocks == 0) blocks = 1;
      if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
      hipLaunchKernelGGL((flat_kernel<Op, U, NT>), dim3((unsigned)blocks), dim3(kThreads), book_bytes, st,
                         op, p, x, y, n);
    });
  } else {
    int64_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL((flat_scalar_kernel<Op>), dim3((unsigned)blocks), dim3(kThreads), book_bytes, st, op, p, x, y, n);
  }
  return check_launch("flat launch");
}

template <class Op>
static int launch_channels(const Op& op, const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                           size_t book_bytes, hipStream_t st) {
  const int64_t n = outer * channels * inner;
  if (n == 0) return 0;
  const int64_t rows = outer * channels;
  const bool vec_ok = aligned16(x, y);

  // rows shape: long, float4-divisible rows.
  if (vec_ok && (inner & 3) == 0 && inner >= 1024 && channels <= 0xffffffffLL) {
    const int64_t inner4 = inner >> 2;
    if constexpr (Op::kHeavy) {
      // float4s per lane per iteration: the widest of {4, 2, 1} whose idle lanes in the last tile of
      // a row stay under 1/8 (more bytes in flight per lane), unless a tuning override is set; then
      // as many iterations per block as still leave >= 8 blocks per CU.
      int u_sel = 1;
      for (int u = 4; u >= 1; u >>= 1) {
        const int64_t per_u = (int64_t)kThreads * u;
        const int64_t cap = ((inner4 + per_u - 1) / per_u) * per_u;
        if ((cap - inner4) * 8 <= cap) { u_sel = u; break; }
      }
      if (g_heavy_unroll) u_sel = g_heavy_unroll;
      const int64_t per = (int64_t)kThreads * u_sel;
      const int64_t tiles = (inner4 + per - 1) / per;
      const int64_t total = rows * tiles;
      if (total <= 0x7fffffffLL && inner4 <= 0x7fffffffLL) {
#define MCTQ_ROWS_PERSIST(U_, NT_)                                                                           \
        do {                                                                                                 \
          static int per_cu = 0;                                                                             \
          if (per_cu == 0) {                                                                                 \
            int nb = 0;                                                                                      \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rows_persist_kernel<Op, U_, NT_>, kThreads, \
                                                             book_bytes) != hipSuccess || nb < 1) nb = 4;    \
            per_cu = nb > 8 ? 8 : nb;                                                                        \
          }                                                                                                  \
          int64_t grid = (int64_t)cu_count() * per_cu;                                                       \
          if (grid > total) grid = total;                                                                    \
          hipLaunchKernelGGL((rows_persist_kernel<Op, U_, NT_>), dim3((unsigned)grid), dim3(kThreads),       \
                             book_bytes, st, op, x, y, (uint32_t)tiles, (uint32_t)total, (uint32_t)inner4,   \
                             (uint32_t)channels);                                                            \
        } while (0)
        if (u_sel == 1) { if (g_nt) MCTQ_ROWS_PERSIST(1, true); else MCTQ_ROWS_PERSIST(1, false); }
        else if (u_sel == 2) { if (g_nt) MCTQ_ROWS_PERSIST(2, true); else MCTQ_ROWS_PERSIST(2, false); }
        else { if (g_nt) MCTQ_ROWS_PERSIST(4, true); else MCTQ_ROWS_PERSIST(4, false); }
#undef MCTQ_ROWS_PERSIST
        return check_launch("rows persistent launch");
      }
    } else {
    // Largest U <= tuned unroll that wastes the fewest lanes in the last tile of a row.
    int best_u = 1;
    int64_t best_waste = -1;
    for (int u = 1; u <= g_unroll; u <<= 1) {
      const int64_t per = (int64_t)kThreads * u;
      const int64_t tiles = (inner4 + per - 1) / per;
      const int64_t waste = tiles * per - inner4;
      if (best_waste < 0 || waste <= best_waste) { best_waste = waste; best_u = u; }
    }
    const int64_t per = (int64_t)kThreads * best_u;
    const int64_t tiles = (inner4 + per - 1) / per;
    if (rows * tiles <= 0x7fffffffLL && inner4 <= 0x7fffffffLL && rows <= 0xffffffffLL) {
      MCTQ_DISPATCH_U_NT(best_u, g_nt, {
        hipLaunchKernelGGL((rows_kernel<Op, U, NT>), dim3((unsigned)(rows * tiles)), dim3(kThreads), book_bytes, st,
                           op, x, y, (uint32_t)tiles, (uint32_t)inner4, (uint32_t)channels);
      });
      return check_launch("rows launch");
    }
    }
  }

  // window shape.
  if (inner > 0x7fffffffLL || channels > 0x7fffffffLL) return fail_arg("inner/channels exceed 2^31-1");
  constexpr int WU = 4;
  const int V = vec_ok ? 4 : 1;
  const uint32_t tile = kThreads * WU * V;
  uint64_t max_rows = (uint64_t)(tile - 1 + (inner - 1)) / (uint64_t)inner + 1;   // rows a tile can touch
  uint64_t entries = (uint64_t)channels <= max_rows ? (uint64_t)channels : max_rows;
  // LDS stride: odd number of words keeps the three/two parameter planes on different banks.
  const uint32_t stride = (uint32_t)entries | 1u;
  const size_t lds = book_bytes + (size_t)stride * Op::kWords * sizeof(float);
  if (lds > 64 * 1024) return fail_arg("parameter window exceeds 64 KiB of LDS");
  const int64_t blocks = (n + tile - 1) / tile;
  if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
  const bool idx32 = n <= (int64_t)0xffffffffLL - (int64_t)tile;
#define MCTQ_WINDOW(V_, NT_, IDX_)                                                                       \
  hipLaunchKernelGGL((window_kernel<Op, WU, V_, NT_, IDX_>), dim3((unsigned)blocks), dim3(kThreads), lds, st, \
                     op, x, y, (IDX_)n, (uint32_t)inner, (uint32_t)channels, stride)
  if (V == 4) {
    if (g_nt) { if (idx32) MCTQ_WINDOW(4, true, uint32_t); else MCTQ_WINDOW(4, true, uint64_t); }
    else      { if (idx32) MCTQ_WINDOW(4, false, uint32_t); else MCTQ_WINDOW(4, false, uint64_t); }
  } else {
    if (idx32) MCTQ_WINDOW(1, false, uint32_t); else MCTQ_WINDOW(1, false, uint64_t);
  }
#undef MCTQ_WINDOW
  return check_launch("window launch");
}

static int lut_class(int n_lut) { return n_lut <= 4 ? 4 : n_lut <= 16 ? 16 : n_lut <= 64 ? 64 : 0; }

static void fill_lut_common(LutCommon& op, const float* thr, float eps, float mult, float cmin, float cmax) {
  op.thr = thr; op.eps = eps; op.mult = mult; op.inv_mult = 1.0f / mult; op.cmin = cmin; op.cmax = cmax;
}

template <int LP>
static LutOp<LP> make_lut_op(const float* thr, float eps, const float* lut, int n_lut, float mult, float cmin, float cmax) {
  LutOp<LP> op;
  fill_lut_common(op, thr, eps, mult, cmin, cmax);
  op.lut = lut; op.n_lut = n_lut;
  return op;
}

// ---- decision table (host) ---------------------------------------------------------------------
static float lut_literal_host(float t, const float* lut, int n) {
  float best_c = lut[0];
  float best_d = fabsf(t - lut[0]);
  for (int j = 1; j < n; ++j) {
    const float d = fabsf(t - lut[j]);
    if (d < best_d) { best_d = d; best_c = lut[j]; }
  }
  return best_c;
}
static uint32_t f2ord(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static float ord2f(uint32_t o) { uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; memcpy(&f, &u, 4); return f; }

static int table_entries(float cmin, float cmax) {
  if (!(cmin < cmax) || cmin != floorf(cmin) || cmax != floorf(cmax)) return -1;
  const double k = 2.0 * ((double)cmax - (double)cmin) + 1.0;
  if (k > 2048.0) return -1;                 // (entries + 1) * 16 B must stay well inside LDS
  return (int)k;
}

static int check_lut_args(const float* lut, int32_t n_lut, float mult) {
  if (!lut) return fail_arg("lut is NULL");
  if (n_lut < 1 || n_lut > 4096) return fail_arg("n_lut must be in [1, 4096]");
  int e = 0;
  if (!(mult > 0.0f) || frexpf(mult, &e) != 0.5f) return fail_arg("mult must be a positive power of two");
  return 0;
}

}  // namespace mctq

using namespace mctq;

extern "C" {

int mctq_abi_version(void) { return MCTQ_ABI_VERSION; }

const char* mctq_last_error(void) { return g_err; }

int mctq_set_tuning(const char* key, int32_t value) {
  if (!key) return fail_arg("key is NULL");
  if (!strcmp(key, "nt")) {
    if (value != 0 && value != 1) return fail_arg("nt must be 0 or 1");
    g_nt = value;
    return 0;
  }
  if (!strcmp(key, "unroll")) {
    if (value != 1 && value != 2 && value != 4 && value != 8) return fail_arg("unroll must be 1, 2, 4 or 8");
    g_unroll = value;
    return 0;
  }
  if (!strcmp(key, "heavy_unroll")) {
    if (value != 0 && value != 1 && value != 2 && value != 4) return fail_arg("heavy_unroll must be 0, 1, 2 or 4");
    g_heavy_unroll = value;
    return 0;
  }
  return fail_arg("unknown tuning key");
}

int mctq_selftest_division(const float* divisors, int32_t n_div, uint64_t* mismatches, void* stream) {
  if (!divisors || !mismatches || n_div < 1) return fail_arg("bad selftest arguments");
  hipLaunchKernelGGL(selftest_division_kernel, dim3(cu_count() * 8), dim3(kThreads), 0, (hipStream_t)stream,
                     divisors, (int)n_div, reinterpret_cast<unsigned long long*>(mismatches));
  return check_launch("selftest launch");
}

int mctq_fq_per_tensor_f32(const float* x, float* y, int64_t n, float scale, int32_t zero_point,
                           int32_t quant_min, int32_t quant_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (quant_min > quant_max) return fail_arg("quant_min > quant_max");
  AffineOp op;
  op.scales = nullptr; op.zps = nullptr;
  op.lo = (float)quant_min; op.hi = (float)quant_max;
  AffineOp::Param p;
  p.s = scale;
  p.inv = 1.0f / scale;            // host IEEE division == ATen's 1.0f / scale
  p.zf = (float)zero_point;
  return launch_flat(op, p, x, y, n, 0, (hipStream_t)stream);
}

int mctq_fq_per_channel_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                            const float* scales, const int32_t* zero_points, int32_t quant_min, int32_t quant_max,
                            void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  if (quant_min > quant_max) return fail_arg("quant_min > quant_max");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !scales || !zero_points)) return fail_arg("NULL pointer");
  AffineOp op;
  op.scales = scales; op.zps = zero_points;
  op.lo = (float)quant_min; op.hi = (float)quant_max;
  return launch_channels(op, x, y, outer, channels, inner, 0, (hipStream_t)stream);
}

int mctq_lut_per_tensor_f32(const float* x, float* y, int64_t n, float thr_div, float thr_mul, const float* lut,
                            int32_t n_lut, float mult, float clip_min, float clip_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (int rc = check_lut_args(lut, n_lut, mult)) return rc;
  hipStream_t st = (hipStream_t)stream;
  switch (lut_class(n_lut)) {
    case 4: { auto op = make_lut_op<4>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max);
              LutOp<4>::Param p = LutOp<4>::make(thr_div, thr_mul); return launch_flat(op, p, x, y, n, 0, st); }
    case 16: { auto op = make_lut_op<16>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max);
               LutOp<16>::Param p = LutOp<16>::make(thr_div, thr_mul); return launch_flat(op, p, x, y, n, 0, st); }
    case 64: { auto op = make_lut_op<64>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max);
               LutOp<64>::Param p = LutOp<64>::make(thr_div, thr_mul); return launch_flat(op, p, x, y, n, 0, st); }
    default: { auto op = make_lut_op<0>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max);
               LutOp<0>::Param p = LutOp<0>::make(thr_div, thr_mul);
               return launch_flat(op, p, x, y, n, (size_t)((n_lut + 3) & ~3) * sizeof(float), st); }
  }
}

int mctq_lut_per_channel_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                             const float* thresholds, float eps, const float* lut, int32_t n_lut, float mult,
                             float clip_min, float clip_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !thresholds)) return fail_arg("NULL pointer");
  if (int rc = check_lut_args(lut, n_lut, mult)) return rc;
  hipStream_t st = (hipStream_t)stream;
  switch (lut_class(n_lut)) {
    case 4: return launch_channels(make_lut_op<4>(thresholds, eps, lut, n_lut, mult, clip_min, clip_max),
                                   x, y, outer, channels, inner, 0, st);
    case 16: return launch_channels(make_lut_op<16>(thresholds, eps, lut, n_lut, mult, clip_min, clip_max),
                                    x, y, outer, channels, inner, 0, st);
    case 64: return launch_channels(make_lut_op<64>(thresholds, eps, lut, n_lut, mult, clip_min, clip_max),
                                    x, y, outer, channels, inner, 0, st);
    default: return launch_channels(make_lut_op<0>(thresholds, eps, lut, n_lut, mult, clip_min, clip_max),
                                    x, y, outer, channels, inner, (size_t)((n_lut + 3) & ~3) * sizeof(float), st);
  }
}

int32_t mctq_lut_table_entries(float clip_min, float clip_max) {
  const int k = table_entries(clip_min, clip_max);
  if (k < 0) return fail_arg("decision table unsupported for this clip range");
  return k;
}

int mctq_lut_build_table(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                         float* table_host) {
  if (!lut_host || !table_host) return fail_arg("NULL pointer");
  if (n_lut < 1 || n_lut > 4096) return fail_arg("n_lut must be in [1, 4096]");
  int e = 0;
  if (!(mult > 0.0f) || frexpf(mult, &e) != 0.5f) return fail_arg("mult must be a positive power of two");
  const int K = table_entries(clip_min, clip_max);
  if (K < 0) return fail_arg("decision table unsupported for this clip range");
  for (int j = 0; j < n_lut; ++j)
    if (!(lut_host[j] == floorf(lut_host[j])) || fabsf(lut_host[j]) > 16777216.0f)
      return fail_arg("decision table needs an integer codebook");
  uint32_t rng = 0x9E3779B9u;
  for (int k = 0; k < K; ++k) {
    const float P = clip_min + 0.5f * (float)k;
    const float lo = fmaxf(clip_min, P - 0.25f), hi = fminf(clip_max, P + 0.25f);
    const float cb = lut_literal_host(lo, lut_host, n_lut), ca = lut_literal_host(hi, lut_host, n_lut);
    float T = -INFINITY;
    if (cb != ca) {
      uint32_t a = f2ord(lo), b = f2ord(hi);          // F(a) == cb, F(b) == ca; smallest b with F == ca
      while (b - a > 1) {
        const uint32_t m = a + (b - a) / 2;
        if (lut_literal_host(ord2f(m), lut_host, n_lut) == ca) b = m; else a = m;
      }
      T = ord2f(b);
      if (lut_literal_host(ord2f(b - 1), lut_host, n_lut) != cb || lut_literal_host(T, lut_host, n_lut) != ca)
        return fail_arg("codebook decision is not a single step");
    }
    // spot-check the single-step model on pseudo-random points of the cell
    for (int r = 0; r < 32; ++r) {
      rng = rng * 1664525u + 1013904223u;
      const uint32_t span = f2ord(hi) - f2ord(lo);
      const float t = ord2f(f2ord(lo) + (span ? rng % (span + 1u) : 0u));
      const float want = lut_literal_host(t, lut_host, n_lut);
      if (want != ((t >= T) ? ca : cb)) return fail_arg("codebook decision is not a single step");
    }
    const float qb = cb / mult, qa = ca / mult;
    const __half hb = __float2half(qb), ha = __float2half(qa);
    if (__half2float(hb) != qb || __half2float(ha) != qa) return fail_arg("codebook centre not exact in fp16");
    uint16_t ub, ua;
    memcpy(&ub, &hb, 2); memcpy(&ua, &ha, 2);
    const uint32_t pair = (uint32_t)ub | ((uint32_t)ua << 16);
    table_host[2 * k + 0] = T;
    memcpy(&table_host[2 * k + 1], &pair, 4);
  }
  table_host[2 * K + 0] = lut_host[0] / mult;        // NaN input: every distance is NaN, argmin = index 0
  table_host[2 * K + 1] = (float)K;
  return 0;
}

static int make_table_op(LutTableOp& op, const float* thr, float eps, const float* table, int32_t entries, float mult,
                         float cmin, float cmax) {
  if (!table) return fail_arg("table is NULL");
  int e = 0;
  if (!(mult > 0.0f) || frexpf(mult, &e) != 0.5f) return fail_arg("mult must be a positive power of two");
  if (entries != table_entries(cmin, cmax)) return fail_arg("entries does not match the clip range");
  fill_lut_common(op, thr, eps, mult, cmin, cmax);
  op.table = table; op.entries = entries; op.koff = 0.5f - 2.0f * cmin;
  return 0;
}

int mctq_lutt_per_tensor_f32(const float* x, float* y, int64_t n, float thr_div, float thr_mul, const float* table,
                             int32_t entries, float mult, float clip_min, float clip_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  LutTableOp op;
  if (int rc = make_table_op(op, nullptr, 0.f, table, entries, mult, clip_min, clip_max)) return rc;
  return launch_flat(op, LutCommon::make(thr_div, thr_mul), x, y, n, (size_t)(((entries + 1) * 2 + 3) & ~3) * 4, (hipStream_t)stream);
}

int mctq_lutt_per_channel_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                              const float* thresholds, float eps, const float* table, int32_t entries, float mult,
                              float clip_min, float clip_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !thresholds)) return fail_arg("NULL pointer");
  LutTableOp op;
  if (int rc = make_table_op(op, thresholds, eps, table, entries, mult, clip_min, clip_max)) return rc;
  return launch_channels(op, x, y, outer, channels, inner, (size_t)(((entries + 1) * 2 + 3) & ~3) * 4, (hipStream_t)stream);
}

}  // extern "C"
