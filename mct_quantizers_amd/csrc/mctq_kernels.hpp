// mctq_kernels.hpp -- gfx950 (MI355X / CDNA4) kernels and launch helpers of libmctq_hip.so.
// Included by the translation units that hold the C ABI (mctq_affine.hip, mctq_codes.hip, mctq_grid.hip,
// mctq_lut_scan.hip, mctq_lut_table.hip, mctq_misc.hip; mctq_qlinear.hip uses only the shared helpers); they are
// compiled in parallel and linked into one library.
//
// Hot path of sony/mct_quantizers' PyTorch inferable quantizers, written for CDNA4:
// one fused load -> scale -> round-half-even -> clamp -> dequant -> store pass (8 algorithmic
// bytes per float32 element) for the affine quantizers, and one fused divide -> clamp -> codebook
// decision -> dequant pass for the LUT quantizers.  The work is elementwise and HBM-bound, so
// there is no MFMA here (the one matrix-core kernel, the integer consumer of the codes, lives in
// mctq_qlinear.hip); what matters is 16-byte-per-lane coalesced traffic (1 KiB per wave
// instruction), enough loads in flight per lane, and keeping the per-channel parameters out of the
// vector memory pipe (SGPR broadcast or an LDS window).
//
// Launch shapes (each templated on the op and on the storage types TI -> TO):
//   flat          per-tensor parameters in kernel arguments (SGPRs).
//   rows          per-channel, long vector-divisible rows: one block = one tile of one
//                 (outer, channel) row; data loads are issued first, then the channel's parameters
//                 arrive through scalar loads (wave-uniform index) -> SGPR broadcast.
//   lastaxis      per-channel along the fastest axis (inner == 1): a lane keeps its N channels while stepping down the
//                 tensor slab by slab; parameters by 16-byte loads from the L1/L2-resident tables once per lane,
//                 inverted by a five-instruction exact reciprocal.
//   shortrows     (affine ops) rows shorter than a tile but at least one lane-vector long, and long 16-bit rows of
//                 launches that fill 7/8 ... 1 round of resident blocks: one block = one contiguous tile; every lane-vector
//                 reads its own row's scale behind the tile's data loads -- no LDS, no barrier.
//   window        what is left (rows shorter than a lane-vector, unaligned tensors, > 2^32 elements, the non-affine
//                 ops' short rows): one block = one contiguous tile; the parameters of the rows that tile touches are
//                 staged in LDS once per block and looked up per element without any per-element division.
// The LUT ops stage their codebook table in LDS after the block's data loads are in flight.
//
// What is instantiated is what the dispatchers below can select (evidence: the launch-variant log of the
// whole GPU suite + every bench configuration + the shape sweeps, profiles/r05/launch_variants_all.log):
//   tuned ops (AffineOp, LutTableOp): lane-vectors per lane U in {1, 2, 4} (16-bit affine: {2, 4}) x cache policy
//     NT in {1, 2}; window tiles of 4 lane-vectors, or 1 when the parameter window of a 4-wide tile would not fit LDS;
//     lastaxis: two slabs per lane without a zero-point table, four with one (AffineOp), two for every other op;
//     shortrows (AffineOp): {zero points or none} x {rows of whole lane-vectors or not} x NT;
//   every other op (integer codes, export grid, literal scan, threshold lists): ONE variant per launch shape.
// Experiments that were measured and not adopted (persistent blocks, the compact decision table, staging ablations) live
// under tools/experiments/, outside the library.
//
// Storage types: float32, float16, bfloat16 in; the affine ops write the input type (as ATen
// does), the LUT ops always write float32 (the reference's op chain promotes).  All arithmetic is
// float32.  Arithmetic contract: include/mctq_hip.h.  Compile with -ffp-contract=off, no fast-math.
#pragma once
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

// ------------------------------------------------------------------------------------------
// Storage: N elements per lane access, N = 16 B / max(sizeof in, sizeof out)
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));

typedef int8_t i8x4 __attribute__((ext_vector_type(4)));
typedef int8_t i8x8 __attribute__((ext_vector_type(8)));
typedef uint8_t u8x4 __attribute__((ext_vector_type(4)));
typedef uint8_t u8x8 __attribute__((ext_vector_type(8)));

template <class T, int N> struct VecT;
template <> struct VecT<int8_t, 4> { typedef i8x4 type; };
template <> struct VecT<int8_t, 8> { typedef i8x8 type; };
template <> struct VecT<uint8_t, 4> { typedef u8x4 type; };
template <> struct VecT<uint8_t, 8> { typedef u8x8 type; };
template <> struct VecT<float, 4> { typedef f32x4 type; };
template <> struct VecT<_Float16, 8> { typedef f16x8 type; };
template <> struct VecT<_Float16, 4> { typedef f16x4 type; };
template <> struct VecT<__bf16, 8> { typedef b16x8 type; };
template <> struct VecT<__bf16, 4> { typedef b16x4 type; };

template <class TI, class TO>
struct IO {
  static constexpr int N = 16 / (int)(sizeof(TI) > sizeof(TO) ? sizeof(TI) : sizeof(TO));
  typedef typename VecT<TI, N>::type VI;
  typedef typename VecT<TO, N>::type VO;

  // NT: 1 = non-temporal loads and stores (streaming: the default),
  //     2 = non-temporal loads, cached stores (the output is consumed right away and fits the aggregate L2)
  template <int NT>
  __device__ __forceinline__ static VI load(const TI* p) {
    const VI* q = reinterpret_cast<const VI*>(p);
    return __builtin_nontemporal_load(q);
  }
  template <int NT>
  __device__ __forceinline__ static void store(TO* p, VO v) {
    VO* q = reinterpret_cast<VO*>(p);
    if (NT == 1) __builtin_nontemporal_store(v, q);
    else *q = v;
  }
  __device__ __forceinline__ static void unpack(VI v, float* f) {
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = (float)v[i];            // exact widening
  }
  __device__ __forceinline__ static VO pack(const float* f) {
    VO o;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if constexpr (sizeof(TO) == 1) o[i] = (TO)(int32_t)f[i];  // integer codes in [-128, 255]: the low byte (int8 and uint8 alike)
      else o[i] = (TO)f[i];                                     // round-to-nearest-even narrowing
    }
    return o;
  }
};

// ------------------------------------------------------------------------------------------
// Ops.  An Op describes: per-channel Param (kWords floats when staged in LDS), fetch(c) that
// builds it from the device tables, an optional Book (codebook / table) set up once per block,
// apply<FAST>(x, param, book) and optionally a batched tile<FAST, NE>().
// ------------------------------------------------------------------------------------------

struct NoBook {};

// 1 / d, correctly rounded, for a POSITIVE d in [2^-100, 2^100] in five VALU instructions: v_rcp_f32 (1 ulp) and two Newton
// steps whose residuals 1 - d * r are exact FMAs -- against the eleven of the compiler's IEEE expansion (two v_div_scale,
// v_rcp, five FMAs, v_div_fmas, v_div_fixup), which only differs in how it treats the exponent range and the special values
// excluded here.  Equal to `1.0f / d` for every one of the 3 355 443 202 float32 bit patterns of that range and its negative
// mirror (mctq_selftest_reciprocal, tests/test_gpu_parity.py; profiles/r06/chanlast2_a.log).  Used where a LANE owns its own
// divisors (channel-last and short-row launches: N reciprocals per lane); wave-uniform divisors keep the plain division.
__device__ __forceinline__ float recip_exact(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
// every one of M values is a positive float in [2^-100, 2^100]: compared as raw bit patterns (negative values, NaNs and
// infinities are large unsigned numbers) -- M min / max operations and two compares
constexpr uint32_t kRecipLo = 0x0d800000u, kRecipHi = 0x71800000u;
template <int M>
__device__ __forceinline__ bool recip_all_in_range(const float (&v)[M]) {
  uint32_t lo = __float_as_uint(v[0]), hi = lo;
#pragma unroll
  for (int j = 1; j < M; ++j) { const uint32_t b = __float_as_uint(v[j]); lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
  return lo >= kRecipLo && hi <= kRecipHi;
}

struct AffineOp {
  static constexpr const char* kName = "AffineOp";
  const float* __restrict__ scales;    // [C] (per-channel launches only)
  const int32_t* __restrict__ zps;     // [C], or NULL for all-zero zero points (symmetric quantizers)
  float lo, hi;                        // clamp domain as floats (exact: |q| < 2^24)

  struct Param { float s, inv, zf; };
  typedef NoBook Book;
  static constexpr int kWords = 3;
  static constexpr bool kHeavy = false;       // a few VALU ops per element: pure streaming
  static constexpr int kFixedU = 0;           // 0: the tuned set of launch variants (U x cache policy); else ONE variant

  __host__ __device__ __forceinline__ static Param make(float s, int32_t zp) {
    Param p;
    p.s = s;
    p.inv = 1.0f / s;                  // correctly rounded IEEE division, as ATen's 1.0f / scale
    p.zf = (float)zp;
    return p;
  }
  __device__ __forceinline__ Param fetch(uint32_t c) const { return make(scales[c], zps ? zps[c] : 0); }
  // N consecutive channels starting at c (c % 4 == 0, tables 16-byte aligned): 16-byte loads of the tables
  template <int N>
  __device__ __forceinline__ void fetch_vec(uint32_t c, Param* p) const {
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < N; j += 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(scales + c + j);
      i32x4 z4 = {0, 0, 0, 0};
      if (zps) z4 = *reinterpret_cast<const i32x4*>(zps + c + j);       // NULL = all zero (symmetric)
#pragma unroll
      for (int i = 0; i < 4; ++i) p[j + i] = make(s4[i], z4[i]);
    }
  }
  // The same N channels for a lane that owns them (channel-last launches): the N reciprocals by recip_exact when every
  // active lane of the wave qualifies (wave-uniform branch), and -- ZP false: the launch has no zero-point table -- zf a
  // compile-time zero, so that the shifted bounds lo - zf / hi - zf of apply() stay the two scalar registers.
  template <int N, bool ZP>
  __device__ __forceinline__ void fetch_lane(uint32_t c, Param* p) const {
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    float sv[N];
#pragma unroll
    for (int j = 0; j < N; j += 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(scales + c + j);
      i32x4 z4 = {0, 0, 0, 0};
      if (ZP) z4 = *reinterpret_cast<const i32x4*>(zps + c + j);
#pragma unroll
      for (int i = 0; i < 4; ++i) { sv[j + i] = s4[i]; p[j + i].s = s4[i]; p[j + i].zf = ZP ? (float)z4[i] : 0.0f; }
    }
    if (__builtin_amdgcn_ballot_w64(!recip_all_in_range(sv)) == 0) {
#pragma unroll
      for (int j = 0; j < N; ++j) p[j].inv = recip_exact(sv[j]);
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) p[j].inv = 1.0f / sv[j];
    }
  }
  __host__ bool tables_aligned16() const { return (((uintptr_t)scales | (uintptr_t)zps) & 15u) == 0; }
  __device__ __forceinline__ static void put(float* lds, uint32_t i, uint32_t stride, const Param& p) {
    lds[i] = p.s; lds[stride + i] = p.inv; lds[2 * stride + i] = p.zf;
  }
  __device__ __forceinline__ static Param get(const float* lds, uint32_t i, uint32_t stride) {
    Param p; p.s = lds[i]; p.inv = lds[stride + i]; p.zf = lds[2 * stride + i]; return p;
  }
  __device__ __forceinline__ uint32_t book_words() const { return 0; }
  __device__ __forceinline__ Book setup(float*) const { return Book(); }
  __device__ __forceinline__ static bool can_fast(const Param&) { return true; }

  // (clamp(r + z, lo, hi) - z) * s with r = rint(x * inv), evaluated as fma(med3(r, lo - z, hi - z), s, +0):
  // four VALU ops per element instead of seven, and the shifted bounds are loop-invariant per parameter set.
  // Identical bit for bit: r and z are integers, so r + z is exact wherever it lies inside [lo, hi] (|.| <= 2^24)
  // and rounds monotonically outside, where both forms saturate to fl(lo - z) / fl(hi - z) -- the very rounding the
  // subtraction q - z performs; v_med3_f32 answers min(lo - z, hi - z) for a NaN as fmax(NaN, lo) answers lo; and the
  // +0 of the fma turns the -0 of a tiny negative x back into the +0 that (q - z) * s yields.
  template <bool FAST = true>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book&) const {
    const float r = __builtin_rintf(x * p.inv);       // v_rndne_f32: ties to even
    const float q = __builtin_amdgcn_fmed3f(r, lo - p.zf, hi - p.zf);   // NaN -> lo - z, +inf -> hi - z, -inf -> lo - z
    float y = __builtin_fmaf(q, p.s, 0.0f);
    // The product is a float32 VALUE before any narrowing: for half-precision outputs the compiler would otherwise fold
    // the fma and the conversion into one v_fma_mix with a single rounding, where ATen rounds to float32 and then to half.
    asm("" : "+v"(y));
    return y;
  }
};

// Same clamp index, left as an integer code (stored as int8 / uint8): the integer domain of the affine
// quantizers for consumers that dequantize themselves (1 B written per element instead of 4).
struct AffineCodesOp : AffineOp {
  static constexpr const char* kName = "AffineCodesOp";
  static constexpr int kFixedU = 4;            // an extension without a reference counterpart: one variant per launch shape
  template <bool FAST = true>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book&) const {
    const float q = __builtin_rintf(x * p.inv) + p.zf;
    return fminf(fmaxf(q, lo), hi);                   // integer-valued, inside the 8-bit code range
  }
};

// Export-time arithmetic (the reference's ONNX branch, `_use_custom_impl and torch.jit.is_tracing()`):
// clip to [lo, hi], TRUE division by the step, round half even, scale back -- a different last-ulp
// contract from ATen's fake-quant (weights_symmetric...py:32-70, weights_uniform...py:34-78,
// activation_symmetric...py:29-54, activation_uniform...py:32-65).  Runs once per export: plain streaming,
// IEEE division per element.
struct GridOp {
  static constexpr const char* kName = "GridOp";
  const float* __restrict__ los;       // [C] lower clip bounds   (per-channel launches only)
  const float* __restrict__ his;       // [C] upper clip bounds
  const float* __restrict__ steps;     // [C] grid steps
  int shifted;                         // 0: rint(c / d) * d;   1: d * rint((c - lo) / d) + lo

  struct Param { float lo, hi, d; };
  typedef NoBook Book;
  static constexpr int kWords = 3;
  static constexpr bool kHeavy = false;
  static constexpr int kFixedU = 4;            // runs once per export: one variant per launch shape

  __host__ __device__ __forceinline__ static Param make(float lo, float hi, float d) {
    Param p; p.lo = lo; p.hi = hi; p.d = d; return p;
  }
  __device__ __forceinline__ Param fetch(uint32_t c) const { return make(los[c], his[c], steps[c]); }
  template <int N>
  __device__ __forceinline__ void fetch_vec(uint32_t c, Param* p) const {
#pragma unroll
    for (int j = 0; j < N; j += 4) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(los + c + j);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(his + c + j);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(steps + c + j);
#pragma unroll
      for (int i = 0; i < 4; ++i) p[j + i] = make(a4[i], b4[i], d4[i]);
    }
  }
  __host__ bool tables_aligned16() const { return (((uintptr_t)los | (uintptr_t)his | (uintptr_t)steps) & 15u) == 0; }
  __device__ __forceinline__ static void put(float* lds, uint32_t i, uint32_t stride, const Param& p) {
    lds[i] = p.lo; lds[stride + i] = p.hi; lds[2 * stride + i] = p.d;
  }
  __device__ __forceinline__ static Param get(const float* lds, uint32_t i, uint32_t stride) {
    Param p; p.lo = lds[i]; p.hi = lds[stride + i]; p.d = lds[2 * stride + i]; return p;
  }
  __device__ __forceinline__ uint32_t book_words() const { return 0; }
  __device__ __forceinline__ Book setup(float*) const { return Book(); }
  __device__ __forceinline__ static bool can_fast(const Param&) { return true; }

  template <bool FAST = true>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book&) const {
    float c = (x < p.lo) ? p.lo : x;                  // torch.where(x < lo, lo, x): NaN and -0.0 == lo pass through
    c = (x > p.hi) ? p.hi : c;                        // torch.where(x > hi, hi, c)
    if (shifted) return p.d * __builtin_rintf((c - p.lo) / p.d) + p.lo;
    return __builtin_rintf(c / p.d) * p.d;
  }
};

// Shared by the two LUT ops: per-channel parameters and the shared-divisor division.
struct LutCommon {
  const float* __restrict__ thr;       // [C] thresholds (per-channel launches only)
  float eps;
  float mult, inv_mult, cmin, cmax;
  int step_round;                      // 0, or MCTQ_DT_F16 / MCTQ_DT_BF16: round the quotient and the scaled
                                       // value to that type (half-precision activations, per-tensor only)

  // divisor d = fl32(thr + eps), multiplier thr; for the fast exact division below also ds = d / mult
  // (exact: mult is a power of two) and r = RN(1/ds); r == 0 selects the plain IEEE division.
  // x / ds == (x / d) * mult with a single rounding, bit-identical to the reference's two steps wherever
  // the codebook decision can depend on the value (no under/overflow there).
  struct Param { float d, t, r, ds; };
  static constexpr int kWords = 4;
  static constexpr bool kHeavy = true;         // launched through the heavy-op dispatch
  static constexpr int kFixedU = 0;            // LutTableOp: the tuned set; the literal-scan and the list ops override it (one variant each)

  __host__ __device__ __forceinline__ static Param make(float d, float t, float mult) {
    Param p; p.d = d; p.t = t;
    const float a = fabsf(d);
    const bool ok = a > 0x1p-60f && a < 0x1p60f;
    p.ds = d / mult;
    p.r = ok ? 1.0f / p.ds : 0.0f;
    return p;
  }
  __device__ __forceinline__ Param fetch(uint32_t c) const {
    const float t = thr[c];
    return make(t + eps, t, mult);
  }
  template <int N>
  __device__ __forceinline__ void fetch_vec(uint32_t c, Param* p) const {
#pragma unroll
    for (int j = 0; j < N; j += 4) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(thr + c + j);
#pragma unroll
      for (int i = 0; i < 4; ++i) p[j + i] = make(t4[i] + eps, t4[i], mult);
    }
  }
  __host__ bool tables_aligned16() const { return ((uintptr_t)thr & 15u) == 0; }
  __device__ __forceinline__ static void put(float* lds, uint32_t i, uint32_t stride, const Param& p) {
    lds[i] = p.d; lds[stride + i] = p.t; lds[2 * stride + i] = p.r; lds[3 * stride + i] = p.ds;
  }
  __device__ __forceinline__ static Param get(const float* lds, uint32_t i, uint32_t stride) {
    Param p; p.d = lds[i]; p.t = lds[stride + i]; p.r = lds[2 * stride + i]; p.ds = lds[3 * stride + i]; return p;
  }
  __device__ __forceinline__ static bool can_fast(const Param& p) { return p.r != 0.0f; }

  // x / ds, correctly rounded, for a divisor shared by many elements: with r = RN(1/ds), q0 = x*r
  // followed by two residual corrections with exact FMA residuals -- the recurrence the compiler's
  // IEEE expansion runs after its v_rcp/Newton steps, minus the per-element reciprocal and scaling
  // (5 VALU ops instead of ~11, no v_div_* wait states).  The caller keeps |x / ds| <= 2^60 (x is
  // clamped first: beyond that the value saturates the clip range anyway); below |q| ~ 2^-31 the
  // codebook decision does not depend on the last bits (every |t - c| with c != 0 rounds to |c|).
  // Verified exhaustively against '/' on the GPU (tests/test_gpu_parity.py::test_fast_division_is_exact).
  __device__ __forceinline__ static float divide_fast(float x, float ds, float r) {
    const float q0 = x * r;
    const float e0 = __builtin_fmaf(-q0, ds, x);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-q1, ds, x);
    return __builtin_fmaf(e1, r, q1);
  }

  __device__ __forceinline__ float narrow(float v) const {
    if (step_round == MCTQ_DT_F16) return (float)(_Float16)v;
    if (step_round == MCTQ_DT_BF16) return (float)(__bf16)v;
    return v;
  }
  // (x / d) * mult as the reference's op chain computes it (quantizer_utils.py:169), incl. the
  // per-op roundings of a half-precision activation tensor.  FAST (wave-uniform divisor with
  // p.r != 0): NaN inputs come out as the saturated minimum and must be handled by the caller.
  template <bool FAST>
  __device__ __forceinline__ float scaled(float x, const Param& p) const {
    if (step_round != 0) return narrow(narrow(x / p.d) * mult);
    if constexpr (FAST) {
      const float xmax = 0x1p60f * fabsf(p.ds);                          // uniform, once per tile
      const float xc = __builtin_amdgcn_fmed3f(x, -xmax, xmax);          // NaN -> -xmax
      return divide_fast(xc, p.ds, p.r);
    } else {
      return (x / p.d) * mult;
    }
  }
};

// Literal codebook scan (any codebook, any bit width): the reference's first-minimum argmin.
template <int LP>
struct RegBook { float c[LP]; };
struct LdsBook { const float* c; int n; };

template <int LP>   // LP > 0: codebook broadcast into LP scalar registers; LP == 0: codebook in LDS
struct LutOp : LutCommon {
  static constexpr const char* kName = LP > 0 ? "LutOp<registers>" : "LutOp<lds>";
  const float* __restrict__ lut;       // [n_lut] device codebook, caller's order
  int n_lut;
  // Fallback path (non-integer codebooks): built in one launch variant only (2 lane-vectors per lane,
  // non-temporal, one tile per block) to keep the library small.
  static constexpr int kFixedU = 2;

  typedef typename std::conditional<(LP > 0), RegBook<(LP > 0 ? LP : 1)>, LdsBook>::type Book;

  __device__ __forceinline__ uint32_t book_words() const { return LP > 0 ? 0u : (uint32_t)((n_lut + 3) & ~3); }

  // Called by every thread of the block before any apply().
  __device__ __forceinline__ Book setup(float* lds) const {
    if constexpr (LP > 0) {
      // One coalesced load per wave: lane j holds lut[j]; every entry is then broadcast to a
      // scalar register with v_readlane (wave-level shuffle), so the scan below reads SGPRs.
      const int lane = threadIdx.x & 63;
      float v = INFINITY;                              // padding never wins a strict '<'
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
    const float v = scaled<FAST>(x, p);
    float t = fminf(fmaxf(v, cmin), cmax);
    // torch.clip keeps NaN: a NaN input (hidden by the clamped numerator of the fast division) or a NaN
    // made by the IEEE division itself (0/0, inf/inf)
    t = (x != x) ? x : t;
    t = (v != v) ? v : t;
    float best_c = b.c[0];
    float best_d = fabsf(t - best_c);
    if constexpr (LP > 0) {
#pragma unroll
      for (int j = 1; j < LP; ++j) {
        const float c = b.c[j];
        const float d = fabsf(t - c);
        const bool lt = d < best_d;                    // strict: first minimum wins; NaN never wins
        best_d = lt ? d : best_d;
        best_c = lt ? c : best_c;
      }
    } else {
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
struct LutTableBook { const f32x2* tab; float nan_q; };

struct LutTableOp : LutCommon {
  static constexpr const char* kName = "LutTableOp";
  const float* __restrict__ table;     // device, (entries + 1) x 2 words
  int entries;
  float koff;                          // 0.5 - 2*clip_min
  float kmax;                          // entries - 1

  typedef LutTableBook Book;
  __device__ __forceinline__ uint32_t book_words() const { return ((uint32_t)(entries + 1) * 2u + 3u) & ~3u; }

  __device__ __forceinline__ Book setup(float* lds) const {
    const f32x2* src = reinterpret_cast<const f32x2*>(table);
    f32x2* dst = reinterpret_cast<f32x2*>(lds);
    for (int j = threadIdx.x; j <= entries; j += kThreads) dst[j] = src[j];
    __syncthreads();
    Book b; b.tab = dst; b.nan_q = dst[entries].x;
    return b;
  }
  // The same staging in two halves, for kernels that issue their data loads in between.  Vector-memory loads return IN
  // ORDER (s_waitcnt vmcnt counts them as a queue): table reads issued BEHIND the tile's data loads could be written to
  // LDS only once all of those had landed, and the block's barrier + table lookups would start after the HBM latency
  // instead of under it.  prefetch() is called BEFORE the data loads (the table is L2-resident and arrives early),
  // commit() after them: it waits for the table words only.  (config 4: 59.3 -> 57.x us)
  struct Prefetch { f32x2 r[8]; };                   // (entries + 1) <= 2048 words of 8 bytes over 256 threads
  __device__ __forceinline__ Prefetch prefetch() const {
    const f32x2* src = reinterpret_cast<const f32x2*>(table);
    Prefetch p;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = (int)threadIdx.x + i * kThreads;
      if (j <= entries) p.r[i] = src[j];
    }
    return p;
  }
  __device__ __forceinline__ Book commit(const Prefetch& p, float* lds) const {
    f32x2* dst = reinterpret_cast<f32x2*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = (int)threadIdx.x + i * kThreads;
      if (j <= entries) dst[j] = p.r[i];
    }
    __syncthreads();
    Book b; b.tab = dst; b.nan_q = dst[entries].x;
    return b;
  }

  // stage 1: scaled value and its table index.  The value is NOT clamped: the index is (integer
  // clamp, one v_med3_i32), and comparing the raw value with the edge entries' thresholds gives the
  // same side as comparing the clipped one (T_0 > clip_min or -inf; T_last <= clip_max).
  template <bool FAST>
  __device__ __forceinline__ void locate(float x, const Param& p, float& v, int& k) const {
    v = scaled<FAST>(x, p);
    // nearest half-integer point (any tie is fine), clamped as a float (one v_med3_f32; NaN -> 0) before the conversion
    k = (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(v, 2.0f, koff), 0.0f, kmax);
  }
  // stage 3: pick the side of the step, dequantize
  template <bool FAST>
  __device__ __forceinline__ float decide(float x, float v, f32x2 e, const Param& p, const Book& b) const {
    const uint32_t pair = __float_as_uint(e.y);
    const uint32_t h = (v >= e.x) ? (pair >> 16) : pair;             // cvt reads the low half only
    float q = __half2float(__ushort_as_half((unsigned short)h));
    // NaN input (all-NaN distances: argmin is index 0).  FAST: the clamped numerator hid it, look at x;
    // otherwise the IEEE chain propagated it (and 0/0, inf/inf) into v.
    const bool nan = (FAST && step_round == 0) ? (x != x) : (v != v);
    q = nan ? b.nan_q : q;
    return q * p.t;
  }

  template <bool FAST = false>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book& b) const {
    float v; int k;
    locate<FAST>(x, p, v, k);
    return decide<FAST>(x, v, b.tab[k], p, b);
  }

  // A whole tile: all indices first, then all LDS reads back to back, then all selects, so one
  // s_waitcnt covers NE lookups instead of one per element.
  template <bool FAST, int NE>
  __device__ __forceinline__ void tile(const float* in, float* out, const Param& p, const Book& b) const {
    float v[NE];
    int k[NE];
    f32x2 e[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) locate<FAST>(in[i], p, v[i], k[i]);
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = b.tab[k[i]];
#pragma unroll
    for (int i = 0; i < NE; ++i) out[i] = decide<FAST>(in[i], v[i], e[i], p, b);
  }
};

// Threshold-list ("steps") codebook quantizer: any INTEGER codebook, any clip range the decision table above is too
// large for (lut_values_bitwidth > 10).  For integer centres at most 2^20 apart from t the float32 distances
// fl(|t - c|) of two centres on the same side of t never tie (they differ by >= 1, the rounding error is < 2^-3), so
// the literal first-minimum scan is a non-decreasing staircase of t whose steps are the hand-overs between ADJACENT
// sorted centres; mctq_lut_build_steps() finds every step's exact float32 threshold T_k by bisection of the literal
// scan over float bit patterns (list-order tie-breaking included) and checks the model on sample points.  The kernel
// counts the thresholds <= t by a branchless binary search over the sorted list in LDS -- log2(P) reads per element
// instead of the scan's 4 VALU ops per codebook entry -- and reads the dequantized centre.
// Layout (floats): T[0 .. P-1] (T[0] unused, padding +inf), Q[0 .. P-1], q for NaN input, P.  P = power of two >= centres.
struct LutStepsBook { const float* T; const float* Q; float nan_q; };

struct LutStepsOp : LutCommon {
  static constexpr const char* kName = "LutStepsOp";
  static constexpr int kFixedU = 4;
  const float* __restrict__ steps;     // device, 2 * P + 2 words
  int P;

  typedef LutStepsBook Book;
  __device__ __forceinline__ uint32_t book_words() const { return ((uint32_t)(2 * P + 2) + 3u) & ~3u; }

  __device__ __forceinline__ Book setup(float* lds) const {
    for (int j = threadIdx.x; j < 2 * P + 2; j += kThreads) lds[j] = steps[j];
    __syncthreads();
    Book b; b.T = lds; b.Q = lds + P; b.nan_q = lds[2 * P];
    return b;
  }

  template <bool FAST>
  __device__ __forceinline__ float clipped(float x, const Param& p) const {
    const float v = scaled<FAST>(x, p);
    float t = fminf(fmaxf(v, cmin), cmax);
    t = (x != x) ? x : t;                              // torch.clip keeps NaN (see LutOp::apply)
    t = (v != v) ? v : t;
    return t;
  }

  template <bool FAST = false>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book& b) const {
    const float t = clipped<FAST>(x, p);
    int idx = 0;
    for (int s = P >> 1; s > 0; s >>= 1) idx += (t >= b.T[idx + s]) ? s : 0;     // NaN: no threshold is <= t
    const float q = (t != t) ? b.nan_q : b.Q[idx];
    return q * p.t;
  }

  // a whole tile level by level: the NE reads of a level are issued back to back
  template <bool FAST, int NE>
  __device__ __forceinline__ void tile(const float* in, float* out, const Param& p, const Book& b) const {
    float t[NE];
    int idx[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { t[i] = clipped<FAST>(in[i], p); idx[i] = 0; }
    for (int s = P >> 1; s > 0; s >>= 1) {
      float th[NE];
#pragma unroll
      for (int i = 0; i < NE; ++i) th[i] = b.T[idx[i] + s];
#pragma unroll
      for (int i = 0; i < NE; ++i) idx[i] += (t[i] >= th[i]) ? s : 0;
    }
    float q[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) q[i] = b.Q[idx[i]];
#pragma unroll
    for (int i = 0; i < NE; ++i) out[i] = ((t[i] != t[i]) ? b.nan_q : q[i]) * p.t;
  }
};

// Long lists (P >= 128) may carry a CELL INDEX behind the list: {G, max thresholds per cell, gscale, clip_min} and G words
// {index of the cell's first threshold | thresholds in the cell << 16}, cell(t) = trunc(clamp((t - clip_min) * gscale)).
// cell() is monotone, so thresholds of earlier cells are <= t, those of later cells > t, and only the (at most 4)
// thresholds of t's own cell are compared: 2 + maxc LDS reads per element instead of log2(P) + 1 dependent ones
// (256 centres: 93 -> 76 us on the config-4 tensor).  A separate op: the short lists keep the lean kernel above.
struct LutCellsBook { const float* T; const float* Q; float nan_q; const uint32_t* cells; int maxc; float gscale, gmax; };

struct LutCellsOp : LutCommon {
  static constexpr const char* kName = "LutCellsOp";
  static constexpr int kFixedU = 4;
  const float* __restrict__ steps;     // device, n_words floats: the list (2 * P + 2), then the cell index
  int P, n_words;

  typedef LutCellsBook Book;
  __device__ __forceinline__ uint32_t book_words() const { return ((uint32_t)n_words + 3u) & ~3u; }

  __device__ __forceinline__ Book setup(float* lds) const {
    for (int j = threadIdx.x; j < n_words; j += kThreads) lds[j] = steps[j];
    __syncthreads();
    const float* h = lds + 2 * P + 2;
    Book b; b.T = lds; b.Q = lds + P; b.nan_q = lds[2 * P];
    b.gmax = h[0] - 1.0f; b.maxc = (int)h[1]; b.gscale = h[2];
    b.cells = reinterpret_cast<const uint32_t*>(h + 4);
    return b;
  }

  template <bool FAST>
  __device__ __forceinline__ float clipped(float x, const Param& p) const {
    const float v = scaled<FAST>(x, p);
    float t = fminf(fmaxf(v, cmin), cmax);
    t = (x != x) ? x : t;                              // torch.clip keeps NaN (see LutOp::apply)
    t = (v != v) ? v : t;
    return t;
  }

  template <bool FAST = false>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book& b) const {
    float in[1] = {x}, out[1];
    tile<FAST, 1>(in, out, p, b);
    return out[0];
  }

  template <bool FAST, int NE>
  __device__ __forceinline__ void tile(const float* in, float* out, const Param& p, const Book& b) const {
    float t[NE];
    int idx[NE], first[NE], n[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      t[i] = clipped<FAST>(in[i], p);
      const float d = t[i] - cmin;                     // the builder evaluates the same two operations (steps_cell);
      const uint32_t e = b.cells[(int)__builtin_amdgcn_fmed3f(d * b.gscale, 0.0f, b.gmax)];      // NaN -> cell 0
      first[i] = (int)(e & 0xffffu); n[i] = (int)(e >> 16); idx[i] = first[i];
    }
    for (int j = 0; j < b.maxc; ++j) {                 // wave-uniform bound (<= 4)
      float th[NE];
#pragma unroll
      for (int i = 0; i < NE; ++i) th[i] = b.T[min(1 + first[i] + j, P - 1)];
#pragma unroll
      for (int i = 0; i < NE; ++i) idx[i] += (j < n[i] && t[i] >= th[i]) ? 1 : 0;
    }
    float q[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) q[i] = b.Q[idx[i]];
#pragma unroll
    for (int i = 0; i < NE; ++i) out[i] = ((t[i] != t[i]) ? b.nan_q : q[i]) * p.t;
  }
};

// one element to its storage type (the scalar tails and the unaligned paths)
template <class TO>
__device__ __forceinline__ TO narrow_to(float v) {
  if constexpr (sizeof(TO) == 1) return (TO)(int32_t)v;
  else return (TO)v;
}

template <class Op, class = void>
struct HasTile : std::false_type {};
template <class Op>
struct HasTile<Op, std::void_t<decltype(&Op::template tile<true, 4>)>> : std::true_type {};

template <class Op, class = void>
struct HasPrefetch : std::false_type {};
template <class Op>
struct HasPrefetch<Op, std::void_t<typename Op::Prefetch>> : std::true_type {};

// NE elements with one Param (all in the same channel).  Batched ops work on at most 16 elements at a
// time (their index / table-entry arrays live in registers).
template <bool FAST, int NE, class Op>
__device__ __forceinline__ void run(const Op& op, const float* in, float* out, const typename Op::Param& p,
                                    const typename Op::Book& b) {
  if constexpr (HasTile<Op>::value) {
    constexpr int CH = NE > 16 ? 16 : NE;
#pragma unroll
    for (int c = 0; c < NE; c += CH) op.template tile<FAST, CH>(in + c, out + c, p, b);
  } else {
#pragma unroll
    for (int i = 0; i < NE; ++i) out[i] = op.template apply<FAST>(in[i], p, b);
  }
}

// U lane-vectors: unpack, run, pack.  Ops without a batched tile() go vector by vector, so the wave can
// start on the first load while the later ones are still in flight (progressive s_waitcnt vmcnt).
template <bool FAST, class Op, class TI, class TO, int U>
__device__ __forceinline__ void run_vectors(const Op& op, const typename IO<TI, TO>::VI (&v)[U],
                                            typename IO<TI, TO>::VO (&r)[U], const typename Op::Param& p,
                                            const typename Op::Book& b) {
  typedef IO<TI, TO> io;
  if constexpr (HasTile<Op>::value) {
    float in[U * io::N], out[U * io::N];
#pragma unroll
    for (int u = 0; u < U; ++u) io::unpack(v[u], in + u * io::N);
    run<FAST, U * io::N>(op, in, out, p, b);
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = io::pack(out + u * io::N);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float in[io::N], out[io::N];
      io::unpack(v[u], in);
      run<FAST, io::N>(op, in, out, p, b);
      r[u] = io::pack(out);
    }
  }
}

// Tile helpers shared by every launch shape: full tiles (wave-uniform test) run straight-line code.
template <class TI, class TO, int U, int NT>
__device__ __forceinline__ void load_tile(typename IO<TI, TO>::VI (&v)[U], const TI* __restrict__ x, int64_t first,
                                          int64_t limit, bool full /* wave-uniform */) {
  typedef IO<TI, TO> io;
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = io::template load<NT>(x + (first + u * kThreads) * io::N);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (first + u * kThreads < limit) v[u] = io::template load<NT>(x + (first + u * kThreads) * io::N);
  }
}

template <bool FAST, class Op, class TI, class TO, int U, int NT>
__device__ __forceinline__ void finish_tile(const Op& op, const typename Op::Param& p, const typename Op::Book& book,
                                            const typename IO<TI, TO>::VI (&w)[U], TO* __restrict__ y, int64_t first,
                                            int64_t limit, bool full) {
  typedef IO<TI, TO> io;
  if (full) {
    if constexpr (HasTile<Op>::value) {
      typename io::VO r[U];
      run_vectors<FAST, Op, TI, TO, U>(op, w, r, p, book);
#pragma unroll
      for (int u = 0; u < U; ++u) io::template store<NT>(y + (first + u * kThreads) * io::N, r[u]);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {                            // compute + store as each load lands
        typename io::VI one[1] = {w[u]};
        typename io::VO res[1];
        run_vectors<FAST, Op, TI, TO, 1>(op, one, res, p, book);
        io::template store<NT>(y + (first + u * kThreads) * io::N, res[0]);
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (first + u * kThreads < limit) {
        typename io::VI one[1] = {w[u]};
        typename io::VO res[1];
        run_vectors<FAST, Op, TI, TO, 1>(op, one, res, p, book);
        io::template store<NT>(y + (first + u * kThreads) * io::N, res[0]);
      }
  }
}

// One tile whose parameters are wave-uniform, FULL known at compile time so the hot (full) path and
// the guarded (row-end) path never share code: the compiler otherwise merges their tails and ends up
// serialising the loads of the hot path.
template <bool FULL, class Op, class TI, class TO, int U, int NT, class GetParam>
__device__ __forceinline__ void one_tile(const Op& op, float* smem, const TI* __restrict__ xs, TO* __restrict__ ys,
                                         int64_t first, int64_t limit, GetParam get_param) {
  typedef IO<TI, TO> io;
  typename io::VI v[U];
  // Issue the data loads early; the parameter fetch (dependent scalar loads + an IEEE divide) and the table's LDS
  // write + barrier then run in the shadow of the HBM latency.  An op whose table is read through vector loads asks for
  // it BEFORE the data loads (in-order return: see LutTableOp::prefetch).
  typename Op::Book book;
  if constexpr (HasPrefetch<Op>::value) {
    const typename Op::Prefetch pf = op.prefetch();
    __builtin_amdgcn_sched_barrier(0);
    load_tile<TI, TO, U, NT>(v, xs, first, limit, FULL);
    __builtin_amdgcn_sched_barrier(0);
    book = op.commit(pf, smem);
  } else {
    load_tile<TI, TO, U, NT>(v, xs, first, limit, FULL);
    __builtin_amdgcn_sched_barrier(0);                     // keep the loads above the fetch in the schedule
    book = op.setup(smem);
  }
  const typename Op::Param p = get_param();
  const bool fast = __builtin_amdgcn_readfirstlane((int)Op::can_fast(p)) != 0;   // wave-uniform
  if (fast) finish_tile<true, Op, TI, TO, U, NT>(op, p, book, v, ys, first, limit, FULL);
  else finish_tile<false, Op, TI, TO, U, NT>(op, p, book, v, ys, first, limit, FULL);
}

// ------------------------------------------------------------------------------------------
// flat: per-tensor parameters.  Block b owns lane-vectors [b*256*U, (b+1)*256*U); lane accesses
// are 16 B (8 B for a 16-bit input widened to float32), consecutive lanes consecutive addresses,
// U independent loads in flight per lane.
// ------------------------------------------------------------------------------------------
template <class Op, class TI, class TO, int U, int NT>
// (Scalar arguments first: with -amdgpu-kernarg-preload-count they arrive in SGPRs at wave start, so the data loads
// are issued without waiting for a kernel-argument fetch; the op struct behind them is only needed once data lands.)
__global__ __launch_bounds__(kThreads) void flat_kernel(const TI* __restrict__ xs, TO* __restrict__ ys, int64_t n,
                                                        Op op, typename Op::Param p) {
  typedef IO<TI, TO> io;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t nv = n / io::N;
  const int64_t base = (int64_t)blockIdx.x * (kThreads * U) + threadIdx.x;
  auto get_param = [&]() { return p; };
  if (((int64_t)blockIdx.x + 1) * (kThreads * U) <= nv)     // full tile (wave-uniform test): no per-lane guards
    one_tile<true, Op, TI, TO, U, NT>(op, smem, xs, ys, base, nv, get_param);
  else
    one_tile<false, Op, TI, TO, U, NT>(op, smem, xs, ys, base, nv, get_param);
  if (blockIdx.x == 0 && nv * io::N < n) {                  // n % N trailing elements (uniform branch)
    __syncthreads();                                         // everyone is done with the block's table
    const typename Op::Book book = op.setup(smem);           // all threads: setup may synchronise
    const int64_t i = nv * io::N + threadIdx.x;
    if (i < n) ys[i] = narrow_to<TO>(op.template apply<false>((float)xs[i], p, book));
  }
}

// flat, for launches that fill 3/4 ... 1 round of resident blocks (8 per CU; 48-64 MiB of traffic on this chip): the same tile as
// flat_kernel<U = 4> under a different order of waits -- the tile's four loads 128 clocks apart, EVERY load landed before the first
// store is issued, and every store acknowledged before the next lane-vector is touched.  All three together are worth 4-6 % on
// such launches (all waves of the launch are resident at once: the chip reads first and writes after, instead of turning the HBM
// around under the read burst); any one or two of them do nothing, and on launches of other sizes the three cost 3-9 %
// (profiles/EXPERIMENTS.md round 6, profiles/r06/flatx*.log: bfloat16 4096^2 12.12 -> 11.38 us, float32 2048 x 4096 12.15 -> 11.47,
// 7/8 of a round 10.82 -> 10.33, 3/4 9.53 -> 9.14; 5/8 8.06 -> 8.31, 9/8 13.5 -> 14.7, two rounds 22.0 -> 23.4).  launch_flat
// takes it for the affine per-tensor launch inside that window only (tuning key "paced").  Arithmetic: flat_kernel's, call for call.
template <bool FAST, class TI, int NT>
__device__ __forceinline__ void paced_tile(const AffineOp& op, const AffineOp::Param& p, const typename IO<TI, TI>::VI (&v)[4],
                                           TI* __restrict__ ys, int64_t first) {
  typedef IO<TI, TI> io;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    typename io::VI one[1] = {v[u]};
    typename io::VO res[1];
    run_vectors<FAST, AffineOp, TI, TI, 1>(op, one, res, p, NoBook());
    io::template store<NT>(ys + (first + u * kThreads) * io::N, res[0]);
    if (u < 3) { __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_sched_barrier(0); }     // vmcnt(0): the store has completed
  }
}

template <class TI, int NT>
__global__ __launch_bounds__(kThreads) void flat_paced_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, int64_t n,
                                                              AffineOp op, AffineOp::Param p) {
  typedef IO<TI, TI> io;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t nv = n / io::N;
  const int64_t base = (int64_t)blockIdx.x * (kThreads * 4) + threadIdx.x;
  if (((int64_t)blockIdx.x + 1) * (kThreads * 4) <= nv) {
    typename io::VI v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = io::template load<NT>(xs + (base + u * kThreads) * io::N);
      if (u < 3) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(2); __builtin_amdgcn_sched_barrier(0); }
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool fast = __builtin_amdgcn_readfirstlane((int)AffineOp::can_fast(p)) != 0;
    __builtin_amdgcn_s_waitcnt(0x0f70);                       // vmcnt(0): all four loads have landed
    __builtin_amdgcn_sched_barrier(0);
    if (fast) paced_tile<true, TI, NT>(op, p, v, ys, base);
    else paced_tile<false, TI, NT>(op, p, v, ys, base);
  } else {
    auto get_param = [&]() { return p; };
    one_tile<false, AffineOp, TI, TI, 4, NT>(op, smem, xs, ys, base, nv, get_param);
  }
  if (blockIdx.x == 0 && nv * io::N < n) {                    // n % N trailing elements (uniform branch)
    const int64_t i = nv * io::N + threadIdx.x;
    if (i < n) ys[i] = narrow_to<TI>(op.template apply<false>((float)xs[i], p, NoBook()));
  }
}

// flat, one element per lane: used when x or y is not vector-aligned.
template <class Op, class TI, class TO>
__global__ __launch_bounds__(kThreads) void flat_scalar_kernel(Op op, typename Op::Param p, const TI* __restrict__ x,
                                                               TO* __restrict__ y, int64_t n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const typename Op::Book book = op.setup(smem);
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
    y[i] = narrow_to<TO>(op.template apply<false>((float)x[i], p, book));
}

// ------------------------------------------------------------------------------------------
// rows: tensor viewed as [rows = outer*C][innerv] lane-vectors.  blockIdx -> (row, tile); the
// row's channel index is wave-uniform, so fetch() compiles to scalar loads and the parameters sit
// in SGPRs for the whole block.
// ------------------------------------------------------------------------------------------
template <class Op, class TI, class TO, int U, int NT>
__global__ __launch_bounds__(kThreads) void rows_kernel(const TI* __restrict__ xs, TO* __restrict__ ys,
                                                        uint32_t tiles_per_row, uint32_t innerv, uint32_t channels, Op op) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t row = blockIdx.x, tile = 0;
  if (tiles_per_row != 1) {                                // uniform branch: skip the division for 1 tile/row
    row = blockIdx.x / tiles_per_row;
    tile = blockIdx.x - row * tiles_per_row;
  }
  const int64_t rbase = (int64_t)row * innerv;
  const int64_t first = rbase + tile * (kThreads * U) + threadIdx.x;
  auto get_param = [&]() {
    uint32_t c = row;
    if (c >= channels) c = row % channels;                 // uniform; outer == 1 needs no modulo
    return op.fetch(c);
  };
  if ((tile + 1) * (kThreads * U) <= innerv)               // wave-uniform
    one_tile<true, Op, TI, TO, U, NT>(op, smem, xs, ys, first, rbase + innerv, get_param);
  else
    one_tile<false, Op, TI, TO, U, NT>(op, smem, xs, ys, first, rbase + innerv, get_param);
}

// ------------------------------------------------------------------------------------------
// rowsteps: rows that are a whole number of 256-lane-vector STEPS but fewer than four of them (float32 rows of 1024 /
// 2048 / 3072 elements, 16-bit rows of 2048 / 4096 / 6144).  rows_kernel gives such rows one- or two-step tiles -- 16 or
// 32 bytes in flight per lane; here a block takes U consecutive steps of the dense [rows][innerv] storage whatever row they
// fall in (64 bytes in flight per lane, one contiguous 16 KiB read per block), and each step -- which lies inside ONE row --
// gets that row's parameters by scalar loads, as in rows_kernel.
// Where it is taken (launch_channels): only where its grid is ONE round of resident blocks (8 per CU) -- 64 MiB-class launches
// such as 4096 x 4096 bfloat16, 8192 x 2048 bfloat16, 4096 x 2048 / 2048 x 3072 float32: 6-15 % faster than rows_kernel there on
// three boxes, 3-12 % slower at half a round or several rounds (profiles/r05/rowsteps_sched.log, profiles/r04/rowsteps_probe_sustained.log).
// What makes it fast is a property of the COMPILED code: the block's first three loads are issued before anything of the
// parameter fetch (row division, scalar loads, IEEE reciprocal), the fourth once the first has landed (16-bit storage) and all
// stores at the end.  Round 5 tried to WRITE that schedule (tools/experiments/rowsteps_sched/: nine source forms -- explicit
// 3 + 1 and 2 + 2 staggers, stores early or late, compile-time steps per row): every one compiles to something 2-10 % slower,
// because LLVM commons the row division of the full and the guarded path above the loads or sinks the loads to their first
// use (sched_barrier binds one basic block; volatile or asm-pinned loads force s_waitcnt vmcnt(0)).  So the source below is
// the form the window was measured with, and tests/test_abi_exports.py ASSERTS the schedule on the code hipcc generates from
// it: a compiler or source change that moves it fails the suite instead of silently removing the window's reason.
// ------------------------------------------------------------------------------------------
template <class Op, class TI, class TO, int U, int NT>
__global__ __launch_bounds__(kThreads) void rowsteps_kernel(const TI* __restrict__ xs, TO* __restrict__ ys,
                                                            uint32_t steps_per_row, uint32_t total_steps, uint32_t channels,
                                                            Op op) {
  typedef IO<TI, TO> io;
  const uint32_t s0 = blockIdx.x * U;
  const int64_t first = (int64_t)s0 * kThreads + threadIdx.x;
  const bool full = s0 + U <= total_steps;                   // wave-uniform
  typename io::VI v[U];
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = io::template load<NT>(xs + (first + u * kThreads) * io::N);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s0 + u < total_steps) v[u] = io::template load<NT>(xs + (first + u * kThreads) * io::N);
  }
  __builtin_amdgcn_sched_barrier(0);                         // loads first; the parameter fetches run under their latency
  uint32_t row = s0 / steps_per_row, rem = s0 - row * steps_per_row;   // one scalar division per block
  typename Op::Param p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (full || s0 + u < total_steps) p[u] = op.fetch(row >= channels ? row % channels : row);
    if (++rem == steps_per_row) { rem = 0; ++row; }
  }
  const typename Op::Book book = typename Op::Book();
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (full || s0 + u < total_steps) {
      typename io::VI one[1] = {v[u]};
      typename io::VO res[1];
      run_vectors<true, Op, TI, TO, 1>(op, one, res, p[u], book);
      io::template store<NT>(ys + (first + u * kThreads) * io::N, res[0]);
    }
}

// n / d for n < 2^24 (exactly representable in float32) and a wave-uniform d with r = 1.0f / d: the float quotient
// is off by at most one, two integer corrections make it exact -- 7 VALU ops instead of the ~25 of a 32-bit
// unsigned division.  The window kernel's positions are offsets inside one tile (+ one row).
__device__ __forceinline__ uint32_t div_small(uint32_t n, uint32_t d, float r) {
  uint32_t q = (uint32_t)((float)n * r);
  q -= (q * d > n) ? 1u : 0u;
  q += ((q + 1u) * d <= n) ? 1u : 0u;
  return q;
}

// ------------------------------------------------------------------------------------------
// lastaxis: the channel axis is the fastest-varying one (inner == 1: NHWC activations, [tokens, hidden],
// weights quantized along their last axis) and C % N == 0.  A lane's N consecutive elements are N
// consecutive channels.  The tensor is cut into SLABS of k whole rows (k * vc lane-vectors, contiguous in
// memory, >= 2048 of them); grid (x, y, z) = (256-lane pieces of a slab, groups of U slabs): the block's lanes
// are laid over one piece of the slab and step DOWN the tensor a slab at a time, so every lane keeps the same
// N channels: their parameters are fetched (16-byte table loads) and inverted once per lane, not per element.
// Round 6 (profiles/r06/chanlast2_*.log, tools/experiments/chanlast2/): the lane's offset inside the slab IS its
// memory offset, so nothing stands in front of the data loads -- the column, which only the parameter fetch needs,
// is divided out under their latency (the round-5 form ran two integer divisions and four exec-masked loads with
// 64-bit multiplies first); N reciprocals per lane by recip_exact (5 instead of 11 VALU instructions each); a launch
// without zero points keeps its clamp bounds in scalar registers; every result first, then the stores back to back.
// Together 3-7 % on 10 of 10 channel-last shapes x 3 storage types, 11 % with zero points; two rows per lane (four
// with zero points: more parameter work per lane to spread) measured best -- looping blocks and a software pipeline
// over the slabs lost 3-10 % everywhere, rotating the pieces over the XCDs or a group-major grid bought nothing.
// ------------------------------------------------------------------------------------------
template <class Op, class TI, class TO, int U, int NT, bool ZP, bool FULL>
__device__ __forceinline__ void lastaxis_body(const Op& op, float* smem, const TI* __restrict__ xs, TO* __restrict__ ys,
                                              uint64_t rows_left, uint32_t vc, uint32_t k, float rvc, uint32_t g,
                                              uint32_t slab) {
  typedef IO<TI, TO> io;
  if constexpr (std::is_same<typename Op::Book, NoBook>::value) {
    if (g >= slab) return;                                   // idle lanes of the slab's last piece (ops with a table stay for its barrier)
  }
  // lanes of slab u that hold a row of the tensor (wave-uniform bound; the last group of a launch may be short)
  uint32_t lim[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (FULL) lim[u] = slab;
    else {
      const uint64_t r = rows_left > (uint64_t)u * k ? rows_left - (uint64_t)u * k : 0;
      lim[u] = (r >= k ? k : (uint32_t)r) * vc;
    }
  }
  typename io::VI v[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if ((FULL && std::is_same<typename Op::Book, NoBook>::value) || g < lim[u])
      v[u] = io::template load<NT>(xs + ((size_t)u * slab + g) * io::N);
  __builtin_amdgcn_sched_barrier(0);                         // the loads first; everything below runs under their latency
  const typename Op::Book book = op.setup(smem);             // every thread of the block (a LUT op's table goes to LDS)
  if (g >= slab) return;
  const uint32_t col = g - div_small(g, vc, rvc) * vc;       // slab < 2^24 lane-vectors (launch_channels)
  typename Op::Param p[io::N];
  if constexpr (std::is_base_of<AffineOp, Op>::value) op.template fetch_lane<io::N, ZP>(col * io::N, p);
  else op.template fetch_vec<io::N>(col * io::N, p);
  typename io::VO r[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (g < lim[u]) {
      float in[io::N], out[io::N];
      io::unpack(v[u], in);
#pragma unroll
      for (int j = 0; j < io::N; ++j) out[j] = op.template apply<false>(in[j], p[j], book);
      r[u] = io::pack(out);
    }
  __builtin_amdgcn_sched_barrier(0);                         // every result, then the stores back to back
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (g < lim[u]) io::template store<NT>(ys + ((size_t)u * slab + g) * io::N, r[u]);
}

template <class Op, class TI, class TO, int U, int NT, bool ZP>
__global__ __launch_bounds__(kThreads) void lastaxis_kernel(const TI* __restrict__ xs, TO* __restrict__ ys, uint64_t rows,
                                                            uint32_t vc, uint32_t k, float rvc, uint32_t groups, Op op) {
  typedef IO<TI, TO> io;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const uint32_t slab = k * vc;
  const uint32_t g = blockIdx.x * kThreads + threadIdx.x;    // lane-vector inside the slab == its offset in memory
  const uint32_t group = blockIdx.y + blockIdx.z * gridDim.y;
  if (group >= groups) return;                               // (uniform: the y x z grid may overshoot)
  const uint64_t row0 = (uint64_t)group * (uint32_t)(U * k);
  const uint64_t rows_left = rows - row0;
  const TI* x0 = xs + row0 * vc * io::N;
  TO* y0 = ys + row0 * vc * io::N;
  if (rows_left >= (uint64_t)U * k)
    lastaxis_body<Op, TI, TO, U, NT, ZP, true>(op, smem, x0, y0, rows_left, vc, k, rvc, g, slab);
  else
    lastaxis_body<Op, TI, TO, U, NT, ZP, false>(op, smem, x0, y0, rows_left, vc, k, rvc, g, slab);
}

// ------------------------------------------------------------------------------------------
// shortrows (affine quantizers): per-channel rows shorter than a tile that are at least one lane-vector long -- 16-bit
// tensors whose rows are not whole 8-element vectors (1020- or 4100-wide) or are shorter than 64 elements, float32 rows of
// 4 ... 31 elements.  Block b owns elements [b * TILE, (b + 1) * TILE) of the dense [rows][inner] storage; a lane-vector
// lies in one row or crosses exactly one row boundary.  No LDS window, no block barrier: after the tile's data loads every
// lane reads its own row's scale (and, where vectors can cross, the next row's) from the L1 / L2-resident tables -- all
// table reads issued back to back -- and inverts it with recip_exact (wave-uniform fallback to the IEEE division).
// (Why it also beat the per-tensor kernel on the same bytes -- 11.75 vs 12.2 us for 64 MiB in 16-bit storage -- was found at the
// end of round 6: its waves wait for ALL their loads before the first store -- the scale reads sit behind the data loads -- and
// some 200 instructions of row arithmetic lie between two stores; see flat_paced_kernel and the `paced` argument below.)
// Measured against window_kernel (tools/experiments/chanlast2/, profiles/r06/chanlast2_a.log): bfloat16 16384 x 1020
// 13.7 -> 12.5 us, 4096 x 4100 13.3 -> 12.3, 1048576 x 16 13.5 -> 13.0, float32 1048576 x 16 23.9 -> 22.1; whole-vector
// 16-bit rows of 64 ... 2040 elements are equal or slower and stay with the window.  Reading the parameters BEFORE the data
// (vector loads return in order) was slower in every case: the data loads must not wait for the row arithmetic.
// ------------------------------------------------------------------------------------------
template <class TI, int U, int NT, bool ZP, bool WHOLE>
__global__ __launch_bounds__(kThreads) void shortrows_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, uint32_t n,
                                                             uint32_t inner, uint32_t channels, float r_inner, float r_channels,
                                                             uint32_t shift /* log2(inner), or 32 */,
                                                             uint32_t paced /* launches of 3/4 ... 1 round: one store in flight per wave */,
                                                             AffineOp op) {
  typedef IO<TI, TI> io;
  constexpr uint32_t N = io::N;
  constexpr uint32_t TILE = kThreads * U * N;
  const uint32_t e0 = blockIdx.x * TILE;
  const uint32_t left = n - e0;
  const uint32_t count = left < TILE ? left : TILE;
  const bool full = left >= TILE;                                    // wave-uniform
  typename io::VI v[U];
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = io::template load<NT>(xs + e0 + (u * kThreads + threadIdx.x) * N);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((u * kThreads + threadIdx.x) * N + N <= count) v[u] = io::template load<NT>(xs + e0 + (u * kThreads + threadIdx.x) * N);
  }
  __builtin_amdgcn_sched_barrier(0);
  uint32_t row0, rem0;
  if (shift < 32) { row0 = e0 >> shift; rem0 = e0 & (inner - 1); }
  else { row0 = e0 / inner; rem0 = e0 - row0 * inner; }
  const uint32_t c0 = row0 < channels ? row0 : row0 % channels;
  const uint32_t nrows = (rem0 + count - 1) / inner + 1;
  const bool wraps = c0 + nrows > channels;                          // some row of the tile starts a new outer slice
  uint32_t split[U];
  float sa[U], sb[U], za[U], zb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    const uint32_t pos = rem0 + ((full || off < count) ? off : 0);   // < 2^24 (launch_channels): exact through float32
    const uint32_t lrow = shift < 32 ? pos >> shift : div_small(pos, inner, r_inner);
    uint32_t c = c0 + lrow;
    if (wraps) c -= div_small(c, channels, r_channels) * channels;
    split[u] = inner - (pos - lrow * inner);                         // elements of the vector that belong to row c (>= N: all)
    sa[u] = op.scales[c];
    if (ZP) za[u] = (float)op.zps[c];
    if (!WHOLE) {
      const uint32_t cn = c + 1 == channels ? 0 : c + 1;
      sb[u] = op.scales[cn];
      if (ZP) zb[u] = (float)op.zps[cn];
    }
  }
  bool in_range = recip_all_in_range(sa);
  if (!WHOLE) in_range = in_range && recip_all_in_range(sb);
  const bool exact = __builtin_amdgcn_ballot_w64(!in_range) == 0;    // wave-uniform
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    if (!full && off >= count) continue;
    AffineOp::Param pa;
    pa.s = sa[u]; pa.inv = exact ? recip_exact(sa[u]) : 1.0f / sa[u]; pa.zf = ZP ? za[u] : 0.0f;
    if (full || off + N <= count) {
      float in[N], out[N];
      io::unpack(v[u], in);
      // Wave-uniform test: a wave with a crossing vector runs the per-element selection for ALL its lanes (lanes whose vector
      // lies in one row select their own set throughout) instead of both forms one after the other -- 16384 x 1020 bfloat16
      // 13.0 -> 12.67 us, 65536 x 252 -7.6 %, 131072 x 100 -5.8 % (profiles/r06/ab_sr1.log).  Taking the next row's parameters
      // from the neighbouring lane by a shuffle instead of a table read was 0.5-5 % slower (ab_sr2.log).
      if (WHOLE || __builtin_amdgcn_ballot_w64(split[u] < N) == 0) {
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) out[j] = op.apply(in[j], pa, NoBook());
      } else {
        AffineOp::Param pb;
        pb.s = sb[u]; pb.inv = exact ? recip_exact(sb[u]) : 1.0f / sb[u]; pb.zf = ZP ? zb[u] : 0.0f;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) out[j] = op.apply(in[j], j < split[u] ? pa : pb, NoBook());
      }
      io::template store<NT>(ys + e0 + off, io::pack(out));
      // PACED (a kernel argument, wave-uniform): every store completed before the next lane-vector is touched.  The kernel waits for
      // all its loads before the first reciprocal anyway (the scale reads are issued behind the data loads and return in order);
      // with this, a launch whose blocks are all resident at once reads first and writes after (flat_paced_kernel's note):
      // float32 2048 x 4096 12.4 -> 11.4 us, 8192 x 1024 12.4 -> 11.4, bfloat16 4096^2 11.7 -> 11.5 (profiles/r06/chanlast2_oneround.log)
      if (paced && u + 1 < U) __builtin_amdgcn_s_waitcnt(0x0f70);
    } else {
      // the tensor's last, partial lane-vector: element by element
      uint32_t rem = inner - split[u], c = c0 + (shift < 32 ? (rem0 + off) >> shift : div_small(rem0 + off, inner, r_inner));
      if (wraps) c -= div_small(c, channels, r_channels) * channels;
      for (uint32_t j = 0; j < N && off + j < count; ++j) {
        ys[e0 + off + j] = narrow_to<TI>(op.apply((float)xs[e0 + off + j], op.fetch(c), NoBook()));
        if (++rem == inner) { rem = 0; if (++c == channels) c = 0; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// window: block b owns elements [b*TILE, (b+1)*TILE), TILE = 256*U*V (V = lane-vector width, or
// 1 for unaligned tensors).  The tile touches rows row0 .. row0+nrows-1 of the [outer*C][inner]
// view; their parameters are staged in LDS (structure-of-arrays, so lanes that read different rows
// hit different banks) and a lane finds its row with ONE 32-bit division per access, then walks
// row boundaries incrementally.  If the whole table is smaller than the tile's row span
// (channel-last layouts: inner == 1, C small) the whole table is staged instead and indexed
// modulo C.
// ------------------------------------------------------------------------------------------
template <class Op, class TI, class TO, int U, bool VEC, int NT, typename IdxT>
__global__ __launch_bounds__(kThreads) void window_kernel(const TI* __restrict__ xs, TO* __restrict__ ys, IdxT n,
                                                          uint32_t inner, uint32_t channels,
                                                          uint32_t stride /* LDS entries per param word */, Op op) {
  typedef IO<TI, TO> io;
  constexpr uint32_t V = VEC ? io::N : 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr uint32_t TILE = kThreads * U * V;
  const typename Op::Book book = op.setup(smem);
  float* tab = smem + op.book_words();

  const IdxT e0 = (IdxT)blockIdx.x * TILE;
  const IdxT left = n - e0;
  const uint32_t count = left < (IdxT)TILE ? (uint32_t)left : TILE;

  // data loads first: the window staging below (division, table reads, barrier) runs under their latency
  typename io::VI v[U];
  if (VEC) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t off = (u * kThreads + threadIdx.x) * V;
      if (off + V <= count) v[u] = io::template load<NT>(xs + e0 + off);
    }
  }

  const IdxT row0 = e0 / inner;                            // uniform, once per block
  const uint32_t rem0 = (uint32_t)(e0 - row0 * inner);
  const uint32_t nrows = (rem0 + count - 1) / inner + 1;
  const bool whole = channels <= nrows;
  const bool same_row = VEC && (inner % V) == 0;            // a lane-vector never straddles two rows
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

  // positions inside the tile are small: exact division through one float multiply when they fit 24 bits (uniform)
  const bool small = (uint64_t)rem0 + TILE < (1u << 24) && (uint64_t)c0 + nrows < (1u << 24);
  const float r_inner = 1.0f / (float)inner, r_channels = 1.0f / (float)channels;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * V;
    if (off >= count) continue;
    const uint32_t pos = rem0 + off;
    uint32_t lrow = small ? div_small(pos, inner, r_inner) : pos / inner;
    uint32_t lrem = pos - lrow * inner;
    uint32_t li = lrow;
    if (whole) {
      const uint32_t cc = c0 + lrow;
      li = small ? cc - div_small(cc, channels, r_channels) * channels : cc % channels;
    }
    if (VEC && off + V <= count) {
      float in[V], out[V];
      io::unpack(v[u], in);
      if (same_row || lrem + V <= inner) {                  // the lane-vector lies inside one row (always, when rows
        const typename Op::Param p = Op::get(tab, li, stride);             // are whole vectors): one parameter set
        bool fast = false;
        if constexpr (Op::kHeavy)          // LUT ops: the exact reciprocal division works per lane as well (r, ds are in the
          fast = __builtin_amdgcn_ballot_w64(!Op::can_fast(p)) == 0;       // window); taken when every active lane qualifies
        if (fast) run<true, (int)V>(op, in, out, p, book);
        else run<false, (int)V>(op, in, out, p, book);
      } else if (inner >= V) {
        // the lane-vector crosses exactly ONE row boundary (rows at least a vector long that are not whole vectors,
        // e.g. 1020 bfloat16 elements): both rows' parameter sets, chosen per element -- straight-line code instead of
        // a per-element walk that the other 63 lanes of the wave wait for (bf16 16384 x 1020: 15.0 -> 13.6 us)
        const uint32_t split = inner - lrem;                  // elements of the vector that belong to the first row
        uint32_t li2 = li + 1;
        if (whole && li2 == channels) li2 = 0;
        const typename Op::Param pa = Op::get(tab, li, stride), pb = Op::get(tab, li2, stride);
#pragma unroll
        for (uint32_t j = 0; j < V; ++j) out[j] = op.template apply<false>(in[j], j < split ? pa : pb, book);
      } else
#pragma unroll
      for (uint32_t j = 0; j < V; ++j) {
        out[j] = op.template apply<false>(in[j], Op::get(tab, li, stride), book);
        if (++lrem == inner) {
          lrem = 0;
          ++li;
          if (whole && li == channels) li = 0;
        }
      }
      io::template store<NT>(ys + e0 + off, io::pack(out));
    } else {
      for (uint32_t j = 0; j < V && off + j < count; ++j) {
        ys[e0 + off + j] = narrow_to<TO>(op.template apply<false>((float)xs[e0 + off + j], Op::get(tab, li, stride), book));
        if (++lrem == inner) {
          lrem = 0;
          ++li;
          if (whole && li == channels) li = 0;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------
// shared state and helpers, defined in mctq_misc.hip
extern thread_local char g_err[256];
extern int g_nt;             // 1 non-temporal loads and stores (+7% on the cold 4096x4096 stream), 2 nt loads + cached stores
extern int64_t g_cached_store_max_bytes;   // with g_nt == 1: outputs up to this size use mode 2 (0 = never)
inline int nt_mode(int64_t out_bytes) {
  if (g_nt != 1) return g_nt;
  return (g_cached_store_max_bytes > 0 && out_bytes <= g_cached_store_max_bytes) ? 2 : 1;
}
extern int g_unroll;
// what the calling thread launched last (mctq_last_launch(): lets a benchmark tie its counters to a kernel variant)
struct LaunchNote { const char* shape; const char* op; int unroll, nt, in_bytes, out_bytes; int64_t count; };
extern thread_local LaunchNote g_note;
// MCTQ_LAUNCH_LOG=<file> in the environment when the library is loaded: every launch VARIANT (the text of
// mctq_last_launch()) is appended to that file the first time the process takes it -- the evidence the set of
// instantiated kernels is pruned against (profiles/r05/launch_variants_all.log: the file itself, sorted).  Off: one predictable
// branch per launch.
extern int g_launch_log;
void log_launch();
template <class Op, class TI, class TO>
inline void note(const char* shape, int unroll, int nt) {
  g_note.shape = shape; g_note.op = Op::kName; g_note.unroll = unroll; g_note.nt = nt;
  g_note.in_bytes = (int)sizeof(TI); g_note.out_bytes = (int)sizeof(TO);
  ++g_note.count;
  if (g_launch_log) log_launch();
}
extern int g_paced;          // launches of 3/4 ... 1 round under the paced order of waits (flat_paced_kernel per tensor; shortrows_kernel's
                             // `paced` argument per channel): 0 never, 1 (default) inside that window, 2 whenever the kernel is taken
extern int g_shortrows;      // rows shorter than a tile through shortrows_kernel: 0 never, 1 (default) where it measured faster, 2 whenever eligible
extern int g_rowsteps;       // short whole-step rows: 0 rows_kernel, 1 rowsteps_kernel, 2 (default) rowsteps_kernel when its grid is one round
extern int g_heavy_unroll;   // 0 = automatic
int fail_arg(const char* msg);
int check_launch(const char* what);
int cu_count();
template <class TI, class TO>
static bool vec_aligned(const void* x, const void* y) {
  typedef IO<TI, TO> io;
  return ((uintptr_t)x % (io::N * sizeof(TI))) == 0 && ((uintptr_t)y % (io::N * sizeof(TO))) == 0;
}

// Launch through hipModuleLaunchKernel with the kernel's hipFunction_t resolved once per (kernel, device): skips
// the host-stub -> device-function lookup every hipLaunchKernel call makes (2.4 vs 2.8 us per launch here,
// profiles/r02/launch_probe.log).  Used for the launch-bound per-tensor path; semantics (stream order, capture,
// error reporting through hipGetLastError) are those of hipLaunchKernelGGL.
// The kernel is a template ARGUMENT (not a function argument): one cache per kernel instantiation -- kernels that
// differ only in U / NT have the same function TYPE.
template <auto kernel, class... Args>
inline void launch_resolved(dim3 grid, dim3 block, size_t shmem, hipStream_t st, Args... args) {
  constexpr int kMaxDevices = 32;
  static hipFunction_t fns[kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices) {
    hipFunction_t fn = fns[dev];
    if (!fn && hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(kernel)) == hipSuccess) fns[dev] = fn;
    if (fn) {
      void* params[] = {(void*)&args...};
      (void)hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)shmem, st, params, nullptr);
      return;
    }
    (void)hipGetLastError();
  }
  hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
}

// Launch-variant dispatch.  NT_ is the runtime cache-policy mode (1: non-temporal loads and stores, 2: non-temporal
// loads + cached stores; see IO::load/store).  An op with kFixedU != 0 is built in ONE variant (U = kFixedU, NT = 1).
#define MCTQ_WITH_MODE(MODE_, ...)                                                          \
  do {                                                                                      \
    if ((MODE_) == 2) { constexpr int NT = 2; __VA_ARGS__; }                                \
    else { constexpr int NT = 1; __VA_ARGS__; }                                             \
  } while (0)

// single-variant ops: NT = 1 whatever the mode
#define MCTQ_WITH_OP_MODE(MODE_, ...)                                                       \
  do {                                                                                      \
    if constexpr (Op::kFixedU != 0) { constexpr int NT = 1; __VA_ARGS__; }                  \
    else MCTQ_WITH_MODE(MODE_, __VA_ARGS__);                                                \
  } while (0)

#define MCTQ_DISPATCH_U_NT(U_, NT_, ...)                                                    \
  do {                                                                                      \
    if constexpr (Op::kFixedU != 0) {                                                       \
      constexpr int U = Op::kFixedU; constexpr int NT = 1; __VA_ARGS__;                     \
    } else if constexpr (std::is_same<TI, float>::value && std::is_same<TO, float>::value) { \
      switch (U_) {                                                                         \
        case 1: { constexpr int U = 1; MCTQ_WITH_MODE(NT_, __VA_ARGS__); } break;           \
        case 2: { constexpr int U = 2; MCTQ_WITH_MODE(NT_, __VA_ARGS__); } break;           \
        default: { constexpr int U = 4; MCTQ_WITH_MODE(NT_, __VA_ARGS__); } break;          \
      }                                                                                     \
    } else {                                                                                \
      switch (U_) {       /* 16-bit storage: 8 elements per lane-vector, so rows are half as many vectors */ \
        case 1: case 2: { constexpr int U = 2; MCTQ_WITH_MODE(NT_, __VA_ARGS__); } break;   \
        default: { constexpr int U = 4; MCTQ_WITH_MODE(NT_, __VA_ARGS__); } break;          \
      }                                                                                     \
    }                                                                                       \
  } while (0)

#define MCTQ_DISPATCH_HEAVY(U_, NT_, ...)                                                   \
  do {                                                                                      \
    if constexpr (Op::kFixedU != 0) {                                                       \
      constexpr int U = Op::kFixedU; constexpr int NT = 1; __VA_ARGS__;                     \
    } else {                                                                                \
      switch (U_) {                                                                         \
        case 1: { constexpr int U = 1; MCTQ_WITH_MODE(NT_, __VA_ARGS__); } break;           \
        case 2: { constexpr int U = 2; MCTQ_WITH_MODE(NT_, __VA_ARGS__); } break;           \
        default: { constexpr int U = 4; MCTQ_WITH_MODE(NT_, __VA_ARGS__); } break;          \
      }                                                                                     \
    }                                                                                       \
  } while (0)

// does the launch carry a zero-point table?  (affine ops: NULL = symmetric quantizers; everything else: not a question)
// Whole-vector per-channel rows of a launch that fills 3/4 ... 1 round of shortrows_kernel's tiles (8 blocks of four waves per CU):
// the window in which paced stores pay (flat_paced_kernel's note; tuning key "paced").
template <class TI>
static bool paced_rows_window(int64_t n, int64_t inner, int64_t channels) {
  constexpr int64_t N = IO<TI, TI>::N, TILE = (int64_t)kThreads * 4 * N;
  if (inner % N != 0 || inner < N || channels <= 1) return false;
  if (n >= (1ll << 32) - TILE || inner + TILE >= (1 << 24) || channels + TILE >= (1 << 24)) return false;
  const int64_t blocks = (n + TILE - 1) / TILE, round = 8LL * cu_count();
  return blocks * 4 >= round * 3 && blocks <= round;
}

template <class Op>
static bool has_zero_points(const Op& op) {
  if constexpr (std::is_base_of<AffineOp, Op>::value) return op.zps != nullptr;
  else return true;
}

template <class TI, class TO, class Op>
static int launch_flat(const Op& op, const typename Op::Param& p, const void* xv, void* yv, int64_t n,
                       size_t book_bytes, hipStream_t st) {
  typedef IO<TI, TO> io;
  if (n == 0) return 0;
  const TI* x = static_cast<const TI*>(xv);
  TO* y = static_cast<TO*>(yv);
  if (!vec_aligned<TI, TO>(x, y)) {
    int64_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > (int64_t)cu_count() * 32) blocks = (int64_t)cu_count() * 32;
    hipLaunchKernelGGL((flat_scalar_kernel<Op, TI, TO>), dim3((unsigned)blocks), dim3(kThreads), book_bytes, st,
                       op, p, x, y, n);
    note<Op, TI, TO>("flat_scalar_kernel", 1, 0);
    return check_launch("flat scalar launch");
  }
  const int64_t nv = n / io::N;
  if constexpr (Op::kHeavy) {
    MCTQ_DISPATCH_HEAVY(g_heavy_unroll ? g_heavy_unroll : 4, nt_mode(n * (int64_t)sizeof(TO)), {
      int64_t blocks = (nv + kThreads * U - 1) / (kThreads * U);
      if (blocks == 0) blocks = 1;
      if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
      hipLaunchKernelGGL((flat_kernel<Op, TI, TO, U, NT>), dim3((unsigned)blocks), dim3(kThreads), book_bytes, st,
                         x, y, n, op, p);
      note<Op, TI, TO>("flat_kernel", U, NT);
    });
    return check_launch("flat launch");
  } else {
    // small tensors: fewer lane-vectors per lane so that the grid still covers the chip (>= 2 blocks per CU)
    int u_sel = g_unroll;
    while (u_sel > 1 && nv < (int64_t)kThreads * u_sel * 2 * cu_count()) u_sel >>= 1;
    if constexpr (std::is_same<Op, AffineOp>::value && std::is_same<TI, TO>::value) {
      // 3/4 ... 1 round of resident blocks (8 blocks of four waves per CU): flat_paced_kernel (see there)
      const int64_t blocks4 = (nv + kThreads * 4 - 1) / (kThreads * 4), round = 8 * (int64_t)cu_count();
      const bool window = blocks4 * 4 >= round * 3 && blocks4 <= round;
      if (u_sel == 4 && nv >= kThreads * 4 && blocks4 <= 0x7fffffffLL && ((g_paced == 1 && window) || g_paced == 2)) {
        MCTQ_WITH_MODE(nt_mode(n * (int64_t)sizeof(TO)), {
          launch_resolved<flat_paced_kernel<TI, NT>>(dim3((unsigned)blocks4), dim3(kThreads), book_bytes, st, x, y, n, op, p);
          note<Op, TI, TO>("flat_paced_kernel", 4, NT);
        });
        return check_launch("flat paced launch");
      }
    }
    MCTQ_DISPATCH_U_NT(u_sel, nt_mode(n * (int64_t)sizeof(TO)), {
      int64_t blocks = (nv + kThreads * U - 1) / (kThreads * U);
      if (blocks == 0) blocks = 1;
      if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
      launch_resolved<flat_kernel<Op, TI, TO, U, NT>>(dim3((unsigned)blocks), dim3(kThreads), book_bytes, st,
                                                       x, y, n, op, p);
      note<Op, TI, TO>("flat_kernel", U, NT);
    });
    return check_launch("flat launch");
  }
}

template <class TI, class TO, class Op>
static int launch_channels(const Op& op, const void* xv, void* yv, int64_t outer, int64_t channels, int64_t inner,
                           size_t book_bytes, hipStream_t st) {
  typedef IO<TI, TO> io;
  const int64_t n = outer * channels * inner;
  if (n == 0) return 0;
  const TI* x = static_cast<const TI*>(xv);
  TO* y = static_cast<TO*>(yv);
  const int64_t rows = outer * channels;
  const bool vec_ok = vec_aligned<TI, TO>(x, y);

  // shortrows shape (affine quantizers only): rows of at least one lane-vector, tile offsets exact in float32.
  // Tuning key "shortrows": 0 never, 1 (default) where it measured faster, 2 whenever eligible.  Measured (profiles/r06/
  // ab_shortrows3.log, sweep_routes_b.log; A / B in one process): 16-bit tensors -- 4-9 % faster than window_kernel on EVERY
  // short- or ragged-row shape (16384 x 1024 12.6 -> 11.8 us, 1048576 x 16 13.6 -> 12.3, 50257 x 768 28.1 -> 26.8, 16384 x 1020
  // 13.6 -> 12.9), and 2-14 % faster than rows_kernel / rowsteps_kernel on long rows while the launch is at most ONE round of
  // resident blocks and at least 7/8 of one (4096^2 12.0 -> 11.75, 256 x 65536 12.45 -> 11.8) but 3-6 % slower on several
  // rounds (8192^2 42.0 -> 43.2) and 4-28 % slower on fractions of a round (2048 x 4096 6.9 -> 7.9); float32 -- equal to the
  // gather launch and to rows_kernel within 1 % everywhere, so only the rows the gather launch does not take (4 ... 31
  // elements: 23.9 -> 22.2 us) come here.
  if constexpr (std::is_same<Op, AffineOp>::value && std::is_same<TI, TO>::value) {
    constexpr int64_t TILE = (int64_t)kThreads * 4 * io::N;
    // (channels > 1: a per-tensor launch through this kernel is equal at 64 MiB and 13-23 % slower below, profiles/r06/ab_tqp.log)
    const bool eligible = vec_ok && inner >= io::N && channels > 1 && n < (1ll << 32) - TILE && inner + TILE < (1 << 24) &&
                          channels + TILE < (1 << 24);
    const bool long_rows = inner % io::N == 0 && inner / io::N >= kThreads;
    // (long rows: only a launch of 7/8 ... 1 round -- below that rows_kernel / rowsteps_kernel win by 4-28 %: 3072 x 4096 +4 %,
    // 2048 x 4096 +16 %, 1024 x 4096 +28 %, 64 x 65536 +23 %, profiles/r06/ab_small.log; short and ragged rows at every size:
    // 4096 x 256 -8 %, 65536 x 16 -14 %.  A FORCED rowsteps_kernel -- tuning key "rowsteps" = 1 -- keeps its long rows.)
    const int64_t sr_blocks = (n + TILE - 1) / TILE, round = 8LL * cu_count();
    const bool one_round = sr_blocks <= round && sr_blocks * 8 > round * 7;
    // float32 rows of a symmetric launch inside the paced window come here too, long or short (rows_kernel / rowsteps_kernel / the
    // gather launch 12.3-12.5 us, this kernel with paced stores 11.3-11.4: profiles/r06/chanlast2_oneround.log, ab_paced_rows.log)
    // (float32 only: on the 16-bit instances the same flag costs 2-4 %, ab_paced_rows.log; rows of one to three whole steps --
    // rowsteps_kernel's -- only from 7/8 of a round: 4096 x 2048 -3 %, 2048 x 3072 at 3/4 of a round +2 %)
    const bool window = sizeof(TI) == 4 && paced_rows_window<TI>(n, inner, channels) && !op.zps;
    const bool paced = sizeof(TI) == 4 && (g_paced == 2 || (g_paced == 1 && window));
    const bool steps_row = long_rows && (inner / io::N) % kThreads == 0 && inner / io::N <= 3 * kThreads;
    const bool f32_window = g_paced == 1 && window && (!long_rows || g_rowsteps != 1) && (!steps_row || sr_blocks * 8 > round * 7);
    const bool rule = sizeof(TI) == 2 ? (!long_rows || (one_round && g_rowsteps != 1)) : ((!long_rows && inner < 32) || f32_window);
    if (eligible && (g_shortrows == 2 || (g_shortrows == 1 && rule))) {
      const bool whole = inner % io::N == 0;
      uint32_t shift = 32;
      if ((inner & (inner - 1)) == 0) { shift = 0; while ((1ll << shift) < inner) ++shift; }
      const float ri = 1.0f / (float)inner, rc = 1.0f / (float)channels;
      const unsigned grid = (unsigned)((n + TILE - 1) / TILE);
#define MCTQ_SHORTROWS(NT_, ZP_, W_)                                                                                   \
      hipLaunchKernelGGL((shortrows_kernel<TI, 4, NT_, ZP_, W_>), dim3(grid), dim3(kThreads), 0, st, x, y, (uint32_t)n,  \
                         (uint32_t)inner, (uint32_t)channels, ri, rc, shift, paced ? 1u : 0u, op)
      MCTQ_WITH_MODE(nt_mode(n * (int64_t)sizeof(TO)), {
        if (op.zps) { if (whole) MCTQ_SHORTROWS(NT, true, true); else MCTQ_SHORTROWS(NT, true, false); }
        else { if (whole) MCTQ_SHORTROWS(NT, false, true); else MCTQ_SHORTROWS(NT, false, false); }
        note<Op, TI, TO>("shortrows_kernel", 4, NT);
      });
#undef MCTQ_SHORTROWS
      return check_launch("shortrows launch");
    }
  }

  // rows shape: long, vector-divisible rows.
  if (vec_ok && (inner % io::N) == 0 && inner / io::N >= kThreads && channels <= 0xffffffffLL && rows <= 0xffffffffLL) {
    const int64_t innerv = inner / io::N;
    if constexpr (Op::kHeavy) {
      // Lane-vectors per lane per tile: the widest of {4, 2, 1} whose idle lanes in the last tile
      // of a row stay under 1/8 (more bytes in flight per lane), unless a tuning override is set.
      int u_sel = 1;
      for (int u = 4; u >= 1; u >>= 1) {
        const int64_t per_u = (int64_t)kThreads * u;
        const int64_t cap = ((innerv + per_u - 1) / per_u) * per_u;
        if ((cap - innerv) * 8 <= cap) { u_sel = u; break; }
      }
      if (g_heavy_unroll) u_sel = g_heavy_unroll;
      if (Op::kFixedU != 0) u_sel = Op::kFixedU;          // the only variant built for these ops
      const int64_t per = (int64_t)kThreads * u_sel;
      const int64_t tiles = (innerv + per - 1) / per;
      const int64_t total = rows * tiles;
      if (total <= 0x7fffffffLL && innerv <= 0x7fffffffLL) {
        MCTQ_DISPATCH_HEAVY(u_sel, nt_mode(n * (int64_t)sizeof(TO)), {
          hipLaunchKernelGGL((rows_kernel<Op, TI, TO, U, NT>), dim3((unsigned)total), dim3(kThreads), book_bytes,
                             st, x, y, (uint32_t)tiles, (uint32_t)innerv, (uint32_t)channels, op);
          note<Op, TI, TO>("rows_kernel", U, NT);
        });
        return check_launch("rows launch");
      }
    } else {
      // Largest U <= tuned unroll that wastes the fewest lanes in the last tile of a row.
      int best_u = 1;
      int64_t best_waste = -1;
      for (int u = 1; u <= g_unroll; u <<= 1) {
        const int64_t per = (int64_t)kThreads * u;
        const int64_t tiles = (innerv + per - 1) / per;
        const int64_t waste = tiles * per - innerv;
        if (best_waste < 0 || waste <= best_waste) { best_waste = waste; best_u = u; }
      }
      if constexpr (std::is_same<Op, AffineOp>::value) {
        // rows of 1 .. 3 whole steps: four steps per block across row boundaries (rowsteps_kernel)
        const int64_t total_steps = rows * (innerv / kThreads);
        // Measured (profiles/r04/rowsteps_probe_sustained.log, 30 shapes x 3 storage types): four steps per block beat
        // rows_kernel's one- / two-step tiles by 5-10 % exactly when the four-step grid is ONE round of resident blocks
        // (8 per CU: 64 MiB launches such as 4096 x 4096 bfloat16 or 4096 x 2048 float32) and lose 3-8 % otherwise
        // (half a round, or several rounds).  Mode 2 (default) takes it in that window only.
        const int64_t rs_blocks = (total_steps + 3) / 4, round = 8LL * cu_count();
        const bool one_round = rs_blocks <= round && rs_blocks * 4 >= round * 3;
        if ((g_rowsteps == 1 || (g_rowsteps == 2 && one_round)) && best_u < 4 && g_unroll >= 4 && innerv % kThreads == 0 &&
            total_steps <= 0xffffffffLL) {
          MCTQ_WITH_MODE(nt_mode(n * (int64_t)sizeof(TO)), {
            constexpr int U = 4;
            hipLaunchKernelGGL((rowsteps_kernel<Op, TI, TO, U, NT>), dim3((unsigned)((total_steps + U - 1) / U)), dim3(kThreads),
                               0, st, x, y, (uint32_t)(innerv / kThreads), (uint32_t)total_steps, (uint32_t)channels, op);
            note<Op, TI, TO>("rowsteps_kernel", U, NT);
          });
          return check_launch("rowsteps launch");
        }
      }
      if (!(std::is_same<TI, float>::value && std::is_same<TO, float>::value) && best_u < 2) best_u = 2;   // built: U = 2, 4
      if (Op::kFixedU != 0) best_u = Op::kFixedU;          // single-variant ops
      const int64_t per = (int64_t)kThreads * best_u;
      const int64_t tiles = (innerv + per - 1) / per;
      if (rows * tiles <= 0x7fffffffLL && innerv <= 0x7fffffffLL) {
        MCTQ_DISPATCH_U_NT(best_u, nt_mode(n * (int64_t)sizeof(TO)), {
          hipLaunchKernelGGL((rows_kernel<Op, TI, TO, U, NT>), dim3((unsigned)(rows * tiles)), dim3(kThreads), book_bytes,
                             st, x, y, (uint32_t)tiles, (uint32_t)innerv, (uint32_t)channels, op);
          note<Op, TI, TO>("rows_kernel", U, NT);
        });
        return check_launch("rows launch");
      }
    }
  }

  // lastaxis shape.
  if (vec_ok && inner == 1 && (channels % io::N) == 0 && channels <= 0x7fffffffLL && op.tables_aligned16()) {
    const int64_t vc = channels / io::N;                              // lane vectors per row
    // rows per slab: enough lanes (>= 2048) that the idle tail of the slab's last block is small; among the next few
    // candidates take the one that wastes the fewest lanes
    int64_t k = (2048 + vc - 1) / vc, best_waste = -1;
    for (int64_t c = k; c < k + 16; ++c) {
      const int64_t waste = (kThreads - (c * vc) % kThreads) % kThreads * 4096 / (c * vc);
      if (best_waste < 0 || waste < best_waste) { best_waste = waste; k = c; }
    }
    const int64_t bps = (k * vc + kThreads - 1) / kThreads;
    if (k * vc < (1 << 24) && bps <= 0x7fffffffLL) {                  // the kernel divides slab offsets through float32
      // slabs per lane: two; four for the affine quantizers with a zero-point table (twice the parameter work per lane to
      // spread: 4096^2 bfloat16 13.7 vs 14.2 us) -- measured on 14 shapes x 3 storage types, profiles/r06/chanlast2_sched.log
      const bool zp4 = std::is_same<Op, AffineOp>::value && has_zero_points(op);
      const int64_t LU = zp4 ? 4 : 2;
      const int64_t groups = (outer + LU * k - 1) / (LU * k);
      const int64_t gy = groups < 65535 ? groups : 65535, gz = (groups + gy - 1) / gy;
      if (groups <= 0xffffffffLL && gz <= 65535) {
        const dim3 grid((unsigned)bps, (unsigned)gy, (unsigned)gz);
        const float rvc = 1.0f / (float)vc;
#define MCTQ_LASTAXIS(U_, NT_, ZP_)                                                                                   \
        hipLaunchKernelGGL((lastaxis_kernel<Op, TI, TO, U_, NT_, ZP_>), grid, dim3(kThreads), book_bytes, st, x, y,   \
                           (uint64_t)outer, (uint32_t)vc, (uint32_t)k, rvc, (uint32_t)groups, op)
        MCTQ_WITH_OP_MODE(nt_mode(n * (int64_t)sizeof(TO)), {
          if constexpr (std::is_same<Op, AffineOp>::value) {
            if (zp4) MCTQ_LASTAXIS(4, NT, true);
            else MCTQ_LASTAXIS(2, NT, false);
          } else if constexpr (std::is_base_of<AffineOp, Op>::value) {
            // (the integer-code op: its zero-point table may be NULL as well -- the ZP = true form reads it unconditionally)
            if (has_zero_points(op)) MCTQ_LASTAXIS(2, NT, true);
            else MCTQ_LASTAXIS(2, NT, false);
          } else {
            MCTQ_LASTAXIS(2, NT, true);
          }
          note<Op, TI, TO>("lastaxis_kernel", (int)LU, NT);
        });
#undef MCTQ_LASTAXIS
        return check_launch("lastaxis launch");
      }
    }
  }

  // window shape.
  if (inner > 0x7fffffffLL || channels > 0x7fffffffLL) return fail_arg("inner/channels exceed 2^31-1");
  const uint32_t V = vec_ok ? io::N : 1;
  // lane-vectors per lane: 4, or ONE when the parameter window of a 4-wide tile would not fit LDS (tiny inner with many
  // channels and a large codebook).  Unaligned tensors (one element per lane access) always fit with 4.
  int wu = 4;
  uint32_t tile = 0, stride = 0;
  size_t lds = 0;
  for (;; wu = 1) {
    tile = kThreads * wu * V;
    const uint64_t max_rows = (uint64_t)(tile - 1 + (inner - 1)) / (uint64_t)inner + 1;   // rows a tile can touch
    const uint64_t entries = (uint64_t)channels <= max_rows ? (uint64_t)channels : max_rows;
    stride = (uint32_t)entries | 1u;       // odd: keeps the parameter planes on different LDS banks
    lds = book_bytes + (size_t)stride * Op::kWords * sizeof(float);
    if (lds <= 64 * 1024 || wu == 1 || !vec_ok) break;
  }
  if (lds > 64 * 1024) return fail_arg("parameter window exceeds 64 KiB of LDS");
  const int64_t blocks = (n + tile - 1) / tile;
  if (blocks > 0x7fffffffLL) return fail_arg("tensor too large for one launch");
  const bool idx32 = n <= (int64_t)0xffffffffLL - (int64_t)tile;
#define MCTQ_WINDOW(WU_, VEC_, NT_, IDX_)                                                                           \
  hipLaunchKernelGGL((window_kernel<Op, TI, TO, WU_, VEC_, NT_, IDX_>), dim3((unsigned)blocks), dim3(kThreads), lds, st, \
                     x, y, (IDX_)n, (uint32_t)inner, (uint32_t)channels, stride, op)
  if (!idx32) {
    // > 4 Gi elements in short rows: the common shape of the affine quantizers only (aligned, 4 lane-vectors, streaming)
    if constexpr (std::is_same<Op, AffineOp>::value) {
      if (!vec_ok || wu != 4) return fail_arg("tensors above 2^32 elements need vector alignment and a small window");
      MCTQ_WINDOW(4, true, 1, uint64_t);
      note<Op, TI, TO>("window_kernel<vector,u64>", 4, 1);
      return check_launch("window launch");
    } else {
      return fail_arg("per-channel rows shorter than 256 lane-vectors above 2^32 elements: affine quantizers only");
    }
  }
  if (!vec_ok) {                                       // one element per lane access: the slow path, one variant
    MCTQ_WINDOW(4, false, 1, uint32_t);
    note<Op, TI, TO>("window_kernel<scalar>", 4, 1);
    return check_launch("window launch");
  }
  MCTQ_WITH_OP_MODE(nt_mode(n * (int64_t)sizeof(TO)), {
    if (wu == 4) MCTQ_WINDOW(4, true, NT, uint32_t);
    else MCTQ_WINDOW(1, true, NT, uint32_t);
    note<Op, TI, TO>("window_kernel<vector>", wu, NT);
  });
#undef MCTQ_WINDOW
  return check_launch("window launch");
}

// ---- LUT helpers (host) ---------------------------------------------------------------------------
// literal scan: codebooks of up to 16 entries live in scalar registers (v_readlane broadcast), longer ones in LDS
inline int lut_class(int n_lut) { return n_lut <= 16 ? 16 : 0; }

inline void fill_lut_common(LutCommon& op, const float* thr, float eps, float mult, float cmin, float cmax,
                            int step_round) {
  op.thr = thr; op.eps = eps; op.mult = mult; op.inv_mult = 1.0f / mult; op.cmin = cmin; op.cmax = cmax;
  op.step_round = step_round;
}

template <int LP>
static LutOp<LP> make_lut_op(const float* thr, float eps, const float* lut, int n_lut, float mult, float cmin, float cmax,
                             int step_round) {
  LutOp<LP> op;
  fill_lut_common(op, thr, eps, mult, cmin, cmax, step_round);
  op.lut = lut; op.n_lut = n_lut;
  return op;
}

inline int table_entries(float cmin, float cmax) {        // same rule as mctq_tb::table_entries (host builder)
  if (!(cmin < cmax) || cmin != floorf(cmin) || cmax != floorf(cmax)) return -1;
  const double k = 2.0 * ((double)cmax - (double)cmin) + 1.0;
  if (k > 2048.0) return -1;
  return (int)k;
}

inline int check_pow2(float mult) {
  int e = 0;
  if (!(mult > 0.0f) || frexpf(mult, &e) != 0.5f) return fail_arg("mult must be a positive power of two");
  return 0;
}
inline int check_lut_args(const float* lut, int32_t n_lut, float mult) {
  if (!lut) return fail_arg("lut is NULL");
  if (n_lut < 1 || n_lut > 4096) return fail_arg("n_lut must be in [1, 4096]");
  return check_pow2(mult);
}

inline int make_table_op(LutTableOp& op, const float* thr, float eps, const float* table, int32_t entries, float mult,
                         float cmin, float cmax, int step_round) {
  if (!table) return fail_arg("table is NULL");
  if (int rc = check_pow2(mult)) return rc;
  if (entries != table_entries(cmin, cmax)) return fail_arg("entries does not match the clip range");
  fill_lut_common(op, thr, eps, mult, cmin, cmax, step_round);
  op.table = table; op.entries = entries; op.koff = 0.5f - 2.0f * cmin; op.kmax = (float)(entries - 1);
  return 0;
}
inline size_t table_bytes(int32_t entries) { return (size_t)(((entries + 1) * 2 + 3) & ~3) * 4; }
// storage-type dispatch: f(TI{}, TO{})
template <class F>
static int with_affine_types(int dtype, F f) {
  switch (dtype) {
    case MCTQ_DT_F32: return f(float(), float());
    case MCTQ_DT_F16: return f(_Float16(), _Float16());
    case MCTQ_DT_BF16: return f(__bf16(), __bf16());
    default: return fail_arg("unknown dtype");
  }
}
// int8 and uint8 codes are the same bytes (the low byte of the clamped integer): ONE 1-byte storage type
template <class F>
static int with_codes_types(int dtype, int code_dtype, F f) {
  if (code_dtype != MCTQ_CODE_I8 && code_dtype != MCTQ_CODE_U8) return fail_arg("unknown code dtype");
  switch (dtype) {
    case MCTQ_DT_F32: return f(float(), uint8_t());
    case MCTQ_DT_F16: return f(_Float16(), uint8_t());
    case MCTQ_DT_BF16: return f(__bf16(), uint8_t());
    default: return fail_arg("unknown dtype");
  }
}
template <class F>
static int with_lut_types(int dtype, F f) {
  switch (dtype) {
    case MCTQ_DT_F32: return f(float(), float());
    case MCTQ_DT_F16: return f(_Float16(), float());
    case MCTQ_DT_BF16: return f(__bf16(), float());
    default: return fail_arg("unknown dtype");
  }
}

}  // namespace mctq
