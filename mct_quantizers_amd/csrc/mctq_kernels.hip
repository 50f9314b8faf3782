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

template <int LP>   // LP > 0: codebook broadcast into LP scalar registers; LP == 0: codebook in LDS
struct LutOp {
  const float* __restrict__ thr;       // [C] thresholds (per-channel launches only)
  float eps;
  const float* __restrict__ lut;       // [n_lut] device codebook, caller's order
  int n_lut;
  float mult, inv_mult, cmin, cmax;

  struct Param { float d, t; };        // divisor fl32(thr + eps), multiplier thr
  static constexpr int kWords = 2;

  __device__ __forceinline__ Param fetch(uint32_t c) const {
    Param p; p.t = thr[c]; p.d = p.t + eps; return p;
  }
  __device__ __forceinline__ static void put(float* lds, uint32_t i, uint32_t stride, const Param& p) {
    lds[i] = p.d; lds[stride + i] = p.t;
  }
  __device__ __forceinline__ static Param get(const float* lds, uint32_t i, uint32_t stride) {
    Param p; p.d = lds[i]; p.t = lds[stride + i]; return p;
  }

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

  __device__ __forceinline__ float apply(float x, const Param& p, const Book& b) const {
    float v = (x / p.d) * mult;                        // IEEE division (quantizer_utils.py:169)
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

template <class Op>
__device__ __forceinline__ f4 apply4(const Op& op, f4 v, const typename Op::Param& p, const typename Op::Book& b) {
  f4 r;
  r.x = op.apply(v.x, p, b);
  r.y = op.apply(v.y, p, b);
  r.z = op.apply(v.z, p, b);
  r.w = op.apply(v.w, p, b);
  return r;
}

// ------------------------------------------------------------------------------------------
// flat: per-tensor parameters.  Block b owns float4s [b*256*U, (b+1)*256*U); lane accesses are
// 16 B, consecutive lanes consecutive addresses, U independent loads in flight per lane.
// ------------------------------------------------------------------------------------------
template <class Op, int U, bool NT>
__global__ __launch_bounds__(kThreads) void flat_kernel(Op op, typename Op::Param p,
                                                        const float* __restrict__ xs, float* __restrict__ ys,
                                                        int64_t n) {
  extern __shared__ float smem[];
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
    for (int u = 0; u < U; ++u) st4<NT>(y + base + u * kThreads, apply4(op, v[u], p, book));
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kThreads;
      if (i < n4) v[u] = ld4<NT>(x + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kThreads;
      if (i < n4) st4<NT>(y + i, apply4(op, v[u], p, book));
    }
  }
  if (blockIdx.x == 0) {                                   // n % 4 trailing elements
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (i < n) ys[i] = op.apply(xs[i], p, book);
  }
}

// flat, one element per lane: used when x or y is not 16-byte aligned.
template <class Op>
__global__ __launch_bounds__(kThreads) void flat_scalar_kernel(Op op, typename Op::Param p,
                                                               const float* __restrict__ x, float* __restrict__ y,
                                                               int64_t n) {
  extern __shared__ float smem[];
  const typename Op::Book book = op.setup(smem);
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
    y[i] = op.apply(x[i], p, book);
}

// ------------------------------------------------------------------------------------------
// rows: tensor viewed as [rows = outer*C][inner4] float4.  blockIdx -> (row, tile); the row's
// channel index is wave-uniform, so fetch() compiles to scalar loads and the parameters sit in
// SGPRs for the whole block.
// ------------------------------------------------------------------------------------------
template <class Op, int U, bool NT>
__global__ __launch_bounds__(kThreads) void rows_kernel(Op op, const float* __restrict__ xs, float* __restrict__ ys,
                                                        uint32_t tiles_per_row, uint32_t inner4, uint32_t channels) {
  extern __shared__ float smem[];
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
    for (int u = 0; u < U; ++u) st4<NT>(y + col + u * kThreads, apply4(op, v[u], p, book));
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = col + u * kThreads;
      if (i < inner4) st4<NT>(y + i, apply4(op, v[u], p, book));
    }
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
  extern __shared__ float smem[];
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
        out[j] = op.apply(in[j], Op::get(tab, li, stride), book);
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
        ys[e0 + off + j] = op.apply(xs[e0 + off + j], Op::get(tab, li, stride), book);
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
static thread_local char g_err[256] = "";
static int g_nt = 1;   // non-temporal loads/stores: +7% on the cold 4096x4096 stream (profiles/)
static int g_unroll = 4;

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

template <int LP>
static LutOp<LP> make_lut_op(const float* thr, float eps, const float* lut, int n_lut, float mult, float cmin, float cmax) {
  LutOp<LP> op;
  op.thr = thr; op.eps = eps; op.lut = lut; op.n_lut = n_lut;
  op.mult = mult; op.inv_mult = 1.0f / mult; op.cmin = cmin; op.cmax = cmax;
  return op;
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
  return fail_arg("unknown tuning key");
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
              LutOp<4>::Param p; p.d = thr_div; p.t = thr_mul; return launch_flat(op, p, x, y, n, 0, st); }
    case 16: { auto op = make_lut_op<16>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max);
               LutOp<16>::Param p; p.d = thr_div; p.t = thr_mul; return launch_flat(op, p, x, y, n, 0, st); }
    case 64: { auto op = make_lut_op<64>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max);
               LutOp<64>::Param p; p.d = thr_div; p.t = thr_mul; return launch_flat(op, p, x, y, n, 0, st); }
    default: { auto op = make_lut_op<0>(nullptr, 0.f, lut, n_lut, mult, clip_min, clip_max);
               LutOp<0>::Param p; p.d = thr_div; p.t = thr_mul;
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

}  // extern "C"
