// mctq_qlinear.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h).
//
// Integer consumer of the quantizers' codes: y = dequant(A) . dequant(W)^T + bias computed on the 8-bit
// clamp indices the fake-quant kernels would have dequantized (mctq_fq_codes_*), with the gfx950 integer
// matrix cores (v_mfma_i32_16x16x64_i8) and exact int32 accumulation:
//     y[m][n] = float( sum_k (qa[m][k] - za) * qw[n][k] ) * (sa * sw[n]) + bias[n]
// For the small batches of inference the product is bound by streaming the weight codes once (1 B per weight
// instead of re-quantizing 4 B -> 4 B and then reading 4 B again in an fp32 GEMM), so the layout serves the
// memory system first: both operands are K-contiguous, per 256-byte K block every lane fetches four
// 16-byte pieces of one row, the four lanes of a row covering one 64-byte sector per load instruction (the
// pieces feed four MFMA steps; A and B use the same k assignment, which is all an exact integer sum needs), the 8 waves of a block interleave K blocks so that a row is read in 2 KiB
// runs, and the partial sums meet in LDS.  One block = 16 output columns x all rows x all of K; no split-K
// atomics, so the result does not depend on scheduling.
#include "mctq_kernels.hpp"

namespace mctq {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kQlKBlock = 256;          // bytes of K per wave step: 4 lane groups x 64 B
extern int g_ql_variant;                // tuning hook "ql_variant": 0 = automatic
// what the calling thread launched last (mctq_last_launch): "qlinear_<kernel>_<tile>" with the code widths
template <bool A_U8>
static void note_ql(const char* shape, int u = 0) {
  g_note.shape = shape; g_note.op = A_U8 ? "u8 x i8" : "i8 x i8"; g_note.unroll = u; g_note.nt = 0;
  g_note.in_bytes = 1; g_note.out_bytes = 4; ++g_note.count;
  if (g_launch_log) log_launch();
}
extern int g_ql_stagger;                // tuning hook "ql_stagger": half of the tiled kernel's waves copy after multiplying (default off)
extern int g_ql_rot;                    // tuning hook "ql_rot": K rotation between the blocks that share a weight tile (default off)
extern int g_ql_band;                   // tuning hook "ql_band": tile rows per band of the tiled kernel, 0 = automatic

// Output form: float32 values, or the next layer's activation codes (the fake-quant arithmetic of
// mctq_fq_codes_per_tensor applied to the float32 value in registers: clamp(rint(v * inv) + zp, lo, hi)).
struct QlOut {
  int mode;                 // 0 float32, 1 int8 codes, 2 uint8 codes
  float inv, zf, lo, hi;
};
__device__ __forceinline__ void ql_store(void* __restrict__ y, int64_t idx, float v, const QlOut& o) {
  if (o.mode == 0) {
    static_cast<float*>(y)[idx] = v;
  } else {
    float q = __builtin_rintf(v * o.inv) + o.zf;
    q = fminf(fmaxf(q, o.lo), o.hi);                  // NaN -> lo, as the codes kernel
    if (o.mode == 1) static_cast<int8_t*>(y)[idx] = (int8_t)(int)q;
    else static_cast<uint8_t*>(y)[idx] = (uint8_t)(int)q;
  }
}

template <bool NT>
__device__ __forceinline__ i32x4 ql_load16(const int8_t* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(p));
  else return *reinterpret_cast<const i32x4*>(p);
}

// 4 x 16 B of one row, 64 B apart, starting at k: piece p of the four lanes that share a row covers one
// contiguous 64-byte sector.  FULL: the whole 256-byte K block lies inside K (no guards, straight-line loads);
// otherwise pieces that start at or beyond K read as zero (K % 16 == 0).
template <bool NT, bool FULL>
__device__ __forceinline__ void ql_load_row(const int8_t* row, int64_t k, int64_t K, i32x4* out) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if constexpr (FULL) {
      out[p] = ql_load16<NT>(row + k + 64 * p);
    } else {
      const i32x4 z = {0, 0, 0, 0};
      out[p] = (k + 64 * p < K) ? ql_load16<NT>(row + k + 64 * p) : z;
    }
  }
}

// 4-bit weights (W4): a row holds K / 2 bytes; each group of 8 consecutive k is one dword whose byte j carries
// code k = j in its low and k = j + 4 in its high nibble (two's complement).  (x << 4) & 0xF0F0F0F0 and
// x & 0xF0F0F0F0 are then the int8 values 16 * code of k = 0..3 and k = 4..7 in natural order: two VALU ops per
// 8 weights, and the factor 16 leaves again by an exact shift of the int32 sum.
typedef int i32x2 __attribute__((ext_vector_type(2)));
template <bool NT, bool FULL>
__device__ __forceinline__ void ql_load_row4(const int8_t* row, int64_t k, int64_t K, i32x2* out) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const i32x2 z = {0, 0};
    const i32x2* q = reinterpret_cast<const i32x2*>(row + (k + 64 * p) / 2);
    if (FULL || k + 64 * p < K) out[p] = NT ? __builtin_nontemporal_load(q) : *q;
    else out[p] = z;
  }
}
__device__ __forceinline__ i32x4 ql_unpack4(i32x2 x) {
  return i32x4{(x[0] << 4) & (int)0xF0F0F0F0, x[0] & (int)0xF0F0F0F0, (x[1] << 4) & (int)0xF0F0F0F0, x[1] & (int)0xF0F0F0F0};
}
__device__ __forceinline__ i32x4 ql_unpack4(i32x4 x) { return x; }       // int8 weights: nothing to do

// MT: 16-row tiles of A per pass (1..4).  A_U8: activation codes are uint8 (re-biased to int8 by ^0x80).
// W4: weights are packed 4-bit codes (above).
template <int kQlWaves, int MT, bool A_U8, bool W_NT, bool W4 = false>
__global__ __launch_bounds__(kQlWaves * 64) void qlinear_i8_kernel(
    const int8_t* __restrict__ a, const int8_t* __restrict__ w, const float* __restrict__ w_scales,
    const int32_t* __restrict__ w_rowsum, const float* __restrict__ bias, void* __restrict__ y,
    int M, int N, int64_t K, int za, float sa, QlOut oq) {
  constexpr int kQlThreads = kQlWaves * 64;
  __shared__ int red[kQlWaves][MT][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  // rows beyond N / M are clamped to the last valid row: they are loaded and multiplied, never stored
  const int8_t* wrow = w + (int64_t)min(n0 + r, N - 1) * (W4 ? K / 2 : K);
  typedef typename std::conditional<W4, i32x2, i32x4>::type WF;     // a weight piece as loaded
  const int64_t kblocks = (K + kQlKBlock - 1) / kQlKBlock;
  const int64_t full_blocks = K / kQlKBlock;
  // this thread's output column in the epilogue is fixed (block size % 16 == 0): fetch its constants now, not
  // at the end of the dependency chain
  const int en = min(n0 + (int)(threadIdx.x & 15), N - 1);
  const int e_corr = za * w_rowsum[en];
  const float e_scale = sa * w_scales[en];
  const float e_bias = bias ? bias[en] : 0.0f;

  for (int m0 = 0; m0 < M; m0 += 16 * MT) {
    i32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = i32x4{0, 0, 0, 0};
    const int8_t* arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) arow[t] = a + (int64_t)min(m0 + 16 * t + r, M - 1) * K;

    WF wf0[4], wf1[4];
    i32x4 af0[MT][4], af1[MT][4];                       // two register buffers, swapped by unrolling
    // blocks walk K from different starting points so that they do not all ask L2 for the same lines of A at once
    const int64_t rot = full_blocks ? (int64_t)blockIdx.x % full_blocks : 0;
    auto fetch = [&](int64_t i, WF* wf, i32x4 (*af)[4]) {               // i-th full K block of this wave
      int64_t kblk = wave + i * kQlWaves + rot;
      if (kblk >= full_blocks) kblk -= full_blocks;
      const int64_t k = kblk * kQlKBlock + 16 * g;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#ifdef MCTQ_QL_ABLATE_A                      // timing experiment (tools/kbench_ql.hip): no activation traffic
#pragma unroll
        for (int p = 0; p < 4; ++p) af[t][p] = i32x4{t, p, t, p};
#else
        ql_load_row<false, true>(arow[t], k, K, af[t]);
#endif
      }
      if constexpr (W4) ql_load_row4<W_NT, true>(wrow, k, K, wf);      // L2-resident activations first, the HBM stream behind them
      else ql_load_row<W_NT, true>(wrow, k, K, wf);
    };
    auto multiply = [&](const WF* wf, const i32x4 (*af)[4]) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const i32x4 wv = ql_unpack4(wf[p]);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          i32x4 av = af[t][p];
          if constexpr (A_U8) av = av ^ (int)0x80808080;          // u8 code c -> int8 (c - 128)
#ifdef MCTQ_QL_ABLATE_MFMA                   // timing experiment: loads only
          acc[t] += av ^ wv;
#else
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, wv, acc[t], 0, 0, 0);
#endif
        }
      }
    };
    // full K blocks wave, wave + W, ...: straight-line loads, next block in flight while this one multiplies
    const int64_t n_i = full_blocks > wave ? (full_blocks - wave + kQlWaves - 1) / kQlWaves : 0;
    int64_t i = 0;
    if (n_i > 0) fetch(0, wf0, af0);
    while (i + 2 < n_i) {
      fetch(i + 1, wf1, af1);
      multiply(wf0, af0);
      fetch(i + 2, wf0, af0);
      multiply(wf1, af1);
      i += 2;
    }
    if (n_i - i == 2) {
      fetch(i + 1, wf1, af1);
      multiply(wf0, af0);
      multiply(wf1, af1);
    } else if (n_i - i == 1) {
      multiply(wf0, af0);
    }
    if (full_blocks != kblocks && wave == kQlWaves - 1) {          // ragged end of K: guarded loads, one wave
      const int64_t k = full_blocks * kQlKBlock + 16 * g;
      if constexpr (W4) ql_load_row4<W_NT, false>(wrow, k, K, wf0);
      else ql_load_row<W_NT, false>(wrow, k, K, wf0);
#pragma unroll
      for (int t = 0; t < MT; ++t) ql_load_row<false, false>(arow[t], k, K, af0[t]);
      multiply(wf0, af0);
    }

    // partial sums of the 8 waves meet in LDS
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][t][lane][i] = acc[t][i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < MT * 256; e += kQlThreads) {
      const int t = e >> 8, rem = e & 255;
      const int mi = rem >> 4, ni = rem & 15;                    // row / column inside the 16 x 16 tile
      const int src_lane = (mi >> 2) * 16 + ni, src_reg = mi & 3;  // C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
      int v = 0;
#pragma unroll
      for (int wv = 0; wv < kQlWaves; ++wv) v += red[wv][t][src_lane][src_reg];
      if constexpr (W4) v >>= 4;                                  // the weights entered as 16 * code: exact
      const int m = m0 + 16 * t + mi, n = n0 + ni;
      if (m < M && n < N) {
        float out = (float)(v - e_corr) * e_scale;
        if (bias) out = out + e_bias;
        ql_store(y, (int64_t)m * N + n, out, oq);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// The same product with the activation codes staged through per-wave LDS (ql_variant 18x, M <= 64 by default).
// In the kernel above every block re-reads ALL of A through L2 as MFMA-fragment-shaped loads: each 16-byte-per-lane
// instruction touches 16 rows x one 64-byte sector, and at M = 64 those 64 MiB of L2 traffic (256 blocks x 256 KiB)
// cost more than the 16 MiB weight stream from HBM (13.1 us vs 6.4 us without them, profiles/r01).  Here a wave
// copies its K block of A global -> LDS with global_load_lds_dwordx4: one instruction = 4 rows x 256 contiguous
// bytes (full lines; the 16-byte chunks of a row are permuted by an XOR swizzle applied on the SOURCE address, so
// the fragment reads below are bank-conflict free), then reads the fragments with ds_read_b128.  Bytes through L2
// are the same; the requests are whole lines instead of half-used sectors.  A wave owns its LDS region: no block
// barrier in the K loop, only vmcnt waits.  MT <= 2: two LDS buffers per wave (copy of block i+1 under the MFMAs of
// block i); MT = 4: one buffer (8 waves x 16 KiB).  The reduction of the 8 waves' partial sums reuses the region.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void ql_lds_void_;
typedef const __attribute__((address_space(1))) void ql_glb_void_;

template <int kQlWaves, int MT, bool A_U8, bool W_NT>
__global__ __launch_bounds__(kQlWaves * 64) void qlinear_i8_lds_kernel(
    const int8_t* __restrict__ a, const int8_t* __restrict__ w, const float* __restrict__ w_scales,
    const int32_t* __restrict__ w_rowsum, const float* __restrict__ bias, void* __restrict__ y,
    int M, int N, int64_t K, int za, float sa, QlOut oq) {
  constexpr int kQlThreads = kQlWaves * 64;
  constexpr int NBUF = kQlWaves * 2 * MT * 4096 <= 128 * 1024 ? 2 : 1;      // double-buffer when it fits 128 KiB
  constexpr int kSlots = MT * 256;                       // 16-byte slots of one A buffer: MT*16 rows x 16 chunks
  constexpr int kRedSlots = kQlWaves * MT * 64;          // int4 slots of the reduction
  constexpr int kLdsSlots = kQlWaves * NBUF * kSlots > kRedSlots ? kQlWaves * NBUF * kSlots : kRedSlots;
  __shared__ i32x4 lds[kLdsSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int8_t* wrow = w + (int64_t)min(n0 + r, N - 1) * K;
  const int64_t kblocks = (K + kQlKBlock - 1) / kQlKBlock;
  const int64_t full_blocks = K / kQlKBlock;
  const int en = min(n0 + (int)(threadIdx.x & 15), N - 1);
  const int e_corr = za * w_rowsum[en];
  const float e_scale = sa * w_scales[en];
  const float e_bias = bias ? bias[en] : 0.0f;
  i32x4* abuf = lds + wave * NBUF * kSlots;

  for (int m0 = 0; m0 < M; m0 += 16 * MT) {
    i32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = i32x4{0, 0, 0, 0};
    // copy instruction j of a K block covers tile rows 4j .. 4j+3: this lane's source row and swizzled chunk
    const int8_t* asrc[MT * 4];
#pragma unroll
    for (int j = 0; j < MT * 4; ++j) {
      const int row = 4 * j + (lane >> 4);
      asrc[j] = a + (int64_t)min(m0 + row, M - 1) * K + 16 * ((lane & 15) ^ (row & 15));
    }
    const int64_t rot = full_blocks ? (int64_t)blockIdx.x % full_blocks : 0;
    auto kbyte = [&](int64_t i) {
      int64_t kblk = wave + i * kQlWaves + rot;
      if (kblk >= full_blocks) kblk -= full_blocks;
      return kblk * kQlKBlock;
    };
    auto copy_a = [&](int64_t i, int buf) {
      const int64_t k = kbyte(i);
#pragma unroll
      for (int j = 0; j < MT * 4; ++j)
        __builtin_amdgcn_global_load_lds((ql_glb_void_*)(asrc[j] + k), (ql_lds_void_*)&abuf[buf * kSlots + j * 64], 16, 0, 0);
    };
    auto load_w = [&](int64_t i, i32x4* wf) { ql_load_row<W_NT, true>(wrow, kbyte(i) + 16 * g, K, wf); };
    auto multiply = [&](const i32x4* wf, int buf) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          i32x4 av = abuf[buf * kSlots + (16 * t + r) * 16 + ((4 * p + g) ^ r)];
          if constexpr (A_U8) av = av ^ (int)0x80808080;
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, wf[p], acc[t], 0, 0, 0);
        }
      }
    };
    const int64_t n_i = full_blocks > wave ? (full_blocks - wave + kQlWaves - 1) / kQlWaves : 0;
    i32x4 wf0[4], wf1[4];
    if constexpr (NBUF == 2) {
      if (n_i > 0) { copy_a(0, 0); load_w(0, wf0); }
      for (int64_t i = 0; i < n_i; i += 2) {
        if (i + 1 < n_i) {
          copy_a(i + 1, 1); load_w(i + 1, wf1);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * 4 + 4) : "memory");    // block i has landed, i+1 in flight
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        multiply(wf0, 0);
        if (i + 1 < n_i) {
          if (i + 2 < n_i) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // fragments of buffer 0 are in registers
            copy_a(i + 2, 0); load_w(i + 2, wf0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * 4 + 4) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          multiply(wf1, 1);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    } else {
      if (n_i > 0) load_w(0, wf0);
      for (int64_t i = 0; i < n_i; ++i) {
        copy_a(i, 0);
        if (i + 1 < n_i) load_w(i + 1, (i & 1) ? wf0 : wf1);
        if (i + 1 < n_i) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // A(i) and W(i) landed; W(i+1) may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        multiply((i & 1) ? wf1 : wf0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // before the next copy overwrites the buffer
      }
    }
    if (full_blocks != kblocks && wave == kQlWaves - 1) {          // ragged end of K: guarded loads straight from global
      const int64_t k = full_blocks * kQlKBlock + 16 * g;
      i32x4 af[4];
      ql_load_row<W_NT, false>(wrow, k, K, wf0);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        ql_load_row<false, false>(a + (int64_t)min(m0 + 16 * t + r, M - 1) * K, k, K, af);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          i32x4 av = af[p];
          if constexpr (A_U8) av = av ^ (int)0x80808080;
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, wf0[p], acc[t], 0, 0, 0);
        }
      }
    }

    // partial sums of the waves meet in LDS (the A buffers are dead now)
    __syncthreads();
    int* red = reinterpret_cast<int*>(lds);                 // [wave][t][lane][4]
#pragma unroll
    for (int t = 0; t < MT; ++t)
      lds[(wave * MT + t) * 64 + lane] = acc[t];
    __syncthreads();
    for (int e = threadIdx.x; e < MT * 256; e += kQlThreads) {
      const int t = e >> 8, rem = e & 255;
      const int mi = rem >> 4, ni = rem & 15;
      const int src_lane = (mi >> 2) * 16 + ni, src_reg = mi & 3;
      int v = 0;
#pragma unroll
      for (int wv = 0; wv < kQlWaves; ++wv) v += red[((wv * MT + t) * 64 + src_lane) * 4 + src_reg];
      const int m = m0 + 16 * t + mi, n = n0 + ni;
      if (m < M && n < N) {
        float out = (float)(v - e_corr) * e_scale;
        if (bias) out = out + e_bias;
        ql_store(y, (int64_t)m * N + n, out, oq);
      }
    }
    __syncthreads();
  }
}

template <int WAVES, int MT, bool A_U8>
static int launch_qlinear_lds(const void* a, const int8_t* w, const float* w_scales, const int32_t* w_rowsum,
                              const float* bias, void* y, int64_t M, int64_t N, int64_t K, int za, float sa,
                              const QlOut& oq, hipStream_t stream) {
  const dim3 grid((unsigned)((N + 15) / 16));
  const bool one_pass = M <= 16 * MT;
  if (one_pass)
    hipLaunchKernelGGL((qlinear_i8_lds_kernel<WAVES, MT, A_U8, true>), grid, dim3(WAVES * 64), 0, stream,
                       (const int8_t*)a, w, w_scales, w_rowsum, bias, y, (int)M, (int)N, K, za, sa, oq);
  else
    hipLaunchKernelGGL((qlinear_i8_lds_kernel<WAVES, MT, A_U8, false>), grid, dim3(WAVES * 64), 0, stream,
                       (const int8_t*)a, w, w_scales, w_rowsum, bias, y, (int)M, (int)N, K, za, sa, oq);
  note_ql<A_U8>(WAVES == 8 ? "qlinear_stream_lds_8waves" : "qlinear_stream_lds_4waves", MT);
  return check_launch("mctq_qlinear_i8 (LDS-staged activations)");
}

template <int WAVES, int MT, bool A_U8, bool W4 = false>
static int launch_qlinear(const void* a, const int8_t* w, const float* w_scales, const int32_t* w_rowsum,
                          const float* bias, void* y, int64_t M, int64_t N, int64_t K, int za, float sa,
                          const QlOut& oq, hipStream_t stream) {
  const dim3 grid((unsigned)((N + 15) / 16));
  const bool one_pass = M <= 16 * MT;             // weights read exactly once: keep them out of the caches
  if (one_pass)
    hipLaunchKernelGGL((qlinear_i8_kernel<WAVES, MT, A_U8, true, W4>), grid, dim3(WAVES * 64), 0, stream,
                       (const int8_t*)a, w, w_scales, w_rowsum, bias, y, (int)M, (int)N, K, za, sa, oq);
  else
    hipLaunchKernelGGL((qlinear_i8_kernel<WAVES, MT, A_U8, false, W4>), grid, dim3(WAVES * 64), 0, stream,
                       (const int8_t*)a, w, w_scales, w_rowsum, bias, y, (int)M, (int)N, K, za, sa, oq);
  note_ql<A_U8>(W4 ? "qlinear_stream_w4" : WAVES == 8 ? "qlinear_stream_8waves" : "qlinear_stream_4waves", MT);
  return check_launch("mctq_qlinear_i8");
}

// ------------------------------------------------------------------------------------------------
// Large row counts (M > 128): a conventional LDS-tiled product.  Block tile BM x BN over K in steps of 128 or
// 256 bytes, 4 waves in a 2 x 2 arrangement, each wave (BM/2) x (BN/2) as 16 x 16 MFMA tiles.  Tiles are stored
// as 16-byte chunks, chunk c of row r at slot c ^ swz(r): the lane groups of a ds_read_b128 then touch 16
// different bank groups.
// ------------------------------------------------------------------------------------------------
template <int CPR>
__device__ __forceinline__ int ql_swizzle(int row) {            // CPR = 16-byte chunks per tile row
  if constexpr (CPR == 8) return (row >> 1) & 7;                // 128-byte rows: two rows span the 64 banks
  else return row & (CPR - 1);                                  // 256-byte rows: every row starts on bank 0
}

// The operands are copied global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no ds_write pass;
// measured 1.3-1.4x over staging through registers).  A wave instruction fills 64 consecutive 16-byte slots, so the LDS image is linear in slot
// order and the swizzle sits on the SOURCE address (slot (row, c') receives chunk c' ^ swz(row) of that row)
// and, identically, on the fragment reads.  One barrier per K step: it retires the copies of the tile about
// to be multiplied and guarantees the other buffer is no longer being read; the next tile's copies are then
// issued and fly while this tile multiplies.
typedef __attribute__((address_space(3))) void ql_lds_void;
typedef const __attribute__((address_space(1))) void ql_glb_void;
__device__ __attribute__((aligned(16))) const int8_t g_ql_zero_chunk[16] = {0};

#ifdef MCTQ_QL_STAMP      // diagnostic build only (tools/kbench_ql_stamps.hip): where a K step of the tiled kernel spends its cycles
__device__ unsigned long long g_ql_stamp[16 * 8];      // per wave of ONE block: wait+barrier, copy issue, multiply, total, steps
#define MCTQ_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
// ST: LDS buffers in the ring (ST - 1 tiles requested ahead).  ST = 2 is the form above; with ST >= 3 the barrier of a K
// step only retires the copies of the tile about to be multiplied (counted vmcnt: copies return in issue order), so
// ST - 2 later tiles stay in flight across it.
// KG: wave groups (1, 2 or 4).  With KG = 2 the block has 8 waves (4: 16 waves, the groups' sums meet pairwise): both groups copy (twice the waves issuing copies -- the
// tiles run at the copy issue rate, which grows with the waves that issue), group g multiplies the g-th half of the
// 64-byte sub-steps of every K tile into its own accumulators, and the two partial sums meet in LDS at the end (exact
// integer sums: the order does not matter).
template <int BM, int BN, int kTileBK, bool A_U8, int ST = 2, int KG = 1>
__global__ __launch_bounds__(256 * KG) void qgemm_i8_glds_kernel(
    const int8_t* __restrict__ a, const int8_t* __restrict__ w, const float* __restrict__ w_scales,
    const int32_t* __restrict__ w_rowsum, const float* __restrict__ bias, void* __restrict__ y,
    int M, int N, int64_t K, int za, float sa, int m_blocks, int n_blocks, int gm, int flags, QlOut oq) {
  const bool rotate = (flags & 1) != 0;              // K rotation (below)
  const bool stagger = (flags & 2) != 0;             // half of the waves request the later tile AFTER multiplying (below)
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int CPR = kTileBK / 16;
  constexpr int SA = BM * CPR, SB = BN * CPR;        // slots of the A and of the B image
  constexpr int T = 256 * KG;                        // threads
  constexpr int LT = (SA + SB) / T;                  // copies per thread per tile
  constexpr int NKS = kTileBK / 64 / KG;             // 64-byte sub-steps of a tile per wave group
  static_assert(ST >= 2 && ST * (SA + SB) * 16 <= 160 * 1024 && (ST - 2) * LT <= 63, "ring depth");
  static_assert((KG == 1 || KG == 2 || KG == 4) && (SA + SB) % T == 0 && LT >= 1 && (kTileBK / 64) % KG == 0, "wave groups");
  static_assert(KG == 1 || (KG / 2) * 4 * TM * TN * 64 <= ST * (SA + SB), "the partial sums of half the groups must fit the ring");
  __shared__ i32x4 lds[ST][SA + SB];

  // Blocks are dealt round-robin to the 8 XCDs, each with its own L2: an XCD takes CONSECUTIVE tiles, ordered in bands
  // of gm tile rows (all gm row tiles of a column tile first), so what one XCD reads -- gm x BM activation rows and its
  // share of the weight rows -- is as small as the host could make it and stays in that XCD's L2 (gm = 1: row-major).
  const int total = m_blocks * n_blocks;
  int id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int per_band = gm * n_blocks;
  const int band = id / per_band, in_band = id - band * per_band;
  const int band_rows = min(gm, m_blocks - band * gm);
  const int mb = band * gm + in_band % band_rows, nb = in_band / band_rows;
  const int m0 = mb * BM, n0 = nb * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = wave >> 2, w4 = wave & 3;           // wave group, position in the 2 x 2 arrangement
  const int wm = w4 >> 1, wn = w4 & 1;
  const int r = lane & 15, g = lane >> 4;

  const int8_t* src[LT];        // source of this thread's j-th slot at k = 0
  int koff[LT];                 // byte offset of that chunk inside the K step (ragged-K test)
#pragma unroll
  for (int j = 0; j < LT; ++j) {
    const int s = j * T + tid;
    const bool is_a = s < SA;
    const int q = is_a ? s : s - SA;
    const int row = q / CPR, c = (q % CPR) ^ ql_swizzle<CPR>(row);
    koff[j] = 16 * c;
    src[j] = is_a ? a + (int64_t)min(m0 + row, M - 1) * K + 16 * c : w + (int64_t)min(n0 + row, N - 1) * K + 16 * c;
  }
  auto copy_tile = [&](int buf, int64_t k0) {
    const bool full = k0 + kTileBK <= K;
#pragma unroll
    for (int j = 0; j < LT; ++j) {
      const int8_t* p = (full || k0 + koff[j] < K) ? src[j] + k0 : g_ql_zero_chunk;     // beyond K: zeros
      __builtin_amdgcn_global_load_lds((ql_glb_void*)p, (ql_lds_void*)&lds[buf][j * T + wave * 64], 16, 0, 0);
    }
  };

  i32x4 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) acc[t][u] = i32x4{0, 0, 0, 0};

  const int64_t kt_n = (K + kTileBK - 1) / kTileBK;
  // The band_rows blocks of a band that share a weight tile run in step; started at the same k, all of them would wait
  // for each line's first fetch from HBM.  Block j of them starts its walk over K at tile j * kt_n / band_rows and wraps:
  // a line is then fetched by one block and found in the XCD's L2 by the others a fraction of the K loop later (an exact
  // integer sum does not depend on the order of its terms).  Measured (profiles/r03/qlinear_rot{0,1}.log): 8-15 % for the
  // two-buffer 4-wave tiles at K = 4096, nothing for the ring tiles, and 10-18 % SLOWER at K >= 8192, where the weight
  // tiles of an XCD no longer stay in its L2 for a quarter of the loop -- off by default (tuning key "ql_rot").
  const int64_t kt_rot = rotate ? (int64_t)(in_band % band_rows) * kt_n / band_rows : 0;
  const auto k_of = [&](int64_t kt) { const int64_t t = kt + kt_rot; return (t >= kt_n ? t - kt_n : t) * (int64_t)kTileBK; };
  copy_tile(0, k_of(0));
  if constexpr (ST > 2) {
#pragma unroll
    for (int i = 1; i < ST - 1; ++i)
      if (i < kt_n) copy_tile(i, k_of(i));
  }
  int e_corr[TN];                                    // epilogue constants of this lane's columns, fetched early
  float e_scale[TN], e_bias[TN];
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int n = min(n0 + wn * (BN / 2) + 16 * u + r, N - 1);
    e_corr[u] = za * w_rowsum[n];
    e_scale[u] = sa * w_scales[n];
    e_bias[u] = bias ? bias[n] : 0.0f;
  }
  int buf = 0, nbuf = ST - 1;                         // ring positions of tile kt and of tile kt + ST - 1
#ifdef MCTQ_QL_STAMP
  unsigned long long st_wait = 0, st_copy = 0, st_mul = 0, st_t0, st_a, st_b;
  MCTQ_STAMP(st_t0);
#endif
  for (int64_t kt = 0; kt < kt_n; ++kt) {
#ifdef MCTQ_QL_STAMP
    MCTQ_STAMP(st_a);
#endif
    if constexpr (ST == 2) {
      __syncthreads();                                // tile kt has landed; buffer buf ^ 1 is free
    } else {
      // tile kt has landed in every wave's share (its copies are older than the (ST - 2) * LT newest); the buffer of
      // tile kt - 1, read before this barrier by every wave, is free
      if (kt + ST - 2 < kt_n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((ST - 2) * LT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#ifdef MCTQ_QL_STAMP
    MCTQ_STAMP(st_b); st_wait += st_b - st_a;
#endif
    // A wave's copies, fragment reads and products are ONE instruction stream (profiles/r03/kbench_ql_stamps.log: per K step
    // of the 4-wave 64 x 64 x 256 tile 810 cycles issuing 8 copies, 930 reading fragments and multiplying, 280 at the
    // barrier).  Experiment "ql_stagger": half of the waves multiply first and copy afterwards (the buffer they fill, tile
    // kt - 1's, is free in either order) so that the address unit is not idle while all waves multiply -- +-3 %
    // (qlinear_stagger{0,1}.log), off by default.
    const bool copy_first = ST == 2 || !stagger || (((KG > 1 ? kg : wave) & 1) == 0);   // two buffers: the copy has only this step to land
    if (copy_first && kt + ST - 1 < kt_n) copy_tile(nbuf, k_of(kt + ST - 1));
#ifdef MCTQ_QL_STAMP
    MCTQ_STAMP(st_a); st_copy += st_a - st_b;
#endif
#pragma unroll
    for (int ksi = 0; ksi < NKS; ++ksi) {
      const int ks = kg * NKS + ksi;
      i32x4 fa[TM], fb[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = wm * (BM / 2) + 16 * t + r;
        fa[t] = lds[buf][row * CPR + ((ks * 4 + g) ^ ql_swizzle<CPR>(row))];
        if constexpr (A_U8) fa[t] = fa[t] ^ (int)0x80808080;
      }
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const int row = wn * (BN / 2) + 16 * u + r;
        fb[u] = lds[buf][SA + row * CPR + ((ks * 4 + g) ^ ql_swizzle<CPR>(row))];
      }
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[t], fb[u], acc[t][u], 0, 0, 0);
    }
#ifdef MCTQ_QL_STAMP
    { i32x4 keep = acc[0][0]; asm volatile("" : "+v"(keep)); acc[0][0] = keep; }     // the products of this step are issued before the stamp
    MCTQ_STAMP(st_b); st_mul += st_b - st_a;
#endif
    if (!copy_first && kt + ST - 1 < kt_n) copy_tile(nbuf, k_of(kt + ST - 1));
#ifdef MCTQ_QL_STAMP
    MCTQ_STAMP(st_a); st_copy += st_a - st_b;
#endif
    buf = buf + 1 == ST ? 0 : buf + 1;
    nbuf = nbuf + 1 == ST ? 0 : nbuf + 1;
  }
#ifdef MCTQ_QL_STAMP
  if (blockIdx.x == gridDim.x / 2 && lane == 0) {
    MCTQ_STAMP(st_a);
    unsigned long long* o = g_ql_stamp + wave * 8;
    o[0] = st_wait; o[1] = st_copy; o[2] = st_mul; o[3] = st_a - st_t0; o[4] = (unsigned long long)kt_n;
  }
#endif

  if constexpr (KG > 1) {             // the upper half of the groups hands its partial sums to the lower half through the
    i32x4* red = &lds[0][0];          // (dead) ring, until group 0 holds the whole sum
#pragma unroll
    for (int h = KG / 2; h >= 1; h >>= 1) {
      __syncthreads();                // every wave has read its last fragments / partial sums, every copy has landed
      if (kg >= h && kg < 2 * h) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) red[((((kg - h) * 4 + w4) * TM + t) * TN + u) * 64 + lane] = acc[t][u];
      }
      __syncthreads();
      if (kg < h) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] += red[(((kg * 4 + w4) * TM + t) * TN + u) * 64 + lane];
      }
    }
    if (kg != 0) return;
  }

#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int n = n0 + wn * (BN / 2) + 16 * u + r;
    if (n >= N) continue;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * (BM / 2) + 16 * t + 4 * g + i;
        if (m < M) {
          float out = (float)(acc[t][u][i] - e_corr[u]) * e_scale[u];
          if (bias) out = out + e_bias[u];
          ql_store(y, (int64_t)m * N + n, out, oq);
        }
      }
    }
  }
}

template <int BM, int BN, int BK, bool A_U8, int ST = 2, int KG = 1>
static int launch_glds(const void* a, const int8_t* w, const float* w_scales, const int32_t* w_rowsum,
                       const float* bias, void* y, int64_t M, int64_t N, int64_t K, int za, float sa,
                       const QlOut& oq, hipStream_t stream) {
  const int mbl = (int)((M + BM - 1) / BM), nbl = (int)((N + BN - 1) / BN);
  // tile rows per band: an XCD's chunk of total / 8 consecutive tiles as a gm x (chunk / gm) rectangle with the least
  // gm * BM + (chunk / gm) * BN, i.e. gm ~ sqrt(chunk * BN / BM)   (tuning key "ql_band" overrides)
  int gm = g_ql_band;
  if (gm <= 0) {
    const double chunk = (double)mbl * nbl / 8.0;
    gm = (int)(sqrt(chunk * BN / BM) + 0.5);
  }
  gm = gm < 1 ? 1 : gm > mbl ? mbl : gm;
  const int flags = (g_ql_rot && gm > 1 ? 1 : 0) | (g_ql_stagger ? 2 : 0);
  hipLaunchKernelGGL((qgemm_i8_glds_kernel<BM, BN, BK, A_U8, ST, KG>), dim3((unsigned)(mbl * nbl)), dim3(256 * KG), 0, stream,
                     (const int8_t*)a, w, w_scales, w_rowsum, bias, y, (int)M, (int)N, K, za, sa, mbl, nbl, gm, flags, oq);
  static char name[48];                              // "qlinear_tiled[_ring]_<BM>x<BN>x<BK>", formatted once per instantiation
  static const bool named = (snprintf(name, sizeof(name), "qlinear_tiled%s%s_%dx%dx%d", ST > 2 || (BM == 128 && BN == 64) ? "_ring" : "",
                                      KG == 2 ? "_8waves" : KG == 4 ? "_16waves" : "", BM, BN, BK), true);
  (void)named;
  note_ql<A_U8>(name, gm);
  return check_launch("mctq_qlinear_i8 (tiled, direct-to-LDS)");
}

// ------------------------------------------------------------------------------------------------
// Many rows AND many columns: the 64 x 64 wave tile above needs 8 KB of LDS reads per 256 MFMA cycles and wave; with
// two blocks per CU and the direct-to-LDS copies that is most of the LDS bandwidth, and the kernel stays near a third
// of the int8 peak.  Here one wave owns (16 TM) x (16 TN) outputs (128 x 128 for the 256 x 256 block tile: 16 KB of
// reads per 1024 MFMA cycles), accumulators in the 512-register file of a 1-wave-per-SIMD block.
// K advances in 64-byte steps through a ring of 4 LDS stages: stage kt is multiplied from registers (its fragments were
// read during stage kt-1), stage kt+1 is certified by the one barrier of the iteration (placed in the middle of the
// MFMA stream, so the matrix pipe has work queued while waves meet), stages kt+2 and kt+3 are in flight.
// Whole tiles only (M % BM == N % BN == K % 128 == 0); other shapes use the kernels above.
// ------------------------------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void ql_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); ql_static_for<I + 1, N>(f); }
}
template <int OFF>
__device__ __forceinline__ void ql_ds_read16(i32x4& v, uint32_t lds_byte) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte), "n"(OFF));
}

// A statement that NAMES the A fragments: everything the compiler itself does with them is ordered against it.
// WAIT: the LDS reads that filled them have returned.  Otherwise: a few idle cycles between the compiler's VALU
// writes to them and the matrix instructions (in asm, so their operand hazards are invisible to the compiler).
template <int TM, bool WAIT>
__device__ __forceinline__ void ql_fence_frags(i32x4* f) {
  static_assert(TM == 4 || TM == 8, "4 or 8 row tiles");
  if constexpr (TM == 8) {
    if constexpr (WAIT)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) :: "memory");
    else
      asm volatile("s_nop 4" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
  } else {
    if constexpr (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) :: "memory");
    else asm volatile("s_nop 4" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
  }
}

template <int TN>
__device__ __forceinline__ void ql_settle_row(i32x4 (&row)[TN]) {
  static_assert(TN == 4 || TN == 8, "row of 4 or 8 tiles");
  if constexpr (TN == 8)
    asm volatile("s_nop 7" : "+a"(row[0]), "+a"(row[1]), "+a"(row[2]), "+a"(row[3]), "+a"(row[4]), "+a"(row[5]), "+a"(row[6]), "+a"(row[7]));
  else
    asm volatile("s_nop 15" : "+a"(row[0]), "+a"(row[1]), "+a"(row[2]), "+a"(row[3]));
}

// products E .. END-1 of a step: tile (E / TN, E % TN) of the wave's TM x TN accumulator grid
template <int TM, int TN, int E, int END>
__device__ __forceinline__ void ql_mfma_run(i32x4 (&acc)[TM][TN], const i32x4* fa, const i32x4* fb) {
  if constexpr (E < END) {
    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[E / TN][E % TN]) : "v"(fa[E / TN]), "v"(fb[E % TN]));
    ql_mfma_run<TM, TN, E + 1, END>(acc, fa, fb);
  }
}
template <int TM, int FB, int Q, int QEND>
__device__ __forceinline__ void ql_frag_reads(i32x4* na, i32x4* nb, uint32_t pa, uint32_t pb) {
  if constexpr (Q < QEND) {
    if constexpr (Q < TM) ql_ds_read16<Q * FB>(na[Q], pa);
    else ql_ds_read16<(Q - TM) * FB>(nb[Q - TM], pb);
    ql_frag_reads<TM, FB, Q + 1, QEND>(na, nb, pa, pb);
  }
}
// second half of a step's products (tiles TM/2 .. TM-1) with the NEXT stage's TM + TN fragment reads spread between them
template <int TM, int TN, int FB, bool NEXT, int E>
__device__ __forceinline__ void ql_second_half(i32x4 (&acc)[TM][TN], const i32x4* fa, const i32x4* fb, i32x4* na,
                                               i32x4* nb, uint32_t pa, uint32_t pb) {
  constexpr int H2 = (TM - TM / 2) * TN, NR = TM + TN;
  if constexpr (E < H2) {
    if constexpr (NEXT) {
      constexpr int first = E == 0 ? 0 : ((E - 1) * NR) / H2 + 1;
      constexpr int last = (E * NR) / H2 < NR - 1 ? (E * NR) / H2 : NR - 1;
      ql_frag_reads<TM, FB, first, last + 1>(na, nb, pa, pb);
    }
    constexpr int t = TM / 2 + E / TN, u = E % TN;
    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[t][u]) : "v"(fa[t]), "v"(fb[u]));
    ql_second_half<TM, TN, FB, NEXT, E + 1>(acc, fa, fb, na, nb, pa, pb);
  }
}

template <int TM, int TN, bool A_U8, int S = 4, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void qgemm_i8_wide_kernel(
    const int8_t* __restrict__ a, const int8_t* __restrict__ w, const float* __restrict__ w_scales,
    const int32_t* __restrict__ w_rowsum, const float* __restrict__ bias, void* __restrict__ y,
    int M, int N, int64_t K, int za, float sa, int m_blocks, int n_blocks, QlOut oq) {
  constexpr int BM = 32 * TM, BN = 32 * TN, CPR = 4;                // S: LDS stages of the ring (S - 1 in flight)
  constexpr int SA = BM * CPR, SB = BN * CPR, SLOTS = SA + SB;      // 16-byte slots of one stage
  constexpr int LA = SA / 256, LB = SB / 256, LT = LA + LB;          // copies per thread per stage
  static_assert(S >= 4 && (S - 2) * LT <= 63 && OCC * S * SLOTS * 16 <= 160 * 1024, "ring depth / blocks per CU");
  __shared__ i32x4 lds[S * SLOTS];

  const int total = m_blocks * n_blocks;
  int id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);    // an XCD takes consecutive tiles ...
  constexpr int GM = 4;                                              // ... which form GM-row bands: shared A rows / W rows stay in its L2
  const int per_band = GM * n_blocks;
  const int band = id / per_band, in_band = id - band * per_band;
  const int band_rows = min(GM, m_blocks - band * GM);
  const int mb = band * GM + in_band % band_rows, nb = in_band / band_rows;
  const int m0 = mb * BM, n0 = nb * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  // copy j of a stage fills slots j * 256 + tid: row 64 j + tid / 4, physical chunk tid & 3 <- source chunk ^ swizzle.
  // swizzle(row) = (-(row >> 2)) & 3: a ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31},
  // {32-35,44-47,52-59}, {36-43,48-51,60-63}; with lane = 16 g + r reading row r, chunk g, the four rows of a group that
  // share (r & 3) -- and with it the 64-byte bank segment -- must sit in four different 16-byte slots:
  // {f(0), f(3), f(1)^1, f(2)^1} and {f(1), f(2), f(0)^1, f(3)^1} are permutations of 0..3 for f(x) = -x mod 4.
  const int crow = tid >> 2, cchunk = (tid & 3) ^ ((0 - (crow >> 2)) & 3);
  const int8_t* pa = a + (int64_t)(m0 + crow) * K + 16 * cchunk;
  const int8_t* pw = w + (int64_t)(n0 + crow) * K + 16 * cchunk;
  const int64_t jstride = 64 * K;
  auto copy_stage = [&](int slot, int kt) {
    i32x4* dst = lds + (slot % S) * SLOTS + wave * 64;
    const int64_t k0 = (int64_t)kt * 64;
#pragma unroll
    for (int j = 0; j < LA; ++j)
      __builtin_amdgcn_global_load_lds((ql_glb_void*)(pa + j * jstride + k0), (ql_lds_void*)(dst + j * 256), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < LB; ++j)
      __builtin_amdgcn_global_load_lds((ql_glb_void*)(pw + j * jstride + k0), (ql_lds_void*)(dst + SA + j * 256), 16, 0, 0);
  };

  // Fragment of tile row 16 t + r, k bytes [16 g, 16 g + 16): slot row * 4 + (g ^ swizzle(row)).
  // Fragment reads and products are inline asm: the compiler would otherwise retire all but the newest direct-to-LDS
  // copies before any LDS read (it cannot tell the stages apart), and it keeps the accumulators in VGPRs in some loop
  // blocks and AGPRs in others, copying 128-256 registers per step.  "a" operands pin them to the AGPR half.
  const uint32_t lane_byte = (uint32_t)(r * CPR + (g ^ ((0 - (r >> 2)) & 3))) * 16u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(ql_lds_void*)lds;
  const uint32_t a_byte = lds0 + lane_byte + (uint32_t)(wm * 16 * TM * CPR) * 16u;
  const uint32_t b_byte = lds0 + lane_byte + (uint32_t)(SA + wn * 16 * TN * CPR) * 16u;
  constexpr uint32_t kStageBytes = SLOTS * 16u;
  constexpr int kFragBytes = 16 * CPR * 16;

  i32x4 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) acc[t][u] = i32x4{0, 0, 0, 0};

  const int kt_n = (int)(K / 64);                      // even and >= 4 (host check)
  constexpr int H1 = TM / 2 * TN, NR = TM + TN;
  i32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];

  // One K step, the same code for every kt (a loop of ONE basic block keeps the register allocation of the accumulators
  // stable): near the end of K the request for stage kt+3 re-fetches the last stage into a ring slot nobody reads again,
  // and the fragment reads of the non-existent stage kt_n land in registers nobody uses.
  const int kt_last = kt_n - 1;
  auto step = [&](int kt, i32x4* fa, const i32x4* fb, i32x4* na, i32x4* nb_) {
    // this stage's fragments (requested half a step ago); the A fragments are operands so that the compiler's own
    // uses of them (the uint8 re-bias below) stay behind the wait
    ql_fence_frags<TM, true>(fa);
    if constexpr (A_U8) {
#pragma unroll
      for (int t = 0; t < TM; ++t) fa[t] = fa[t] ^ (int)0x80808080;     // uint8 codes -> int8 (za carries the -128)
      ql_fence_frags<TM, false>(fa);
    }
    ql_mfma_run<TM, TN, 0, H1>(acc, fa, fb);
    // stage kt+1 complete in LDS for every wave (own copies retired, then the barrier); stage kt+2 may still fly
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((S - 3) * LT) : "memory");
    copy_stage(kt + S - 1, min(kt + S - 1, kt_last));    // into the slot of kt-1, whose fragments were consumed a step ago
    const uint32_t so = (uint32_t)((kt + 1) % S) * kStageBytes;
    ql_second_half<TM, TN, kFragBytes, true, 0>(acc, fa, fb, na, nb_, a_byte + so, b_byte + so);
  };

  ql_static_for<0, S - 1>([&](auto i_) { copy_stage(decltype(i_)::value, min((int)decltype(i_)::value, kt_last)); });
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((S - 2) * LT) : "memory");
  ql_frag_reads<TM, kFragBytes, 0, NR>(fa0, fb0, a_byte, b_byte);
  for (int kt = 0; kt < kt_n; kt += 2) {                // kt_n is even (host check)
    step(kt, fa0, fb0, fa1, fb1);
    step(kt + 1, fa1, fb1, fa0, fb0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // The compiler does not see the matrix pipe behind the asm statements: without a barrier that NAMES the
  // accumulators it may read a tile (v_accvgpr_read) right behind the asm of that tile's last product, before the
  // result has left the pipe.  One statement per row of tiles, after the last product, 8 x 8 idle cycles in all.
  ql_static_for<0, TM>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    ql_settle_row<TN>(acc[t]);
  });

  auto epilogue = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int n = n0 + wn * 16 * TN + 16 * u + r;
      const int e_corr = za * w_rowsum[n];
      const float e_scale = sa * w_scales[n];
      const float e_bias = bias ? bias[n] : 0.0f;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t idx = (int64_t)(m0 + wm * 16 * TM + 16 * t + 4 * g + i) * N + n;
          float out = (float)(acc[t][u][i] - e_corr) * e_scale;
          if (bias) out = out + e_bias;
          if constexpr (MODE == 0) {
            static_cast<float*>(y)[idx] = out;
          } else {
            float q = fminf(fmaxf(__builtin_rintf(out * oq.inv) + oq.zf, oq.lo), oq.hi);
            if constexpr (MODE == 1) static_cast<int8_t*>(y)[idx] = (int8_t)(int)q;
            else static_cast<uint8_t*>(y)[idx] = (uint8_t)(int)q;
          }
        }
      }
    }
  };
  if (oq.mode == 0) epilogue(std::integral_constant<int, 0>{});
  else if (oq.mode == 1) epilogue(std::integral_constant<int, 1>{});
  else epilogue(std::integral_constant<int, 2>{});
}

// ------------------------------------------------------------------------------------------------
// The same 256 x 256 block tile with EIGHT waves in two groups that alternate between a memory phase and a product
// phase.  Issuing a direct-to-LDS copy holds a wave's instruction stream for ~100 cycles; with one wave per SIMD
// the matrix pipe idles meanwhile (counters on the 4-wave kernel above: pipe 47 % busy, half of the wave time
// spent waiting to issue).  Here waves w and w + 4 share a SIMD and belong to different groups: while one requests the
// next stage (4 copies) and reads its 12 fragments, the other issues its 32 products; a workgroup barrier swaps the
// roles twice per K step.  Wave tile 128 x 64 (TM 8 x TN 4), fragments single-buffered, ring of 4 stages with the
// request running 3 stages ahead.
//   slot:      2k            2k+1           2k+2
//   group 0:   MEM(k)    |   CMP(k)     |   MEM(k+1)
//   group 1:   CMP(k-1)  |   MEM(k)     |   CMP(k)
// Stage k+1 is certified by the barrier that ends slot 2k+1 (every wave first retires its own copies of it); the
// slot overwritten by MEM(k) held stage k-1, whose last fragment reads (group 1, slot 2k-1) were awaited before
// the barrier that ended that slot.
// ------------------------------------------------------------------------------------------------
template <bool A_U8>
__global__ __launch_bounds__(512, 1) void qgemm_i8_pp_kernel(
    const int8_t* __restrict__ a, const int8_t* __restrict__ w, const float* __restrict__ w_scales,
    const int32_t* __restrict__ w_rowsum, const float* __restrict__ bias, void* __restrict__ y,
    int M, int N, int64_t K, int za, float sa, int m_blocks, int n_blocks, QlOut oq) {
  constexpr int TM = 8, TN = 4, BM = 256, BN = 256, CPR = 4, S = 4;
  constexpr int SA = BM * CPR, SB = BN * CPR, SLOTS = SA + SB;      // 2048 16-byte slots per stage
  constexpr int LA = SA / 512, LB = SB / 512, LT = LA + LB;          // copies per thread per stage: 2 + 2
  __shared__ i32x4 lds[S * SLOTS];

  const int total = m_blocks * n_blocks;
  int id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  constexpr int GM = 4;
  const int per_band = GM * n_blocks;
  const int band = id / per_band, in_band = id - band * per_band;
  const int band_rows = min(GM, m_blocks - band * GM);
  const int mb = band * GM + in_band % band_rows, nb = in_band / band_rows;
  const int m0 = mb * BM, n0 = nb * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = (wave >> 1) & 3;
  const int r = lane & 15, g = lane >> 4;

  // copy j of a stage fills slots j * 512 + tid: row 128 j + tid / 4, physical chunk tid & 3 (swizzle as above)
  const int crow = tid >> 2, cchunk = (tid & 3) ^ ((0 - (crow >> 2)) & 3);
  const int8_t* pa = a + (int64_t)(m0 + crow) * K + 16 * cchunk;
  const int8_t* pw = w + (int64_t)(n0 + crow) * K + 16 * cchunk;
  const int64_t jstride = 128 * K;
  auto copy_stage = [&](int slot, int kt) {
    i32x4* dst = lds + (slot & (S - 1)) * SLOTS + wave * 64;
    const int64_t k0 = (int64_t)kt * 64;
#pragma unroll
    for (int j = 0; j < LA; ++j)
      __builtin_amdgcn_global_load_lds((ql_glb_void*)(pa + j * jstride + k0), (ql_lds_void*)(dst + j * 512), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < LB; ++j)
      __builtin_amdgcn_global_load_lds((ql_glb_void*)(pw + j * jstride + k0), (ql_lds_void*)(dst + SA + j * 512), 16, 0, 0);
  };

  const uint32_t lane_byte = (uint32_t)(r * CPR + (g ^ ((0 - (r >> 2)) & 3))) * 16u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(ql_lds_void*)lds;
  const uint32_t a_byte = lds0 + lane_byte + (uint32_t)(wm * 16 * TM * CPR) * 16u;
  const uint32_t b_byte = lds0 + lane_byte + (uint32_t)(SA + wn * 16 * TN * CPR) * 16u;
  constexpr uint32_t kStageBytes = SLOTS * 16u;
  constexpr int kFragBytes = 16 * CPR * 16;

  i32x4 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) acc[t][u] = i32x4{0, 0, 0, 0};

  const int kt_n = (int)(K / 64), kt_last = kt_n - 1;
  i32x4 fa[TM], fb[TN];

  auto mem_phase = [&](int kt) {
    const uint32_t so = (uint32_t)(kt & (S - 1)) * kStageBytes;
    ql_frag_reads<TM, kFragBytes, 0, TM + TN>(fa, fb, a_byte + so, b_byte + so);
    copy_stage(kt + 3, min(kt + 3, kt_last));
  };
  auto cmp_phase = [&]() {
    if constexpr (A_U8) {
#pragma unroll
      for (int t = 0; t < TM; ++t) fa[t] = fa[t] ^ (int)0x80808080;
      ql_fence_frags<TM, false>(fa);
    }
    ql_mfma_run<TM, TN, 0, TM * TN>(acc, fa, fb);
  };

  copy_stage(0, 0);
  copy_stage(1, 1);
  copy_stage(2, 2);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LT) : "memory");
  if (wave < 4) {
    for (int kt = 0; kt < kt_n; ++kt) {
      mem_phase(kt);
      ql_fence_frags<TM, true>(fa);                                        // fragments in registers (lgkmcnt 0)
      asm volatile("s_barrier" ::: "memory");                              // b(2k)
      cmp_phase();
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LT) : "memory");   // own copies of stage k+1 retired; b(2k+1)
    }
  } else {
    asm volatile("s_barrier" ::: "memory");                                // b0
    for (int kt = 0; kt < kt_n; ++kt) {
      mem_phase(kt);
      ql_fence_frags<TM, true>(fa);
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LT) : "memory");   // b(2k+1)
      cmp_phase();
      if (kt + 1 < kt_n) asm volatile("s_barrier" ::: "memory");           // b(2k+2)
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  ql_static_for<0, TM>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    ql_settle_row<TN>(acc[t]);
  });

  auto epilogue = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int n = n0 + wn * 16 * TN + 16 * u + r;
      const int e_corr = za * w_rowsum[n];
      const float e_scale = sa * w_scales[n];
      const float e_bias = bias ? bias[n] : 0.0f;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t idx = (int64_t)(m0 + wm * 16 * TM + 16 * t + 4 * g + i) * N + n;
          float out = (float)(acc[t][u][i] - e_corr) * e_scale;
          if (bias) out = out + e_bias;
          if constexpr (MODE == 0) {
            static_cast<float*>(y)[idx] = out;
          } else {
            float q = fminf(fmaxf(__builtin_rintf(out * oq.inv) + oq.zf, oq.lo), oq.hi);
            if constexpr (MODE == 1) static_cast<int8_t*>(y)[idx] = (int8_t)(int)q;
            else static_cast<uint8_t*>(y)[idx] = (uint8_t)(int)q;
          }
        }
      }
    }
  };
  if (oq.mode == 0) epilogue(std::integral_constant<int, 0>{});
  else if (oq.mode == 1) epilogue(std::integral_constant<int, 1>{});
  else epilogue(std::integral_constant<int, 2>{});
}

template <bool A_U8>
static int launch_pp(const void* a, const int8_t* w, const float* w_scales, const int32_t* w_rowsum,
                     const float* bias, void* y, int64_t M, int64_t N, int64_t K, int za, float sa,
                     const QlOut& oq, hipStream_t stream) {
  if (M % 256 != 0 || N % 256 != 0 || K % 64 != 0 || K < 256) return fail_arg("ping-pong tiled kernel needs whole 256 x 256 tiles, K % 64 == 0 and K >= 256");
  const int mbl = (int)(M / 256), nbl = (int)(N / 256);
  hipLaunchKernelGGL((qgemm_i8_pp_kernel<A_U8>), dim3((unsigned)(mbl * nbl)), dim3(512), 0, stream,
                     (const int8_t*)a, w, w_scales, w_rowsum, bias, y, (int)M, (int)N, K, za, sa, mbl, nbl, oq);
  note_ql<A_U8>("qlinear_pingpong_256x256");
  return check_launch("mctq_qlinear_i8 (ping-pong tiles)");
}

template <int TM, int TN, bool A_U8, int S = 4, int OCC = 1>
static int launch_wide(const void* a, const int8_t* w, const float* w_scales, const int32_t* w_rowsum,
                       const float* bias, void* y, int64_t M, int64_t N, int64_t K, int za, float sa,
                       const QlOut& oq, hipStream_t stream) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  if (M % BM != 0 || N % BN != 0 || K % 128 != 0 || K < 256) return fail_arg("wide tiled kernel needs whole tiles, K % 128 == 0 and K >= 256");
  const int mbl = (int)(M / BM), nbl = (int)(N / BN);
  hipLaunchKernelGGL((qgemm_i8_wide_kernel<TM, TN, A_U8, S, OCC>), dim3((unsigned)(mbl * nbl)), dim3(256), 0, stream,
                     (const int8_t*)a, w, w_scales, w_rowsum, bias, y, (int)M, (int)N, K, za, sa, mbl, nbl, oq);
  static const char* const kName = TM == 8 && TN == 8 ? "qlinear_wide_256x256" : TM == 4 && TN == 8 ? "qlinear_wide_128x256"
      : TM == 8 ? "qlinear_wide_256x128" : "qlinear_wide_128x128";
  note_ql<A_U8>(kName, S);
  return check_launch("mctq_qlinear_i8 (wide tiles)");
}

}  // namespace mctq

using namespace mctq;

static int qlinear_dispatch(const void* a_codes, int32_t a_code_dtype, int32_t a_zero_point, float a_scale,
                            const int8_t* w_codes, const float* w_scales, const int32_t* w_rowsum, const float* bias,
                            void* y, const QlOut& oq, int64_t M, int64_t N, int64_t K, void* stream) {
  if (M < 0 || N < 0 || K < 0) return fail_arg("negative extent");
  if (a_code_dtype != MCTQ_CODE_I8 && a_code_dtype != MCTQ_CODE_U8) return fail_arg("bad a_code_dtype");
  if (M == 0 || N == 0) return 0;
  if (!a_codes || !w_codes || !w_scales || !w_rowsum || !y) return fail_arg("NULL pointer");
  if (K % 16 != 0) return fail_arg("K must be a multiple of 16");
  if ((((uintptr_t)a_codes | (uintptr_t)w_codes) & 15u) != 0) return fail_arg("code matrices must be 16-byte aligned");
  if (K > (1 << 15)) return fail_arg("K > 32768 could overflow the int32 accumulator");
  if (M > INT32_MAX / 2 || N > INT32_MAX / 2) return fail_arg("M or N too large");
  const bool u8 = a_code_dtype == MCTQ_CODE_U8;
  const int za = u8 ? a_zero_point - 128 : a_zero_point;
  const hipStream_t s = (hipStream_t)stream;
#define MCTQ_QLL(W_, MT_)                                                                                          \
  (u8 ? launch_qlinear_lds<W_, MT_, true>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s) \
      : launch_qlinear_lds<W_, MT_, false>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s))
#define MCTQ_QG(BM_, BN_, BK_)                                                                                    \
  (u8 ? launch_glds<BM_, BN_, BK_, true>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s)  \
      : launch_glds<BM_, BN_, BK_, false>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s))
#define MCTQ_QW(TM_, TN_)                                                                                         \
  (u8 ? launch_wide<TM_, TN_, true>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s)      \
      : launch_wide<TM_, TN_, false>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s))
#define MCTQ_QG16(BM_, BN_, BK_, ST_)                                                                                      \
  (u8 ? launch_glds<BM_, BN_, BK_, true, ST_, 4>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s)  \
      : launch_glds<BM_, BN_, BK_, false, ST_, 4>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s))
#define MCTQ_QG8(BM_, BN_, BK_, ST_)                                                                                       \
  (u8 ? launch_glds<BM_, BN_, BK_, true, ST_, 2>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s)  \
      : launch_glds<BM_, BN_, BK_, false, ST_, 2>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s))
#define MCTQ_QPP()                                                                                                    \
  (u8 ? launch_pp<true>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s)                      \
      : launch_pp<false>(a_codes, w_codes, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s))
  // tuning key "ql_variant": force ONE of the kernels the automatic choice below can select (tests run each of them over
  // ragged shapes against the integer oracle); the experiment variants of round 3 are not built any more
  switch (g_ql_variant) {
    case 181: return MCTQ_QLL(8, 1);                  // weight streaming, activation codes through per-wave LDS: 16 / 32 / 64 rows per pass
    case 182: return MCTQ_QLL(8, 2);
    case 184: return MCTQ_QLL(8, 4);
    case 83233: return MCTQ_QG8(32, 32, 256, 3);      // 8-wave blocks (two wave groups), ring of three LDS buffers
    case 86433: return MCTQ_QG8(64, 32, 256, 3);
    case 86633: return MCTQ_QG8(64, 64, 128, 3);
    case 812613: return MCTQ_QG8(128, 64, 128, 3);
    case 166623: return MCTQ_QG16(64, 64, 256, 3);    // 16-wave blocks (four wave groups)
    case 1612623: return MCTQ_QG16(128, 64, 256, 3);
    case 612: return MCTQ_QG(64, 128, 128);           // 4-wave blocks, two LDS buffers
    case 1212: return MCTQ_QG(128, 128, 128);
    case 662: return MCTQ_QG(64, 64, 256);
    case 2544: return MCTQ_QW(4, 4);                  // wave-wide 128 x 128 / 128 x 256 tiles
    case 2548: return MCTQ_QW(4, 8);
    case 2560: return MCTQ_QPP();                     // 256 x 256 ping-pong tiles
    default: break;
  }
  const int64_t cus = cu_count();
  const auto blocks = [&](int64_t bm, int64_t bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  // Few rows: the weight-streaming kernels (activation codes staged through per-wave LDS: full-line copies instead of
  // fragment-shaped L2 reads; profiles/r02/qlinear_probe.log) are candidates of the cost model below.
  // Many rows and columns, whole tiles: the 256 x 256 ping-pong kernel (2.2-2.3 POP/s against 1.5 for the 128 x 128
  // tiles; profiles/r02/qgemm_wide_probe.log) or, when there are too few such tiles for the chip, 128 x 256 wave-wide
  // tiles -- weighed by how full their last round of blocks is (one block per CU; the 128 x 128 kernel fits two).
  // Only where the 128 x 128 kernel would fill the chip: smaller problems keep its finer tiles.
  if (K % 128 == 0 && K >= 256 && N % 256 == 0 && M % 128 == 0 && blocks(128, 128) >= 2 * cus) {
    const auto fill = [&](int64_t nb, int64_t slots) { return (double)nb / (double)(((nb + slots - 1) / slots) * slots); };
    const double old_rate = 1.5 * fill(blocks(128, 128), 2 * cus);
    const double pp_rate = M % 256 == 0 ? 2.25 * fill(blocks(256, 256), cus) : 0.0;
    const double w48_rate = 2.0 * fill(blocks(128, 256), cus);
    if (pp_rate >= w48_rate && pp_rate > old_rate) return MCTQ_QPP();
    if (w48_rate > old_rate) return MCTQ_QW(4, 8);
  }
  // Between the two regimes every kernel runs at the CU's intake of direct-to-LDS copies, an issue rate: about 36 KiB/us
  // with one resident block per CU, 48-58 with two, 64 with three (profiles/r03/qlinear_tile_sweep.log,
  // qlinear_small_tiles.log, EXPERIMENTS.md).  So the time of a launch is (operand bytes of a block) x (blocks the busiest
  // CU takes) / rate, and the choice is the kernel that makes that least: the weight-streaming kernel (a block = 16
  // columns x all rows, 64 rows per pass), 32 x 32 ... 128 x 128 tiles with two or three LDS buffers and four or eight
  // waves (eight / sixteen: two / four wave groups that all copy and each multiply a share of a K tile -- more waves
  // issuing copies raise the rate by 10-30 %), and the asm-pinned 128 x 128 kernel where the problem is whole tiles.  Earlier candidates win ties (smaller tiles first).
  {
    const auto tiles_cost = [&](int bm, int bn, int occ, double r1, double r2, double r3) {   // us, up to a common constant
      const double kib = (double)(bm + bn) * (double)K / 1024.0, rate[4] = {1.0, r1, r2, r3};
      const int64_t per_cu = (blocks(bm, bn) + cus - 1) / cus, full = per_cu / occ, rem = per_cu % occ;
      return (double)(full * occ) * kib / rate[occ] + (rem ? (double)rem * kib / rate[rem] : 0.0);
    };
    enum { kStream1, kStream2, kStream, kT33, kT63, kT66R, kT66, kT66S, kT126, kT126X, kT612, kT1212, kW44 };
    double best = 1e300;
    int pick = kT66;
    const auto consider = [&](int id, double c) { if (c < best) { best = c; pick = id; } };
    // weight streaming: a block = 16 weight rows + 16 MT activation rows per pass; MT = 1 fits two blocks per CU.  Rates
    // fitted on profiles/r03/qlinear_small_m.log (16 / 32 rows) and qlinear_small_tiles.log (64 ... 128 rows).
    const int64_t s_per_cu = ((N + 15) / 16 + cus - 1) / cus;
    const double k_kib = (double)K / 1024.0;
    if (M <= 16)
      consider(kStream1, (double)(s_per_cu / 2 * 2) * 32.0 * k_kib / 36.5 + (double)(s_per_cu % 2) * 32.0 * k_kib / 29.0);
    else if (M <= 32)
      consider(kStream2, (double)s_per_cu * 48.0 * k_kib / 38.0);
    else if (M <= 128)
      consider(kStream, (double)s_per_cu * (double)((M + 63) / 64) * 80.0 * k_kib / 47.0);
    // 8-wave blocks (two wave groups, 3-buffer ring): rates fitted on profiles/r03/qlinear_8waves.log
    consider(kT33, tiles_cost(32, 32, 3, 42, 63, 65));
    consider(kT63, tiles_cost(64, 32, 2, 50, 61, 0));
    consider(kT66R, tiles_cost(64, 64, 1, 48, 0, 0));           // 16 waves (qlinear_16waves.log): one block per CU
    consider(kT66, tiles_cost(64, 64, 2, 36, 54, 0));           // 4 waves, two buffers, two blocks per CU
    consider(kT66S, tiles_cost(64, 64, 3, 36, 56, 59));         // 8 waves, 128-byte K steps, three blocks per CU
    consider(kT126, tiles_cost(128, 64, 2, 44, 54, 0));
    consider(kT126X, tiles_cost(128, 64, 1, 48, 0, 0));         // 16 waves, 256-byte K steps: one block per CU
    consider(kT612, tiles_cost(64, 128, 3, 36, 50, 64));
    consider(kT1212, tiles_cost(128, 128, 2, 30, 48, 0));
    if (K % 128 == 0 && K >= 256 && N % 128 == 0 && M % 128 == 0) consider(kW44, tiles_cost(128, 128, 2, 38, 43, 0));
    switch (pick) {
      case kStream1: return MCTQ_QLL(8, 1);
      case kStream2: return MCTQ_QLL(8, 2);
      case kStream: return MCTQ_QLL(8, 4);
      case kT33: return MCTQ_QG8(32, 32, 256, 3);
      case kT63: return MCTQ_QG8(64, 32, 256, 3);
      case kT66R: return MCTQ_QG16(64, 64, 256, 3);
      case kT66S: return MCTQ_QG8(64, 64, 128, 3);
      case kT126: return MCTQ_QG8(128, 64, 128, 3);
      case kT126X: return MCTQ_QG16(128, 64, 256, 3);
      case kT612: return MCTQ_QG(64, 128, 128);
      case kT1212: return MCTQ_QG(128, 128, 128);
      case kW44: return MCTQ_QW(4, 4);
      default: return MCTQ_QG(64, 64, 256);
    }
  }
#undef MCTQ_QG
#undef MCTQ_QG8
#undef MCTQ_QG16
#undef MCTQ_QW
#undef MCTQ_QPP
#undef MCTQ_QLL
}

extern "C" {

int mctq_qlinear_i8(const void* a_codes, int32_t a_code_dtype, int32_t a_zero_point, float a_scale,
                    const int8_t* w_codes, const float* w_scales, const int32_t* w_rowsum, const float* bias,
                    float* y, int64_t M, int64_t N, int64_t K, void* stream) {
  QlOut oq;
  oq.mode = 0; oq.inv = oq.zf = oq.lo = oq.hi = 0.0f;
  return qlinear_dispatch(a_codes, a_code_dtype, a_zero_point, a_scale, w_codes, w_scales, w_rowsum, bias, y, oq, M, N, K,
                          stream);
}

int mctq_qlinear_i8_codes(const void* a_codes, int32_t a_code_dtype, int32_t a_zero_point, float a_scale,
                          const int8_t* w_codes, const float* w_scales, const int32_t* w_rowsum, const float* bias,
                          void* y_codes, int32_t y_code_dtype, float y_scale, int32_t y_zero_point, int32_t y_quant_min,
                          int32_t y_quant_max, int64_t M, int64_t N, int64_t K, void* stream) {
  if (y_code_dtype != MCTQ_CODE_I8 && y_code_dtype != MCTQ_CODE_U8) return fail_arg("bad y_code_dtype");
  if (y_quant_min > y_quant_max) return fail_arg("quant_min > quant_max");
  if (y_code_dtype == MCTQ_CODE_I8 ? (y_quant_min < -128 || y_quant_max > 127) : (y_quant_min < 0 || y_quant_max > 255))
    return fail_arg("clamp domain does not fit the code type");
  QlOut oq;
  oq.mode = y_code_dtype == MCTQ_CODE_I8 ? 1 : 2;
  oq.inv = 1.0f / y_scale;                           // host IEEE division == the codes kernel's
  oq.zf = (float)y_zero_point; oq.lo = (float)y_quant_min; oq.hi = (float)y_quant_max;
  return qlinear_dispatch(a_codes, a_code_dtype, a_zero_point, a_scale, w_codes, w_scales, w_rowsum, bias, y_codes, oq, M,
                          N, K, stream);
}


int mctq_qlinear_w4a8(const void* a_codes, int32_t a_code_dtype, int32_t a_zero_point, float a_scale,
                      const uint8_t* w_codes4, const float* w_scales, const int32_t* w_rowsum, const float* bias,
                      void* y, int32_t y_code_dtype, float y_scale, int32_t y_zero_point, int32_t y_quant_min,
                      int32_t y_quant_max, int64_t M, int64_t N, int64_t K, void* stream) {
  if (M < 0 || N < 0 || K < 0) return fail_arg("negative extent");
  if (a_code_dtype != MCTQ_CODE_I8 && a_code_dtype != MCTQ_CODE_U8) return fail_arg("bad a_code_dtype");
  QlOut oq;
  oq.mode = 0; oq.inv = oq.zf = oq.lo = oq.hi = 0.0f;
  if (y_code_dtype >= 0) {
    if (y_code_dtype != MCTQ_CODE_I8 && y_code_dtype != MCTQ_CODE_U8) return fail_arg("bad y_code_dtype");
    if (y_quant_min > y_quant_max) return fail_arg("quant_min > quant_max");
    if (y_code_dtype == MCTQ_CODE_I8 ? (y_quant_min < -128 || y_quant_max > 127) : (y_quant_min < 0 || y_quant_max > 255))
      return fail_arg("clamp domain does not fit the code type");
    oq.mode = y_code_dtype == MCTQ_CODE_I8 ? 1 : 2;
    oq.inv = 1.0f / y_scale;
    oq.zf = (float)y_zero_point; oq.lo = (float)y_quant_min; oq.hi = (float)y_quant_max;
  }
  if (M == 0 || N == 0) return 0;
  if (!a_codes || !w_codes4 || !w_scales || !w_rowsum || !y) return fail_arg("NULL pointer");
  if (K % 16 != 0) return fail_arg("K must be a multiple of 16");
  if ((((uintptr_t)a_codes) & 15u) != 0 || (((uintptr_t)w_codes4) & 7u) != 0)
    return fail_arg("a_codes must be 16-byte and w_codes4 8-byte aligned");
  if (K > (1 << 15)) return fail_arg("K > 32768 could overflow the int32 accumulator");
  if (M > INT32_MAX / 2 || N > INT32_MAX / 2) return fail_arg("M or N too large");
  const bool u8 = a_code_dtype == MCTQ_CODE_U8;
  const int za = u8 ? a_zero_point - 128 : a_zero_point;
  const int8_t* w = reinterpret_cast<const int8_t*>(w_codes4);
  const hipStream_t s = (hipStream_t)stream;
#define MCTQ_QL4(MT_)                                                                                               \
  (u8 ? launch_qlinear<8, MT_, true, true>(a_codes, w, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s)     \
      : launch_qlinear<8, MT_, false, true>(a_codes, w, w_scales, w_rowsum, bias, y, M, N, K, za, a_scale, oq, s))
  if (M <= 16) return MCTQ_QL4(1);
  if (M <= 32) return MCTQ_QL4(2);
  return MCTQ_QL4(4);
#undef MCTQ_QL4
}

#ifdef MCTQ_QL_STAMP
int mctq_debug_ql_stamps(unsigned long long* out16x8) {      // diagnostic build only
  return -(int)hipMemcpyFromSymbol(out16x8, HIP_SYMBOL(mctq::g_ql_stamp), sizeof(unsigned long long) * 16 * 8);
}
#endif

}  // extern "C"
