// LutCompactOp -- the compact decision table of round 4, kept OUTSIDE the shipped library: an experiment that
// measured equal to the decision table under the bench protocol (profiles/r04/cfg4_lut_experiments.md).
// Include after mctq_kernels.hpp; build: tools/experiments/lut_compact/README.md.
#pragma once
#include "mctq_kernels.hpp"

namespace mctq {

// Compact decision table (mctq_table_builder.h: build_compact): the same cells k and the same exact thresholds as
// LutTableOp, stored as one BYTE per cell (index j of the first step at or above the cell) plus the list of the codebook's
// steps {T_j, half2(q below, q above)} -- 648 bytes instead of 4 KB for 16 centres on an 8-bit clip range; two dependent
// LDS reads per element instead of one.  Built to test whether the 4 KB staged by every block is what keeps the table
// kernel behind the affine one on config 4.  Measured (profiles/r04/cfg4_lut_experiments.md): under bench.py's cold,
// sustained protocol the two forms are equal at U = 4 (58.6-59.2 vs 58.9 us) and one-step tiles, which the small table makes
// affordable and which win 3 us in short half-warm runs, LOSE 3 us there -- so the decision table stays the default and this
// op is selected by MCTQ_COMPACT_LUT=1 only.
struct LutCompactBook { const uint8_t* cell; const f32x2* step; float nan_q; };

struct LutCompactOp : LutCommon {
  static constexpr const char* kName = "LutCompactOp";
  const uint32_t* __restrict__ blob;   // device, n_words words
  int entries;                         // K cells
  int n_words;
  float koff;                          // 0.5 - 2*clip_min
  float kmax;                          // entries - 1

  typedef LutCompactBook Book;
  __device__ __forceinline__ uint32_t book_words() const { return ((uint32_t)n_words + 3u) & ~3u; }
  __device__ __forceinline__ Book book_at(float* lds) const {
    const uint32_t cw = ((uint32_t)entries + 3u) >> 2;
    Book b;
    b.cell = reinterpret_cast<const uint8_t*>(lds);
    b.step = reinterpret_cast<const f32x2*>(lds + cw);
    b.nan_q = lds[n_words - 2];
    return b;
  }
  __device__ __forceinline__ Book setup(float* lds) const {
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds);
    for (int j = threadIdx.x; j < n_words; j += kThreads) dst[j] = blob[j];
    __syncthreads();
    return book_at(lds);
  }
  // requested BEFORE the tile's data loads, written to LDS after them (in-order return of vector loads: LutTableOp::prefetch)
  struct Prefetch { uint32_t r[5]; };                // n_words <= 512 + 512 + 2 over 256 threads
  __device__ __forceinline__ Prefetch prefetch() const {
    Prefetch p;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int j = (int)threadIdx.x + i * kThreads;
      if (j < n_words) p.r[i] = blob[j];
    }
    return p;
  }
  __device__ __forceinline__ Book commit(const Prefetch& p, float* lds) const {
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int j = (int)threadIdx.x + i * kThreads;
      if (j < n_words) dst[j] = p.r[i];
    }
    __syncthreads();
    return book_at(lds);
  }

  template <bool FAST>
  __device__ __forceinline__ void locate(float x, const Param& p, float& v, int& k) const {   // as LutTableOp::locate
    v = scaled<FAST>(x, p);
    k = (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(v, 2.0f, koff), 0.0f, kmax);
  }
  template <bool FAST>
  __device__ __forceinline__ float decide(float x, float v, f32x2 e, const Param& p, const Book& b) const {
    const uint32_t pair = __float_as_uint(e.y);
    const uint32_t h = (v >= e.x) ? (pair >> 16) : pair;
    float q = __half2float(__ushort_as_half((unsigned short)h));
    const bool nan = (FAST && step_round == 0) ? (x != x) : (v != v);    // see LutTableOp::decide
    q = nan ? b.nan_q : q;
    return q * p.t;
  }
  template <bool FAST = false>
  __device__ __forceinline__ float apply(float x, const Param& p, const Book& b) const {
    float v; int k;
    locate<FAST>(x, p, v, k);
    return decide<FAST>(x, v, b.step[b.cell[k]], p, b);
  }
  // a whole tile level by level: the NE reads of a level are issued back to back
  template <bool FAST, int NE>
  __device__ __forceinline__ void tile(const float* in, float* out, const Param& p, const Book& b) const {
    float v[NE];
    int k[NE];
    uint32_t j[NE];
    f32x2 e[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) locate<FAST>(in[i], p, v[i], k[i]);
#pragma unroll
    for (int i = 0; i < NE; ++i) j[i] = b.cell[k[i]];
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = b.step[j[i]];
#pragma unroll
    for (int i = 0; i < NE; ++i) out[i] = decide<FAST>(in[i], v[i], e[i], p, b);
  }
};

inline int make_compact_op(LutCompactOp& op, const float* thr, float eps, const void* blob, int32_t n_words, float mult,
                           float cmin, float cmax, int step_round) {
  if (!blob) return fail_arg("compact table is NULL");
  if (int rc = check_pow2(mult)) return rc;
  const int entries = table_entries(cmin, cmax);
  if (entries < 0) return fail_arg("decision table unsupported for this clip range");
  const int cw = (entries + 3) / 4;
  if (n_words < cw + 4 || n_words > cw + 2 * 256 + 2 || ((n_words - cw) & 1)) return fail_arg("n_words does not match the clip range");
  fill_lut_common(op, thr, eps, mult, cmin, cmax, step_round);
  op.blob = static_cast<const uint32_t*>(blob); op.entries = entries; op.n_words = n_words;
  op.koff = 0.5f - 2.0f * cmin; op.kmax = (float)(entries - 1);
  return 0;
}
inline size_t compact_bytes(int32_t n_words) { return (size_t)((n_words + 3) & ~3) * 4; }


}  // namespace mctq
