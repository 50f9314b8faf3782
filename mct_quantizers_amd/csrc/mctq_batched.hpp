// mctq_batched.hpp -- the batched launches of libmctq_hip.so, compiled as TWO translation units so that neither is the
// build's critical path: mctq_batched.hip (MCTQ_BATCHED_PART 1: the affine lists, mctq_fq_batched / mctq_fq_batch_*) and
// mctq_batched_lut.hip (MCTQ_BATCHED_PART 2: the decision-table LUT lists, mctq_lutt_batch_*).  The tile code and the
// policies are shared; each part instantiates only the kernels its entry points launch.
//
//
// A LIST of affine fake-quantizations in one launch.  The reference re-quantizes every wrapped layer's weights
// on every forward (pytorch/quantize_wrapper.py:228-240: one quantizer call per weight attribute), i.e. tens of
// launches per model forward, each paying its own launch cost and ~2 us of ramp/drain (profiles/r02).  Here one grid
// covers all of them.
//
// Block -> (tensor, tile) in O(1): the grid is cut into CHUNKS of 2^shift blocks, every tensor starts on a chunk
// boundary (its last chunk may hold a few idle blocks), and a byte / half-word per chunk names the tensor.  A block
// therefore reads ONE map word and ONE 64-byte descriptor by scalar loads before it can issue its data loads
// (round 2 scanned up to 32 descriptors per block: 0.6 us on every block's critical path, 17 % of the launch at 32
// tensors -- profiles/r03/batched_shape_probe_before.log).
//
// Two sources for the descriptors, one kernel body:
//   KernargSrc  descriptors + chunk map travel in the kernel arguments (<= 4 KiB: 48 tensors, 960 chunks): no device
//               table, no memcpy, legal under hipGraph capture -- mctq_fq_batched().
//   TableSrc    a caller-owned device copy of a table packed on the host by mctq_fq_batch_pack(): any number of
//               tensors in ONE launch (a whole model's weights) -- mctq_fq_batch_run().
//
// Per tile (256 * U lane-vectors of one tensor, contiguous, never crossing tensors):
//   - the tile lies inside ONE (outer, channel) row  -> the row's scale / zero point arrive by scalar loads and
//     sit in SGPRs, exactly as rows_kernel does (Linear / conv weights quantized along axis 0, per-tensor items);
//   - the tile covers exactly TWO rows (rows longer than a tile that do not divide into tiles): both rows' parameters in
//     SGPRs, a lane-vector selects by its position relative to the boundary;
//   - otherwise (rows shorter than a tile) every lane-vector finds its row with one float-reciprocal division
//     (div_small) and reads its parameters from the (L1/L2-resident) tables; vectors that straddle rows go element by
//     element.
// Arithmetic: AffineOp (mctq_kernels.hpp), the same expression as every other affine entry point.
#pragma once
#ifndef MCTQ_BATCHED_PART
#error "include through mctq_batched.hip / mctq_batched_lut.hip"
#endif
#include "mctq_kernels.hpp"

#include <vector>

using namespace mctq;

namespace mctq {

constexpr int kMaxBatch = 48;            // descriptors per kernel-argument launch
constexpr int kMaxChunksK = 960;         // chunk-map bytes in the kernel arguments
constexpr uint32_t kMaxChunksT = 16384;  // chunk-map entries of a packed table (2 bytes each)
#ifndef MCTQ_BATCH_U
#define MCTQ_BATCH_U 4                   // lane-vectors per lane and tile; 2 / 8 are timing experiments (tools/batched_unroll_probe.py)
#endif
constexpr int kBatchU = MCTQ_BATCH_U;
#ifndef MCTQ_BATCH_EXACT_RECIP
// Per-lane scales of the several-rows path inverted by recip_exact: 1 = in the one-tensor gather launch only (default: +1 % there,
// -1 % on the 54-weight list launch, profiles/r06/batched_ab.log), 2 = everywhere, 0 = nowhere (timing experiments, tools/build_variant.py)
#define MCTQ_BATCH_EXACT_RECIP 1
#endif

struct __attribute__((aligned(16))) BatchItem {   // 64 bytes
  const void* x;
  void* y;
  const float* scales;
  const int32_t* zps;
  uint32_t n;            // elements, < 2^31 - tile
  uint32_t inner;
  uint32_t channels;
  uint32_t tile_begin;   // first block of this tensor in the launch (a multiple of the chunk size)
  uint32_t tiles;        // blocks that have work; the rest of the last chunk is idle
  uint32_t reserved;
  float lo, hi;
};
static_assert(sizeof(BatchItem) == 64, "descriptor layout");

struct KernargSrc {
  BatchItem it[kMaxBatch];
  uint32_t map[kMaxChunksK / 4];         // one byte per chunk
  uint32_t shift;
  __device__ __forceinline__ uint32_t lookup(uint32_t chunk) const { return (map[chunk >> 2] >> ((chunk & 3u) * 8u)) & 0xffu; }
  __device__ __forceinline__ const void* head(uint32_t i) const { return &it[i]; }
};
static_assert(sizeof(KernargSrc) <= 4096, "kernel arguments");

struct TableSrc {
  const BatchItem* __restrict__ it;      // device
  const uint32_t* __restrict__ map;      // device, one half-word per chunk
  uint32_t shift;
  __device__ __forceinline__ uint32_t lookup(uint32_t chunk) const { return (map[chunk >> 1] >> ((chunk & 1u) * 16u)) & 0xffffu; }
  __device__ __forceinline__ const void* head(uint32_t i) const { return &it[i]; }
};

// The descriptor's pointers arrive through scalar loads, so the compiler cannot infer their address space (it would
// emit flat_load / flat_store and a vector load + readfirstlane for the row parameters): say it.  x / y are global
// memory; the parameter tables are not written by this kernel, so uniform reads of them go through the scalar cache.
#define MCTQ_GLOBAL __attribute__((address_space(1)))
#define MCTQ_CONST __attribute__((address_space(4)))

template <class TI, class TO, int NT>
struct GIO {
  typedef IO<TI, TO> io;
  __device__ __forceinline__ static typename io::VI load(const TI MCTQ_GLOBAL* p) {
    const typename io::VI MCTQ_GLOBAL* q = reinterpret_cast<const typename io::VI MCTQ_GLOBAL*>(p);
    if (NT != 0) return __builtin_nontemporal_load(q);
    return *q;
  }
  __device__ __forceinline__ static void store(TO MCTQ_GLOBAL* p, typename io::VO v) {
    typename io::VO MCTQ_GLOBAL* q = reinterpret_cast<typename io::VO MCTQ_GLOBAL*>(p);
    if (NT == 1) __builtin_nontemporal_store(v, q);
    else *q = v;
  }
};

// ---- what a tile does with its elements: a POLICY per operation ---------------------------------------------------
// init() runs after the tile's data loads have been issued (it may stage a table in LDS and synchronise the block);
// uniform(c) / lane(c) build channel c's parameter set from a wave-uniform / per-lane index; run<UNI, N>() quantizes N
// elements that share one set (UNI: the set is wave-uniform); pick() selects between two sets.
template <class TI_, class TO_, bool EXACT = false>
struct AffinePol {
  typedef TI_ TI;
  typedef TO_ TO;
  typedef AffineOp::Param Param;
  AffineOp op;
  const float MCTQ_CONST* s_uniform;     // same tables, two views: wave-uniform reads (scalar loads) ...
  const int32_t MCTQ_CONST* z_uniform;
  const float MCTQ_GLOBAL* s_lane;       // ... and per-lane reads
  const int32_t MCTQ_GLOBAL* z_lane;
  __device__ __forceinline__ explicit AffinePol(const BatchItem& it) {
    s_uniform = (const float MCTQ_CONST*)it.scales; z_uniform = (const int32_t MCTQ_CONST*)it.zps;
    s_lane = (const float MCTQ_GLOBAL*)it.scales; z_lane = (const int32_t MCTQ_GLOBAL*)it.zps;
    op.scales = nullptr; op.zps = nullptr; op.lo = it.lo; op.hi = it.hi;
  }
  __device__ __forceinline__ void pre() {}
  __device__ __forceinline__ void init(float*) {}
  __device__ __forceinline__ Param uniform(uint32_t c) const { return AffineOp::make(s_uniform[c], z_uniform ? z_uniform[c] : 0); }
  __device__ __forceinline__ Param lane(uint32_t c) const { return AffineOp::make(s_lane[c], z_lane ? z_lane[c] : 0); }
  // The parameter sets of a tile's U lane-vectors in two steps (round 6): raws() issues all table reads back to back and decides
  // ONCE per wave whether the lanes may invert their scales with recip_exact (5 VALU instructions instead of the 11 of the IEEE
  // expansion); finish() builds one set right where it is used, so that no reciprocal is kept alive across the tile.
  struct Raw { float s, zf; };
  template <int U>
  __device__ __forceinline__ bool raws(const uint32_t (&c)[U], Raw (&r)[U]) const {
    float sv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { sv[u] = s_lane[c[u]]; r[u].s = sv[u]; r[u].zf = z_lane ? (float)z_lane[c[u]] : 0.0f; }
    return EXACT && __builtin_amdgcn_ballot_w64(!recip_all_in_range(sv)) == 0;
  }
  __device__ __forceinline__ static Param finish(const Raw& r, bool exact /* wave-uniform */) {
    Param p; p.s = r.s; p.zf = r.zf;
    p.inv = exact ? recip_exact(r.s) : 1.0f / r.s;
    return p;
  }
  template <bool UNI, int N>
  __device__ __forceinline__ void run(const float* in, float* out, const Param& p) const {
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = op.apply(in[j], p, NoBook());
  }
  __device__ __forceinline__ static Param pick(bool first, const Param& a, const Param& b) {
    Param p; p.s = first ? a.s : b.s; p.inv = first ? a.inv : b.inv; p.zf = first ? a.zf : b.zf; return p;
  }
};

// LUT quantizers with a decision table (LutTableOp, mctq_kernels.hpp): 96-byte descriptor = the 64 bytes above
// (scales -> thresholds or NULL, zps -> the decision table, lo / hi unused) + the codebook's constants.
struct __attribute__((aligned(16))) LutBatchItem {
  BatchItem b;
  float mult, cmin, cmax;
  float eps;             // per channel: divisor fl32(thresholds[c] + eps)
  float thr_div, thr_mul;   // per tensor (b.scales == NULL)
  int32_t entries, step_round;
};
static_assert(sizeof(LutBatchItem) == 96, "LUT descriptor layout");

template <class TI_>
struct LutPol {
  typedef TI_ TI;
  typedef float TO;
  typedef LutCommon::Param Param;
  LutTableOp op;
  LutTableBook book;
  const float MCTQ_CONST* t_uniform;
  const float MCTQ_GLOBAL* t_lane;
  const f32x2 MCTQ_GLOBAL* table;
  float thr_div, thr_mul;
  __device__ __forceinline__ explicit LutPol(const LutBatchItem& it) {
    t_uniform = (const float MCTQ_CONST*)it.b.scales; t_lane = (const float MCTQ_GLOBAL*)it.b.scales;
    table = (const f32x2 MCTQ_GLOBAL*)it.b.zps;
    thr_div = it.thr_div; thr_mul = it.thr_mul;
    op.thr = nullptr; op.eps = it.eps; op.mult = it.mult; op.inv_mult = 1.0f / it.mult; op.cmin = it.cmin; op.cmax = it.cmax;
    op.step_round = it.step_round; op.table = nullptr; op.entries = it.entries;
    op.koff = 0.5f - 2.0f * it.cmin; op.kmax = (float)(it.entries - 1);
  }
  // the table is requested BEFORE the tile's data loads and written to LDS after them: vector loads return in order, so
  // a table read queued behind the data loads could not be staged before all of them had landed (LutTableOp::prefetch)
  f32x2 pf[8];
  __device__ __forceinline__ void pre() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = (int)threadIdx.x + i * kThreads;
      if (j <= op.entries) pf[i] = table[j];
    }
  }
  __device__ __forceinline__ void init(float* smem) {      // every thread of the block: the table goes to LDS
    f32x2* dst = reinterpret_cast<f32x2*>(smem);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = (int)threadIdx.x + i * kThreads;
      if (j <= op.entries) dst[j] = pf[i];
    }
    __syncthreads();
    book.tab = dst; book.nan_q = dst[op.entries].x;
  }
  __device__ __forceinline__ Param uniform(uint32_t c) const {
    if (!t_uniform) return LutCommon::make(thr_div, thr_mul, op.mult);
    const float t = t_uniform[c];
    return LutCommon::make(t + op.eps, t, op.mult);
  }
  __device__ __forceinline__ Param lane(uint32_t c) const {
    if (!t_lane) return LutCommon::make(thr_div, thr_mul, op.mult);
    const float t = t_lane[c];
    return LutCommon::make(t + op.eps, t, op.mult);
  }
  typedef Param Raw;                       // (nothing to gain here: the LUT sets carry their own divisions)
  template <int U>
  __device__ __forceinline__ bool raws(const uint32_t (&c)[U], Raw (&r)[U]) const {
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = lane(c[u]);
    return false;
  }
  __device__ __forceinline__ static Param finish(const Raw& r, bool) { return r; }
  template <bool UNI, int N>
  __device__ __forceinline__ void run(const float* in, float* out, const Param& p) const {
    // the exact reciprocal division (LutCommon::divide_fast) is per element: it works with per-lane divisors too, as long
    // as every active lane's divisor qualifies (wave-uniform test either way)
    const bool fast = UNI ? __builtin_amdgcn_readfirstlane((int)LutCommon::can_fast(p)) != 0
                          : __builtin_amdgcn_ballot_w64(!LutCommon::can_fast(p)) == 0;
    if (fast) op.template tile<true, N>(in, out, p, book);
    else op.template tile<false, N>(in, out, p, book);
  }
  __device__ __forceinline__ static Param pick(bool first, const Param& a, const Param& b) {
    Param p; p.d = first ? a.d : b.d; p.t = first ? a.t : b.t; p.r = first ? a.r : b.r; p.ds = first ? a.ds : b.ds; return p;
  }
};

template <bool FULL, class Pol, int U, int NT>
__device__ __forceinline__ void batched_tile(const BatchItem& it, Pol& pol, float* smem, const uint32_t e0, const uint32_t count) {
  typedef typename Pol::TI TI;
  typedef typename Pol::TO TO;
  typedef typename Pol::Param Param;
  typedef IO<TI, TO> io;
  typedef GIO<TI, TO, NT> gio;
  constexpr uint32_t N = io::N;
  const TI MCTQ_GLOBAL* __restrict__ x = (const TI MCTQ_GLOBAL*)it.x;
  TO MCTQ_GLOBAL* __restrict__ y = (TO MCTQ_GLOBAL*)it.y;
  const uint32_t inner = it.inner, channels = it.channels;

  // (an op's table reads first,) then the data loads; the LDS staging / row search / parameter fetch run under their latency
  pol.pre();
  __builtin_amdgcn_sched_barrier(0);
  typename io::VI v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    if (FULL || off + N <= count) v[u] = gio::load(x + e0 + off);
  }
  __builtin_amdgcn_sched_barrier(0);
  pol.init(smem);

  uint32_t row0 = 0, rem0 = e0;
  if (channels > 1) {                                        // uniform; per-tensor items have one row
    row0 = e0 / inner;
    rem0 = e0 - row0 * inner;
  }
  const uint32_t c0 = row0 < channels ? row0 : row0 % channels;     // outer == 1 (weights along axis 0): no modulo

  if (channels == 1 || rem0 + count <= inner) {
    // ---- one row: parameters in SGPRs ----
    const Param p = pol.uniform(c0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t off = (u * kThreads + threadIdx.x) * N;
      if (FULL || off + N <= count) {
        float in[N], out[N];
        io::unpack(v[u], in);
        pol.template run<true, (int)N>(in, out, p);
        gio::store(y + e0 + off, io::pack(out));
      } else {
        for (uint32_t j = 0; j < N && off + j < count; ++j) {
          float in1[1] = {(float)x[e0 + off + j]}, out1[1];
          pol.template run<true, 1>(in1, out1, p);
          y[e0 + off + j] = (TO)out1[0];
        }
      }
    }
    return;
  }

  if (rem0 + count <= 2 * inner) {
    // ---- exactly two rows (rows at least half a tile long): both parameter sets in SGPRs,
    //      a lane-vector picks by its position relative to the row boundary -- no per-lane division or table read ----
    const uint32_t c1 = c0 + 1 == channels ? 0 : c0 + 1;
    const Param p0 = pol.uniform(c0), p1 = pol.uniform(c1);
    const uint32_t bnd = inner - rem0;                       // elements of the tile that belong to row0 (0 < bnd < count)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t off = (u * kThreads + threadIdx.x) * N;
      if (!FULL && off >= count) continue;
      if (FULL || off + N <= count) {
        float in[N], out[N];
        io::unpack(v[u], in);
        if (off + N <= bnd || off >= bnd) {
          pol.template run<false, (int)N>(in, out, Pol::pick(off + N <= bnd, p0, p1));
        } else {                                              // the vector straddles the boundary (inner % N != 0)
#pragma unroll
          for (uint32_t j = 0; j < N; ++j) pol.template run<false, 1>(in + j, out + j, off + j < bnd ? p0 : p1);
        }
        gio::store(y + e0 + off, io::pack(out));
      } else {
        for (uint32_t j = 0; j < N && off + j < count; ++j) {
          float in1[1] = {(float)x[e0 + off + j]}, out1[1];
          pol.template run<false, 1>(in1, out1, off + j < bnd ? p0 : p1);
          y[e0 + off + j] = (TO)out1[0];
        }
      }
    }
    return;
  }

  // ---- several rows in the tile (inner < tile / 2): per lane-vector parameters ----
  // Pass 1 finds every vector's row (positions inside the tile are < 2^24: one float multiply + two integer
  // corrections) and builds its parameter set (table reads issued back to back); pass 2 applies.
  const float r_inner = 1.0f / (float)inner;
  const uint32_t nrows = (rem0 + count - 1) / inner + 1;      // uniform
  const bool wraps = c0 + nrows > channels;                   // uniform: some row of the tile starts a new outer slice
  const bool small_c = (uint64_t)channels + nrows < (1u << 24);
  const float r_channels = 1.0f / (float)channels;
  uint32_t cc[U], rr[U];
  typename Pol::Raw pv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    const uint32_t pos = rem0 + ((FULL || off < count) ? off : 0);
    const uint32_t lrow = div_small(pos, inner, r_inner);
    rr[u] = pos - lrow * inner;
    uint32_t c = c0 + lrow;
    if (wraps) c = small_c ? c - div_small(c, channels, r_channels) * channels : c % channels;
    cc[u] = c;
  }
  const bool exact = pol.template raws<U>(cc, pv);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    if (!FULL && off >= count) continue;
    uint32_t rem = rr[u], c = cc[u];
    if (FULL || off + N <= count) {
      float in[N], out[N];
      io::unpack(v[u], in);
      if (rem + N <= inner) {                               // the vector lies in one row
        pol.template run<false, (int)N>(in, out, Pol::finish(pv[u], exact));
      } else if (inner >= N) {                              // it crosses exactly one row boundary: two sets, chosen per element
        const uint32_t split = inner - rem;
        const Param pa = Pol::finish(pv[u], exact), pb = pol.lane(c + 1 == channels ? 0 : c + 1);
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) pol.template run<false, 1>(in + j, out + j, Pol::pick(j < split, pa, pb));
      } else {
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
          pol.template run<false, 1>(in + j, out + j, pol.lane(c));
          if (++rem == inner) { rem = 0; if (++c == channels) c = 0; }
        }
      }
      gio::store(y + e0 + off, io::pack(out));
    } else {
      for (uint32_t j = 0; j < N && off + j < count; ++j) {
        float in1[1] = {(float)x[e0 + off + j]}, out1[1];
        pol.template run<false, 1>(in1, out1, pol.lane(c));
        y[e0 + off + j] = (TO)out1[0];
        if (++rem == inner) { rem = 0; if (++c == channels) c = 0; }
      }
    }
  }
}

// Block -> its descriptor: one map word, then the 64-byte head in ONE scalar load (left to itself the compiler fetches
// tile_begin / tiles first and the pointers behind the idle-block test: one more dependent round trip in front of the
// data loads).
template <class Src>
__device__ __forceinline__ bool batched_head(const Src& src, BatchItem& out, uint32_t& index) {
  index = src.lookup(blockIdx.x >> src.shift);
  typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
  union { u32x16 w; BatchItem it; } d;
  d.w = *reinterpret_cast<const u32x16*>(src.head(index));
  asm volatile("" : "+s"(d.w));
  out = d.it;
  return blockIdx.x - out.tile_begin < out.tiles;            // false: idle block at the end of the tensor's last chunk
}

template <class TI, class TO, int U, int NT, class Src>
__device__ __forceinline__ void batched_block(const Src& src) {
  constexpr uint32_t TILE = kThreads * U * IO<TI, TO>::N;
  BatchItem it;
  uint32_t index;
  if (!batched_head(src, it, index)) return;
  const uint32_t e0 = (blockIdx.x - it.tile_begin) * TILE;
  const uint32_t left = it.n - e0;
  typedef AffinePol<TI, TO, MCTQ_BATCH_EXACT_RECIP == 2> Pol;
  Pol pol(it);
  if (left >= TILE) batched_tile<true, Pol, U, NT>(it, pol, nullptr, e0, TILE);   // wave-uniform: straight-line code
  else batched_tile<false, Pol, U, NT>(it, pol, nullptr, e0, left);
}

template <class TI, class TO, int U, int NT>
__global__ __launch_bounds__(kThreads) void batched_kernel(const KernargSrc src) {
  batched_block<TI, TO, U, NT>(src);
}

// Three scalar arguments (not a struct): with -amdgpu-kernarg-preload-count they arrive in SGPRs at wave start, so the
// map word can be requested in the block's first instructions instead of behind a kernel-argument load.
template <class TI, class TO, int U, int NT>
__global__ __launch_bounds__(kThreads) void batched_table_kernel(const BatchItem* __restrict__ it,
                                                                 const uint32_t* __restrict__ map, uint32_t shift) {
  TableSrc src;
  src.it = it; src.map = map; src.shift = shift;
  batched_block<TI, TO, U, NT>(src);
}

// ONE tensor through the same tile code, its descriptor in nine scalar kernel arguments (all preloaded into SGPRs: no
// map word, no descriptor load): the single-tensor entry point's route for float32 rows that are neither long and
// vector-divisible (rows_kernel) nor the fastest axis -- the per-lane-vector parameter path beats window_kernel's LDS
// window + block barrier there (mctq_affine.hip).
template <class TI, class TO, int U, int NT>
__global__ __launch_bounds__(kThreads) void batched_one_kernel(const TI* __restrict__ x, TO* __restrict__ y,
                                                               const float* __restrict__ scales,
                                                               const int32_t* __restrict__ zps, uint32_t n, uint32_t inner,
                                                               uint32_t channels, float lo, float hi) {
  constexpr uint32_t TILE = kThreads * U * IO<TI, TO>::N;
  BatchItem it;
  it.x = x; it.y = y; it.scales = scales; it.zps = zps; it.n = n; it.inner = inner; it.channels = channels;
  it.tile_begin = 0; it.tiles = gridDim.x; it.reserved = 0; it.lo = lo; it.hi = hi;
  const uint32_t e0 = blockIdx.x * TILE;
  const uint32_t left = n - e0;
  typedef AffinePol<TI, TO, MCTQ_BATCH_EXACT_RECIP != 0> Pol;
  Pol pol(it);
  if (left >= TILE) batched_tile<true, Pol, U, NT>(it, pol, nullptr, e0, TILE);
  else batched_tile<false, Pol, U, NT>(it, pol, nullptr, e0, left);
}

// The same grid for LUT quantizers with a decision table: all LUT weights of a model, or a group of LUT activation
// batches, in one launch.  Output float32 (the reference's chain promotes), table staged in dynamic LDS per block.
struct LutTableSrc {
  const LutBatchItem* __restrict__ it;
  const uint32_t* __restrict__ map;
  uint32_t shift;
  __device__ __forceinline__ uint32_t lookup(uint32_t chunk) const { return (map[chunk >> 1] >> ((chunk & 1u) * 16u)) & 0xffffu; }
  __device__ __forceinline__ const void* head(uint32_t i) const { return &it[i].b; }
};

template <class TI, int U, int NT>
__global__ __launch_bounds__(kThreads) void batched_lut_table_kernel(const LutBatchItem* __restrict__ items,
                                                                     const uint32_t* __restrict__ map, uint32_t shift) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr uint32_t TILE = kThreads * U * IO<TI, float>::N;
  LutTableSrc src;
  src.it = items; src.map = map; src.shift = shift;
  BatchItem it;
  uint32_t index;
  if (!batched_head(src, it, index)) return;
  LutBatchItem full;
  full.b = it;
  {                                                          // the codebook constants: 32 more bytes, off the critical path
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    union { u32x8 w; float f[8]; int32_t i[8]; } t;
    t.w = *reinterpret_cast<const u32x8*>(reinterpret_cast<const uint8_t*>(&items[index]) + sizeof(BatchItem));
    full.mult = t.f[0]; full.cmin = t.f[1]; full.cmax = t.f[2]; full.eps = t.f[3]; full.thr_div = t.f[4]; full.thr_mul = t.f[5];
    full.entries = t.i[6]; full.step_round = t.i[7];
  }
  const uint32_t e0 = (blockIdx.x - it.tile_begin) * TILE;
  const uint32_t left = it.n - e0;
  LutPol<TI> pol(full);
  if (left >= TILE) batched_tile<true, LutPol<TI>, U, NT>(it, pol, smem, e0, TILE);
  else batched_tile<false, LutPol<TI>, U, NT>(it, pol, smem, e0, left);
}

template <class TI, class TO>
static constexpr uint32_t batch_tile_elems() { return kThreads * kBatchU * IO<TI, TO>::N; }
static uint32_t tile_elems_of(int dtype) {
  return dtype == MCTQ_DT_F32 ? batch_tile_elems<float, float>() : batch_tile_elems<_Float16, _Float16>();
}

#if MCTQ_BATCHED_PART == 1
template <class TI, class TO>
static int launch_batch(const KernargSrc& src, uint32_t grid, int64_t out_bytes, hipStream_t st) {
  MCTQ_WITH_MODE(nt_mode(out_bytes), {
    hipLaunchKernelGGL((batched_kernel<TI, TO, kBatchU, NT>), dim3(grid), dim3(kThreads), 0, st, src);
    note<AffineOp, TI, TO>("batched_kernel", kBatchU, NT);
  });
  return check_launch("batched launch");
}
template <class TI, class TO>
static int launch_batch(const TableSrc& src, uint32_t grid, int64_t out_bytes, hipStream_t st) {
  MCTQ_WITH_MODE(nt_mode(out_bytes), {
    hipLaunchKernelGGL((batched_table_kernel<TI, TO, kBatchU, NT>), dim3(grid), dim3(kThreads), 0, st, src.it, src.map, src.shift);
    note<AffineOp, TI, TO>("batched_kernel<table>", kBatchU, NT);
  });
  return check_launch("batched launch");
}
template <class Src>
static int launch_batch_dt(int dt, const Src& src, uint32_t grid, int64_t out_bytes, hipStream_t st) {
  if (dt == MCTQ_DT_F32) return launch_batch<float, float>(src, grid, out_bytes, st);
  if (dt == MCTQ_DT_F16) return launch_batch<_Float16, _Float16>(src, grid, out_bytes, st);
  return launch_batch<__bf16, __bf16>(src, grid, out_bytes, st);
}

#endif
#if MCTQ_BATCHED_PART == 2
template <class TI>
static int launch_lut_batch(const LutTableSrc& src, uint32_t grid, size_t lds, int64_t out_bytes, hipStream_t st) {
  MCTQ_WITH_MODE(nt_mode(out_bytes), {
    hipLaunchKernelGGL((batched_lut_table_kernel<TI, kBatchU, NT>), dim3(grid), dim3(kThreads), lds, st, src.it, src.map, src.shift);
    note<LutTableOp, TI, float>("batched_lut_kernel<table>", kBatchU, NT);
  });
  return check_launch("batched LUT launch");
}

#endif

#if MCTQ_BATCHED_PART == 1
// float32 [rows][inner] with per-channel parameters, x / y 16-byte aligned, n < 2^31 - tile (checked by the caller)
int fq_gather_one_f32(const void* x, void* y, int64_t outer, int64_t channels, int64_t inner, const float* scales,
                      const int32_t* zps, int32_t qmin, int32_t qmax, hipStream_t st) {
  const int64_t n = outer * channels * inner;
  if (n == 0) return 0;
  constexpr uint32_t tile_e = kThreads * kBatchU * 4;
  const uint32_t grid = (uint32_t)((n + tile_e - 1) / tile_e);
  MCTQ_WITH_MODE(nt_mode(n * 4), {
    hipLaunchKernelGGL((batched_one_kernel<float, float, kBatchU, NT>), dim3(grid), dim3(kThreads), 0, st,
                       static_cast<const float*>(x), static_cast<float*>(y), scales, zps, (uint32_t)n, (uint32_t)inner,
                       (uint32_t)channels, (float)qmin, (float)qmax);
    note<AffineOp, float, float>("gather_kernel", kBatchU, NT);
  });
  return check_launch("gather launch");
}

// ---- host: which tensors one grid can take, and how the grid is cut ---------------------------------------
static int validate_items(const mctq_fq_item* items, int32_t n_items) {
  if (n_items < 0) return fail_arg("n_items < 0");
  if (n_items > 0 && !items) return fail_arg("items is NULL");
  for (int32_t k = 0; k < n_items; ++k) {
    const mctq_fq_item& d = items[k];
    if (d.outer < 0 || d.channels < 0 || d.inner < 0) return fail_arg("negative extent");
    if (d.quant_min > d.quant_max) return fail_arg("quant_min > quant_max");
    if (d.outer * d.channels * d.inner > 0 && (!d.x || !d.y || !d.scales)) return fail_arg("NULL pointer");
    if (d.dtype != MCTQ_DT_F32 && d.dtype != MCTQ_DT_F16 && d.dtype != MCTQ_DT_BF16 && d.dtype != MCTQ_DT_F64)
      return fail_arg("unknown dtype");
    if (d.dtype == MCTQ_DT_F64 && (d.flags & MCTQ_FQ_ITEM_PER_TENSOR) && !d.zero_points)
      return fail_arg("a float64 per-tensor item needs a zero_points pointer");
  }
  return 0;
}

// One grid takes: float32 / float16 / bfloat16 tensors with 16-byte aligned x and y, fewer than 2^31 elements, and
// rows of at least 32 elements (or one parameter set for the whole tensor) -- or any rows when the tensor is small
// (<= 2^20 elements: depthwise 3x3 weights, [C, 1, 3, 3], are tiny and many; one more launch each would cost more
// than the element-by-element path of their row-straddling vectors).  LARGE channel-last layouts keep their own
// kernels (lastaxis / window), float64 its own path: launched one by one on the same stream.
static bool batchable(const mctq_fq_item& d) {
  if (d.dtype == MCTQ_DT_F64) return false;
  const int64_t n = d.outer * d.channels * d.inner;
  const bool aligned = (((uintptr_t)d.x | (uintptr_t)d.y) & 15u) == 0;
  if (!aligned || n >= (1ll << 31) - (int64_t)tile_elems_of(d.dtype)) return false;
  if (d.channels > 0x7fffffffLL || d.inner > 0x7fffffffLL) return false;
  return d.outer * d.channels == 1 || d.inner >= 32 || n <= (1ll << 20);
}

static int launch_single(const mctq_fq_item& d, void* stream) {
  const int64_t n = d.outer * d.channels * d.inner;
  if (n == 0) return 0;
  if ((d.flags & MCTQ_FQ_ITEM_PER_TENSOR) && d.zero_points)
    return mctq_fq_per_tensor_tqp(d.x, d.y, n, d.dtype, d.scales, d.zero_points, d.quant_min, d.quant_max, stream);
  if ((d.flags & MCTQ_FQ_ITEM_PER_TENSOR) && d.dtype == MCTQ_DT_F64)
    return fail_arg("a float64 per-tensor item needs a zero_points pointer");
  return mctq_fq_per_channel(d.x, d.y, d.outer, d.channels, d.inner, d.dtype, d.scales, d.zero_points, d.quant_min,
                             d.quant_max, stream);
}

static void fill_item(BatchItem& b, const mctq_fq_item& d, uint32_t tiles) {
  const int64_t n = d.outer * d.channels * d.inner;
  b.x = d.x; b.y = d.y; b.scales = d.scales; b.zps = d.zero_points;
  b.n = (uint32_t)n;
  const bool one_row = d.outer * d.channels == 1;
  b.inner = one_row ? (uint32_t)n : (uint32_t)d.inner;
  b.channels = one_row ? 1u : (uint32_t)d.channels;
  b.tile_begin = 0; b.tiles = tiles; b.reserved = 0;
  b.lo = (float)d.quant_min; b.hi = (float)d.quant_max;
}

#endif
// Smallest chunk shift with sum over tensors of ceil(tiles / 2^shift) <= max_chunks.
static uint32_t chunk_shift(const uint32_t* tiles, int n, uint32_t max_chunks) {
  for (uint32_t s = 0;; ++s) {
    uint64_t chunks = 0;
    for (int k = 0; k < n; ++k) chunks += ((uint64_t)tiles[k] + (1u << s) - 1) >> s;
    if (chunks <= max_chunks) return s;
  }
}
#if MCTQ_BATCHED_PART == 1

// ---- packed table (mctq_fq_batch_pack / mctq_fq_batch_run) -------------------------------------------------
constexpr uint32_t kTableMagic = 0x4d435451u;   // "MCTQ"
struct TableGroup { uint32_t dtype, items_off, map_off, n_items, grid, shift; int64_t out_bytes; };
struct TableHeader {
  uint32_t magic, version, total_bytes, n_groups, n_singles, singles_off, pad0, pad1;
  TableGroup g[3];
};
static_assert(sizeof(TableHeader) == 128, "table header layout");

}  // namespace mctq

extern "C" {

int mctq_fq_batched(const mctq_fq_item* items, int32_t n_items, void* stream) {
  // validate everything before the first launch: a bad descriptor must not leave the list half done
  if (int rc = validate_items(items, n_items)) return rc;
  hipStream_t st = (hipStream_t)stream;
  bool singles = false;
  for (int dt = MCTQ_DT_F32; dt <= MCTQ_DT_BF16; ++dt) {
    const uint32_t tile_e = tile_elems_of(dt);
    const int64_t esz = dt == MCTQ_DT_F32 ? 4 : 2;
    int32_t k = 0;
    while (k < n_items) {
      // next group: up to kMaxBatch batchable tensors of this storage type whose chunk map fits the kernel arguments
      KernargSrc a;
      uint32_t tiles[kMaxBatch];
      int m = 0;
      int64_t out_bytes = 0;
      uint64_t total_tiles = 0;
      for (; k < n_items && m < kMaxBatch; ++k) {
        const mctq_fq_item& d = items[k];
        if (d.dtype != dt) { if (d.dtype == MCTQ_DT_F64 && dt == MCTQ_DT_F32) singles = true; continue; }
        const int64_t n = d.outer * d.channels * d.inner;
        if (n == 0) continue;
        if (!batchable(d)) { singles = true; continue; }
        const uint32_t t = (uint32_t)((n + tile_e - 1) / tile_e);
        if (total_tiles + t + ((uint64_t)kMaxBatch << 21) > 0x7fffffffull) break;    // grid limit (incl. chunk padding)
        fill_item(a.it[m], d, t);
        tiles[m++] = t;
        total_tiles += t;
        out_bytes += n * esz;
      }
      if (m == 0) continue;
      a.shift = chunk_shift(tiles, m, kMaxChunksK);
      memset(a.map, 0, sizeof(a.map));
      uint32_t chunk = 0;
      uint8_t* map = reinterpret_cast<uint8_t*>(a.map);
      for (int j = 0; j < m; ++j) {
        a.it[j].tile_begin = chunk << a.shift;
        const uint32_t c = (tiles[j] + (1u << a.shift) - 1) >> a.shift;
        memset(map + chunk, j, c);
        chunk += c;
      }
      if (int rc = launch_batch_dt(dt, a, chunk << a.shift, out_bytes, st)) return rc;
    }
  }
  if (singles)
    for (int32_t k = 0; k < n_items; ++k)                     // what one grid cannot take: one launch each, same stream
      if (!batchable(items[k]))
        if (int rc = launch_single(items[k], stream)) return rc;
  return 0;
}

int64_t mctq_fq_batch_pack(const mctq_fq_item* items, int32_t n_items, void* host_table, int64_t capacity) {
  if (int rc = validate_items(items, n_items)) return rc;
  // sizes first
  uint32_t count[3] = {0, 0, 0}, n_singles = 0;
  for (int32_t k = 0; k < n_items; ++k) {
    const mctq_fq_item& d = items[k];
    if (d.outer * d.channels * d.inner == 0) continue;
    if (batchable(d)) ++count[d.dtype - MCTQ_DT_F32]; else ++n_singles;
  }
  for (int g = 0; g < 3; ++g)
    if (count[g] > 0xffffu) return fail_arg("more than 65535 tensors of one storage type");
  uint32_t shift[3] = {0, 0, 0}, chunks[3] = {0, 0, 0};
  uint32_t* tl = n_items ? (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n_items) : nullptr;
  if (n_items && !tl) return fail_arg("out of host memory");
  for (int g = 0; g < 3; ++g) {
    if (!count[g]) continue;
    const uint32_t tile_e = tile_elems_of(MCTQ_DT_F32 + g);
    int m = 0;
    uint64_t total = 0;
    for (int32_t k = 0; k < n_items; ++k) {
      const mctq_fq_item& d = items[k];
      const int64_t n = d.outer * d.channels * d.inner;
      if (d.dtype != MCTQ_DT_F32 + g || n == 0 || !batchable(d)) continue;
      tl[m] = (uint32_t)((n + tile_e - 1) / tile_e);
      total += tl[m++];
    }
    shift[g] = chunk_shift(tl, m, kMaxChunksT);
    for (int j = 0; j < m; ++j) chunks[g] += (tl[j] + (1u << shift[g]) - 1) >> shift[g];
    if (total + ((uint64_t)m << shift[g]) > 0x7fffffffull) { free(tl); return fail_arg("too many tiles for one launch"); }
  }
  auto align16 = [](uint64_t v) { return (v + 15u) & ~(uint64_t)15u; };
  uint64_t off = sizeof(TableHeader);
  uint64_t items_off[3], map_off[3];
  for (int g = 0; g < 3; ++g) {
    items_off[g] = off; off += (uint64_t)count[g] * sizeof(BatchItem);
    map_off[g] = off; off = align16(off + (uint64_t)chunks[g] * 2u + 2u);
  }
  const uint64_t singles_off = off;
  off = align16(off + (uint64_t)n_singles * sizeof(mctq_fq_item));
  if (off > 0x7fffffffull) { free(tl); return fail_arg("table too large"); }
  if (!host_table || capacity < (int64_t)off) { free(tl); return (int64_t)off; }     // size query

  uint8_t* base = static_cast<uint8_t*>(host_table);
  memset(base, 0, (size_t)off);
  TableHeader* h = reinterpret_cast<TableHeader*>(base);
  h->magic = kTableMagic; h->version = MCTQ_ABI_VERSION; h->total_bytes = (uint32_t)off;
  h->n_singles = n_singles; h->singles_off = (uint32_t)singles_off;
  for (int g = 0; g < 3; ++g) {
    if (!count[g]) continue;
    TableGroup& tg = h->g[h->n_groups++];
    tg.dtype = MCTQ_DT_F32 + g; tg.items_off = (uint32_t)items_off[g]; tg.map_off = (uint32_t)map_off[g];
    tg.n_items = count[g]; tg.shift = shift[g]; tg.out_bytes = 0;
    const uint32_t tile_e = tile_elems_of(tg.dtype);
    BatchItem* bi = reinterpret_cast<BatchItem*>(base + items_off[g]);
    uint16_t* map = reinterpret_cast<uint16_t*>(base + map_off[g]);
    uint32_t chunk = 0;
    int m = 0;
    for (int32_t k = 0; k < n_items; ++k) {
      const mctq_fq_item& d = items[k];
      const int64_t n = d.outer * d.channels * d.inner;
      if (d.dtype != (int32_t)tg.dtype || n == 0 || !batchable(d)) continue;
      const uint32_t t = (uint32_t)((n + tile_e - 1) / tile_e);
      fill_item(bi[m], d, t);
      bi[m].tile_begin = chunk << tg.shift;
      const uint32_t c = (t + (1u << tg.shift) - 1) >> tg.shift;
      for (uint32_t j = 0; j < c; ++j) map[chunk + j] = (uint16_t)m;
      chunk += c;
      tg.out_bytes += n * (tg.dtype == MCTQ_DT_F32 ? 4 : 2);
      ++m;
    }
    tg.grid = chunk << tg.shift;
  }
  mctq_fq_item* sg = reinterpret_cast<mctq_fq_item*>(base + singles_off);
  uint32_t s = 0;
  for (int32_t k = 0; k < n_items; ++k)
    if (items[k].outer * items[k].channels * items[k].inner != 0 && !batchable(items[k])) sg[s++] = items[k];
  free(tl);
  return (int64_t)off;
}

int mctq_fq_batch_run(const void* host_table, const void* device_table, void* stream) {
  if (!host_table) return fail_arg("host_table is NULL");
  const uint8_t* base = static_cast<const uint8_t*>(host_table);
  const TableHeader* h = reinterpret_cast<const TableHeader*>(base);
  if (h->magic != kTableMagic || h->version != (uint32_t)MCTQ_ABI_VERSION || h->n_groups > 3)
    return fail_arg("not a table packed by this library version (mctq_fq_batch_pack)");
  if (h->n_groups && (!device_table || ((uintptr_t)device_table & 15u))) return fail_arg("device_table is NULL or not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const uint8_t* dev = static_cast<const uint8_t*>(device_table);
  for (uint32_t g = 0; g < h->n_groups; ++g) {
    const TableGroup& tg = h->g[g];
    TableSrc src;
    src.it = reinterpret_cast<const BatchItem*>(dev + tg.items_off);
    src.map = reinterpret_cast<const uint32_t*>(dev + tg.map_off);
    src.shift = tg.shift;
    if (int rc = launch_batch_dt((int)tg.dtype, src, tg.grid, tg.out_bytes, st)) return rc;
  }
  const mctq_fq_item* sg = reinterpret_cast<const mctq_fq_item*>(base + h->singles_off);
  for (uint32_t s = 0; s < h->n_singles; ++s)
    if (int rc = launch_single(sg[s], stream)) return rc;
  return 0;
}

}  // extern "C"
#endif  // affine part

#if MCTQ_BATCHED_PART == 2
}  // namespace mctq

extern "C" {

// ---- LUT quantizers with a decision table: the same table-driven grid -------------------------------------------
namespace {

int validate_lut_items(const mctq_lut_item* items, int32_t n_items) {
  if (n_items < 0) return fail_arg("n_items < 0");
  if (n_items > 0 && !items) return fail_arg("items is NULL");
  for (int32_t k = 0; k < n_items; ++k) {
    const mctq_lut_item& d = items[k];
    if (d.outer < 0 || d.channels < 0 || d.inner < 0) return fail_arg("negative extent");
    if (d.outer * d.channels * d.inner > 0 && (!d.x || !d.y || !d.table)) return fail_arg("NULL pointer");
    if (d.dtype != MCTQ_DT_F32 && d.dtype != MCTQ_DT_F16 && d.dtype != MCTQ_DT_BF16) return fail_arg("unknown dtype");
    if (d.step_round != 0 && d.step_round != MCTQ_DT_F16 && d.step_round != MCTQ_DT_BF16) return fail_arg("bad step_round");
    if (d.step_round != 0 && d.thresholds) return fail_arg("step_round is a per-tensor option");
    if (check_pow2(d.mult)) return MCTQ_E_ARG;
    if (d.entries != table_entries(d.clip_min, d.clip_max)) return fail_arg("entries does not match the clip range");
  }
  return 0;
}

uint32_t lut_tile_elems(int dtype) {      // 256 lanes x 4 vectors x N, N = 16 B / 4 B (the float32 output decides)
  (void)dtype;
  return kThreads * kBatchU * 4;
}

bool lut_batchable(const mctq_lut_item& d) {
  const int64_t n = d.outer * d.channels * d.inner;
  const uintptr_t xa = d.dtype == MCTQ_DT_F32 ? 15u : 7u;    // a lane-vector: 4 elements in, 4 float32 out
  if (((uintptr_t)d.x & xa) || ((uintptr_t)d.y & 15u)) return false;
  if (n >= (1ll << 31) - (int64_t)lut_tile_elems(d.dtype) || d.channels > 0x7fffffffLL || d.inner > 0x7fffffffLL) return false;
  return !d.thresholds || d.outer * d.channels == 1 || d.inner >= 32 || n <= (1ll << 20);
}

int launch_lut_single(const mctq_lut_item& d, void* stream) {
  const int64_t n = d.outer * d.channels * d.inner;
  if (n == 0) return 0;
  if (!d.thresholds)
    return mctq_lutt_per_tensor(d.x, d.y, n, d.dtype, d.step_round, d.thr_div, d.thr_mul, d.table, d.entries, d.mult,
                                d.clip_min, d.clip_max, stream);
  return mctq_lutt_per_channel(d.x, d.y, d.outer, d.channels, d.inner, d.dtype, d.thresholds, d.eps, d.table, d.entries,
                               d.mult, d.clip_min, d.clip_max, stream);
}

constexpr uint32_t kLutTableMagic = 0x4d43544cu;   // "MCTL"
struct LutTableGroup { uint32_t dtype, items_off, map_off, n_items, grid, shift, lds_bytes, pad; int64_t out_bytes; };
struct LutTableHeader {
  uint32_t magic, version, total_bytes, n_groups, n_singles, singles_off, pad0, pad1;
  LutTableGroup g[3];
};
static_assert(sizeof(LutTableHeader) == 32 + 3 * 40, "LUT table header layout");

}  // namespace

int64_t mctq_lutt_batch_pack(const mctq_lut_item* items, int32_t n_items, void* host_table, int64_t capacity) {
  if (int rc = validate_lut_items(items, n_items)) return rc;
  uint32_t count[3] = {0, 0, 0}, n_singles = 0, shift[3] = {0, 0, 0}, chunks[3] = {0, 0, 0};
  for (int32_t k = 0; k < n_items; ++k) {
    const mctq_lut_item& d = items[k];
    if (d.outer * d.channels * d.inner == 0) continue;
    if (lut_batchable(d)) ++count[d.dtype - MCTQ_DT_F32]; else ++n_singles;
  }
  for (int g = 0; g < 3; ++g)
    if (count[g] > 0xffffu) return fail_arg("more than 65535 tensors of one storage type");
  std::vector<uint32_t> tl((size_t)(n_items > 0 ? n_items : 1));
  for (int g = 0; g < 3; ++g) {
    if (!count[g]) continue;
    const uint32_t tile_e = lut_tile_elems(MCTQ_DT_F32 + g);
    int m = 0;
    uint64_t total = 0;
    for (int32_t k = 0; k < n_items; ++k) {
      const mctq_lut_item& d = items[k];
      const int64_t n = d.outer * d.channels * d.inner;
      if (d.dtype != MCTQ_DT_F32 + g || n == 0 || !lut_batchable(d)) continue;
      tl[m] = (uint32_t)((n + tile_e - 1) / tile_e);
      total += tl[m++];
    }
    shift[g] = chunk_shift(tl.data(), m, kMaxChunksT);
    for (int j = 0; j < m; ++j) chunks[g] += (tl[j] + (1u << shift[g]) - 1) >> shift[g];
    if (total + ((uint64_t)m << shift[g]) > 0x7fffffffull) return fail_arg("too many tiles for one launch");
  }
  auto align16 = [](uint64_t v) { return (v + 15u) & ~(uint64_t)15u; };
  uint64_t off = align16(sizeof(LutTableHeader));
  uint64_t items_off[3], map_off[3];
  for (int g = 0; g < 3; ++g) {
    items_off[g] = off; off += (uint64_t)count[g] * sizeof(LutBatchItem);
    map_off[g] = off; off = align16(off + (uint64_t)chunks[g] * 2u + 2u);
  }
  const uint64_t singles_off = off;
  off = align16(off + (uint64_t)n_singles * sizeof(mctq_lut_item));
  if (off > 0x7fffffffull) return fail_arg("table too large");
  if (!host_table || capacity < (int64_t)off) return (int64_t)off;     // size query

  uint8_t* base = static_cast<uint8_t*>(host_table);
  memset(base, 0, (size_t)off);
  LutTableHeader* h = reinterpret_cast<LutTableHeader*>(base);
  h->magic = kLutTableMagic; h->version = MCTQ_ABI_VERSION; h->total_bytes = (uint32_t)off;
  h->n_singles = n_singles; h->singles_off = (uint32_t)singles_off;
  for (int g = 0; g < 3; ++g) {
    if (!count[g]) continue;
    LutTableGroup& tg = h->g[h->n_groups++];
    tg.dtype = MCTQ_DT_F32 + g; tg.items_off = (uint32_t)items_off[g]; tg.map_off = (uint32_t)map_off[g];
    tg.n_items = count[g]; tg.shift = shift[g]; tg.out_bytes = 0; tg.lds_bytes = 0;
    const uint32_t tile_e = lut_tile_elems(tg.dtype);
    LutBatchItem* bi = reinterpret_cast<LutBatchItem*>(base + items_off[g]);
    uint16_t* map = reinterpret_cast<uint16_t*>(base + map_off[g]);
    uint32_t chunk = 0;
    int m = 0;
    for (int32_t k = 0; k < n_items; ++k) {
      const mctq_lut_item& d = items[k];
      const int64_t n = d.outer * d.channels * d.inner;
      if (d.dtype != (int32_t)tg.dtype || n == 0 || !lut_batchable(d)) continue;
      const uint32_t t = (uint32_t)((n + tile_e - 1) / tile_e);
      LutBatchItem& b = bi[m];
      const bool one_row = !d.thresholds || d.outer * d.channels == 1;
      b.b.x = d.x; b.b.y = d.y; b.b.scales = d.thresholds; b.b.zps = reinterpret_cast<const int32_t*>(d.table);
      b.b.n = (uint32_t)n; b.b.inner = one_row ? (uint32_t)n : (uint32_t)d.inner; b.b.channels = one_row ? 1u : (uint32_t)d.channels;
      b.b.tile_begin = chunk << tg.shift; b.b.tiles = t; b.b.reserved = 0; b.b.lo = 0.f; b.b.hi = 0.f;
      b.mult = d.mult; b.cmin = d.clip_min; b.cmax = d.clip_max; b.eps = d.eps; b.thr_div = d.thr_div; b.thr_mul = d.thr_mul;
      b.entries = d.entries; b.step_round = d.step_round;
      const uint32_t c = (t + (1u << tg.shift) - 1) >> tg.shift;
      for (uint32_t j = 0; j < c; ++j) map[chunk + j] = (uint16_t)m;
      chunk += c;
      tg.out_bytes += n * 4;
      const uint32_t lds = (uint32_t)table_bytes(d.entries);
      if (lds > tg.lds_bytes) tg.lds_bytes = lds;
      ++m;
    }
    tg.grid = chunk << tg.shift;
  }
  mctq_lut_item* sg = reinterpret_cast<mctq_lut_item*>(base + singles_off);
  uint32_t sidx = 0;
  for (int32_t k = 0; k < n_items; ++k)
    if (items[k].outer * items[k].channels * items[k].inner != 0 && !lut_batchable(items[k])) sg[sidx++] = items[k];
  return (int64_t)off;
}

int mctq_lutt_batch_run(const void* host_table, const void* device_table, void* stream) {
  if (!host_table) return fail_arg("host_table is NULL");
  const uint8_t* base = static_cast<const uint8_t*>(host_table);
  const LutTableHeader* h = reinterpret_cast<const LutTableHeader*>(base);
  if (h->magic != kLutTableMagic || h->version != (uint32_t)MCTQ_ABI_VERSION || h->n_groups > 3)
    return fail_arg("not a table packed by this library version (mctq_lutt_batch_pack)");
  if (h->n_groups && (!device_table || ((uintptr_t)device_table & 15u))) return fail_arg("device_table is NULL or not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const uint8_t* dev = static_cast<const uint8_t*>(device_table);
  for (uint32_t g = 0; g < h->n_groups; ++g) {
    const LutTableGroup& tg = h->g[g];
    LutTableSrc src;
    src.it = reinterpret_cast<const LutBatchItem*>(dev + tg.items_off);
    src.map = reinterpret_cast<const uint32_t*>(dev + tg.map_off);
    src.shift = tg.shift;
    int rc;
    if (tg.dtype == MCTQ_DT_F32) rc = launch_lut_batch<float>(src, tg.grid, tg.lds_bytes, tg.out_bytes, st);
    else if (tg.dtype == MCTQ_DT_F16) rc = launch_lut_batch<_Float16>(src, tg.grid, tg.lds_bytes, tg.out_bytes, st);
    else rc = launch_lut_batch<__bf16>(src, tg.grid, tg.lds_bytes, tg.out_bytes, st);
    if (rc) return rc;
  }
  const mctq_lut_item* sg = reinterpret_cast<const mctq_lut_item*>(base + h->singles_off);
  for (uint32_t k = 0; k < h->n_singles; ++k)
    if (int rc = launch_lut_single(sg[k], stream)) return rc;
  return 0;
}

}  // extern "C"
#endif  // LUT part
