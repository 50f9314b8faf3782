// chanlast2.hip -- EXPERIMENT, not part of the shipped library: the channel-last (lastaxis) and the short-row
// (window / gather) launches of 16-bit tensors rewritten around three ideas, each of which can be switched separately:
//   (a) nothing in front of the data loads: no integer division, no 64-bit multiply, no divergent branch per load
//       (the shipped lastaxis kernel runs two v_rcp_iflag divisions and four exec-masked loads with 64-bit mads first);
//   (b) an exact reciprocal in five VALU instructions (v_rcp_f32 + two Newton steps with exact FMA residuals) instead of the
//       eleven of the IEEE expansion -- a lane of the channel-last kernel computes N of them;
//   (c) parameter reads issued BEFORE the data loads in the short-row kernel (vector loads return in order) and no LDS window,
//       no block barrier.
// Entry points: mctq_x_recip_check (exhaustive check of (b)), mctq_x_lastaxis, mctq_x_shortrows.  Driver: run.py beside this file.
// Build: python tools/build_variant.py chanlast2 -> tools/ablate/libmctq_hip_chanlast2.so
#include "mctq_kernels.hpp"

namespace mctq {

// 1 / d, correctly rounded, for 2^-100 <= |d| <= 2^100 (checked exhaustively by mctq_x_recip_check)
__device__ __forceinline__ float recip_nr2(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ bool recip_ok(float d) {
  const float a = fabsf(d);
  return a >= 0x1p-100f && a <= 0x1p100f;
}
// every one of n values is a POSITIVE float in [2^-100, 2^100]: the raw bit patterns as unsigned integers (negative values, NaNs
// and infinities are large) -- n min / max operations and two compares instead of 2n compares
template <int M>
__device__ __forceinline__ bool recip_all_ok(const float (&v)[M]) {
  uint32_t lo = __float_as_uint(v[0]), hi = lo;
#pragma unroll
  for (int j = 1; j < M; ++j) { const uint32_t b = __float_as_uint(v[j]); lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
  return lo >= 0x0d800000u && hi <= 0x71800000u;
}

__global__ void recip_check_kernel(unsigned long long* out /* [0] mismatches, [1] first bad pattern + 1, [2] checked */) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0, seen = 0, first = 0;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
    const float d = __uint_as_float((uint32_t)b);
    if (!recip_ok(d)) continue;
    ++seen;
    const float want = 1.0f / d;
    const float got = recip_nr2(d);
    if (__float_as_uint(want) != __float_as_uint(got)) { ++bad; if (!first) first = b + 1; }
  }
  if (seen) atomicAdd(&out[2], seen);
  if (bad) { atomicAdd(&out[0], bad); atomicMax(&out[1], first); }
}

template <int RECIP>
__device__ __forceinline__ float recip_mode(float s, bool all_ok) {
  if (RECIP == 2) return s;                          // timing bound only (wrong results)
  if (RECIP == 1 && all_ok) return recip_nr2(s);
  return 1.0f / s;
}

// ------------------------------------------------------------------------------------------------------------------
// lastaxis2: grid (bps, groups_lo, groups_hi); block (bx, group) owns lanes [bx*256, bx*256+256) of a SLAB of k rows
// (k * vc lane-vectors, contiguous) for U slabs k rows apart.  The lane's offset inside the slab is its memory offset.
// ------------------------------------------------------------------------------------------------------------------
template <class TI, int U, int NT, int RECIP, bool HASZP, bool FULL, int SCHED>
__device__ __forceinline__ void lastaxis2_body(const TI* __restrict__ xs, TI* __restrict__ ys, uint64_t rows_left, uint32_t vc,
                                               uint32_t k, float rvc, float lo, float hi, const float* __restrict__ scales,
                                               const int32_t* __restrict__ zps, uint32_t g, uint32_t slab) {
  typedef IO<TI, TI> io;
  constexpr int N = io::N;
  typedef typename io::VI VI;
  uint32_t lim[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (FULL) lim[u] = slab;
    else {
      const uint64_t r = rows_left > (uint64_t)u * k ? rows_left - (uint64_t)u * k : 0;
      lim[u] = (r >= k ? k : (uint32_t)r) * vc;
    }
  }
  const VI* px = reinterpret_cast<const VI*>(xs) + g;
  VI* py = reinterpret_cast<VI*>(ys) + g;
  VI v[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (FULL || g < lim[u]) v[u] = __builtin_nontemporal_load(px + (size_t)u * slab);
  __builtin_amdgcn_sched_barrier(0);
  // the lane's N channels
  const uint32_t col = g - div_small(g, vc, rvc) * vc;
  float s[N], inv[N], blo[N], bhi[N];
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  // RECIP 3 / 4: timing bounds only (wrong results) -- 3: ONE dword of the table per lane, used for all N elements (the table's
  // bytes gone, its dependent load kept); 4: no table access at all (a constant scale)
#pragma unroll
  for (int j = 0; j < N; j += 4) {
    f32x4 s4;
    if (RECIP == 3) { const float one = scales[(size_t)col * N]; s4 = f32x4{one, one, one, one}; }
    else if (RECIP == 4) s4 = f32x4{0.03f, 0.03f, 0.03f, 0.03f};
    else s4 = *reinterpret_cast<const f32x4*>(scales + (size_t)col * N + j);
    i32x4 z4 = {0, 0, 0, 0};
    if (HASZP) z4 = *reinterpret_cast<const i32x4*>(zps + (size_t)col * N + j);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[j + i] = s4[i];
      if (HASZP) { const float zf = (float)z4[i]; blo[j + i] = lo - zf; bhi[j + i] = hi - zf; }
    }
  }
  const bool ok = RECIP == 1 && recip_all_ok(s);
  const bool all_ok = RECIP == 1 && __builtin_amdgcn_ballot_w64(!ok) == 0;
  if (all_ok) {
#pragma unroll
    for (int j = 0; j < N; ++j) inv[j] = recip_mode<RECIP>(s[j], true);
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) inv[j] = recip_mode<RECIP>(s[j], false);
  }
  typename io::VO res[U];
  if (SCHED == 5) { __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_sched_barrier(0); }     // everything landed first
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (FULL || g < lim[u]) {
      float in[N], out[N];
      io::unpack(v[u], in);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float r = __builtin_rintf(in[j] * inv[j]);
        const float q = HASZP ? __builtin_amdgcn_fmed3f(r, blo[j], bhi[j]) : __builtin_amdgcn_fmed3f(r, lo, hi);
        float y = __builtin_fmaf(q, s[j], 0.0f);
        asm("" : "+v"(y));
        out[j] = y;
      }
      if (SCHED == 1) res[u] = io::pack(out);
      else io::template store<NT>(reinterpret_cast<TI*>(py + (size_t)u * slab), io::pack(out));
      if ((SCHED == 4 || SCHED == 5) && u + 1 < U) { __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_sched_barrier(0); }   // PACED: vmcnt(0)
    }
  }
  if (SCHED == 1) {                                 // every result first, then the stores back to back
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (FULL || g < lim[u]) io::template store<NT>(reinterpret_cast<TI*>(py + (size_t)u * slab), res[u]);
  }
}

// SCHED 0: a row's result is stored as soon as it exists; 1: all results, then all stores; 2: the block's column stripe rotates with
// the group (block -> XCD is round robin: without it XCD j always reads stripe j of every slab); 3: grid x = group, y = stripe
template <class TI, int U, int NT, int RECIP, bool HASZP, int SCHED>
__global__ __launch_bounds__(kThreads) void lastaxis2_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, uint64_t rows,
                                                             uint32_t vc, uint32_t k, float rvc, float lo, float hi,
                                                             uint32_t groups, const float* __restrict__ scales,
                                                             const int32_t* __restrict__ zps) {
  const uint32_t slab = k * vc;
  uint32_t bx = blockIdx.x, group = blockIdx.y + blockIdx.z * gridDim.y;
  if (SCHED == 3) { bx = blockIdx.y; group = blockIdx.x; }
  if (group >= groups) return;
  if (SCHED == 2) { const uint32_t nb = gridDim.x; bx += group % nb; if (bx >= nb) bx -= nb; }
  const uint32_t g = bx * kThreads + threadIdx.x;
  if (g >= slab) return;
  const uint64_t grow0 = (uint64_t)group * (uint32_t)(U * k);
  const uint64_t rows_left = rows - grow0;
  typedef IO<TI, TI> io;
  const TI* x0 = xs + grow0 * vc * io::N;
  TI* y0 = ys + grow0 * vc * io::N;
  if (rows_left >= (uint64_t)U * k)
    lastaxis2_body<TI, U, NT, RECIP, HASZP, true, SCHED>(x0, y0, rows_left, vc, k, rvc, lo, hi, scales, zps, g, slab);
  else
    lastaxis2_body<TI, U, NT, RECIP, HASZP, false, SCHED>(x0, y0, rows_left, vc, k, rvc, lo, hi, scales, zps, g, slab);
}

// the same lanes looping over `groups` slabs-of-U with a stride of gridDim.y * gridDim.z groups; parameters once per lane.
// PIPE: the next group's loads are issued before the current group is computed and stored.
template <class TI, int U, int NT, bool HASZP, bool PIPE>
__global__ __launch_bounds__(kThreads) void lastaxis2_loop_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, uint64_t rows,
                                                                  uint32_t vc, uint32_t k, float rvc, float lo, float hi,
                                                                  uint32_t groups, const float* __restrict__ scales,
                                                                  const int32_t* __restrict__ zps) {
  typedef IO<TI, TI> io;
  constexpr int N = io::N;
  typedef typename io::VI VI;
  const uint32_t slab = k * vc;
  const uint32_t g = blockIdx.x * kThreads + threadIdx.x;
  const uint32_t G = gridDim.y * gridDim.z;
  uint32_t group = blockIdx.y + blockIdx.z * gridDim.y;
  if (group >= groups || g >= slab) return;
  const size_t gstride = (size_t)U * slab;                 // lane-vectors per group
  auto loadg = [&](VI (&v)[U], uint32_t grp) {
    const uint64_t grow0 = (uint64_t)grp * (uint32_t)(U * k);
    const uint64_t rows_left = rows - grow0;
    const VI* px = reinterpret_cast<const VI*>(xs) + (size_t)grp * gstride + g;
    if (rows_left >= (uint64_t)U * k) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(px + (size_t)u * slab);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t r = rows_left > (uint64_t)u * k ? rows_left - (uint64_t)u * k : 0;
        if (g < (r >= k ? k : (uint32_t)r) * vc) v[u] = __builtin_nontemporal_load(px + (size_t)u * slab);
      }
    }
  };
  VI v[U];
  loadg(v, group);
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t col = g - div_small(g, vc, rvc) * vc;
  float s[N], inv[N], blo[N], bhi[N];
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int j = 0; j < N; j += 4) {
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(scales + (size_t)col * N + j);
    i32x4 z4 = {0, 0, 0, 0};
    if (HASZP) z4 = *reinterpret_cast<const i32x4*>(zps + (size_t)col * N + j);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[j + i] = s4[i];
      if (HASZP) { const float zf = (float)z4[i]; blo[j + i] = lo - zf; bhi[j + i] = hi - zf; }
    }
  }
  if (__builtin_amdgcn_ballot_w64(!recip_all_ok(s)) == 0) {
#pragma unroll
    for (int j = 0; j < N; ++j) inv[j] = recip_nr2(s[j]);
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) inv[j] = 1.0f / s[j];
  }
  auto finish = [&](const VI (&v)[U], uint32_t grp) {
    const uint64_t grow0 = (uint64_t)grp * (uint32_t)(U * k);
    const uint64_t rows_left = rows - grow0;
    VI* py = reinterpret_cast<VI*>(ys) + (size_t)grp * gstride + g;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t r = rows_left > (uint64_t)u * k ? rows_left - (uint64_t)u * k : 0;
      if (g < (r >= k ? k : (uint32_t)r) * vc) {
        float in[N], out[N];
        io::unpack(v[u], in);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const float rr = __builtin_rintf(in[j] * inv[j]);
          const float q = HASZP ? __builtin_amdgcn_fmed3f(rr, blo[j], bhi[j]) : __builtin_amdgcn_fmed3f(rr, lo, hi);
          float y = __builtin_fmaf(q, s[j], 0.0f);
          asm("" : "+v"(y));
          out[j] = y;
        }
        io::template store<NT>(reinterpret_cast<TI*>(py + (size_t)u * slab), io::pack(out));
      }
    }
  };
  if (!PIPE) {
    for (;;) {
      finish(v, group);
      group += G;
      if (group >= groups) break;
      loadg(v, group);
    }
  } else {
    for (;;) {
      const uint32_t next = group + G;
      VI w[U];
      if (next < groups) loadg(w, next);
      finish(v, group);
      if (next >= groups) break;
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = w[u];
      group = next;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// lastaxis3: the channel-last tensor cut like the short-row launch -- block b owns U x 256 CONSECUTIVE lane-vectors (one contiguous
// 8 or 16 KiB read) whatever rows they fall in; every lane-vector fetches its own N channels' scales (no sharing between a lane's
// vectors unless the row length divides the step) and inverts them right where they are used.
// ------------------------------------------------------------------------------------------------------------------
template <class TI, int U, int NT, bool HASZP>
__global__ __launch_bounds__(kThreads) void lastaxis3_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, uint32_t n_lv,
                                                             uint32_t vc, float rvc, float lo, float hi,
                                                             const float* __restrict__ scales, const int32_t* __restrict__ zps) {
  typedef IO<TI, TI> io;
  constexpr int N = io::N;
  typedef typename io::VI VI;
  const uint32_t b0 = blockIdx.x * (U * kThreads);
  const bool full = b0 + U * kThreads <= n_lv;
  VI v[U];
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs) + b0 + u * kThreads + threadIdx.x);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (b0 + u * kThreads + threadIdx.x < n_lv) v[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs) + b0 + u * kThreads + threadIdx.x);
  }
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t base = b0 % vc;                                   // uniform
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  f32x4 s4[U][N / 4];
  i32x4 z4[U][N / 4];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint32_t c = base + u * kThreads + threadIdx.x;                // < vc + U * 256: exact through float32
    c -= div_small(c, vc, rvc) * vc;
#pragma unroll
    for (int j = 0; j < N / 4; ++j) {
      s4[u][j] = *reinterpret_cast<const f32x4*>(scales + (size_t)c * N + 4 * j);
      if (HASZP) z4[u][j] = *reinterpret_cast<const i32x4*>(zps + (size_t)c * N + 4 * j);
    }
  }
  bool ok = true;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < N / 4; ++j) { float t[4] = {s4[u][j][0], s4[u][j][1], s4[u][j][2], s4[u][j][3]}; ok = ok && recip_all_ok(t); }
  const bool exact = __builtin_amdgcn_ballot_w64(!ok) == 0;
  VI r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (!full && b0 + u * kThreads + threadIdx.x >= n_lv) continue;
    float in[N], out[N];
    io::unpack(v[u], in);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float sc = s4[u][j / 4][j % 4];
      const float inv = exact ? recip_nr2(sc) : 1.0f / sc;
      const float zf = HASZP ? (float)z4[u][j / 4][j % 4] : 0.0f;
      const float q = __builtin_amdgcn_fmed3f(__builtin_rintf(in[j] * inv), lo - zf, hi - zf);
      float y = __builtin_fmaf(q, sc, 0.0f);
      asm("" : "+v"(y));
      out[j] = y;
    }
    r[u] = io::pack(out);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (full || b0 + u * kThreads + threadIdx.x < n_lv) io::template store<NT>(ys + ((size_t)b0 + u * kThreads + threadIdx.x) * N, r[u]);
}

// lastaxis4: contiguous tiles as lastaxis3, for row lengths that make a lane's columns REPEAT across its U vectors: 256 % vc == 0 (one column
// for all U vectors: SETS = 1) or vc == 512 (columns alternate: SETS = 2).  The table traffic per byte of data is then that of the slab
// kernel (N scales per lane and SETS), the reads are one contiguous 16 KiB piece per block on a 1-D grid, as in shortrows_kernel.
template <class TI, int NT, bool HASZP, int SETS>
__global__ __launch_bounds__(kThreads) void lastaxis4_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, uint32_t n_lv,
                                                             uint32_t vc, float lo, float hi,
                                                             const float* __restrict__ scales, const int32_t* __restrict__ zps) {
  typedef IO<TI, TI> io;
  constexpr int N = io::N, U = 4;
  typedef typename io::VI VI;
  const uint32_t b0 = blockIdx.x * (U * kThreads);
  const bool full = b0 + U * kThreads <= n_lv;
  VI v[U];
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs) + b0 + u * kThreads + threadIdx.x);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (b0 + u * kThreads + threadIdx.x < n_lv) v[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs) + b0 + u * kThreads + threadIdx.x);
  }
  __builtin_amdgcn_sched_barrier(0);
  // b0 is a multiple of 1024 lane-vectors: with 256 % vc == 0 or vc == 512 it is a multiple of vc as well
  float s[SETS][N], inv[SETS][N], zf[SETS][N];
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  float all[SETS * N];
#pragma unroll
  for (int k = 0; k < SETS; ++k) {
    const uint32_t c = SETS == 1 ? (threadIdx.x & (vc - 1)) : threadIdx.x + k * kThreads;     // vc a power of two (256 % vc == 0)
#pragma unroll
    for (int j = 0; j < N; j += 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(scales + (size_t)c * N + j);
      i32x4 z4 = {0, 0, 0, 0};
      if (HASZP) z4 = *reinterpret_cast<const i32x4*>(zps + (size_t)c * N + j);
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[k][j + i] = s4[i]; all[k * N + j + i] = s4[i]; zf[k][j + i] = HASZP ? (float)z4[i] : 0.0f; }
    }
  }
  const bool exact = __builtin_amdgcn_ballot_w64(!recip_all_ok(all)) == 0;
#pragma unroll
  for (int k = 0; k < SETS; ++k)
#pragma unroll
    for (int j = 0; j < N; ++j) inv[k][j] = exact ? recip_nr2(s[k][j]) : 1.0f / s[k][j];
  VI r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (!full && b0 + u * kThreads + threadIdx.x >= n_lv) continue;
    constexpr int dummy = 0; (void)dummy;
    const int k = SETS == 1 ? 0 : (u & 1);
    float in[N], out[N];
    io::unpack(v[u], in);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float q = __builtin_amdgcn_fmed3f(__builtin_rintf(in[j] * inv[k][j]), lo - zf[k][j], hi - zf[k][j]);
      float y = __builtin_fmaf(q, s[k][j], 0.0f);
      asm("" : "+v"(y));
      out[j] = y;
    }
    r[u] = io::pack(out);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (full || b0 + u * kThreads + threadIdx.x < n_lv) io::template store<NT>(ys + ((size_t)b0 + u * kThreads + threadIdx.x) * N, r[u]);
}

template <class TI>
static int launch_lastaxis4(const void* xv, void* yv, int64_t rows, int64_t channels, const float* scales, const int32_t* zps,
                            int32_t qmin, int32_t qmax, int nt, hipStream_t st) {
  typedef IO<TI, TI> io;
  const TI* x = static_cast<const TI*>(xv);
  TI* y = static_cast<TI*>(yv);
  if (channels % io::N) return fail_arg("channels % N");
  const int64_t vc = channels / io::N, n_lv = rows * vc;
  const int sets = (vc <= 256 && 256 % vc == 0) ? 1 : (vc == 512 ? 2 : 0);
  if (!sets) return fail_arg("row length does not repeat over a tile");
  if (n_lv >= (1ll << 32) - 2048) return fail_arg("too large for the experiment");
  const float lo = (float)qmin, hi = (float)qmax;
  const unsigned grid = (unsigned)((n_lv + 4 * kThreads - 1) / (4 * kThreads));
#define LA4(NT_, Z_, S_) hipLaunchKernelGGL((lastaxis4_kernel<TI, NT_, Z_, S_>), dim3(grid), dim3(kThreads), 0, st, x, y, (uint32_t)n_lv, \
                                             (uint32_t)vc, lo, hi, scales, zps)
  if (sets == 1) { if (nt == 2) { if (zps) LA4(2, true, 1); else LA4(2, false, 1); } else { if (zps) LA4(1, true, 1); else LA4(1, false, 1); } }
  else { if (nt == 2) { if (zps) LA4(2, true, 2); else LA4(2, false, 2); } else { if (zps) LA4(1, true, 2); else LA4(1, false, 2); } }
  return check_launch("lastaxis4");
}

template <class TI>
static int launch_lastaxis3(int u_sel, const void* xv, void* yv, int64_t rows, int64_t channels, const float* scales,
                            const int32_t* zps, int32_t qmin, int32_t qmax, int nt, hipStream_t st) {
  typedef IO<TI, TI> io;
  const TI* x = static_cast<const TI*>(xv);
  TI* y = static_cast<TI*>(yv);
  if (channels % io::N) return fail_arg("channels % N");
  const int64_t vc = channels / io::N, n_lv = rows * vc;
  if (n_lv >= (1ll << 32) - 2048 || vc + 2048 >= (1 << 24)) return fail_arg("too large for the experiment");
  const float rvc = 1.0f / (float)vc, lo = (float)qmin, hi = (float)qmax;
  const unsigned grid = (unsigned)((n_lv + u_sel * kThreads - 1) / (u_sel * kThreads));
#define LA3(U_, NT_, Z_) hipLaunchKernelGGL((lastaxis3_kernel<TI, U_, NT_, Z_>), dim3(grid), dim3(kThreads), 0, st, x, y, (uint32_t)n_lv, \
                                             (uint32_t)vc, rvc, lo, hi, scales, zps)
  if (u_sel == 4) { if (nt == 2) { if (zps) LA3(4, 2, true); else LA3(4, 2, false); } else { if (zps) LA3(4, 1, true); else LA3(4, 1, false); } }
  else if (u_sel == 2) { if (nt == 2) { if (zps) LA3(2, 2, true); else LA3(2, 2, false); } else { if (zps) LA3(2, 1, true); else LA3(2, 1, false); } }
  else return fail_arg("U");
  return check_launch("lastaxis3");
}

template <class TI>
static int launch_lastaxis2(int mode, int u_sel, int loops, const void* xv, void* yv, int64_t rows, int64_t channels,
                            const float* scales, const int32_t* zps, int32_t qmin, int32_t qmax, int nt, hipStream_t st) {
  typedef IO<TI, TI> io;
  const TI* x = static_cast<const TI*>(xv);
  TI* y = static_cast<TI*>(yv);
  if (channels % io::N) return fail_arg("channels % N");
  const int64_t vc = channels / io::N;
  int64_t k = (2048 + vc - 1) / vc, best_waste = -1;
  for (int64_t c = k; c < k + 16; ++c) {
    const int64_t waste = (kThreads - (c * vc) % kThreads) % kThreads * 4096 / (c * vc);
    if (best_waste < 0 || waste < best_waste) { best_waste = waste; k = c; }
  }
  if (k * vc >= (1 << 24)) return fail_arg("slab too large for the experiment");
  const int64_t bps = (k * vc + kThreads - 1) / kThreads;
  const int64_t U = u_sel;
  const int64_t groups = (rows + U * k - 1) / (U * k);
  int64_t gl = groups;
  if (mode == 4 || mode == 5) gl = (groups + loops - 1) / loops;          // groups per grid "column"
  const int64_t gy = gl < 65535 ? gl : 65535, gz = (gl + gy - 1) / gy;
  const dim3 grid((unsigned)bps, (unsigned)gy, (unsigned)gz);
  const float rvc = 1.0f / (float)vc, lo = (float)qmin, hi = (float)qmax;
#define LA2(U_, NT_, R_, Z_) hipLaunchKernelGGL((lastaxis2_kernel<TI, U_, NT_, R_, Z_, 0>), grid, dim3(kThreads), 0, st, x, y, \
                                                 (uint64_t)rows, (uint32_t)vc, (uint32_t)k, rvc, lo, hi, (uint32_t)groups, scales, zps)
#define LA2S(U_, NT_, Z_, S_) hipLaunchKernelGGL((lastaxis2_kernel<TI, U_, NT_, 1, Z_, S_>), (S_ == 3 ? dim3((unsigned)groups, (unsigned)bps, 1) : grid), \
                                                  dim3(kThreads), 0, st, x, y, (uint64_t)rows, (uint32_t)vc, (uint32_t)k, rvc, lo, hi, (uint32_t)groups, scales, zps)
#define LA2L(U_, NT_, Z_, P_) hipLaunchKernelGGL((lastaxis2_loop_kernel<TI, U_, NT_, Z_, P_>), grid, dim3(kThreads), 0, st, x, y, \
                                                  (uint64_t)rows, (uint32_t)vc, (uint32_t)k, rvc, lo, hi, (uint32_t)groups, scales, zps)
#define BY_Z(M_, ...) do { if (zps) { constexpr bool Z = true; __VA_ARGS__; } else { constexpr bool Z = false; __VA_ARGS__; } } while (0)
#define BY_NT(...) do { if (nt == 2) { constexpr int NT = 2; __VA_ARGS__; } else { constexpr int NT = 1; __VA_ARGS__; } } while (0)
#define BY_U(...) do { if (u_sel == 2) { constexpr int UU = 2; __VA_ARGS__; } else if (u_sel == 4) { constexpr int UU = 4; __VA_ARGS__; } else return fail_arg("U"); } while (0)
  switch (mode) {
    case 1: BY_U(BY_NT(BY_Z(0, LA2(UU, NT, 0, Z)))); break;
    case 2: BY_U(BY_NT(BY_Z(0, LA2(UU, NT, 1, Z)))); break;
    case 3: BY_U(BY_NT(BY_Z(0, LA2(UU, NT, 2, Z)))); break;
    case 6: BY_U(BY_NT(BY_Z(0, LA2(UU, NT, 3, Z)))); break;
    case 7: BY_U(BY_NT(BY_Z(0, LA2(UU, NT, 4, Z)))); break;
    case 11: BY_U(BY_NT(BY_Z(0, LA2S(UU, NT, Z, 1)))); break;
    case 12: BY_U(BY_NT(BY_Z(0, LA2S(UU, NT, Z, 2)))); break;
    case 13: BY_U(BY_NT(BY_Z(0, LA2S(UU, NT, Z, 3)))); break;
    case 14: BY_U(BY_NT(BY_Z(0, LA2S(UU, NT, Z, 4)))); break;      // paced stores
    case 15: BY_U(BY_NT(BY_Z(0, LA2S(UU, NT, Z, 5)))); break;      // all loads landed first, then paced stores
    case 4: BY_U(BY_NT(BY_Z(0, LA2L(UU, NT, Z, false)))); break;
    case 5: BY_U(BY_NT(BY_Z(0, LA2L(UU, NT, Z, true)))); break;
    default: return fail_arg("mode");
  }
  return check_launch("lastaxis2");
}

// ------------------------------------------------------------------------------------------------------------------
// lastaxis5: the parameter table SHARED by the block's four waves through LDS.  The slab kernel's lanes each read their N scales
// from the table (N * 4 bytes per lane and U rows: as many L1 accesses as a row of data); here a block is ONE 64-lane piece of
// the slab x (4 waves x U slabs): the four waves own the same columns in different slabs, the piece's 64 * N scales are read from
// memory once per block (threads 0 .. 64*N/4-1, 16 bytes each) into LDS, every lane takes its N from there.  PREINV: the loader
// threads also invert (one reciprocal per scale and block instead of one per scale and wave).
// Grid (pieces of 64 lane-vectors per slab, groups of 4 * U slabs).
// ------------------------------------------------------------------------------------------------------------------
template <class TI, int U, int NT, bool HASZP, bool PREINV>
__global__ __launch_bounds__(kThreads) void lastaxis5_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, uint64_t rows,
                                                             uint32_t vc, uint32_t k, float rvc, float lo, float hi,
                                                             uint32_t groups, const float* __restrict__ scales,
                                                             const int32_t* __restrict__ zps) {
  typedef IO<TI, TI> io;
  constexpr int N = io::N, Q = N / 4, W = kThreads / 64;
  typedef typename io::VI VI;
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  __shared__ f32x4 sh_s[Q][64];
  __shared__ f32x4 sh_inv[PREINV ? Q : 1][64];
  __shared__ i32x4 sh_z[HASZP ? Q : 1][64];
  const uint32_t slab = k * vc;
  const uint32_t group = blockIdx.y + blockIdx.z * gridDim.y;
  if (group >= groups) return;                                     // (the whole block)
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x * 64 + lane;
  const uint64_t grow0 = (uint64_t)group * (uint32_t)(W * U * k);
  const uint64_t rows_left = rows - grow0;
  const VI* px = reinterpret_cast<const VI*>(xs + grow0 * vc * N) + g;
  VI* py = reinterpret_cast<VI*>(ys + grow0 * vc * N) + g;
  uint32_t lim[U], si[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    si[u] = u * W + wave;                                          // adjacent waves: adjacent slabs
    const uint64_t r = rows_left > (uint64_t)si[u] * k ? rows_left - (uint64_t)si[u] * k : 0;
    lim[u] = (r >= k ? k : (uint32_t)r) * vc;
  }
  VI v[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (g < lim[u]) v[u] = __builtin_nontemporal_load(px + (size_t)si[u] * slab);
  __builtin_amdgcn_sched_barrier(0);
  // the piece's table -> LDS
  {
    const uint32_t t = threadIdx.x;
    const bool zpart = HASZP && t >= 64 * Q && t < 128 * Q;
    const uint32_t tt = zpart ? t - 64 * Q : t;
    const uint32_t L = tt / Q, h = tt % Q;
    const uint32_t gl = blockIdx.x * 64 + L;
    if (tt < 64 * Q && (t < 64 * Q || zpart) && gl < slab) {
      const uint32_t col = gl - div_small(gl, vc, rvc) * vc;
      if (zpart) {
        sh_z[h][L] = *reinterpret_cast<const i32x4*>(zps + (size_t)col * N + 4 * h);
      } else {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(scales + (size_t)col * N + 4 * h);
        sh_s[h][L] = s4;
        if (PREINV) {
          float sv[4] = {s4[0], s4[1], s4[2], s4[3]};
          const bool ok = recip_all_ok(sv);
          f32x4 i4;
#pragma unroll
          for (int i = 0; i < 4; ++i) i4[i] = ok ? recip_nr2(sv[i]) : 1.0f / sv[i];
          sh_inv[h][L] = i4;
        }
      }
    }
  }
  __syncthreads();
  float s[N], inv[N], blo[N], bhi[N];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const f32x4 s4 = sh_s[q][lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[4 * q + i] = s4[i];
    if (PREINV) {
      const f32x4 i4 = sh_inv[q][lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) inv[4 * q + i] = i4[i];
    }
    if (HASZP) {
      const i32x4 z4 = sh_z[q][lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float zf = (float)z4[i]; blo[4 * q + i] = lo - zf; bhi[4 * q + i] = hi - zf; }
    }
  }
  if (!PREINV) {
    const bool ok = g < slab ? recip_all_ok(s) : true;
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
#pragma unroll
      for (int j = 0; j < N; ++j) inv[j] = recip_nr2(s[j]);
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) inv[j] = 1.0f / s[j];
    }
  }
  VI res[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (g < lim[u]) {
      float in[N], out[N];
      io::unpack(v[u], in);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float r = __builtin_rintf(in[j] * inv[j]);
        const float q = HASZP ? __builtin_amdgcn_fmed3f(r, blo[j], bhi[j]) : __builtin_amdgcn_fmed3f(r, lo, hi);
        float y = __builtin_fmaf(q, s[j], 0.0f);
        asm("" : "+v"(y));
        out[j] = y;
      }
      res[u] = io::pack(out);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (g < lim[u]) io::template store<NT>(reinterpret_cast<TI*>(py + (size_t)si[u] * slab), res[u]);
}

template <class TI>
static int launch_lastaxis5(int mode, int u_sel, int minlv, const void* xv, void* yv, int64_t rows, int64_t channels,
                            const float* scales, const int32_t* zps, int32_t qmin, int32_t qmax, int nt, hipStream_t st) {
  typedef IO<TI, TI> io;
  const TI* x = static_cast<const TI*>(xv);
  TI* y = static_cast<TI*>(yv);
  if (channels % io::N) return fail_arg("channels % N");
  const int64_t vc = channels / io::N;
  if (minlv < 64) minlv = 512;                                           // lane-vectors per slab, at least
  int64_t k = (minlv + vc - 1) / vc, best_waste = -1;
  for (int64_t c = k; c < k + 16; ++c) {
    const int64_t waste = (64 - (c * vc) % 64) % 64 * 4096 / (c * vc);
    if (best_waste < 0 || waste < best_waste) { best_waste = waste; k = c; }
  }
  if (k * vc >= (1 << 24)) return fail_arg("slab too large for the experiment");
  const int64_t pieces = (k * vc + 63) / 64, U = u_sel, W = kThreads / 64;
  const int64_t groups = (rows + W * U * k - 1) / (W * U * k);
  const int64_t gy = groups < 65535 ? groups : 65535, gz = (groups + gy - 1) / gy;
  const dim3 grid((unsigned)pieces, (unsigned)gy, (unsigned)gz);
  const float rvc = 1.0f / (float)vc, lo = (float)qmin, hi = (float)qmax;
#define LA5(U_, NT_, Z_, P_) hipLaunchKernelGGL((lastaxis5_kernel<TI, U_, NT_, Z_, P_>), grid, dim3(kThreads), 0, st, x, y,                                                  (uint64_t)rows, (uint32_t)vc, (uint32_t)k, rvc, lo, hi, (uint32_t)groups, scales, zps)
  switch (mode) {
    case 31: BY_U(BY_NT(BY_Z(0, LA5(UU, NT, Z, false)))); break;
    case 32: BY_U(BY_NT(BY_Z(0, LA5(UU, NT, Z, true)))); break;
    default: return fail_arg("mode");
  }
  return check_launch("lastaxis5");
}

// ------------------------------------------------------------------------------------------------------------------
// shortrows: block b owns elements [b*TILE, (b+1)*TILE) of the dense [rows][inner] storage, inner >= N; a lane-vector lies
// in one row or crosses one boundary.  ORDER 0: data loads, then the parameter reads (the shipped gather path's order);
// ORDER 1: parameter reads first (they return first), reciprocals under the data loads' latency.
// ------------------------------------------------------------------------------------------------------------------
template <class TI, int U, int NT, int ORDER, int RECIP, bool HASZP, bool SAMEROW>
__global__ __launch_bounds__(kThreads) void shortrows_x_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, uint32_t n,
                                                             uint32_t inner, uint32_t channels, float r_inner, float r_channels,
                                                             uint32_t sh, float lo, float hi, const float* __restrict__ scales,
                                                             const int32_t* __restrict__ zps) {
  typedef IO<TI, TI> io;
  constexpr uint32_t N = io::N;
  typedef typename io::VI VI;
  constexpr uint32_t TILE = kThreads * U * N;
  const uint32_t e0 = blockIdx.x * TILE;
  const uint32_t left = n - e0;
  const uint32_t count = left < TILE ? left : TILE;
  const bool full = left >= TILE;                                    // uniform
  uint32_t row0, rem0;
  if (sh < 32) { row0 = e0 >> sh; rem0 = e0 & (inner - 1); }
  else { row0 = e0 / inner; rem0 = e0 - row0 * inner; }
  const uint32_t c0 = row0 < channels ? row0 : row0 % channels;
  const uint32_t nrows = (rem0 + count - 1) / inner + 1;
  const bool wraps = c0 + nrows > channels;

  VI v[U];
  auto data_loads = [&]() {
    if (full) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs + e0) + u * kThreads + threadIdx.x);
        if (ORDER == 4 && u + 1 < U) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(2); __builtin_amdgcn_sched_barrier(0); }   // loads 128 clocks apart
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t off = (u * kThreads + threadIdx.x) * N;
        if (off + N <= count) v[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs + e0) + u * kThreads + threadIdx.x);
      }
    }
  };
  if (ORDER == 0 || ORDER == 2 || ORDER == 3 || ORDER == 4) { data_loads(); __builtin_amdgcn_sched_barrier(0); }

  uint32_t ca[U], cb[U], split[U];
  float sa[U], sb[U], za[U], zb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    const uint32_t pos = rem0 + ((full || off < count) ? off : 0);
    const uint32_t lrow = sh < 32 ? pos >> sh : div_small(pos, inner, r_inner);
    const uint32_t rem = pos - lrow * inner;
    uint32_t c = c0 + lrow;
    if (wraps) c = c - div_small(c, channels, r_channels) * channels;
    ca[u] = c;
    split[u] = inner - rem;                                          // elements of the vector that belong to row c (>= N: all)
    sa[u] = scales[c];
    if (HASZP) za[u] = (float)zps[c];
    if (!SAMEROW) {
      cb[u] = c + 1 == channels ? 0 : c + 1;
      sb[u] = scales[cb[u]];
      if (HASZP) zb[u] = (float)zps[cb[u]];
    }
  }
  if (ORDER == 1) { __builtin_amdgcn_sched_barrier(0); data_loads(); __builtin_amdgcn_sched_barrier(0); }

  bool ok = recip_all_ok(sa);
  if (!SAMEROW) ok = ok && recip_all_ok(sb);
  const bool all_ok = RECIP == 1 && __builtin_amdgcn_ballot_w64(!ok) == 0;
  float ia[U], ib[U];
  if (all_ok) {
#pragma unroll
    for (int u = 0; u < U; ++u) { ia[u] = recip_nr2(sa[u]); if (!SAMEROW) ib[u] = recip_nr2(sb[u]); }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) { ia[u] = 1.0f / sa[u]; if (!SAMEROW) ib[u] = 1.0f / sb[u]; }
  }
  if (ORDER == 3 || ORDER == 4) { __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    if (!full && off >= count) continue;
    const float loa = HASZP ? lo - za[u] : lo, hia = HASZP ? hi - za[u] : hi;
    if (full || off + N <= count) {
      float in[N], out[N];
      io::unpack(v[u], in);
      if (SAMEROW || split[u] >= N) {
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
          const float r = __builtin_rintf(in[j] * ia[u]);
          float y = __builtin_fmaf(__builtin_amdgcn_fmed3f(r, loa, hia), sa[u], 0.0f);
          asm("" : "+v"(y));
          out[j] = y;
        }
      } else {
        const float lob = HASZP ? lo - zb[u] : lo, hib = HASZP ? hi - zb[u] : hi;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
          const bool a = j < split[u];
          const float r = __builtin_rintf(in[j] * (a ? ia[u] : ib[u]));
          float y = __builtin_fmaf(__builtin_amdgcn_fmed3f(r, a ? loa : lob, a ? hia : hib), a ? sa[u] : sb[u], 0.0f);
          asm("" : "+v"(y));
          out[j] = y;
        }
      }
      io::template store<NT>(ys + e0 + off, io::pack(out));
      if ((ORDER == 2 || ORDER == 3 || ORDER == 4) && u + 1 < U) { __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_sched_barrier(0); }   // PACED: vmcnt(0)
    } else {
      // the tensor's last, partial lane-vector: element by element
      AffineOp op; op.scales = scales; op.zps = zps; op.lo = lo; op.hi = hi;
      uint32_t rem = inner - split[u], c = ca[u];
      for (uint32_t j = 0; j < N && off + j < count; ++j) {
        ys[e0 + off + j] = narrow_to<TI>(op.apply((float)xs[e0 + off + j], op.fetch(c), NoBook()));
        if (++rem == inner) { rem = 0; if (++c == channels) c = 0; }
      }
    }
  }
}

template <class TI>
static int launch_shortrows(int mode, const void* xv, void* yv, int64_t rows, int64_t inner, int64_t channels,
                            const float* scales, const int32_t* zps, int32_t qmin, int32_t qmax, int nt, hipStream_t st) {
  typedef IO<TI, TI> io;
  const TI* x = static_cast<const TI*>(xv);
  TI* y = static_cast<TI*>(yv);
  const int64_t n = rows * inner;
  if (inner < io::N) return fail_arg("inner < N");
  constexpr int U = 4;
  constexpr uint32_t TILE = kThreads * U * io::N;
  if (n >= (1ll << 32) - TILE || (uint64_t)inner + TILE >= (1u << 24) || channels >= (1 << 23)) return fail_arg("too large for the experiment");
  const uint32_t grid = (uint32_t)((n + TILE - 1) / TILE);
  uint32_t sh = 32;
  if ((inner & (inner - 1)) == 0) { sh = 0; while ((1ll << sh) < inner) ++sh; }
  const bool same = inner % io::N == 0;
  const float ri = 1.0f / (float)inner, rc = 1.0f / (float)channels, lo = (float)qmin, hi = (float)qmax;
#define SR(NT_, O_, R_, Z_, S_) hipLaunchKernelGGL((shortrows_x_kernel<TI, U, NT_, O_, R_, Z_, S_>), dim3(grid), dim3(kThreads), 0, st, x, y, \
                                                    (uint32_t)n, (uint32_t)inner, (uint32_t)channels, ri, rc, sh, lo, hi, scales, zps)
#define BY_S(...) do { if (same) { constexpr bool S = true; __VA_ARGS__; } else { constexpr bool S = false; __VA_ARGS__; } } while (0)
  switch (mode) {
    case 1: BY_NT(BY_Z(0, BY_S(SR(NT, 0, 0, Z, S)))); break;      // data first, IEEE reciprocal: the gather path as shipped for float32
    case 2: BY_NT(BY_Z(0, BY_S(SR(NT, 0, 1, Z, S)))); break;      // data first, fast reciprocal
    case 3: BY_NT(BY_Z(0, BY_S(SR(NT, 1, 0, Z, S)))); break;      // parameters first, IEEE
    case 4: BY_NT(BY_Z(0, BY_S(SR(NT, 1, 1, Z, S)))); break;      // parameters first, fast reciprocal
    case 5: BY_NT(BY_Z(0, BY_S(SR(NT, 2, 1, Z, S)))); break;      // data first, fast reciprocal, paced stores
    case 6: BY_NT(BY_Z(0, BY_S(SR(NT, 3, 1, Z, S)))); break;      // data first, fast reciprocal, all loads landed first, paced stores
    case 7: BY_NT(BY_Z(0, BY_S(SR(NT, 4, 1, Z, S)))); break;      // the three ingredients of flat_paced_kernel: loads apart, all landed first, paced stores
    default: return fail_arg("mode");
  }
  return check_launch("shortrows");
}

}  // namespace mctq

using namespace mctq;

extern "C" int mctq_x_recip_check(unsigned long long* dev_out3, void* stream) {
  hipLaunchKernelGGL(recip_check_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, dev_out3);
  return check_launch("recip_check");
}

// ------------------------------------------------------------------------------------------------------------------
// flat_x: the per-tensor launch (block b owns 4 x 256 consecutive lane-vectors, scale / bounds in scalar registers) with something
// BETWEEN the arrival of the data and the arithmetic, to find what gives shortrows_kernel its edge over flat_kernel on identical
// work (edge.py: 11.72 vs 12.11 us, same output bits).  MODE 0: nothing (flat_kernel's shape); 1: one dependent dword load per block
// from a 16 KiB table, a different word for every block (shortrows_kernel's scale read); 2: the same load, one hot word for all blocks;
// 3: s_sleep SLEEP after the loads have landed; 4: MODE 1's load issued BEFORE the data loads.
// The loaded word d enters as fma(d, 0, scale): the scale exactly (table entries are finite), a true dependency for the compiler.
// ------------------------------------------------------------------------------------------------------------------
template <class TI, int NT, int MODE, int SLEEP>
__global__ __launch_bounds__(kThreads) void flat_x_kernel(const TI* __restrict__ xs, TI* __restrict__ ys, uint32_t n_lv, float scale,
                                                          float lo, float hi, const float* __restrict__ table) {
  typedef IO<TI, TI> io;
  constexpr int N = io::N, U = 4;
  typedef typename io::VI VI;
  const uint32_t b0 = blockIdx.x * (U * kThreads) + threadIdx.x;
  if ((MODE >= 5 && MODE <= 16) && blockIdx.x * (U * kThreads) + U * kThreads <= n_lv) {
    // full tile, no masks: the shipped flat_kernel's shape; 5: every store waited for before the next vector is touched (PACED);
    // 6: unpaced (control); 7: the first store unpaced, then paced
    // 10: mode 8 with the loads 64 clocks apart; 11: mode 8 with lane-masked loads; 13: mode 8 with lane-masked arithmetic + stores
    VI w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 11) { if (b0 + u * kThreads < n_lv) w[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs) + b0 + u * kThreads); }
      else w[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs) + b0 + u * kThreads);
      if ((MODE == 10 || MODE >= 14) && u + 1 < U) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(SLEEP > 0 ? SLEEP : 1); __builtin_amdgcn_sched_barrier(0); }   // 14: spaced loads only; 15: + all loads first; 16: + paced stores only
    }
    __builtin_amdgcn_sched_barrier(0);
    const float inv0 = 1.0f / scale;
    if (MODE == 8 || MODE == 9 || (MODE >= 10 && MODE <= 13) || MODE == 15) { __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_sched_barrier(0); }   // 8 / 9: ALL loads landed first
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 13 && !(b0 + u * kThreads < n_lv)) continue;
      float in[N], out[N];
      io::unpack(w[u], in);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float q = __builtin_amdgcn_fmed3f(__builtin_rintf(in[j] * inv0), lo, hi);
        float y = __builtin_fmaf(q, scale, 0.0f);
        asm("" : "+v"(y));
        out[j] = y;
      }
      io::template store<NT>(ys + ((size_t)b0 + u * kThreads) * N, io::pack(out));
      if ((MODE == 5 || MODE == 8 || (MODE >= 10 && MODE <= 13) || MODE == 16 || (MODE == 7 && u > 0)) && u + 1 < U) { __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_sched_barrier(0); }
    }
    return;
  }
  float d = 0.0f;
  if (MODE == 4) d = table[(blockIdx.x * 2u) & 4095u];
  VI v[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (b0 + u * kThreads < n_lv) v[u] = __builtin_nontemporal_load(reinterpret_cast<const VI*>(xs) + b0 + u * kThreads);
  __builtin_amdgcn_sched_barrier(0);
  if (MODE == 1) d = table[(blockIdx.x * 2u) & 4095u];
  if (MODE == 2) d = table[0];
  if (MODE == 3) {
    __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0) (gfx9 encoding: vmcnt low 4 bits 0, high bits 15:14 0; expcnt 7, lgkmcnt 15)
    __builtin_amdgcn_s_sleep(SLEEP);
    __builtin_amdgcn_sched_barrier(0);
  }
  const float s = MODE == 0 || MODE == 3 ? scale : __builtin_fmaf(d, 0.0f, scale);
  const float inv = 1.0f / s;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (b0 + u * kThreads < n_lv) {
      float in[N], out[N];
      io::unpack(v[u], in);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float q = __builtin_amdgcn_fmed3f(__builtin_rintf(in[j] * inv), lo, hi);
        float y = __builtin_fmaf(q, s, 0.0f);
        asm("" : "+v"(y));
        out[j] = y;
      }
      io::template store<NT>(ys + ((size_t)b0 + u * kThreads) * N, io::pack(out));
    }
  }
}

template <class TI>
static int launch_flat_x(int mode, int sleep, const void* xv, void* yv, int64_t n, float scale, int32_t qmin, int32_t qmax, int nt,
                         const float* table, hipStream_t st) {
  typedef IO<TI, TI> io;
  if (n % io::N || n / io::N >= (1ll << 32) - 4096) return fail_arg("n");
  const uint32_t n_lv = (uint32_t)(n / io::N);
  const unsigned grid = (n_lv + 4 * kThreads - 1) / (4 * kThreads);
  const TI* x = static_cast<const TI*>(xv);
  TI* y = static_cast<TI*>(yv);
  const float lo = (float)qmin, hi = (float)qmax;
#define FX(NT_, M_, S_) hipLaunchKernelGGL((flat_x_kernel<TI, NT_, M_, S_>), dim3(grid), dim3(kThreads), 0, st, x, y, n_lv, scale, lo, hi, table)
#define FXN(M_, S_) do { if (nt == 2) FX(2, M_, S_); else FX(1, M_, S_); } while (0)
  switch (mode) {
    case 0: FXN(0, 0); break;
    case 1: FXN(1, 0); break;
    case 2: FXN(2, 0); break;
    case 4: FXN(4, 0); break;
    case 5: FXN(5, 0); break;
    case 6: FXN(6, 0); break;
    case 7: FXN(7, 0); break;
    case 8: FXN(8, 0); break;
    case 9: FXN(9, 0); break;
    case 10:
      switch (sleep) {
        case 2: FXN(10, 2); break;
        case 4: FXN(10, 4); break;
        case 8: FXN(10, 8); break;
        default: FXN(10, 1); break;
      }
      break;
    case 14: if (sleep == 4) FXN(14, 4); else FXN(14, 1); break;
    case 15: FXN(15, 1); break;
    case 16: FXN(16, 1); break;
    case 11: FXN(11, 0); break;
    case 13: FXN(13, 0); break;
    case 3:
      switch (sleep) {
        case 4: FXN(3, 4); break;
        case 8: FXN(3, 8); break;
        case 16: FXN(3, 16); break;
        case 32: FXN(3, 32); break;
        default: return fail_arg("sleep");
      }
      break;
    default: return fail_arg("mode");
  }
  return check_launch("flat_x");
}

extern "C" int mctq_x_flat(int32_t mode, int32_t sleep, const void* x, void* y, int64_t n, int32_t dtype, float scale, int32_t qmin,
                           int32_t qmax, int32_t nt, const float* table, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case MCTQ_DT_F32: return launch_flat_x<float>(mode, sleep, x, y, n, scale, qmin, qmax, nt, table, st);
    case MCTQ_DT_F16: return launch_flat_x<_Float16>(mode, sleep, x, y, n, scale, qmin, qmax, nt, table, st);
    case MCTQ_DT_BF16: return launch_flat_x<__bf16>(mode, sleep, x, y, n, scale, qmin, qmax, nt, table, st);
    default: return fail_arg("dtype");
  }
}

// mode 1: early loads + IEEE reciprocal; 2: + fast exact reciprocal; 3: no reciprocal (timing bound, wrong results);
// 4: mode 2 looping over `loops` groups per block; 5: the same with the next group's loads in flight (software pipeline)
extern "C" int mctq_x_lastaxis(int32_t mode, int32_t u, int32_t loops, const void* x, void* y, int64_t rows, int64_t channels,
                               int32_t dtype, const float* scales, const int32_t* zps, int32_t qmin, int32_t qmax, int32_t nt,
                               void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (loops < 1) loops = 1;
  if (mode == 22) {                                      // contiguous tiles, repeating columns (lastaxis4)
    switch (dtype) {
      case MCTQ_DT_F32: return launch_lastaxis4<float>(x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      case MCTQ_DT_F16: return launch_lastaxis4<_Float16>(x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      case MCTQ_DT_BF16: return launch_lastaxis4<__bf16>(x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      default: return fail_arg("dtype");
    }
  }
  if (mode == 31 || mode == 32) {                        // table shared through LDS (lastaxis5); `loops` = lane-vectors per slab, at least
    switch (dtype) {
      case MCTQ_DT_F32: return launch_lastaxis5<float>(mode, u, loops, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      case MCTQ_DT_F16: return launch_lastaxis5<_Float16>(mode, u, loops, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      case MCTQ_DT_BF16: return launch_lastaxis5<__bf16>(mode, u, loops, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      default: return fail_arg("dtype");
    }
  }
  if (mode == 21) {                                      // contiguous tiles (lastaxis3)
    switch (dtype) {
      case MCTQ_DT_F32: return launch_lastaxis3<float>(u, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      case MCTQ_DT_F16: return launch_lastaxis3<_Float16>(u, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      case MCTQ_DT_BF16: return launch_lastaxis3<__bf16>(u, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
      default: return fail_arg("dtype");
    }
  }
  switch (dtype) {
    case MCTQ_DT_F32: return launch_lastaxis2<float>(mode, u, loops, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
    case MCTQ_DT_F16: return launch_lastaxis2<_Float16>(mode, u, loops, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
    case MCTQ_DT_BF16: return launch_lastaxis2<__bf16>(mode, u, loops, x, y, rows, channels, scales, zps, qmin, qmax, nt, st);
    default: return fail_arg("dtype");
  }
}

extern "C" int mctq_x_shortrows(int32_t mode, const void* x, void* y, int64_t rows, int64_t inner, int64_t channels, int32_t dtype,
                                const float* scales, const int32_t* zps, int32_t qmin, int32_t qmax, int32_t nt, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case MCTQ_DT_F32: return launch_shortrows<float>(mode, x, y, rows, inner, channels, scales, zps, qmin, qmax, nt, st);
    case MCTQ_DT_F16: return launch_shortrows<_Float16>(mode, x, y, rows, inner, channels, scales, zps, qmin, qmax, nt, st);
    case MCTQ_DT_BF16: return launch_shortrows<__bf16>(mode, x, y, rows, inner, channels, scales, zps, qmin, qmax, nt, st);
    default: return fail_arg("dtype");
  }
}
