// rows_persist_kernel / flat_loop_kernel -- the persistent-block launch shapes of rounds 1-4, kept OUTSIDE the shipped library
// (VERDICT r04 #3): measured 2-5 % slower than one tile per block for every op (profiles/r01, profiles/r04/cfg4_lut_experiments.md).
// Include after mctq_kernels.hpp (namespace mctq).
#pragma once
#include "mctq_kernels.hpp"

namespace mctq {

// ------------------------------------------------------------------------------------------
// Heavy ops: persistent blocks.  The work is cut into tiles of 256*U lane-vectors (tiles never
// cross a row); block b takes tiles b, b+grid, b+2*grid, ... so every CU finishes at the same time
// (a one-block-per-row grid leaves the last round of blocks mostly empty), prefetches the next
// tile's loads before it computes the current one, and pays the table / codebook staging once.
// Full tiles run straight-line code so the LDS table reads of a tile are issued back to back.
// ------------------------------------------------------------------------------------------
template <class Op, class TI, class TO, int U, int NT>
__global__ __launch_bounds__(kThreads) void rows_persist_kernel(Op op, const TI* __restrict__ xs, TO* __restrict__ ys,
                                                                uint32_t tiles_per_row, uint32_t total_tiles,
                                                                uint32_t innerv, uint32_t channels) {
  typedef IO<TI, TO> io;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr uint32_t TILE = kThreads * U;
  const uint32_t G = gridDim.x;
  uint32_t item = blockIdx.x;                                // grid <= total_tiles
  // two tiles of loads in flight per lane: vA = the tile after the current one, vB = the one after that
  typename io::VI vA[U], vB[U];
  auto issue = [&](typename io::VI (&v)[U], uint32_t it) {
    const uint32_t row = it / tiles_per_row, tile = it - row * tiles_per_row;
    const int64_t rbase = (int64_t)row * innerv;
    load_tile<TI, TO, U, NT>(v, xs, rbase + tile * TILE + threadIdx.x, rbase + innerv, (tile + 1) * TILE <= innerv);
  };
  issue(vA, item);
  if (item + G < total_tiles) issue(vB, item + G);
  const typename Op::Book book = op.setup(smem);
  // The parameters of a tile are fetched one iteration ahead (scalar loads + one IEEE reciprocal), so
  // their latency sits under the previous tile's compute instead of in front of the next loads.
  auto params_of = [&](uint32_t it) {
    const uint32_t row = it / tiles_per_row;
    return op.fetch(row >= channels ? row % channels : row);
  };
  typename Op::Param p_next = params_of(item);
  for (; item < total_tiles; item += G) {
    const uint32_t row = item / tiles_per_row;
    const uint32_t tile = item - row * tiles_per_row;
    const typename Op::Param p = p_next;
    const bool fast = __builtin_amdgcn_readfirstlane((int)Op::can_fast(p)) != 0;
    typename io::VI w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { w[u] = vA[u]; vA[u] = vB[u]; }
    if (item + 2 * G < total_tiles) issue(vB, item + 2 * G);
    if (item + G < total_tiles) p_next = params_of(item + G);
    const int64_t rbase = (int64_t)row * innerv;
    const int64_t first = rbase + tile * TILE + threadIdx.x;
    const bool full = (tile + 1) * TILE <= innerv;
    if (fast) finish_tile<true, Op, TI, TO, U, NT>(op, p, book, w, ys, first, rbase + innerv, full);
    else finish_tile<false, Op, TI, TO, U, NT>(op, p, book, w, ys, first, rbase + innerv, full);
  }
}

template <class Op, class TI, class TO, int U, int NT>
__global__ __launch_bounds__(kThreads) void flat_loop_kernel(Op op, typename Op::Param p, const TI* __restrict__ xs,
                                                             TO* __restrict__ ys, int64_t n) {
  typedef IO<TI, TO> io;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int64_t TILE = (int64_t)kThreads * U;
  const int64_t nv = n / io::N;
  const int64_t tiles = (nv + TILE - 1) / TILE;
  typename io::VI v[U];
  load_tile<TI, TO, U, NT>(v, xs, (int64_t)blockIdx.x * TILE + threadIdx.x, nv, ((int64_t)blockIdx.x + 1) * TILE <= nv);
  const typename Op::Book book = op.setup(smem);
  const bool fast = Op::can_fast(p);                        // kernel argument: uniform
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    typename io::VI w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = v[u];
    const int64_t tn = t + gridDim.x;
    if (tn < tiles) load_tile<TI, TO, U, NT>(v, xs, tn * TILE + threadIdx.x, nv, (tn + 1) * TILE <= nv);
    const bool full = (t + 1) * TILE <= nv;
    if (fast) finish_tile<true, Op, TI, TO, U, NT>(op, p, book, w, ys, t * TILE + threadIdx.x, nv, full);
    else finish_tile<false, Op, TI, TO, U, NT>(op, p, book, w, ys, t * TILE + threadIdx.x, nv, full);
  }
  if (blockIdx.x == 0) {
    const int64_t i = nv * io::N + threadIdx.x;
    if (i < n) ys[i] = (TO)op.template apply<false>((float)xs[i], p, book);
  }
}


}  // namespace mctq
