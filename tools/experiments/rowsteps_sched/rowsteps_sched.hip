// rowsteps_sched.hip -- EXPERIMENT, not part of the shipped library: the four-steps-per-block kernel of
// short whole-step rows (rowsteps_kernel) with its instruction schedule WRITTEN instead of left to the compiler.
// Round 4 found the shipped form fast (11.8 us on 4096^2 bfloat16) and cleaner source forms slow (12.3-13.2 us).
// First pass of this experiment (profiles/r05/rowsteps_sched_hoisted.log): every written schedule lost 6-10 % -- because LLVM
// commons the row arithmetic (s0 / steps_per_row: an integer division with uniform branches) of the full and the guarded
// path into a place ABOVE everybody's loads, whatever the source order; the shipped form happens to keep its loads first.
// Second pass (this file): steps per row as a COMPILE-TIME constant SPR (1, 2 or 3: the only rows the kernel is for), so that the
// row arithmetic is a shift / multiply-high and its place does not matter; then the schedules are compared again.
// MODE (full blocks; the one partial block at the end of a launch takes the guarded path of every mode):
//   0  the shipped source form, run-time steps_per_row (as compiled here)
//   1  pinned 3 + 1: loads 0-2 | parameters | result 0 | load 3 | results 1-3 | four stores
//   2  pinned 3 + 1, each result stored as soon as it exists
//   3  pinned 2 + 2: loads 0-1 | parameters | result 0 | load 2 | result 1 | load 3 | results 2-3 | four stores
//   4  four loads | parameters | results as the loads land | four stores
//   5  four loads | parameters | each result stored as soon as it exists
//   6  the shipped source form with compile-time SPR
// (In modes 1-5 the plain loads are SUNK to their first use, below the parameter fetch's uniform branches, whatever the source
// order: sched_barrier constrains one basic block, not MachineSink.  Volatile loads stay put but are each followed by
// s_waitcnt vmcnt(0); an asm use of the loaded registers forces the wait as well.  A C++ source cannot pin this schedule.)
// Build: python tools/build_variant.py rowsteps_sched -> tools/ablate/libmctq_hip_rowsteps_sched.so; driver: run.py beside this file.
#include "mctq_kernels.hpp"

namespace mctq {

#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE, class TI, int NT, int SPR>     // SPR 0: run-time steps_per_row
__global__ __launch_bounds__(kThreads) void rowsteps_x_kernel(const TI* __restrict__ xs, TI* __restrict__ ys,
                                                              uint32_t steps_per_row_rt, uint32_t total_steps, uint32_t channels,
                                                              AffineOp op) {
  typedef IO<TI, TI> io;
  constexpr int U = 4;
  const uint32_t spr = SPR ? (uint32_t)SPR : steps_per_row_rt;
  const uint32_t s0 = blockIdx.x * U;
  const int64_t first = (int64_t)s0 * kThreads + threadIdx.x;
  const bool full = s0 + U <= total_steps;                   // wave-uniform
  const NoBook book;
  auto ld = [&](int u) { return io::template load<NT>(xs + (first + u * kThreads) * io::N); };
  auto st = [&](int u, typename io::VO r) { io::template store<NT>(ys + (first + u * kThreads) * io::N, r); };
  auto res = [&](typename io::VI v, const AffineOp::Param& p) {
    typename io::VI one[1] = {v};
    typename io::VO out[1];
    run_vectors<true, AffineOp, TI, TI, 1>(op, one, out, p, book);
    return out[0];
  };
  if (MODE == 0 || MODE == 6 || !full) {                      // the shipped source form (also every mode's partial block)
    typename io::VI v[U];
    if (full) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld(u);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (s0 + u < total_steps) v[u] = ld(u);
    }
    SB();
    uint32_t row = s0 / spr, rem = s0 - row * spr;
    AffineOp::Param p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (full || s0 + u < total_steps) p[u] = op.fetch(row >= channels ? row % channels : row);
      if (++rem == spr) { rem = 0; ++row; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (full || s0 + u < total_steps) st(u, res(v[u], p[u]));
    return;
  }
  typename io::VI v0, v1, v2, v3;
  typename io::VO r0, r1, r2, r3;
  AffineOp::Param p0, p1, p2, p3;
  auto params = [&]() {                                       // one division per block, then the steps walk the rows (as shipped)
    uint32_t row = s0 / spr, rem = s0 - row * spr;
    auto next = [&]() { const uint32_t c = row >= channels ? row % channels : row; if (++rem == spr) { rem = 0; ++row; } return c; };
    p0 = op.fetch(next()); p1 = op.fetch(next()); p2 = op.fetch(next()); p3 = op.fetch(next());
  };
  if (MODE == 1 || MODE == 2) {
    v0 = ld(0); v1 = ld(1); v2 = ld(2); SB();
    params(); SB();
    r0 = res(v0, p0); SB();
    v3 = ld(3); SB();
    if (MODE == 2) { st(0, r0); SB(); }
    r1 = res(v1, p1); SB();
    if (MODE == 2) { st(1, r1); SB(); }
    r2 = res(v2, p2); SB();
    if (MODE == 2) { st(2, r2); SB(); }
    r3 = res(v3, p3); SB();
    if (MODE != 2) { st(0, r0); st(1, r1); st(2, r2); }
    st(3, r3);
  } else if (MODE == 3) {
    v0 = ld(0); v1 = ld(1); SB();
    params(); SB();
    r0 = res(v0, p0); SB();
    v2 = ld(2); SB();
    r1 = res(v1, p1); SB();
    v3 = ld(3); SB();
    r2 = res(v2, p2); SB();
    r3 = res(v3, p3); SB();
    st(0, r0); st(1, r1); st(2, r2); st(3, r3);
  } else {                                                    // MODE 4, 5
    v0 = ld(0); v1 = ld(1); v2 = ld(2); v3 = ld(3); SB();
    params(); SB();
    r0 = res(v0, p0); SB();
    if (MODE != 4) { st(0, r0); SB(); }
    r1 = res(v1, p1); SB();
    if (MODE != 4) { st(1, r1); SB(); }
    r2 = res(v2, p2); SB();
    if (MODE != 4) { st(2, r2); SB(); }
    r3 = res(v3, p3); SB();
    if (MODE == 4) { st(0, r0); st(1, r1); st(2, r2); }
    st(3, r3);
  }
}

template <int MODE, class TI, int NT>
static int launch_nt(const void* x, void* y, int64_t total_steps, uint32_t spr, int64_t channels, const AffineOp& op, hipStream_t st) {
  const dim3 grid((unsigned)((total_steps + 3) / 4));
#define GO(SPR_) hipLaunchKernelGGL((rowsteps_x_kernel<MODE, TI, NT, SPR_>), grid, dim3(kThreads), 0, st, (const TI*)x, (TI*)y, spr, \
                                    (uint32_t)total_steps, (uint32_t)channels, op)
  if (MODE == 0) GO(0);
  else if (spr == 1) GO(1);
  else if (spr == 2) GO(2);
  else if (spr == 3) GO(3);
  else return fail_arg("compile-time modes: rows of 1, 2 or 3 steps");
#undef GO
  return check_launch("rowsteps experiment launch");
}

template <int MODE, class TI>
static int launch_x(const void* x, void* y, int64_t rows, int64_t innerv, const float* scales, int qmin, int qmax, int nt, hipStream_t st) {
  AffineOp op; op.scales = scales; op.zps = nullptr; op.lo = (float)qmin; op.hi = (float)qmax;
  const uint32_t spr = (uint32_t)(innerv / kThreads);
  if (nt == 2) return launch_nt<MODE, TI, 2>(x, y, rows * spr, spr, rows, op, st);
  return launch_nt<MODE, TI, 1>(x, y, rows * spr, spr, rows, op, st);
}

}  // namespace mctq

using namespace mctq;

// rows x inner elements, per-channel along dim 0 (channels == rows); dtype MCTQ_DT_F32 / MCTQ_DT_BF16; inner = 2 or 3 whole
// 256-lane-vector steps; nt 1 or 2
extern "C" int mctq_x_rowsteps(int32_t mode, const void* x, void* y, int64_t rows, int64_t inner, int32_t dtype, const float* scales,
                               int32_t qmin, int32_t qmax, int32_t nt, void* stream) {
  const int64_t n_per = dtype == MCTQ_DT_F32 ? 4 : 8;
  if (inner % (n_per * kThreads) != 0) return fail_arg("inner must be whole steps");
  const int64_t innerv = inner / n_per;
  hipStream_t st = (hipStream_t)stream;
#define CASE(M_)                                                                                          \
  case M_: return dtype == MCTQ_DT_F32 ? launch_x<M_, float>(x, y, rows, innerv, scales, qmin, qmax, nt, st)   \
                                       : launch_x<M_, __bf16>(x, y, rows, innerv, scales, qmin, qmax, nt, st);
  switch (mode) {
    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6)
    default: return fail_arg("mode");
  }
#undef CASE
}
