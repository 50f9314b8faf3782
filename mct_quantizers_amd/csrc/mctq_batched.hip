// mctq_batched.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h)
//
// A LIST of affine fake-quantizations in one launch.  The reference re-quantizes every wrapped layer's weights
// on every forward (pytorch/quantize_wrapper.py:228-240: one quantizer call per weight attribute), i.e. tens of
// launches per model forward, each paying its own launch cost and ~2 us of ramp/drain (profiles/r02).  Here the
// descriptors of up to kMaxBatch tensors travel in the kernel arguments (no device-side table, no memcpy: legal
// under hipGraph capture) and one grid covers all of them; block -> (tensor, tile) by a scan of the
// wave-uniform tile offsets.
//
// Per tile (256 * U lane-vectors of one tensor, never crossing tensors):
//   - the tile lies inside ONE (outer, channel) row  -> the row's scale / zero point arrive by scalar loads and
//     sit in SGPRs, exactly as rows_kernel does (Linear / conv weights quantized along axis 0, per-tensor items);
//   - the tile covers exactly TWO rows (rows longer than a tile that do not divide into tiles): both rows' parameters in
//     SGPRs, a lane-vector selects by its position relative to the boundary;
//   - otherwise every lane-vector finds its row with one 32-bit division and reads its parameters from the
//     (L1/L2-resident) tables; vectors that straddle rows go element by element.
// Arithmetic: AffineOp (mctq_kernels.hpp), the same expression as every other affine entry point.
#include "mctq_kernels.hpp"

using namespace mctq;

namespace mctq {

constexpr int kMaxBatch = 32;

struct BatchItem {
  const void* x;
  void* y;
  const float* scales;
  const int32_t* zps;
  uint32_t n;            // elements, < 2^31
  uint32_t inner;
  uint32_t channels;
  uint32_t tile_begin;   // first tile (= block) of this tensor in the launch
  float lo, hi;
};

struct BatchArgs {
  BatchItem it[kMaxBatch];
  int n_items;
};

template <class TI, class TO, int U, int NT>
__global__ __launch_bounds__(kThreads) void batched_kernel(const BatchArgs a) {
  typedef IO<TI, TO> io;
  constexpr uint32_t N = io::N;
  constexpr uint32_t TILE = kThreads * U * N;
  int i = 0;
  for (int j = 1; j < a.n_items; ++j)                       // wave-uniform scan over <= 32 kernel-argument words
    if (blockIdx.x >= a.it[j].tile_begin) i = j;
  const BatchItem& it = a.it[i];
  const TI* __restrict__ x = static_cast<const TI*>(it.x);
  TO* __restrict__ y = static_cast<TO*>(it.y);
  const uint32_t n = it.n, inner = it.inner, channels = it.channels;
  const uint32_t e0 = (blockIdx.x - it.tile_begin) * TILE;
  const uint32_t count = n - e0 < TILE ? n - e0 : TILE;
  AffineOp op;
  op.scales = it.scales; op.zps = it.zps; op.lo = it.lo; op.hi = it.hi;
  const NoBook book;

  // data loads first; the row search / parameter fetch below runs under their latency
  typename io::VI v[U];
  const bool full = count == TILE;                          // wave-uniform
  if (full) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = io::template load<NT>(x + e0 + (u * kThreads + threadIdx.x) * N);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t off = (u * kThreads + threadIdx.x) * N;
      if (off + N <= count) v[u] = io::template load<NT>(x + e0 + off);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  const uint32_t row0 = e0 / inner;                         // uniform
  const uint32_t row_last = (e0 + count - 1) / inner;
  if (row0 == row_last) {
    // ---- one row: parameters in SGPRs ----
    const uint32_t c = channels > 1 ? row0 % channels : 0;
    const AffineOp::Param p = op.fetch(c);
    if (full) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float in[N], out[N];
        io::unpack(v[u], in);
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) out[j] = op.apply(in[j], p, book);
        io::template store<NT>(y + e0 + (u * kThreads + threadIdx.x) * N, io::pack(out));
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t off = (u * kThreads + threadIdx.x) * N;
        if (off + N <= count) {
          float in[N], out[N];
          io::unpack(v[u], in);
#pragma unroll
          for (uint32_t j = 0; j < N; ++j) out[j] = op.apply(in[j], p, book);
          io::template store<NT>(y + e0 + off, io::pack(out));
        } else {
          for (uint32_t j = 0; j < N && off + j < count; ++j)
            y[e0 + off + j] = (TO)op.apply((float)x[e0 + off + j], p, book);
        }
      }
    }
    return;
  }

  if (row_last == row0 + 1) {
    // ---- exactly two rows (rows at least a tile long, e.g. 11008-wide Linear weights): both parameter sets in SGPRs,
    //      a lane-vector picks by its position relative to the row boundary -- no per-lane division or table read ----
    const uint32_t c0 = channels > 1 ? row0 % channels : 0;
    const uint32_t c1 = channels > 1 ? (c0 + 1 == channels ? 0 : c0 + 1) : 0;
    const AffineOp::Param p0 = op.fetch(c0), p1 = op.fetch(c1);
    const uint32_t bnd = (row0 + 1) * inner - e0;             // elements of the tile that belong to row0 (0 < bnd < count)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t off = (u * kThreads + threadIdx.x) * N;
      if (off >= count) continue;
      if (off + N <= count) {
        float in[N], out[N];
        io::unpack(v[u], in);
        if (off + N <= bnd || off >= bnd) {
          const bool first = off + N <= bnd;
          AffineOp::Param p;
          p.s = first ? p0.s : p1.s; p.inv = first ? p0.inv : p1.inv; p.zf = first ? p0.zf : p1.zf;
#pragma unroll
          for (uint32_t j = 0; j < N; ++j) out[j] = op.apply(in[j], p, book);
        } else {                                              // the vector straddles the boundary (inner % N != 0)
#pragma unroll
          for (uint32_t j = 0; j < N; ++j) out[j] = op.apply(in[j], off + j < bnd ? p0 : p1, book);
        }
        io::template store<NT>(y + e0 + off, io::pack(out));
      } else {
        for (uint32_t j = 0; j < N && off + j < count; ++j)
          y[e0 + off + j] = (TO)op.apply((float)x[e0 + off + j], off + j < bnd ? p0 : p1, book);
      }
    }
    return;
  }

  // ---- several rows in the tile: per lane-vector parameters ----
  // Pass 1 finds every vector's row and issues its table reads (U independent loads in flight, not U dependent
  // round trips to L2); pass 2 inverts the scales and applies.
  uint32_t cc[U], rr[U];
  float sv[U];
  int32_t zv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    const uint32_t pos = e0 + (off < count ? off : 0);
    const uint32_t row = pos / inner;
    rr[u] = pos - row * inner;
    cc[u] = channels > 1 ? row % channels : 0;
    sv[u] = op.scales[cc[u]];
    zv[u] = op.zps ? op.zps[cc[u]] : 0;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = (u * kThreads + threadIdx.x) * N;
    if (off >= count) continue;
    uint32_t rem = rr[u], c = cc[u];
    if (off + N <= count) {
      float in[N], out[N];
      io::unpack(v[u], in);
      if (rem + N <= inner) {                               // the vector lies in one row
        const AffineOp::Param p = AffineOp::make(sv[u], zv[u]);
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) out[j] = op.apply(in[j], p, book);
      } else {
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
          out[j] = op.apply(in[j], op.fetch(c), book);
          if (++rem == inner) { rem = 0; if (++c == channels) c = 0; }
        }
      }
      io::template store<NT>(y + e0 + off, io::pack(out));
    } else {
      for (uint32_t j = 0; j < N && off + j < count; ++j) {
        y[e0 + off + j] = (TO)op.apply((float)x[e0 + off + j], op.fetch(c), book);
        if (++rem == inner) { rem = 0; if (++c == channels) c = 0; }
      }
    }
  }
}

template <class TI, class TO>
static int launch_batch(const BatchArgs& a, uint32_t tiles, int64_t out_bytes, hipStream_t st) {
  constexpr int U = 4;
  MCTQ_WITH_MODE(nt_mode(out_bytes) == 0 ? 1 : nt_mode(out_bytes), false, {
    hipLaunchKernelGGL((batched_kernel<TI, TO, U, NT>), dim3(tiles), dim3(kThreads), 0, st, a);
    note<AffineOp, TI, TO>("batched_kernel", U, NT);
  });
  return check_launch("batched launch");
}

template <class TI, class TO>
static uint32_t batch_tile_elems() { return kThreads * 4 * IO<TI, TO>::N; }

}  // namespace mctq

extern "C" {

int mctq_fq_batched(const mctq_fq_item* items, int32_t n_items, void* stream) {
  if (n_items < 0) return fail_arg("n_items < 0");
  if (n_items > 0 && !items) return fail_arg("items is NULL");
  hipStream_t st = (hipStream_t)stream;
  // validate everything before the first launch: a bad descriptor must not leave the list half done
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
  for (int dt = MCTQ_DT_F32; dt <= MCTQ_DT_BF16; ++dt) {
    BatchArgs a;
    a.n_items = 0;
    uint32_t tiles = 0;
    int64_t out_bytes = 0;
    const uint32_t tile_e = dt == MCTQ_DT_F32 ? batch_tile_elems<float, float>() : batch_tile_elems<_Float16, _Float16>();
    const size_t esz = dt == MCTQ_DT_F32 ? 4 : 2;
    auto flush = [&]() -> int {
      if (a.n_items == 0) return 0;
      int rc;
      if (dt == MCTQ_DT_F32) rc = launch_batch<float, float>(a, tiles, out_bytes, st);
      else if (dt == MCTQ_DT_F16) rc = launch_batch<_Float16, _Float16>(a, tiles, out_bytes, st);
      else rc = launch_batch<__bf16, __bf16>(a, tiles, out_bytes, st);
      a.n_items = 0; tiles = 0; out_bytes = 0;
      return rc;
    };
    for (int32_t k = 0; k < n_items; ++k) {
      const mctq_fq_item& d = items[k];
      if (d.dtype != dt) continue;
      const int64_t n = d.outer * d.channels * d.inner;
      if (n == 0) continue;
      const bool aligned = (((uintptr_t)d.x | (uintptr_t)d.y) & 15u) == 0;
      if (!aligned || n >= (1ll << 31) - (int64_t)tile_e || d.channels > 0x7fffffffLL || d.inner > 0x7fffffffLL) {
        // not batchable (unaligned view, huge tensor): the single-tensor entry point, same stream
        if (int rc = mctq_fq_per_channel(d.x, d.y, d.outer, d.channels, d.inner, d.dtype, d.scales, d.zero_points,
                                         d.quant_min, d.quant_max, stream)) return rc;
        continue;
      }
      const uint32_t t = (uint32_t)((n + tile_e - 1) / tile_e);
      if (a.n_items == kMaxBatch || (uint64_t)tiles + t > 0x7fffffffu) {
        if (int rc = flush()) return rc;
      }
      BatchItem& b = a.it[a.n_items++];
      b.x = d.x; b.y = d.y; b.scales = d.scales; b.zps = d.zero_points;
      b.n = (uint32_t)n; b.inner = (uint32_t)d.inner; b.channels = (uint32_t)d.channels;
      b.tile_begin = tiles;
      b.lo = (float)d.quant_min; b.hi = (float)d.quant_max;
      tiles += t;
      out_bytes += n * (int64_t)esz;
    }
    if (int rc = flush()) return rc;
  }
  for (int32_t k = 0; k < n_items; ++k) {                   // float64 tensors: one launch each
    const mctq_fq_item& d = items[k];
    if (d.dtype != MCTQ_DT_F64 || d.outer * d.channels * d.inner == 0) continue;
    int rc;
    if ((d.flags & MCTQ_FQ_ITEM_PER_TENSOR) && d.zero_points)
      rc = mctq_fq_per_tensor_tqp(d.x, d.y, d.outer * d.channels * d.inner, d.dtype, d.scales, d.zero_points, d.quant_min,
                                  d.quant_max, stream);
    else if (d.flags & MCTQ_FQ_ITEM_PER_TENSOR)
      return fail_arg("a float64 per-tensor item needs a zero_points pointer");
    else
      rc = mctq_fq_per_channel(d.x, d.y, d.outer, d.channels, d.inner, d.dtype, d.scales, d.zero_points, d.quant_min,
                               d.quant_max, stream);
    if (rc) return rc;
  }
  return 0;
}

}  // extern "C"
