// mctq_lut_table.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h); kernels: mctq_kernels.hpp
#include "mctq_kernels.hpp"

using namespace mctq;

extern "C" {

// ---- LUT, decision table --------------------------------------------------------------------------

int32_t mctq_lut_table_entries(float clip_min, float clip_max) {
  const int k = table_entries(clip_min, clip_max);
  if (k < 0) return fail_arg("decision table unsupported for this clip range");
  return k;
}

int mctq_lut_build_table(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                         float* table_host) {
  if (!lut_host || !table_host) return fail_arg("NULL pointer");
  if (n_lut < 1 || n_lut > 4096) return fail_arg("n_lut must be in [1, 4096]");
  if (int rc = check_pow2(mult)) return rc;
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

int mctq_lutt_per_tensor(const void* x, float* y, int64_t n, int32_t dtype, int32_t step_round, float thr_div,
                         float thr_mul, const float* table, int32_t entries, float mult, float clip_min,
                         float clip_max, void* stream) {
  if (n < 0) return fail_arg("n < 0");
  if (n > 0 && (!x || !y)) return fail_arg("x or y is NULL");
  if (step_round != 0 && step_round != MCTQ_DT_F16 && step_round != MCTQ_DT_BF16) return fail_arg("bad step_round");
  LutTableOp op;
  if (int rc = make_table_op(op, nullptr, 0.f, table, entries, mult, clip_min, clip_max, step_round)) return rc;
  const LutCommon::Param p = LutCommon::make(thr_div, thr_mul, mult);
  return with_lut_types(dtype, [&](auto ti, auto to) {
    return launch_flat<decltype(ti), decltype(to)>(op, p, x, y, n, table_bytes(entries), (hipStream_t)stream);
  });
}

int mctq_lutt_per_channel(const void* x, float* y, int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                          const float* thresholds, float eps, const float* table, int32_t entries, float mult,
                          float clip_min, float clip_max, void* stream) {
  if (outer < 0 || channels < 0 || inner < 0) return fail_arg("negative extent");
  const int64_t n = outer * channels * inner;
  if (n > 0 && (!x || !y || !thresholds)) return fail_arg("NULL pointer");
  LutTableOp op;
  if (int rc = make_table_op(op, thresholds, eps, table, entries, mult, clip_min, clip_max, 0)) return rc;
  return with_lut_types(dtype, [&](auto ti, auto to) {
    return launch_channels<decltype(ti), decltype(to)>(op, x, y, outer, channels, inner, table_bytes(entries),
                                                       (hipStream_t)stream);
  });
}

int mctq_lutt_per_tensor_f32(const float* x, float* y, int64_t n, float thr_div, float thr_mul, const float* table,
                             int32_t entries, float mult, float clip_min, float clip_max, void* stream) {
  return mctq_lutt_per_tensor(x, y, n, MCTQ_DT_F32, 0, thr_div, thr_mul, table, entries, mult, clip_min, clip_max, stream);
}

int mctq_lutt_per_channel_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                              const float* thresholds, float eps, const float* table, int32_t entries, float mult,
                              float clip_min, float clip_max, void* stream) {
  return mctq_lutt_per_channel(x, y, outer, channels, inner, MCTQ_DT_F32, thresholds, eps, table, entries, mult,
                               clip_min, clip_max, stream);
}

}  // extern "C"
