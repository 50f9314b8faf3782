// mctq_misc.hip -- part of libmctq_hip.so: shared state, tuning hook, diagnostics.
#include "mctq_kernels.hpp"

#include <stdlib.h>

#include <mutex>
#include <set>
#include <string>

namespace mctq {

thread_local char g_err[256] = "";
thread_local LaunchNote g_note = {"", "", 0, 0, 0, 0, 0};
static thread_local char g_note_text[160] = "";
int g_nt = 1;
int64_t g_cached_store_max_bytes = 32ll << 20;   // outputs that fit the aggregate L2 stay cached for their consumer
int g_unroll = 4;
int g_heavy_unroll = 0;
int g_rowsteps = 2;           // 2: rowsteps_kernel only where it measured faster (see launch_channels)
int g_paced = 1;              // launches of 3/4 ... 1 round under the paced order of waits: 0 never, 1 inside the window, 2 whenever the kernel is taken (launch_flat, launch_channels)
int g_shortrows = 1;          // short / ragged rows through shortrows_kernel: 0 never, 1 the measured rule, 2 whenever eligible (launch_channels)
int g_ql_variant = 0;
int g_ql_band = 0;
int g_ql_stagger = 0;         // tiled consumer kernel: half of the waves copy after multiplying (experiment: no gain)
int g_ql_rot = 0;            // tiled consumer kernel: blocks that share a weight tile start at different K offsets (experiment: no gain)

static const char* launch_log_path() {
  const char* p = getenv("MCTQ_LAUNCH_LOG");
  return (p && p[0]) ? p : nullptr;
}
int g_launch_log = launch_log_path() != nullptr;

int fail_arg(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return MCTQ_E_ARG;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -(int)e;
  }
  return 0;
}
int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else cus = 256;
  }
  return cus;
}

// Self-test: the fast division of the LUT ops against the compiler's IEEE '/' for EVERY float32 numerator.
__global__ __launch_bounds__(kThreads) void selftest_division_kernel(const float* __restrict__ divisors, int n_div,
                                                                     unsigned long long* __restrict__ mismatches) {
  const uint64_t stride = (uint64_t)gridDim.x * kThreads;
  for (int j = 0; j < n_div; ++j) {
    const LutCommon::Param p = LutCommon::make(divisors[j], divisors[j], 1.0f);
    unsigned int bad = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * kThreads + threadIdx.x; b < (1ull << 32); b += stride) {
      const float x = __uint_as_float((uint32_t)b);
      const float slow = x / p.d;
      const float xmax = 0x1p60f * fabsf(p.ds);
      const float fast = LutCommon::can_fast(p) ? LutCommon::divide_fast(__builtin_amdgcn_fmed3f(x, -xmax, xmax), p.ds, p.r) : x / p.d;
      const float a = fabsf(slow);
      bool ok;
      if (a >= 0x1p-40f && a < 0x1p59f) ok = __float_as_uint(slow) == __float_as_uint(fast);    // exact domain
      else if (slow != slow) ok = LutCommon::can_fast(p) ? (x != x && fast <= -0x1p58f) || (x == x && fast != fast) || fabsf(fast) >= 0x1p58f
                                                         : fast != fast;      // NaN numerator: saturated low (caller checks x)
      else if (a >= 0x1p59f) ok = fabsf(fast) >= 0x1p58f && (slow < 0) == (fast < 0);             // clamps alike
      else ok = fabsf(fast) < 0x1p-39f;                                                           // stays tiny
      bad += ok ? 0u : 1u;
    }
    if (bad) atomicAdd(&mismatches[j], (unsigned long long)bad);
  }
}

// Self-test: recip_exact against the compiler's IEEE 1.0f / d for EVERY float32 bit pattern of its range (both signs).
__global__ __launch_bounds__(kThreads) void selftest_reciprocal_kernel(unsigned long long* __restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * kThreads;
  unsigned long long bad = 0, seen = 0, first = 0;
  for (uint64_t b = (uint64_t)blockIdx.x * kThreads + threadIdx.x; b < (1ull << 32); b += stride) {
    const uint32_t mag = (uint32_t)b & 0x7fffffffu;
    if (mag < kRecipLo || mag > kRecipHi) continue;
    ++seen;
    const float d = __uint_as_float((uint32_t)b);
    if (__float_as_uint(1.0f / d) != __float_as_uint(recip_exact(d))) { ++bad; if (!first) first = b + 1; }
  }
  if (seen) atomicAdd(&out[0], seen);
  if (bad) { atomicAdd(&out[1], bad); atomicMax(&out[2], first); }
}

}  // namespace mctq

using namespace mctq;

extern "C" {

int mctq_abi_version(void) { return MCTQ_ABI_VERSION; }

const char* mctq_last_error(void) { return g_err; }

const char* mctq_last_launch(void) {
  if (!g_note.shape[0]) return "";
  snprintf(g_note_text, sizeof(g_note_text), "%s<%s,in%dB,out%dB,U=%d,NT=%d>", g_note.shape, g_note.op, g_note.in_bytes,
           g_note.out_bytes, g_note.unroll, g_note.nt);
  return g_note_text;
}

int64_t mctq_launch_count(void) { return g_note.count; }

}  // extern "C"

namespace mctq {
// first use of a launch variant by this process -> one line in $MCTQ_LAUNCH_LOG (O_APPEND: several processes may share it)
void log_launch() {
  static std::mutex mu;
  static std::set<std::string> seen;
  const char* path = launch_log_path();
  if (!path) return;
  const std::string text = mctq_last_launch();
  std::lock_guard<std::mutex> lock(mu);
  if (!seen.insert(text).second) return;
  if (FILE* f = fopen(path, "a")) {
    fprintf(f, "%s\n", text.c_str());
    fclose(f);
  }
}
}  // namespace mctq

extern "C" {

#ifndef MCTQ_BUILD_ID
#define MCTQ_BUILD_ID "unstamped"
#endif
// "MCTQ_BUILD_ID=<id>" is also found by scanning the file (hip/build.py: needs_build) without loading it
const char* mctq_build_id(void) {
  static const char text[] = "MCTQ_BUILD_ID=" MCTQ_BUILD_ID;
  return text + sizeof("MCTQ_BUILD_ID=") - 1;
}

int mctq_set_tuning(const char* key, int32_t value) {
  if (!key) return fail_arg("key is NULL");
  if (!strcmp(key, "nt")) {
    if (value != 1 && value != 2) return fail_arg("nt must be 1 or 2");
    g_nt = value;
    return 0;
  }
  if (!strcmp(key, "cached_store_max_mb")) {
    if (value < 0) return fail_arg("cached_store_max_mb must be >= 0");
    g_cached_store_max_bytes = (int64_t)value << 20;
    return 0;
  }
  if (!strcmp(key, "unroll")) {
    if (value != 1 && value != 2 && value != 4) return fail_arg("unroll must be 1, 2 or 4");
    g_unroll = value;
    return 0;
  }
  if (!strcmp(key, "heavy_unroll")) {
    if (value != 0 && value != 1 && value != 2 && value != 4)
      return fail_arg("heavy_unroll must be 0, 1, 2 or 4");
    g_heavy_unroll = value;
    return 0;
  }
  if (!strcmp(key, "paced")) {
    if (value != 0 && value != 1 && value != 2) return fail_arg("paced must be 0, 1 or 2");
    g_paced = value;
    return 0;
  }
  if (!strcmp(key, "shortrows")) {
    if (value != 0 && value != 1 && value != 2) return fail_arg("shortrows must be 0, 1 or 2");
    g_shortrows = value;
    return 0;
  }
  if (!strcmp(key, "rowsteps")) {
    if (value != 0 && value != 1 && value != 2) return fail_arg("rowsteps must be 0, 1 or 2");
    g_rowsteps = value;
    return 0;
  }
  if (!strcmp(key, "ql_variant")) {
    // 0 = automatic; otherwise one of the kernels the automatic choice can select (csrc/mctq_qlinear.hip: qlinear_dispatch)
    static const int ok[] = {0, 181, 182, 184, 83233, 86433, 166623, 86633, 812613, 1612623, 612, 1212, 662, 2544, 2548, 2560};
    bool found = false;
    for (int v : ok) found = found || v == value;
    if (!found) return fail_arg("ql_variant must be 0 (automatic) or one of 181 182 184 (streaming), 83233 86433 86633 812613 (8-wave ring tiles), 166623 1612623 (16-wave), 612 1212 662 (two-buffer tiles), 2544 2548 (wide), 2560 (ping-pong)");
    g_ql_variant = value;
    return 0;
  }
  if (!strcmp(key, "ql_stagger")) {
    if (value != 0 && value != 1) return fail_arg("ql_stagger must be 0 or 1");
    g_ql_stagger = value;
    return 0;
  }
  if (!strcmp(key, "ql_rot")) {
    if (value != 0 && value != 1) return fail_arg("ql_rot must be 0 or 1");
    g_ql_rot = value;
    return 0;
  }
  if (!strcmp(key, "ql_band")) {
    if (value < 0 || value > 4096) return fail_arg("ql_band must be 0 (automatic) or the number of tile rows per band");
    g_ql_band = value;
    return 0;
  }
  return fail_arg("unknown tuning key");
}

int mctq_selftest_division(const float* divisors, int32_t n_div, uint64_t* mismatches, void* stream) {
  if (!divisors || !mismatches || n_div < 1) return fail_arg("bad selftest arguments");
  hipLaunchKernelGGL(selftest_division_kernel, dim3(cu_count() * 8), dim3(kThreads), 0, (hipStream_t)stream,
                     divisors, (int)n_div, reinterpret_cast<unsigned long long*>(mismatches));
  return check_launch("selftest launch");
}

int mctq_selftest_reciprocal(uint64_t* out3, void* stream) {
  if (!out3) return fail_arg("bad selftest arguments");
  hipLaunchKernelGGL(selftest_reciprocal_kernel, dim3(cu_count() * 8), dim3(kThreads), 0, (hipStream_t)stream,
                     reinterpret_cast<unsigned long long*>(out3));
  return check_launch("selftest launch");
}

}  // extern "C"
