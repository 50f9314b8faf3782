/*
 * mctq_hip.h -- C ABI of libmctq_hip.so, the MI355X (gfx950) implementation of the
 * sony/mct_quantizers PyTorch inferable-quantizer hot path.
 *
 * Every entry point replaces one tensor-runtime call site of the reference
 * (paths relative to /root/reference/mct_quantizers/):
 *
 *   mctq_fq_per_tensor_f32    torch.fake_quantize_per_tensor_affine at
 *                             pytorch/quantizers/weights_inferable_quantizers/weights_symmetric_inferable_quantizer.py:147
 *                             .../weights_uniform_inferable_quantizer.py:161
 *                             pytorch/quantizers/activation_inferable_quantizers/activation_symmetric_inferable_quantizer.py:113
 *                             .../activation_uniform_inferable_quantizer.py:124
 *   mctq_fq_per_channel_f32   torch.fake_quantize_per_channel_affine at
 *                             .../weights_symmetric_inferable_quantizer.py:139, .../weights_uniform_inferable_quantizer.py:153
 *   mctq_lut_per_tensor_f32   lut_quantizer (pytorch/quantizer_utils.py:95-139) with a scalar threshold, called from
 *                             .../activation_lut_pot_inferable_quantizer.py:86 and
 *                             .../weights_lut_symmetric_inferable_quantizer.py:114 (per_channel=False)
 *   mctq_lut_per_channel_f32  lut_quantizer with a per-channel threshold, .../weights_lut_symmetric_inferable_quantizer.py:114
 *   mctq_grid_per_*_f32       the export-time functions quantize_*_torch behind `_use_custom_impl and
 *                             torch.jit.is_tracing()` (file:line at the declarations below)
 *   (dtype-generic forms, the decision-table LUT entry points and the extensions -- integer codes, the integer
 *    consumer mctq_qlinear_* -- are documented at their declarations; extensions have no reference call site)
 *
 * Conventions
 *   - x, y, scales, zero_points, thresholds and lut are DEVICE pointers owned by the caller
 *     (torch's caching allocator); the library never allocates, frees, copies or synchronises.
 *   - Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 *     the call returns immediately; it is legal inside hipGraph stream capture.
 *   - A tensor is addressed in its dense storage order as [outer][channels][inner], float32.
 *     The channel of linear element i is (i / inner) % channels.
 *   - Return value: 0 on success, a negative hipError_t on a HIP failure, MCTQ_E_ARG (-10001)
 *     on an invalid argument.  mctq_last_error() gives the message for the calling thread.
 *   - Arithmetic contract (bit exact with the reference on finite inputs, |x/scale| < 2^31):
 *       affine: inv = 1.0f/scale (correctly rounded); q = clamp(rint(x*inv) + zp, qmin, qmax);
 *               y = (q - zp) * scale           rint = round-half-to-even
 *       lut   : t = clamp((x / thr_div) * mult, clip_min, clip_max)  (true IEEE division, NaN kept);
 *               j = first index minimising fl32(|t - lut[j]|); y = (lut[j] / mult) * thr_mul
 *     Outside that domain the kernels saturate: +inf -> qmax, -inf and NaN -> qmin.
 */
#ifndef MCTQ_HIP_H
#define MCTQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v8 (round 5): the compact-decision-table entry points (mctq_lut_compact_words, mctq_lut_build_compact, mctq_lutc_*) and the
 * tuning keys "heavy_persistent", "nt" = 0, "unroll" / "heavy_unroll" = 8 and the experiment codes of "ql_variant" left the
 * library with the kernels behind them (measured, not adopted: tools/experiments/); nothing else changed. */
/* Size limit (since v8): a tensor of more than 2^32 - 8192 elements whose per-channel rows are shorter than 256 lane-vectors is
 * taken in ONE launch by the affine fake-quantizers only (16-byte aligned x / y); the LUT, integer-code and export-grid entry
 * points return MCTQ_E_ARG for it ("per-channel rows shorter than 256 lane-vectors above 2^32 elements: affine quantizers
 * only") -- the Python layer (hip/ops.py: _split_rows) cuts such a tensor into row blocks below the limit and issues one launch
 * per block, which is what a caller of the C ABI has to do as well.  Long rows and per-tensor launches have no such limit. */
/* v9 (round 6): + mctq_selftest_reciprocal, + tuning keys "shortrows", "paced"; the channel-last (lastaxis) launch and the new launch of
 * short / ragged per-channel rows (shortrows) invert a lane's own scales with a five-instruction exact reciprocal (no signature
 * changed; results are bit-identical). */
#define MCTQ_ABI_VERSION 9
#define MCTQ_E_ARG (-10001)

/* storage types of x (and of y for the affine entry points); arithmetic is always float32 */
#define MCTQ_DT_F32 0
#define MCTQ_DT_F16 1
#define MCTQ_DT_BF16 2
#define MCTQ_DT_F64 3   /* float64 tensors: ATen's double arithmetic, see "float64" below */

/* storage types of the integer-code outputs */
#define MCTQ_CODE_I8 0
#define MCTQ_CODE_U8 1
#define MCTQ_CODE_I4 2   /* two codes per byte: element 2j in the low nibble, 2j+1 in the high one */
#define MCTQ_CODE_U4 3

/* ABI version of the loaded library (== MCTQ_ABI_VERSION it was built with). */
int mctq_abi_version(void);

/* Content hash of the sources, headers and compiler flags this binary was built from (hip/build.py: tree_build_id()):
 * the loader refuses a library whose id differs from the tree's, so a stale binary cannot be used silently. */
const char* mctq_build_id(void);

/* Message of the last failing call on this thread ("" if none). */
const char* mctq_last_error(void);

/* Diagnostic: which kernel variant the calling thread's last successful elementwise launch used, e.g.
 * "rows_kernel<AffineOp,in4B,out4B,U=4,NT=1>" ("" before the first launch).  Benchmarks use it to tie profiler
 * counters (profiles/pmc_traffic.json) to the variant they were measured on.  Valid until the next call.
 * With MCTQ_LAUNCH_LOG=<file> in the environment when the library is loaded, every variant is also appended to that file
 * the first time the process takes it (the evidence the set of instantiated kernels was cut against). */
const char* mctq_last_launch(void);

/* Diagnostic: number of kernel launches the library has enqueued from the calling thread since it was loaded (every
 * launch that mctq_last_launch() would name counts once; the one-by-one tail of a batched call counts per tensor).
 * Tests use the difference around a model forward to show how many launches the forward issued: the reference issues
 * one per wrapped weight per forward (pytorch/quantize_wrapper.py:228-240); an accelerated model one per storage type. */
int64_t mctq_launch_count(void);

/* y[i] = (clamp(rint(x[i] * (1/scale)) + zero_point, quant_min, quant_max) - zero_point) * scale, i < n. */
int mctq_fq_per_tensor_f32(const float* x, float* y, int64_t n,
                           float scale, int32_t zero_point, int32_t quant_min, int32_t quant_max,
                           void* stream);

/* Same, with scale/zero point taken per channel: scales[channels] float32, zero_points[channels] int32
 * (zero_points may be NULL, meaning all zero: the symmetric quantizers). */
int mctq_fq_per_channel_f32(const float* x, float* y,
                            int64_t outer, int64_t channels, int64_t inner,
                            const float* scales, const int32_t* zero_points,
                            int32_t quant_min, int32_t quant_max,
                            void* stream);

/*
 * The same two operations for x and y stored as float32, float16 or bfloat16 (dtype = MCTQ_DT_*; y has x's
 * type, as ATen's fake_quantize kernels: float32 arithmetic, one round-to-nearest-even narrowing at the store).
 */
int mctq_fq_per_tensor(const void* x, void* y, int64_t n, int32_t dtype,
                       float scale, int32_t zero_point, int32_t quant_min, int32_t quant_max,
                       void* stream);

int mctq_fq_per_channel(const void* x, void* y,
                        int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                        const float* scales, const int32_t* zero_points,
                        int32_t quant_min, int32_t quant_max,
                        void* stream);

/*
 * float64 (dtype = MCTQ_DT_F64, accepted by mctq_fq_per_tensor, mctq_fq_per_channel, mctq_fq_per_tensor_tqp,
 * mctq_lut_per_tensor, mctq_lut_per_channel and mctq_lut_per_tensor_f64; the reference passes double tensors
 * straight to ATen at the call sites listed at the top).  ATen's double path is not float32 arithmetic on wider
 * storage, and the package reproduces it as measured against the reference (tests/golden/cases_f64.*):
 *   q = clamp(rint(x * (double)(1.0f / scale)) + zp, qmin, qmax)      double product, double rounding
 *   per tensor (float or tensor qparams):  y = (double)((float)(q - zp) * scale)
 *   per channel:                           y = (double)(q - zp) * (double)scale
 *   LUT: the scaled value, the clip and the distances |t - lut[j]| are evaluated in double, the result
 *        (lut[j] / mult) * thr_mul in float32 -- y is float32.
 */

/*
 * Per-tensor fake-quant whose scale and zero point are 1-element DEVICE arrays (read by the kernel through scalar
 * loads; no device->host copy): the tensor-qparams overload of torch.fake_quantize_per_tensor_affine, called by the
 * per-tensor weights quantizers (weights_symmetric_inferable_quantizer.py:147-151, weights_uniform...py:161-165)
 * and recorded as such by an fx trace of a wrapper (saved-model flow, pytorch/load_model.py:23-34).
 */
int mctq_fq_per_tensor_tqp(const void* x, void* y, int64_t n, int32_t dtype,
                           const float* scale, const int32_t* zero_point, int32_t quant_min, int32_t quant_max,
                           void* stream);

/*
 * A LIST of affine fake-quantizations in one call -- one launch per group of up to 48 tensors of one storage
 * type.  PytorchQuantizationWrapper.forward re-quantizes each wrapped layer's weights on every forward
 * (pytorch/quantize_wrapper.py:228-240); a model has tens of such layers, and each separate launch pays its own
 * host cost and ~2 us of ramp/drain on the GPU.  `items` is a HOST array (it is consumed before the call returns:
 * the descriptors and the block -> tensor map travel in the kernel arguments, so the call is legal under hipGraph
 * capture); every pointer
 * inside an item is a DEVICE pointer with the meaning it has in mctq_fq_per_channel.  Per-tensor quantization is
 * outer = channels = 1, inner = n with 1-element device scales / zero_points and flags = MCTQ_FQ_ITEM_PER_TENSOR.  Tensors the batched kernel cannot
 * take (x or y not 16-byte aligned, >= 2^31 elements, float64, more than 2^20 elements in rows shorter than 32) are
 * launched one by one on the same stream, after the batched launches.
 * All items are validated before anything is launched.
 */
typedef struct mctq_fq_item {
  const void* x;
  void* y;
  int64_t outer, channels, inner;
  const float* scales;           /* device float32[channels] */
  const int32_t* zero_points;    /* device int32[channels], or NULL = all zero */
  int32_t quant_min, quant_max;
  int32_t dtype;                 /* MCTQ_DT_*: storage type of x and y */
  int32_t flags;                 /* MCTQ_FQ_ITEM_PER_TENSOR or 0 */
} mctq_fq_item;

/* The item is a per-tensor quantization (torch.fake_quantize_per_tensor_affine with tensor qparams), not a
 * per-channel one that happens to have one channel.  Only float64 tensors can tell the difference (see "float64"). */
#define MCTQ_FQ_ITEM_PER_TENSOR 1

int mctq_fq_batched(const mctq_fq_item* items, int32_t n_items, void* stream);

/*
 * The same list as ONE launch per storage type, whatever its length (a whole model's weights): the descriptors are
 * packed on the host into a table whose DEVICE copy the kernel reads by scalar loads.
 *   mctq_fq_batch_pack  writes the table for `items` into host_table (capacity bytes) and returns its size in bytes;
 *                       with host_table == NULL or a capacity that is too small it writes nothing and returns the size
 *                       needed.  Negative: MCTQ_E_ARG.  Pure host code (no HIP call).
 *   mctq_fq_batch_run   launches a packed table: host_table is the packed bytes (launch geometry, and the tensors
 *                       that are launched one by one, are read from it), device_table a 16-byte aligned device copy
 *                       of the same bytes, made and owned by the caller (the library never allocates or copies).
 * The table holds the items' device pointers: pack again (and refresh the device copy) when one of them changes.
 * Results are bit-identical to mctq_fq_batched and to the single-tensor entry points.
 */
int64_t mctq_fq_batch_pack(const mctq_fq_item* items, int32_t n_items, void* host_table, int64_t capacity);
int mctq_fq_batch_run(const void* host_table, const void* device_table, void* stream);

/*
 * The same table-driven launch for LUT quantizers that have a decision table (mctq_lut_build_table): all LUT weights of a
 * model per forward (weights_lut_symmetric_inferable_quantizer.py:114-122 is called once per wrapped layer,
 * quantize_wrapper.py:228-240), or a group of LUT activation batches.  Per-channel items give `thresholds` (device
 * float32[channels]) and `eps`; per-tensor items give thresholds = NULL and the host floats thr_div / thr_mul with the
 * meaning they have in mctq_lutt_per_tensor (step_round likewise).  y is float32.  Items one grid cannot take (x not
 * vector-aligned, >= 2^31 elements, more than 2^20 elements in rows shorter than 32) are launched one by one.
 * Results are bit-identical to mctq_lutt_per_tensor / mctq_lutt_per_channel.
 */
typedef struct mctq_lut_item {
  const void* x;
  float* y;
  int64_t outer, channels, inner;
  const float* thresholds;       /* device float32[channels], or NULL: per tensor */
  const float* table;            /* device decision table, (entries + 1) x 2 words */
  int32_t entries;
  float eps;
  float thr_div, thr_mul;
  float mult, clip_min, clip_max;
  int32_t dtype;                 /* MCTQ_DT_F32 / F16 / BF16: storage type of x */
  int32_t step_round;
} mctq_lut_item;

int64_t mctq_lutt_batch_pack(const mctq_lut_item* items, int32_t n_items, void* host_table, int64_t capacity);
int mctq_lutt_batch_run(const void* host_table, const void* device_table, void* stream);

/*
 * Integer-code output of the affine quantizers: codes[i] = clamp(rint(x[i] * (1/scale)) + zero_point, quant_min,
 * quant_max), stored as int8 (MCTQ_CODE_I8, domain within [-128, 127]) or uint8 (MCTQ_CODE_U8, within [0, 255]).
 * (codes - zero_point) * scale equals the fake-quantized value of mctq_fq_* bit for bit.  This is the
 * "integer domain" the reference never materialises; consumers that dequantize inside their GEMM
 * read 1 B per element instead of 4.
 * MCTQ_CODE_I4 (domain within [-8, 7], two's-complement nibbles) / MCTQ_CODE_U4 (within [0, 15]): two codes per
 * byte in storage order, element 2j in the low nibble and 2j + 1 in the high nibble; `codes` then holds n / 2
 * bytes (4-byte aligned).  Supported layouts: per tensor with n % 8 == 0; per channel with inner % 8 == 0, or
 * inner == 1 with channels % 8 == 0; anything else returns MCTQ_E_ARG.
 */
int mctq_fq_codes_per_tensor(const void* x, void* codes, int64_t n, int32_t dtype, int32_t code_dtype,
                             float scale, int32_t zero_point, int32_t quant_min, int32_t quant_max,
                             void* stream);

int mctq_fq_codes_per_channel(const void* x, void* codes,
                              int64_t outer, int64_t channels, int64_t inner, int32_t dtype, int32_t code_dtype,
                              const float* scales, const int32_t* zero_points,
                              int32_t quant_min, int32_t quant_max,
                              void* stream);

/*
 * LUT (codebook) quantizer with one threshold for the whole tensor.
 *   thr_div : float32(threshold + eps), the divisor of quantizer_utils.py:169
 *   thr_mul : float32(threshold), the final multiplier of quantizer_utils.py:137
 *   lut     : device float32[n_lut] codebook in the caller's list order (1 <= n_lut <= 4096)
 *   mult    : 2^(lut_values_bitwidth - signed); clip_min/clip_max: the clamp range of :162-167
 */
int mctq_lut_per_tensor_f32(const float* x, float* y, int64_t n,
                            float thr_div, float thr_mul,
                            const float* lut, int32_t n_lut,
                            float mult, float clip_min, float clip_max,
                            void* stream);

/* LUT quantizer with thresholds[channels] (device float32); the divisor is fl32(thresholds[c] + eps). */
int mctq_lut_per_channel_f32(const float* x, float* y,
                             int64_t outer, int64_t channels, int64_t inner,
                             const float* thresholds, float eps,
                             const float* lut, int32_t n_lut,
                             float mult, float clip_min, float clip_max,
                             void* stream);

/*
 * LUT quantizers for x stored as float32 / float16 / bfloat16 (dtype); y is ALWAYS float32, as the
 * reference's op chain promotes to float32 once the float32 codebook enters (quantizer_utils.py:131-137).
 * step_round (per-tensor only): 0, or MCTQ_DT_F16 / MCTQ_DT_BF16 to round the quotient x/thr_div and the
 * scaled value to that type, which is what the reference's chain does to a half-precision activation
 * divided by a Python-float threshold (activation_lut_pot_inferable_quantizer.py:86-91); thr_div must then
 * already be rounded to that type by the caller.  The literal-scan entry points (mctq_lut_*) accept all four
 * storage types (MCTQ_DT_F64: see "float64"); the decision-table ones (mctq_lutt_*) float32 / float16 / bfloat16.
 */
int mctq_lut_per_tensor(const void* x, float* y, int64_t n, int32_t dtype, int32_t step_round,
                        float thr_div, float thr_mul,
                        const float* lut, int32_t n_lut,
                        float mult, float clip_min, float clip_max,
                        void* stream);

int mctq_lut_per_channel(const void* x, float* y,
                         int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                         const float* thresholds, float eps,
                         const float* lut, int32_t n_lut,
                         float mult, float clip_min, float clip_max,
                         void* stream);

/* float64 input with a DOUBLE divisor: the activation LUT quantizer divides a double tensor by the Python float
 * threshold + eps (activation_lut_pot_inferable_quantizer.py:86-91), which stays a double; thr_mul = float32(threshold). */
int mctq_lut_per_tensor_f64(const double* x, float* y, int64_t n,
                            double thr_div, float thr_mul,
                            const float* lut, int32_t n_lut,
                            float mult, float clip_min, float clip_max,
                            void* stream);

/*
 * float64 tensors through a sorted threshold list evaluated in DOUBLE (integer codebooks, centres and clip bounds
 * within 2^20): the reference's chain runs quotient, clip and distances in double for a double tensor
 * (quantizer_utils.py:126-134 under type promotion), so the staircase's thresholds are doubles.
 *   mctq_lut_steps_f64_bytes  upper bound of the blob size for a codebook of n_lut entries
 *   mctq_lut_build_steps_f64  host code: fills steps_host (double T[P], float Q[P], float q_nan, float P) and *p_out = P
 *                             by bisecting the literal double scan; MCTQ_E_ARG when the codebook does not qualify
 *   mctq_luts_per_tensor_f64 / _per_channel_f64   the launches; `steps` is a DEVICE copy of that blob.  Arguments as
 *                             mctq_lut_per_tensor_f64 / mctq_lut_per_channel with dtype MCTQ_DT_F64; y is float32.
 */
int32_t mctq_lut_steps_f64_bytes(int32_t n_lut);
int mctq_lut_build_steps_f64(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                             void* steps_host, int32_t* p_out);
int mctq_luts_per_tensor_f64(const double* x, float* y, int64_t n, double thr_div, float thr_mul, const void* steps,
                             int32_t P, float mult, float clip_min, float clip_max, void* stream);
int mctq_luts_per_channel_f64(const double* x, float* y, int64_t outer, int64_t channels, int64_t inner,
                              const float* thresholds, float eps, const void* steps, int32_t P, float mult,
                              float clip_min, float clip_max, void* stream);

/*
 * Decision-table form of the LUT quantizer (integer codebooks, clip range of at most 1023.5 units).
 *
 * The literal scan above costs ~4 VALU ops per codebook entry per element.  For an integer codebook the
 * result of that scan, as a function of the scaled value t, is a staircase whose steps sit within a few
 * ulps of the half-integer points clip_min + k/2.  mctq_lut_build_table() -- host code, no GPU -- runs the
 * literal scan in float32 and records for every point the exact threshold of its step and the dequantized
 * centres on both sides; the mctq_lutt_* kernels stage that table in LDS and decide each element with one
 * LDS read, one compare and one select, bit-identically to the literal scan for every float32 input.
 *
 *   mctq_lut_table_entries : number K of table points for a clip range (= 2*(clip_max-clip_min)+1), or
 *                            MCTQ_E_ARG if the range is not integral or too large for LDS.
 *   mctq_lut_build_table   : lut_host[n_lut] is a HOST array in the caller's list order; table_host receives
 *                            2*(K+1) 32-bit words: K entries {T_k (float32), half2(q_below, q_above)} and one
 *                            trailer {q for NaN input (float32), K}.  Fails (MCTQ_E_ARG) for a non-integer
 *                            codebook; callers then use the literal kernels.
 *   mctq_lutt_per_tensor_f32 / mctq_lutt_per_channel_f32 : as mctq_lut_per_*_f32, with `table` (DEVICE copy
 *                            of table_host) and `entries` = K in place of the codebook.
 */
int32_t mctq_lut_table_entries(float clip_min, float clip_max);

int mctq_lut_build_table(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                         float* table_host);

int mctq_lutt_per_tensor_f32(const float* x, float* y, int64_t n,
                             float thr_div, float thr_mul,
                             const float* table, int32_t entries,
                             float mult, float clip_min, float clip_max,
                             void* stream);

int mctq_lutt_per_channel_f32(const float* x, float* y,
                              int64_t outer, int64_t channels, int64_t inner,
                              const float* thresholds, float eps,
                              const float* table, int32_t entries,
                              float mult, float clip_min, float clip_max,
                              void* stream);

int mctq_lutt_per_tensor(const void* x, float* y, int64_t n, int32_t dtype, int32_t step_round,
                         float thr_div, float thr_mul,
                         const float* table, int32_t entries,
                         float mult, float clip_min, float clip_max,
                         void* stream);

int mctq_lutt_per_channel(const void* x, float* y,
                          int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                          const float* thresholds, float eps,
                          const float* table, int32_t entries,
                          float mult, float clip_min, float clip_max,
                          void* stream);

/*
 * Threshold-list ("steps") form of the LUT quantizer: INTEGER codebooks whose clip range is too large for the decision
 * table (lut_values_bitwidth > 10; clip bounds and centres within +-2^20).  For such codebooks the literal scan's
 * result is a non-decreasing staircase of the scaled value with one hand-over per pair of adjacent sorted centres;
 * mctq_lut_build_steps() -- host code -- finds each hand-over's exact float32 threshold by bisecting the literal
 * scan over float bit patterns and checks the model on sample points; the mctq_luts_* kernels count the thresholds
 * <= t with a branchless binary search in LDS (log2 of the codebook size reads per element instead of 4 VALU ops per
 * entry), bit-identically to the literal scan (tested for all 2^32 inputs).
 *   mctq_lut_steps_words : upper bound of the array size in floats for a codebook of n_lut entries (lists of 128
 *                          thresholds or more carry a cell index behind the 2 * P + 2 words: see LutCellsOp).
 *   mctq_lut_build_steps : fills steps_host (HOST) and *n_words with the actual size 2 * P + 2; MCTQ_E_ARG for a
 *                          non-integer codebook or one that fails the staircase check (use the literal kernels then).
 *   mctq_luts_per_tensor / _per_channel : as mctq_lutt_*, with `steps` (DEVICE copy) and `n_words` in place of the table.
 */
int32_t mctq_lut_steps_words(int32_t n_lut);

int mctq_lut_build_steps(const float* lut_host, int32_t n_lut, float mult, float clip_min, float clip_max,
                         float* steps_host, int32_t* n_words);

int mctq_luts_per_tensor(const void* x, float* y, int64_t n, int32_t dtype, int32_t step_round,
                         float thr_div, float thr_mul,
                         const float* steps, int32_t n_words,
                         float mult, float clip_min, float clip_max,
                         void* stream);

int mctq_luts_per_channel(const void* x, float* y,
                          int64_t outer, int64_t channels, int64_t inner, int32_t dtype,
                          const float* thresholds, float eps,
                          const float* steps, int32_t n_words,
                          float mult, float clip_min, float clip_max,
                          void* stream);

/*
 * Export-time arithmetic: what the reference's quantizers compute while an ONNX export traces them
 * (`self._use_custom_impl and torch.jit.is_tracing()`), a different last-ulp contract from the fake-quant
 * entry points above: clip, TRUE division by the step, round half even, scale back.
 *     c = x < lo ? lo : x;   c = x > hi ? hi : c;                 (NaN and signed zeros pass through)
 *     y = shifted ? step * rint((c - lo) / step) + lo  :  rint(c / step) * step
 * Replaces (relative to mct_quantizers/pytorch/quantizers/):
 *   weights_inferable_quantizers/weights_symmetric_inferable_quantizer.py:32-70  quantize_sym_weights_torch
 *       (lo = -thr, hi = thr - scale, step = scale = thr / 2^(n-1); also weights_pot_inferable_quantizer.py:106-122)
 *   weights_inferable_quantizers/weights_uniform_inferable_quantizer.py:34-78   quantize_uniform_weights_torch
 *       (lo, hi = range adjusted to contain 0, step = (hi - lo) / (2^n - 1), shifted = 0)
 *   activation_inferable_quantizers/activation_symmetric_inferable_quantizer.py:29-54  quantize_sym_activations_torch
 *   activation_inferable_quantizers/activation_uniform_inferable_quantizer.py:32-65   quantize_uniform_activations_torch
 *       (shifted = 1)
 * float32 in and out; per-channel tables are DEVICE float32[channels]; layout arguments as mctq_fq_per_channel.
 */
int mctq_grid_per_tensor_f32(const float* x, float* y, int64_t n,
                             float lo, float hi, float step, int32_t shifted,
                             void* stream);

int mctq_grid_per_channel_f32(const float* x, float* y,
                              int64_t outer, int64_t channels, int64_t inner,
                              const float* los, const float* his, const float* steps, int32_t shifted,
                              void* stream);

/*
 * Per-tensor codes with a layout change: x [batch][channels][pixels] (NCHW, pixels = H * W) -> codes
 * [batch][pixels][channels] (NHWC), int8 / uint8, same arithmetic as mctq_fq_codes_per_tensor.  What a pointwise
 * (1x1) convolution consumer feeds mctq_qlinear_i8 with (rows = pixels, K = channels) when the activation arrives
 * in PyTorch's default layout.
 */
int mctq_fq_codes_nchw_to_nhwc(const void* x, void* codes, int64_t batch, int64_t channels, int64_t pixels, int32_t dtype,
                               int32_t code_dtype, float scale, int32_t zero_point, int32_t quant_min, int32_t quant_max,
                               void* stream);

/*
 * Integer consumer of the codes (extension; the reference has no counterpart): the product a wrapped
 * torch.nn.Linear computes on fake-quantized operands -- PytorchQuantizationWrapper.forward
 * (pytorch/quantize_wrapper.py:231-257: quantize the weight, then self.layer(x)) fed by an activation holder
 * (pytorch/activation_quantization_holder.py:53) -- evaluated on the 8-bit clamp indices instead:
 *     y[m][n] = float( sum_k (a[m][k] - a_zero_point) * w[n][k] ) * (a_scale * w_scales[n]) + bias[n]
 * with exact int32 accumulation on the integer matrix cores; float32 multiply, then float32 add, each
 * rounded once.  a_codes [M][K] int8 or uint8 (a_code_dtype = MCTQ_CODE_I8 / MCTQ_CODE_U8, from
 * mctq_fq_codes_per_tensor), w_codes [N][K] int8 with zero point 0 (symmetric / power-of-two weights, from
 * mctq_fq_codes_per_channel with axis 0 or _per_tensor), w_scales float32[N], w_rowsum int32[N] =
 * sum_k w[n][k] (computed once per weight), bias float32[N] or NULL, y float32 [M][N]; all DEVICE pointers,
 * code matrices 16-byte aligned, K % 16 == 0, K <= 32768.  The result differs from the float32 product of
 * the dequantized operands only by that product's own rounding (it is the exact sum, scaled once).
 */
int mctq_qlinear_i8(const void* a_codes, int32_t a_code_dtype, int32_t a_zero_point, float a_scale,
                    const int8_t* w_codes, const float* w_scales, const int32_t* w_rowsum, const float* bias,
                    float* y, int64_t M, int64_t N, int64_t K, void* stream);

/*
 * Same product, but the result leaves as the NEXT layer's activation codes: the float32 value y[m][n] above is
 * quantized in registers exactly as mctq_fq_codes_per_tensor would quantize it,
 *     code = clamp(rint(y * (1.0f / y_scale)) + y_zero_point, y_quant_min, y_quant_max),
 * and stored as int8 / uint8 (y_code_dtype) -- for chains activation holder -> wrapped Linear -> activation
 * holder -> wrapped Linear, where the float32 tensor between two layers is only ever quantized again.
 */
int mctq_qlinear_i8_codes(const void* a_codes, int32_t a_code_dtype, int32_t a_zero_point, float a_scale,
                          const int8_t* w_codes, const float* w_scales, const int32_t* w_rowsum, const float* bias,
                          void* y_codes, int32_t y_code_dtype, float y_scale, int32_t y_zero_point, int32_t y_quant_min,
                          int32_t y_quant_max, int64_t M, int64_t N, int64_t K, void* stream);

/*
 * The same consumer for 4-bit weights (quantizers with num_bits <= 4), streamed at half a byte per weight: the
 * weight-streaming kernel only, meant for few rows (every M is computed correctly; beyond ~64 rows the int8 tiled
 * path is faster).  w_codes4 [N][K / 2] holds the codes in the CONSUMER layout, not the storage-order packing of
 * MCTQ_CODE_I4: each group of 8 consecutive k is 4 bytes, byte j = (code[k = j] & 0xF) | (code[k = j + 4] << 4),
 * two's-complement nibbles in [-8, 7]; 8-byte aligned rows (K % 16 == 0).  w_rowsum[n] = sum_k code[n][k].
 * y_code_dtype < 0: y is float32 [M][N]; otherwise y holds the next layer's codes as in mctq_qlinear_i8_codes.
 */
int mctq_qlinear_w4a8(const void* a_codes, int32_t a_code_dtype, int32_t a_zero_point, float a_scale,
                      const uint8_t* w_codes4, const float* w_scales, const int32_t* w_rowsum, const float* bias,
                      void* y, int32_t y_code_dtype, float y_scale, int32_t y_zero_point, int32_t y_quant_min,
                      int32_t y_quant_max, int64_t M, int64_t N, int64_t K, void* stream);

/*
 * Tuning hook (benchmarks only): selects the launch variant used by later calls on any thread.
 *   key "nt"     : 1 = non-temporal loads and stores (default), 2 = non-temporal loads with cached stores
 *   key "cached_store_max_mb" : with nt = 1, outputs of at most this many MiB are stored through the caches
 *                  (mode 2) so that a consumer launched right after finds them in L2 / the Infinity Cache; default 32, 0 = never
 *   key "unroll" : upper bound of the 16-byte accesses in flight per lane the affine kernels choose from (1, 2 or 4; default 4)
 *   key "heavy_unroll" : the decision-table LUT kernels' lane-vectors per tile (0 = automatic, 1, 2 or 4)
 *   key "rowsteps" : per-channel rows of two or three whole 256-lane-vector steps: 0 = one- / two-step tiles inside a row
 *                  (rows_kernel), 1 = four steps per block across row boundaries (rowsteps_kernel), 2 (default) = rowsteps_kernel
 *                  when its grid is one round of resident blocks, where it measured 5-10 % faster (64 MiB launches)
 *   key "ql_variant" : mctq_qlinear_i8 launch shape: 0 = automatic (cost model, csrc/mctq_qlinear.hip: qlinear_dispatch), or ONE of
 *                  the kernels that choice can select: 181 / 182 / 184 (weight streaming, 16 / 32 / 64 rows per pass), 83233, 86433,
 *                  86633, 812613 (8-wave ring tiles 32x32 ... 128x64), 166623, 1612623 (16-wave), 612, 1212, 662 (two-buffer tiles),
 *                  2544, 2548 (wave-wide tiles), 2560 (256 x 256 ping-pong)
 *   key "ql_band" : tile rows per XCD band of the tiled kernel (0 = automatic); "ql_rot", "ql_stagger": experiments of the
 *                  tiled kernel (K rotation between blocks sharing a weight tile; half of the waves copy after multiplying), 0 / 1, default 0
 *   key "paced" : the affine per-tensor launch through flat_paced_kernel (flat_kernel's tile under another order of waits: loads
 *                  128 clocks apart, every load landed before the first store, every store completed before the next lane-vector):
 *                  0 = never, 1 (default) = launches that fill 3/4 ... 1 round of resident blocks, where it measured 4-6 % faster
 *                  (48-64 MiB of traffic on this chip; 3-9 % slower outside), 2 = every launch with a full four-vector tile.
 *                  The same key sends symmetric float32 per-channel launches of that window (whole-vector rows) through
 *                  shortrows_kernel with paced stores: 2048 x 4096 12.4 -> 11.3 us, 32768 x 256 12.3 -> 11.3 (16-bit: not taken, +3 %)
 *   key "shortrows" : affine per-channel tensors through shortrows_kernel (per-lane parameter reads behind the tile's data loads,
 *                  no LDS window): 0 = never, 1 (default) = where it measured faster (16-bit storage: every short or ragged row
 *                  shape, and long rows of launches that fill 7/8 ... 1 round of resident blocks; float32: rows of 4 ... 31
 *                  elements), 2 = every eligible tensor (rows of at least one lane-vector, fewer than 2^24 elements per row)
 * Only variants a default dispatcher can select are instantiated; every value of every key is exercised by the GPU tests.
 * Returns 0, or MCTQ_E_ARG for an unknown key/value.  Numerical results never depend on it.
 */
int mctq_set_tuning(const char* key, int32_t value);

/*
 * Diagnostic: checks the LUT kernels' shared-divisor division (reciprocal + two exact FMA residual
 * corrections) against the compiler's IEEE division for ALL 2^32 float32 numerators, for each of
 * divisors[n_div] (device float32).  mismatches[n_div] (device uint64, zeroed by the caller) receives the
 * number of numerators whose quotient differs inside the domain where the last bit can matter
 * (2^-40 <= |x/d| < 2^59); outside it the two results must agree on sign and saturation (a NaN numerator comes
 * out saturated low: the kernels test the input itself for NaN).
 */
int mctq_selftest_division(const float* divisors, int32_t n_div, uint64_t* mismatches, void* stream);

/*
 * Diagnostic: checks the five-instruction reciprocal the channel-last and short-row launches use for a lane's own scales
 * (v_rcp_f32 + two Newton steps with exact FMA residuals; csrc/mctq_kernels.hpp: recip_exact) against the compiler's IEEE
 * 1.0f / d for EVERY float32 bit pattern with 2^-100 <= |d| <= 2^100 (outside that range the kernels use the IEEE division).
 * out3 (device uint64[3], zeroed by the caller) receives {patterns checked, mismatches, one mismatching pattern + 1}.
 */
int mctq_selftest_reciprocal(uint64_t* out3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCTQ_HIP_H */
