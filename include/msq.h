/*
 * msq.h -- C ABI of libmsq_hip.so, the MI355X (gfx950) native back end of the
 * MicroScopiQ outlier-aware microscaling quant/dequant + mixed-precision Linear
 * hot path.
 *
 * Drop-in boundary.  The reference routes its native work through the pybind
 * module `custom_extensions.funcs` (number_system/mx/custom_extensions.py:19,
 * number_system/mx/cpp/funcs.cpp:218-226) when `custom_cuda=True`
 * (elemwise_ops.py:117-129, mx_ops.py:363-422).  Each entry point below names
 * the reference interface it replaces.  Entry points marked NEW have no
 * reference counterpart (the reference is fake-quant only) and
 * are specified by the fake-quant maths of utils/quant.py:147-266.
 *
 * Conventions (reference: cpp/funcs.h:11-13, cpp/mx.cu:25-26, common.cuh:211-219)
 *   - plain pointers + sizes, no torch types; all buffers are DEVICE memory,
 *     contiguous, owned and allocated by the caller; the library never
 *     allocates, frees or synchronises (graph-capture safe);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns an int status: MSQ_OK or a negative MSQ_ERR_*;
 *     the reference's AT_ASSERTM / exit() paths become error codes, never exit;
 *     msq_last_error() returns a thread-local message for the last failure;
 *   - stateless and re-entrant; rounding-mode ints are the reference's
 *     RoundingMode {nearest=0 (half away), floor=1, even=2} (formats.py:15-18);
 *   - tensors are described as [pre, axis_len, post] with the quantisation
 *     blocks running along the middle axis (a 2-D weight [O,I] with the
 *     reference harness' axes=[0] is pre=1, axis_len=O, post=I; axes=[-1] is
 *     pre=O, axis_len=I, post=1).
 */
#ifndef MSQ_H
#define MSQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSQ_OK 0
#define MSQ_ERR_BAD_ARG (-1)
#define MSQ_ERR_UNSUPPORTED (-2)
#define MSQ_ERR_LAUNCH (-3)

/* rounding modes: formats.py:15-18 == cpp/common.cuh:130-134 */
#define MSQ_RD_NEAREST 0
#define MSQ_RD_FLOOR 1
#define MSQ_RD_EVEN 2

/* element formats: ids 1..10 are the reference's ElemFormat values (formats.py:25-38);
 * posit ids are an extension: MSQ_FMT_POSIT(n, es). */
#define MSQ_FMT_INT8 1
#define MSQ_FMT_INT4 2
#define MSQ_FMT_INT2 3
#define MSQ_FMT_FP8_E5M2 4
#define MSQ_FMT_FP8_E4M3 5
#define MSQ_FMT_FP6_E3M2 6
#define MSQ_FMT_FP6_E2M3 7
#define MSQ_FMT_FP4_E2M1 8
#define MSQ_FMT_FP16 9
#define MSQ_FMT_BF16 10
#define MSQ_FMT_POSIT(n, es) (0x100 | ((n) << 2) | (es))

/* outlier quantiser variants */
#define MSQ_VARIANT_QUANT 0 /* utils/quant.py:147-266 (canonical) */
#define MSQ_VARIANT_MXOPS 1 /* number_system/mx/mx_ops.py:210-330 (used by MXLinear) */

/* status bits written to *status_flag by the outlier kernels */
#define MSQ_STATUS_NAN 1 /* one of the reference's NaN asserts (utils/quant.py:225-250) would fire */
#define MSQ_STATUS_INEXACT 2 /* pack: a value is not exactly code * 2^scale (never silently wrong) */
#define MSQ_STATUS_TIMEOUT 4 /* msq_gptq_block: a workgroup gave up waiting at the grid barrier (CUs held by another stream or process): the result is invalid */

/* GEMM-ready plane kinds (what the fused kernel converts in-register with the CDNA4
 * v_cvt_scalef32_pk_bf16_{fp4,fp8,bf8} instructions) */
#define MSQ_PLANE_NONE 0
#define MSQ_PLANE_FP4 1  /* e2m1 nibbles                      (inlier formats fp4, int2)           */
#define MSQ_PLANE_FP8 2  /* OCP e4m3 bytes                    (outlier formats fp8_e4m3, fp4, fp6, int4) */
#define MSQ_PLANE_BF8 3  /* OCP e5m2 bytes                    (outlier format fp8_e5m2)            */
#define MSQ_PLANE_BF16 4 /* final values, bf16, no scale      (posit / int8 outliers; any format when
                            the inlier plane is NONE: the whole fake-quant value)                 */
#define MSQ_PLANE_U8 5   /* unified: ONE e4m3 byte per weight + one E8M0 scale per 32 k; inlier plane
                            NONE (inlier codes are widened to e4m3 at pack time)                   */
#define MSQ_PLANE_U8X 6  /* unified e4m3 byte + 1 extension bit per weight (a 4th fraction bit: holds
                            posit<8,1> outliers exactly); the extension plane travels as inl_plane   */

/* packed layouts */
#define MSQ_LAYOUT_PLANES 0  /* "MSQ-T1": inlier plane + outlier plane + two scales per block          */
#define MSQ_LAYOUT_UNIFIED 1 /* "MSQ-U1": one 8-bit code plane (+ extension bits) + one scale per 32 k */

int msq_version(void);
const char* msq_last_error(void);

/* formats.py:65-129 _get_format_params (host, no GPU needed).  kind: 0 float/int, 1 posit. */
int msq_format_id(const char* name);
int msq_format_params(int fmt, int* ebits, int* mbits, int* emax, float* max_norm, float* min_norm,
                      int* kind);

/* replaces quantize_elemwise_func_cuda (cpp/funcs.cpp:183-200, cpp/elemwise.cu:12-75).
 * dtype: 0 = f32, 1 = f16, 2 = bf16 (the reference supports f32 and f16). */
int msq_quantize_elemwise(const void* in, void* out, int64_t n, int dtype, int bits, int exp_bits,
                          float max_norm, int rmode, int saturate_normals, int allow_denorm,
                          void* stream);

/* NEW: elementwise round to a NAMED format id (posit<n,es> included) with the saturating,
 * denorm-keeping codec MicroScopiQ uses (elemwise_ops.py:84-174 with saturate_normals=True,
 * allow_denorm=True; posit: number_system/posit/Posit.py:221-385 round-to-nearest-even). */
int msq_quantize_format(const float* in, float* out, int64_t n, int fmt, int rmode, void* stream);

/* replaces quantize_mx_func_cuda (cpp/funcs.cpp:138-159, cpp/mx.cu:13-72):
 * max_values is [pre, post] = max |.| over the whole axis (the caller has already
 * reshaped to blocks, mx_ops.py:397-415). */
int msq_quantize_mx(const float* in, float* out, const float* max_values, int64_t pre,
                    int64_t axis_len, int64_t post, int scale_bits, int elem_ebits, int elem_mbits,
                    float elem_max_norm, int flush_fp32_subnorms, int rmode, void* stream);

/* replaces quantize_mx_by_tile_func_cuda (cpp/funcs.cpp:161-181, cpp/mx.cu:76-170):
 * tiles of tile_size along the axis, ragged last tile NOT padded, shared scale from
 * the max biased exponent computed in-kernel (single pass). */
int msq_quantize_mx_by_tile(const float* in, float* out, int64_t pre, int64_t axis_len, int64_t post,
                            int tile_size, int scale_bits, int elem_ebits, int elem_mbits,
                            float elem_max_norm, int flush_fp32_subnorms, int rmode, void* stream);
/* the same with the divisor of the reference's PYTHON path, `2**shared_exp + 1e-6` in fp32 (number_system/mx/mx_ops.py:444,
 * taken when mx_specs["custom_cuda"] is False): ties of the scaled element round DOWN for every scale below 2^5.  The
 * native kernel above (cpp/mx.cuh:132) and the upstream OCP-MX KATs divide by the scale itself. */
int msq_quantize_mx_by_tile_py(const float* in, float* out, int64_t pre, int64_t axis_len, int64_t post,
                               int tile_size, int scale_bits, int elem_ebits, int elem_mbits,
                               float elem_max_norm, int flush_fp32_subnorms, int rmode, void* stream);
/* Both Python-path traits as switches.  py_exponent: the block's shared exponent is floor(torch.log2(max|A|)) in fp32
 * (mx_ops.py:66-77 on the Python path) -- one HIGHER than the exponent field for the up to 88 largest floats under a power of two,
 * where the native kernel (cpp/shared_exp.cuh:14-53) reads the field itself -- and, under round = "floor", the element's private
 * exponent likewise (elemwise_ops.py:139-144).  py_divisor: as msq_quantize_mx_by_tile_py.  (0, 0) = msq_quantize_mx_by_tile,
 * (1, 1) = msq_quantize_mx_by_tile_py; `_quantize_mx` of the package calls (its divisor switch, 1). */
int msq_quantize_mx_by_tile_ex(const float* in, float* out, int64_t pre, int64_t axis_len, int64_t post,
                               int tile_size, int scale_bits, int elem_ebits, int elem_mbits,
                               float elem_max_norm, int flush_fp32_subnorms, int rmode, int py_divisor, int py_exponent, void* stream);

/* `_quantize_mx` (number_system/mx/mx_ops.py:332-457) on a HALF-PRECISION tensor, as the reference executes it: its native kernel
 * takes float32 only (cpp/mx.cu:124-125), so a Half / BFloat16 tensor runs the Python path op by op in the tensor dtype (ATen:
 * each op in fp32, result rounded to the dtype): floor(log2) rounds values just under a power of two up, `2**e + 1e-6` (:444) is
 * rounded to the dtype, 2**e can leave the fp16 range.  1-11 % of the elements differ from "upcast, fp32, round once".  in / out:
 * [pre, axis_len, post] tensors of dtype 1 = fp16 / 2 = bf16, blocks of `block` (8 ... 128) along the axis, last block zero
 * padded; elem_fmt = a float / int MSQ_FMT_* id; status_flag (device int, may be NULL) receives MSQ_STATUS_NAN when a shared
 * exponent exceeds the scale range (the reference stores NaN).  One pass, no casts: the KV-cache MX variant (config 4). */
int msq_quantize_mx_lowp(const void* in, void* out, int dtype, int64_t pre, int64_t axis_len, int64_t post, int block,
                         int scale_bits, int elem_fmt, int rmode, int flush_fp32_subnorms, int* status_flag, void* stream);

/* replace reduce_sum_inner_dim / reduce_max_inner_dim (cpp/funcs.cpp:203-215, cpp/reduce.cu:19-93):
 * out[outer] = sum / max over the innermost `inner` elements. */
int msq_reduce_sum_inner(const float* in, float* out, int64_t outer, int64_t inner, void* stream);
int msq_reduce_max_inner(const float* in, float* out, int64_t outer, int64_t inner, void* stream);

/* NEW -- the MicroScopiQ fake-quant (utils/quant.py:147-266 quantize_mx_outlier_v1 and
 * :23-146 quantize_mx_outlier_hessian; variant 1 = mx_ops.py:210-330) fused into one
 * read + one write.  Optional outputs (NULL to skip):
 *   mask  uint8 [pre,axis_len,post]  (utils/quant.py:460-495 _extract_outlier_indices)
 *   e_in / e_out  float [pre,nblk,post] clamped shared exponents (NaN kept)
 *   num_outliers  int8 [ceil(nblk/block)*post] (utils/quant.py:66; needs pre == 1)
 *   status_flag   int (device): MSQ_STATUS_* bits OR-ed in
 *   workspace     device scratch of msq_outlier_workspace_bytes() bytes: variant 1 needs it; variant 0 uses it for fp16 / bf16 tensors
 *                 computed in their dtype (MSQ_DTYPE_*_NATIVE: the list of waves the packed kernels hand back to the op-by-op kernel,
 *                 csrc/msq_quant_lowp.hip) -- NULL there runs the op-by-op kernel alone: same results, a third of the speed
 * in/out dtype: 0 = f32 (bit-exact vs the reference); 1 = fp16 / 2 = bf16 tensors (read as f32 values, computed in f32, one
 *   round-to-nearest-even on the way out: the same bits as upcasting, running dtype 0 and casting back), built for
 *   round-to-nearest with float / int inlier formats, variant 0; the remaining combinations return
 *   MSQ_ERR_UNSUPPORTED (the Python shim upcasts).
 *   MSQ_DTYPE_F16_NATIVE / MSQ_DTYPE_BF16_NATIVE = fp16 / bf16 tensors COMPUTED IN THAT DTYPE: every torch op of
 *   utils/quant.py:147-266 rounded back to the tensor dtype as ATen's CPU half kernels do -- what the reference's RTN
 *   harness executes on an fp16 checkpoint (llm/llama.py:238) -- bit-exact vs the reference on half tensors
 *   (variant 0, float / int element formats, no num_outliers; otherwise MSQ_ERR_UNSUPPORTED). */
#define MSQ_DTYPE_F16_NATIVE 0x11
#define MSQ_DTYPE_BF16_NATIVE 0x12
int64_t msq_outlier_workspace_bytes(int64_t pre, int64_t axis_len, int64_t post, int block, int variant);
/* floor(log2(v)) as torch evaluates it on a Half (dtype 1) / BFloat16 (dtype 2) tensor: floor(R(log2f(v))), which
 * rounds values just under a power of two up to the next exponent (utils/quant.py:525-529 and elemwise_ops.py:139 on
 * half tensors).  v: non-negative values of that dtype held as f32.  Diagnostic entry (exhaustively tested). */
int msq_floor_log2_lowp(const float* v, float* out, int64_t n, int dtype, void* stream);
int msq_outlier_fakequant(const void* in, void* out, uint8_t* mask, float* e_in, float* e_out,
                          int8_t* num_outliers, int* status_flag, void* workspace,
                          int64_t workspace_bytes, int dtype, int64_t pre,
                          int64_t axis_len, int64_t post, int block, int inlier_fmt,
                          int outlier_fmt, int inlier_scale_bits, int outlier_scale_bits,
                          float std_dev, int rmode, int flush_fp32_subnorms, int variant,
                          void* stream);

/* ---------------------------------------------------------------------------
 * NEW -- packed ("GEMM-ready", tile-major) weight format and the fused
 * unpack-dequant-GEMM.  The reference has no packed format (llm/opt.py:254-264
 * "TODO: perform packing on GPU", Quant3Linear undefined); the contract is the
 * fake-quant value: unpack(pack(W)) == quantize_mx_outlier_v1(W) bit for bit.
 *
 * Layout (DESIGN.md "MSQ-T1"): W is [N, K] (out_features, in_features), blocks of
 * `block` run along K (the reference's axes=[-1]); N % 64 == 0, K % 64 == 0.
 * The tensor is cut into 64(n) x 64(k) tiles, tile index = (n/64) * (K/64) + k/64;
 * inside a tile the data is stored exactly as one wavefront consumes it for
 * v_mfma_f32_16x16x32_bf16 (lane l: column c = l & 15, k-group g = l >> 4; fragment
 * (nf, kf) = 8 elements n = 16 nf + c, k = 32 kf + 8 g + j):
 *   inlier plane : per tile 2 slots (kf) of 64 lanes x 16 B; dword nf of a lane = 8 e2m1 nibbles (j)
 *   outlier plane: per tile 4 slots (kf*2 + nf/2) of 64 lanes x 16 B (8-bit kinds: 2 dwords per
 *                  fragment) or 8 slots (kf*4 + nf) for the bf16 kind
 *   scale plane  : per tile 16 B per lane group: byte nf*4 + kf*2 + {0: inlier E8M0, 1: outlier E8M0
 *                  (= e_out - e_in + 127)}; one group per column (16 groups) when block >= 32,
 *                  one per lane (64 groups) when block < 32.
 * Zero is always stored as +0 so that inlier and outlier parts combine with a bitwise OR.
 *
 * Unified layout ("MSQ-U1", MSQ_PLANE_U8 / _U8X with in_kind NONE): every fake-quant value of a 32-k
 * group of one column is code * 2^scale with ONE shared E8M0 scale and an e4m3 code (inlier e2m1 codes
 * widen exactly; outlier e4m3 codes are kept as they are whenever the outlier holds the group maximum):
 *   code plane   : as the 8-bit outlier plane above (4 slots per tile)
 *   scale plane  : per tile 16 groups (columns c) x 8 B: byte kf*4 + nf
 *   extension    : (U8X only, passed as inl_plane) per tile 2 slots (kf) of 64 lanes x 4 B: the bit of
 *                  element j of fragment nf sits at bit (3 + 16 (j & 1) + 4 nf + (j >> 1)) mod 32, so that
 *                  rotating by 4 nf + (j >> 1) and masking with 0x00080008 drops it onto bf16 mantissa bit 3
 * 8.25 (U8) / 9.25 (U8X) bits per weight.  The pack kernel decodes every code exactly as the GEMM will
 * and raises MSQ_STATUS_INEXACT if a group cannot be represented (range of more than ~2^15 inside one
 * group); callers then keep the two-plane layout for that tensor.
 * ------------------------------------------------------------------------- */
int msq_packed_kinds(int inlier_fmt, int outlier_fmt, int* in_kind, int* out_kind);
/* layout-aware form; MSQ_ERR_UNSUPPORTED when the formats cannot use the unified layout */
int msq_packed_kinds_layout(int inlier_fmt, int outlier_fmt, int layout, int* in_kind, int* out_kind);
int msq_packed_sizes(int64_t N, int64_t K, int block, int in_kind, int out_kind, int64_t* inl_bytes,
                     int64_t* out_bytes, int64_t* scale_bytes, int64_t* workspace_bytes);

/* quantise W [N,K] f32 (utils/quant.py:147-266 semantics, axes=[-1]) and emit the planes.
 * status_flag (device int, may be NULL) receives MSQ_STATUS_* bits. */
int msq_outlier_pack(const float* W, void* inl_plane, void* out_plane, void* scale_plane,
                     int* status_flag, void* workspace, int64_t workspace_bytes, int64_t N, int64_t K,
                     int block, int inlier_fmt, int outlier_fmt, int inlier_scale_bits,
                     int outlier_scale_bits, float std_dev, int rmode, int flush_fp32_subnorms,
                     int variant, int layout, void* stream);

/* pack a dense tensor that already holds fake-quant VALUES (any quantiser: blocks along out_features as
 * in the reference harness default llm/llama.py:229-237, the GPTQ solver's output llm/gptq.py:166, ...) into a
 * single-plane kind with in_kind MSQ_PLANE_NONE: MSQ_PLANE_U8, MSQ_PLANE_U8X or MSQ_PLANE_BF16.  Nothing is
 * rounded: MSQ_STATUS_INEXACT is raised if a value does not fit the kind (callers try U8, U8X, BF16 in turn). */
int msq_pack_values(const float* Wq, void* inl_plane, void* out_plane, void* scale_plane, int* status_flag,
                    int64_t N, int64_t K, int in_kind, int out_kind, void* stream);

/* planes -> dense dequantised W [N,K]; out_dtype 0 = f32, 2 = bf16 (both exact). */
int msq_outlier_unpack(const void* inl_plane, const void* out_plane, const void* scale_plane, void* W_out,
                       int out_dtype, int64_t N, int64_t K, int block, int in_kind, int out_kind,
                       void* stream);

/* fused unpack-dequant-GEMM: Y[M,N] = X[M,K] (bf16) . W^T (+ bias f32 [N] or NULL), fp32 accumulate on
 * v_mfma_f32_16x16x32_bf16; y_dtype 0 = f32, 1 = fp16, 2 = bf16 (the 16-bit kinds share one kernel: the epilogue converts either way).  Replaces the dense F.linear the reference
 * runs on the fake-quantised weight (number_system/mx/linear.py:91, llm/llama.py:255-256).
 * Shapes: N % 256 == 0, K % 64 == 0, any M >= 0.  M <= 32 (<= 64 for the 4096 x 4096 class, <= 36 / 48 for the unified layouts
 * with more than 16384 / 8192 columns) takes the decode kernels, which stream the packed weight once: for the unified layouts
 * with more than 8192 columns one launch writes Y directly (one block per 64-column strip over all of K, its waves' k-runs summed
 * in LDS), otherwise one wave per (64 columns, k-chunk) and fp32 partial tiles through `workspace`.
 * For larger M with a grid that does not fill the chip the GEMM splits K over several workgroups and reduces fp32
 * partial tiles from `workspace` (msq_qlinear_workspace_bytes(); NULL or too small = single pass, never an error).
 * Results are bit-identical from run to run on every path (fixed summation order). */
int64_t msq_qlinear_workspace_bytes(int64_t M, int64_t N, int64_t K);
/* Which kernel family msq_qlinear_bf16 (mx_wf < 0, plane kind out_kind) or msq_qlinear_mx_w4a8 (mx_wf = weight operand format 0 .. 3)
 * picks for a shape -- host logic only, no launch, no device needed: decode kernels (weight streaming, M <= 32-64), the 128- / 64-row
 * GEMM with split-K (k_qgemm3 / k_mxgemm), or the hand-allocated kernels k_qgemm256 / k_mxgemm256 in their 256-row or 128-row
 * form (whichever covers the grid in fewer rounds over the 256 CUs).  Negative: shape not supported.
 * No reference counterpart (the reference has one dense F.linear, number_system/mx/linear.py:91); for tests and capacity planning. */
#define MSQ_KERNEL_DECODE  0
#define MSQ_KERNEL_GEMM128 1
#define MSQ_KERNEL_T256    2
#define MSQ_KERNEL_T128    3
#define MSQ_KERNEL_PERSISTENT 4   /* k_qgemm256p: persistent 256-row blocks, stream-K over the part-filled last round (MSQ_GEMM_256=3) */
#define MSQ_KERNEL_STREAMK 5      /* k_qgemm_sk (round 6): 32 < M <= 256 on wide projections -- one strip (or strip pair) per block, K cut over the block's
                                   * waves, partial tiles summed in LDS in a fixed order: no split-K planes, one launch (MSQ_GEMM_SK=0 disables, 1-3 force a form) */
int msq_qlinear_kernel_choice(int64_t M, int64_t N, int64_t K, int out_kind, int mx_wf);
/* The GEMM kernel instantiation the same call would launch, as text ("k_qgemm256<6, uint16_t, 16>"): the dispatcher's own decision function,
 * tuning switches (MSQ_GEMM_256 / MSQ_MX_256) included; y_dtype 0 = float32 output, else 16-bit.  For measurement labels (bench.py). */
int msq_qlinear_kernel_name(int64_t M, int64_t N, int64_t K, int out_kind, int mx_wf, int y_dtype, char* buf, int cap);
/* Tuning switches without the environment (thread-safe: an atomic per key): key "MSQ_GEMM_256" (-1 = the rules, 0 = k_qgemm3 only, 1 / 2 =
 * force the 256- / 128-row form of k_qgemm256, 3 = the persistent kernel where its plan applies) or "MSQ_MX_256" (-1, 0, 1, 2 likewise for
 * k_mxgemm256); value INT_MIN hands the key back to the environment variable of the same name, which is otherwise read per call (for
 * single-threaded A / B runs).  "MSQ_MX_LOWP_PAIR4" (1 default / 0): msq_quantize_mx_lowp on a strided axis of whole 32-blocks through
 * k_mx_lowp_pair4 (a block row cut over four waves) or k_mx_lowp_pair.  The other test / A-B switches likewise (override, else the environment per
 * call): "MSQ_ACT_ROWS" (0: the mx_ops activation quantiser as two launches), "MSQ_MX_PACK_BLOCK" (1: one lane per block in msq_mx_pack_a8),
 * "MSQ_VEC_GENERIC" (1: the vector ops through the run-time-parameter rounding), "MSQ_RMS_RPB" (rows per block of the register RMSNorm kernel), "MSQ_PACK_TWO_PASS" (1: msq_outlier_pack in two passes),
 * "MSQ_OUTLIER_LOWP_PK" (0: msq_outlier_fakequant in the tensor dtype through k_outlier_lowp only, not the packed kernels).
 * Returns MSQ_ERR_UNSUPPORTED for an unknown key. */
int msq_set_tuning(const char* key, int value);
/* Schedule of the persistent fused GEMM k_qgemm256p (csrc/msq_gemm256p.hip) for a shape -- host arithmetic only, no device needed; for tests
 * and capacity planning.  The T = ceil(M / 256) (N / 256) output tiles are dealt to P resident workgroups (one per CU; cus = CU count,
 * 0 = 256): `full` = T / P whole rounds, and the R = T - full P tiles of the part-filled last round as a stream of R (K / 64) K-steps cut
 * into runs of q K-steps, one run per workgroup (stream-K); a tile cut between runs is summed through 256 KiB fp32 slots in the
 * workspace: ws_bytes = 4096 + 262144 per run, 0 when no tile is cut.  Returns 0, or -1 when the kernel does not apply (N % 256, K % 128).
 * No reference counterpart (the reference has one dense F.linear, number_system/mx/linear.py:91). */
int msq_qgemm256p_plan(int64_t M, int64_t N, int64_t K, int cus, int* P, int* full, int* R, int* q, int64_t* ws_bytes);
/* The work list of workgroup b under such a plan (KT = K / 64): up to `cap` segments, 6 ints each {tile id, first K-step, end K-step, role,
 * first peer, last peer}; role 0 = writes Y, 1 = leaves its fp32 partial tile in workspace slot b, 2 = adds the slots of workgroups
 * peer0 .. peer1 (in that order) and writes Y.  Returns the number of segments.  The same code runs on the device. */
int msq_qgemm256p_segments(int b, int P, int full, int R, int KT, int q, int* out, int cap);
int msq_qlinear_bf16(const void* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                     const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int block,
                     int in_kind, int out_kind, void* workspace, int64_t workspace_bytes, void* stream);
/* The same call on fp16 activations (an fp16 model, llm/llama.py:33) at the decode sizes (M <= 32; <= 64 for the 4096 x 4096 class):
 * the weight-streaming kernels convert them to bf16 (round to nearest even = x.to(bfloat16)) while loading: no cast launch in front of
 * every projection.  Other shapes: MSQ_ERR_UNSUPPORTED (cast to bf16 and call msq_qlinear_bf16). */
int msq_qlinear_f16x(const void* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                     const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int block,
                     int in_kind, int out_kind, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * W4A8 Linear -- NEW; replaces the activation quantisation + F.linear of
 * number_system/mx/linear.py:66-73,91 (MXLinear with an 8-bit a_elem_format; BASELINE config 3).
 *   msq_act_quant_bf16: X [M,K] f32 -> MicroScopiQ outlier-aware MX fake-quant along K
 *       (variant 0 = utils/quant.py:147-266, 1 = mx_ops.py:210-330 as MXLinear uses it), written as
 *       bf16.  Exact for element formats of <= 8 bits (MSQ_STATUS_INEXACT is raised otherwise).
 *   msq_qlinear_w4a8: the same followed by the fused dequant-GEMM on packed weights;
 *       Y [M,N] = Q_a(X) . W^T + bias.  Workspace from msq_qlinear_w4a8_workspace_bytes.
 * ------------------------------------------------------------------------- */
int64_t msq_act_quant_workspace_bytes(int64_t M, int64_t K, int block, int variant);
int msq_act_quant_bf16(const float* X, void* Xq, int* status_flag, void* workspace, int64_t workspace_bytes,
                       int64_t M, int64_t K, int block, int inlier_fmt, int outlier_fmt,
                       int inlier_scale_bits, int outlier_scale_bits, float std_dev, int rmode,
                       int flush_fp32_subnorms, int variant, void* stream);
/* the same with bfloat16 activations X (a bf16 model): no cast pass in front; the mx_ops variant needs block 32 or 64 */
int msq_act_quant_bf16_x16(const void* X, void* Xq, int* status_flag, void* workspace, int64_t workspace_bytes,
                       int64_t M, int64_t K, int block, int inlier_fmt, int outlier_fmt,
                       int inlier_scale_bits, int outlier_scale_bits, float std_dev, int rmode,
                       int flush_fp32_subnorms, int variant, void* stream);
int64_t msq_qlinear_w4a8_workspace_bytes(int64_t M, int64_t N, int64_t K, int a_block, int a_variant);
int msq_qlinear_w4a8(const float* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                     const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int w_block,
                     int in_kind, int out_kind, int a_block, int a_inlier_fmt, int a_outlier_fmt,
                     int a_inlier_scale_bits, int a_outlier_scale_bits, float a_std_dev, int a_rmode,
                     int a_flush_fp32_subnorms, int a_variant, int* status_flag, void* workspace,
                     int64_t workspace_bytes, void* stream);
int msq_qlinear_w4a8_x16(const void* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                     const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int w_block,
                     int in_kind, int out_kind, int a_block, int a_inlier_fmt, int a_outlier_fmt,
                     int a_inlier_scale_bits, int a_outlier_scale_bits, float a_std_dev, int a_rmode,
                     int a_flush_fp32_subnorms, int a_variant, int* status_flag, void* workspace,
                     int64_t workspace_bytes, void* stream);   /* bfloat16 X */

/* ---------------------------------------------------------------------------
 * MX-native W4A8 Linear -- NEW (BASELINE config 3, "CDNA4 fp8 MFMA path"): plain OCP-MX operands (block 32 along K,
 * number_system/mx/mx_ops.py:332-457 _quantize_mx; native semantics cpp/mx.cuh, cpp/shared_exp.cuh, scale_bits 8,
 * round to nearest): e4m3 activation codes x e2m1 weight codes with E8M0 block scales, multiplied by
 * v_mfma_scale_f32_16x16x128_f8f6f4 without any dequantisation.
 *   msq_mx_pack_a8: X [M,K] f32 -> codes [M*K] bytes (row-major) + scales [M*K/32] bytes.        K % 128 == 0
 *   msq_mx_pack_w4: W [N,K] f32 -> codes [N*K/2] bytes in MFMA operand order (tile 64 n x 128 k, slot nf = 16 n,
 *                   lane (n % 16, (k % 128) / 32) holds 32 k = 16 B) + scales [N*K/32] bytes ([tile][lane][nf]).
 *                   N % 64 == 0 (N % 256 == 0 for the GEMM), K % 128 == 0.  4.25 bits per weight.
 *   msq_qlinear_mx_w4a8: Y [M,N] = dq(X) . dq(W)^T + bias, y_dtype 0 = f32 / 1 = fp16 / 2 = bf16.  The MFMA sums the 128
 *                   products of one instruction with ~15 bits relative to the largest term: tolerance 1e-4 max|y|.
 * status_flag receives MSQ_STATUS_NAN when a block holds Inf / NaN or its scale overflows.
 * ------------------------------------------------------------------------- */
int msq_mx_pack_a8(const float* X, void* codes, void* scales, int* status_flag, int64_t M, int64_t K,
                   int flush_fp32_subnorms, void* stream);
int msq_mx_pack_a8_bf16(const void* X, void* codes, void* scales, int* status_flag, int64_t M, int64_t K,
                        int flush_fp32_subnorms, void* stream);
/* ... and with float16 activations (an fp16 model; buffers 16-byte aligned) */
int msq_mx_pack_a8_f16(const void* X, void* codes, void* scales, int* status_flag, int64_t M, int64_t K,
                       int flush_fp32_subnorms, void* stream);
/* float16 -> bfloat16, round to nearest even (= Tensor.to(torch.bfloat16)): the activation cast in front of msq_qlinear_bf16 for
 * an fp16 model at prefill sizes, one bandwidth-bound launch (the decode sizes need none: msq_qlinear_f16x) */
int msq_cast_f16_bf16(const void* x, void* y, int64_t n, void* stream);
/* Measurement aid, no reference counterpart: a plain read stream over `bytes` of `buf` (16-byte non-temporal loads, `inflight` = 4 or 8 per
 * lane, `blocks` workgroups of 256, grid-stride), results folded away (`sink`: 4 writable bytes, never written in practice).  bench.py times
 * it on the same cold bytes as the decode kernels (`decode_cold.frac_of_read_stream`: what the weight-streaming kernels reach of the read
 * rate this box gives a stream of that size at that moment). */
int msq_read_stream_probe(const void* buf, int64_t bytes, int blocks, int inflight, void* sink, void* stream);   /* X holds bfloat16: same codes as casting to f32 first */
int msq_mx_pack_w4(const float* W, void* codes, void* scales, int* status_flag, int64_t N, int64_t K,
                   int flush_fp32_subnorms, void* stream);
int64_t msq_qlinear_mx_w4a8_workspace_bytes(int64_t M, int64_t N, int64_t K);   /* > 0 for small M (decode with partial planes, split-K); 0 for the single-launch decode */
int msq_qlinear_mx_w4a8(const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales,
                        const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* MicroScopiQ weights on the same path -- NEW: the fake-quant VALUES of any quantiser of this library (inliers +
 * outliers of utils/quant.py:147-266, the mx_ops variant, GPTQ output) as ONE e4m3 code per weight + one E8M0 scale
 * per 32 k (8.25 bits/weight, scale rule of MSQ-U1) in the fp8 operand order of the scaled MFMA: lane (n % 16, kg)
 * holds k = 16 kg .. +15 and 64 + 16 kg .. +15 of the 128-k tile, stored as two 1 KiB half-slots per (tile, nf).
 * Every code is decoded back and compared with the value: status_flag receives MSQ_STATUS_INEXACT when a value is
 * not representable (the caller then keeps the MSQ-T1 / MSQ-U1 planes and msq_qlinear_bf16).
 *   msq_mx_pack_w8: Wq [N,K] f32 -> codes [N*K] bytes + scales [N*K/32] bytes.  N % 64 == 0, K % 128 == 0.
 *   msq_qlinear_mx_w8a8: as msq_qlinear_mx_w4a8 with that weight operand (same workspace size). */
int msq_mx_pack_w8(const float* Wq, void* codes, void* scales, int* status_flag, int64_t N, int64_t K, void* stream);
int msq_qlinear_mx_w8a8(const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales,
                        const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* MX-FP6 weights on the same path -- NEW: plain OCP-MX fp6_e3m2 / fp6_e2m3 weight codes (number_system/mx/formats.py:76-79,
 * mx_ops.py:332-457 native semantics, block 32 along K) as a true 6-bit plane in the fp6 operand order of the scaled MFMA:
 * lane (n % 16, kg) of slot nf holds k = 32 kg .. +31 of the 128-k tile as 32 six-bit codes (little endian) = 24 bytes, stored
 * as a 16-byte and an 8-byte piece, 1.5 KiB per (tile, nf); one E8M0 scale byte per 32 weights: 6.25 bits per weight.
 *   msq_mx_pack_w6: W [N,K] f32 -> codes [N*K*3/4] bytes + scales [N*K/32] bytes.  N % 64 == 0, K % 128 == 0;
 *                   w_format MSQ_FMT_FP6_E3M2 or MSQ_FMT_FP6_E2M3.
 *   msq_qlinear_mx_w6a8: as msq_qlinear_mx_w4a8 with that weight operand (same workspace size).  Activations whose codes
 *                   hold fp6 values (every fp6 value is an e4m3 value) give the W6A6 product of the fp6 spec exactly. */
int msq_mx_pack_w6(const float* W, void* codes, void* scales, int* status_flag, int64_t N, int64_t K, int w_format,
                   int flush_fp32_subnorms, void* stream);
/* msq_mx_pack_a6: as msq_mx_pack_a8, but X is quantised to MX-FP6 (a_format, fp6 block scale) and the fp6 VALUES are
 * stored as e4m3 codes with that scale (exact) for the fp8 activation operand. */
int msq_mx_pack_a6(const float* X, void* codes, void* scales, int* status_flag, int64_t M, int64_t K, int a_format,
                   int flush_fp32_subnorms, void* stream);
int msq_qlinear_mx_w6a8(const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales,
                        const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int w_format,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * KV-cache group quantisation at the GEAR hook (BASELINE config 4).  Replaces
 * kv_quant/GEARLM/Simulated/compress_function.py:8-38 fake_groupwise_token_asymmetric_quantization (along_tokens = 0:
 * groups of `group_size` consecutive head.dim entries of one token; group_size must divide H * D) and :41-70
 * fake_groupwise_channel_asymmetric_quantization_new (along_tokens = 1: groups of `group_size` consecutive tokens of one
 * channel; group_size must divide S), called from compress_insert_function (:428-517) at the attention hook
 * (modeling_llama_new.py:944-1030).  in / out: [B, H, S, D] contiguous cache tensors (the reference's permute / view /
 * float / type(dtype) round trip happens inside); dtype 0 = f32, 1 = f16, 2 = bf16; asymmetric min / max quantisation to
 * `quantize_bit` bits computed in fp32 with the reference's op order; bit-exact, the NaNs of constant groups included.
 * The MX variants of the cache use msq_quantize_mx_by_tile / msq_outlier_fakequant on the same axes:
 * K [pre = B H, axis = S, post = D], V [pre = B H S, axis = D, post = 1].
 * ------------------------------------------------------------------------- */
int msq_kv_group_quant(const void* in, void* out, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                       int quantize_bit, int64_t group_size, int along_tokens, void* stream);

/* ---------------------------------------------------------------------------
 * Bfloat-rounded vector ops around the MX Linear.  The reference emulates every non-GEMM op as the
 * torch op followed by quantize_elemwise_op (number_system/mx/vector_ops.py); each entry below is that whole chain as
 * ONE launch with the rounding Q() = quantize_elemwise (bits, exp_bits, max_norm, rmode, saturate_normals = false,
 * allow_denorm) applied after every step exactly where the reference applies it (bits = 0: no rounding).
 *   msq_vec_layernorm  replaces mx.LayerNorm's forward (layernorm.py:18-42 -> norm_utils.py:27-113) over the last axis of
 *                      x [rows, H]; row sums in ATen's order for a contiguous inner dimension.  H <= 40704.
 *   msq_vec_gelu       replaces mx.gelu (activations.py:460-512); first_order = the x * sigmoid(1.702 x) variant.
 *   msq_vec_add        replaces mx.simd_add (simd_ops.py:85-106); b == NULL adds the (unrounded) constant b_scalar.
 * All tensors f32, contiguous.  The exp inside gelu is the device's expf (torch uses Sleef's): results can differ from the
 * CPU reference by one unit of the rounded format on isolated elements (none observed on the fixtures).
 * ------------------------------------------------------------------------- */
int msq_vec_round(const float* x, float* out, int64_t n, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm,
                  int force_codec, void* stream);   /* Q() alone; force_codec = 1 bypasses the bfloat fast path (tests) */
int msq_vec_layernorm(const float* x, const float* weight, const float* bias, float* out, int64_t rows, int64_t H,
                      float eps, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm, void* stream);
int msq_vec_gelu(const float* x, float* out, int64_t n, int first_order, int bits, int exp_bits, float max_norm,
                 int rmode, int allow_denorm, void* stream);
int msq_vec_add(const float* a, const float* b, float b_scalar, float* out, int64_t n, int bits, int exp_bits,
                float max_norm, int rmode, int allow_denorm, void* stream);

/* Round 5 -- the activation PRODUCERS in front of the MX Linear, able to hand their result on as the MX-FP8 operand of the scaled-MFMA GEMM
 * (plain MX quantisation of the activation along the last axis, quantize_mx_op mx_ops.py:460-490 with a_elem_format fp8_e4m3 -- the operand of
 * msq_qlinear_mx_w4a8 / _w8a8 / _w6a8; the reference's own MXLinear applies its outlier variant there, linear.py:66-73: msq_act_quant_bf16) in the SAME launch: e4m3 codes [rows * H] row-major + scale bytes [rows * H / 32],
 * the layout and the bytes of msq_mx_pack_a8 on the producer's output (one shared code path, csrc/msq_mx_pack_core.h) -- the float32
 * intermediate never reaches memory unless `out` asks for it.
 *   msq_vec_rmsnorm             replaces mx.RMSNorm's forward (layernorm.py:177 -> RMSNormFunction.forward :98-128): x = Q(x), ms = Q(Q(sum
 *                               Q(x x)) / H), inv = Q(1 / Q(sqrt(Q(ms + eps)))), out = Q(Q(Q(w) Q(x inv)) + Q(b)); bias == NULL = zeros
 *                               (Llama's RMSNorm).  Row sum in ATen's order.  H <= 40704.
 *   msq_vec_rmsnorm_mx_pack_a8  the same + the pack; out may be NULL (packed operand only).  H % 128 == 0, codes 8-byte aligned.
 *   msq_vec_silu                replaces mx.silu (activations.py:76 -> :420-434): q = Q(x), out = Q(q Q(1 / Q(Q(exp(-q)) + 1))).
 *   msq_vec_mul                 replaces mx.simd_mul on two tensors of one shape (simd_ops.py:445 -> :154-187): Q(Q(a) Q(b)).
 *   msq_vec_silu_mul_mx_pack_a8 simd_mul(silu(gate), up) -- the gated-MLP activation -- on rows of I values at row strides ld_gate / ld_up
 *                               (elements; gate and up may be the halves of one [M, 2 I] projection output), to out [M, I] (may be NULL)
 *                               and / or packed (codes / scales may both be NULL; then I only needs ld % 4 == 0, I % 8 == 0).  Pack: I % 128 == 0,
 *                               16-byte aligned buffers.
 * The exp inside silu is the device's expf (see msq_vec_gelu).  status_flag as msq_mx_pack_a8's. */
int msq_vec_rmsnorm(const float* x, const float* weight, const float* bias, float* out, int64_t rows, int64_t H, float eps,
                    int bits, int exp_bits, float max_norm, int rmode, int allow_denorm, void* stream);
int msq_vec_rmsnorm_mx_pack_a8(const float* x, const float* weight, const float* bias, float* out, void* codes, void* scales,
                               int* status_flag, int64_t rows, int64_t H, float eps, int bits, int exp_bits, float max_norm, int rmode,
                               int allow_denorm, int flush_fp32_subnorms, void* stream);
/* ... reading float16 (x_dtype 1) / bfloat16 (2) activations as they are (every 16-bit value is a float32 value: the results of casting first,
 * without the cast pass); weight / bias stay float32; codes / scales may both be NULL (then `out` is required: the plain producer on a 16-bit input). */
int msq_vec_rmsnorm_mx_pack_a8_x16(const void* x, int x_dtype, const float* weight, const float* bias, float* out, void* codes, void* scales,
                                   int* status_flag, int64_t rows, int64_t H, float eps, int bits, int exp_bits, float max_norm, int rmode,
                                   int allow_denorm, int flush_fp32_subnorms, void* stream);
int msq_vec_silu_mul_mx_pack_a8_x16(const void* gate, const void* up, int x_dtype, int64_t ld_gate, int64_t ld_up, float* out, void* codes,
                                    void* scales, int* status_flag, int64_t M, int64_t I, int bits, int exp_bits, float max_norm, int rmode,
                                    int allow_denorm, int flush_fp32_subnorms, void* stream);   /* row strides % 8 == 0 */
int msq_vec_silu(const float* x, float* out, int64_t n, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm, void* stream);
int msq_vec_mul(const float* a, const float* b, float* out, int64_t n, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm,
                void* stream);
int msq_vec_silu_mul_mx_pack_a8(const float* gate, const float* up, int64_t ld_gate, int64_t ld_up, float* out, void* codes, void* scales,
                                int* status_flag, int64_t M, int64_t I, int bits, int exp_bits, float max_norm, int rmode,
                                int allow_denorm, int flush_fp32_subnorms, void* stream);

/* ---------------------------------------------------------------------------
 * GPTQ column block with MicroScopiQ pruning -- replaces the inner column loop of llm/gptq.py:106-165 (per column:
 * quantize_mx_outlier_hessian on [O, 1], zero the num_outliers least important entries, error feedback to the columns on
 * the right) for up to 128 columns as ONE launch.
 *   Wt  [cols][O]   the block's columns as they stand when the block starts (column-major copy of W[:, c0:c1])
 *   U   pointer to U[c0][c0] of the upper Cholesky factor of the inverse Hessian (llm/gptq.py:99-103), row stride ldu
 *   Qt, Et [cols][O] outputs: quantised + pruned columns and error columns (w - q) / d (the caller applies
 *                   W[:, c1:] -= Et^T . U[c0:c1, c1:], llm/gptq.py:163, with a library GEMM)
 *   loss (double, device) += sum (w - q)^2 / d^2 / 2;  pruned (u64, device) += entries zeroed by the pruning step
 * Quantiser = utils/quant.py:23-146 along the output rows (block 8 / 16 / 32 / 64, float / int inlier formats), bit-identical
 * to msq_outlier_fakequant on the same column; num_outliers follows the reference's every-block-th-block count (:66).
 * "The n least important" is resolved exactly (radix select on the importance bits); ties go to the LOWEST ROW INDEX
 * (torch.topk leaves the order among equal values unspecified).  O <= 51200.  status_flag receives MSQ_STATUS_NAN.
 * ------------------------------------------------------------------------- */
int64_t msq_gptq_block_workspace_bytes(int64_t O, int cols);
int msq_gptq_block(const float* Wt, const float* U, int ldu, float* Qt, float* Et, double* loss, unsigned long long* pruned,
                   int* status_flag, void* workspace, int64_t workspace_bytes, int64_t O, int cols, int block,
                   int inlier_fmt, int outlier_fmt, int inlier_scale_bits, int outlier_scale_bits, float std_dev, int rmode,
                   int flush_fp32_subnorms, void* stream);

#ifdef __cplusplus
}
#endif
#endif
