/* libsupir_hip.so -- C ABI of the MI355X (gfx950) kernels behind SUPIR's restoration-guided EDM sampling path.
 *
 * The reference (Fanghua-Yu/SUPIR) has no native code of its own: its compute arrives through torch.nn modules that
 * dispatch to cuDNN / cuBLAS / xformers.  Each entry point below replaces one of those dispatches; the comment on each
 * names the reference call sites (file:line relative to the reference root) it stands in for.  INTEGRATION.md shows
 * the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless stated otherwise;
 *   - caller owns every buffer (including workspaces); the library allocates nothing.  Its only mutable state is the per-thread
 *     last-HIP-error code read by supir_last_hip_error (the process-global kernel-variant switches of the measurement tools,
 *     `supir_debug_knob`, exist only in libsupir_hip_tools.so, built with -DSUPIR_TOOLS; the product libraries do not export the symbol).  Optional per-launch requests (next-weight prefetch, GroupNorm partials
 *     from the producer) are an ARGUMENT: the *_ex entry points take a `const supir_launch_hints*` (may be NULL) and the entry
 *     points without it carry no request (ABI 2 removed the thread-local one-shot setters of ABI 1); results never depend on a
 *     prefetch request;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous and stream ordered, never synchronise;
 *   - return 0 on success, <0 on error (SUPIR_ERR_*); never throws;
 *   - "bf16" buffers are raw 16-bit bfloat16 (IEEE binary16 in the f16 build, see supir_elem_type); activations are NHWC / token-major: element (b, y, x, c) of a
 *     [B,H,W,C] feature map lives at ((b*H + y)*W + x)*ld + c where ld >= C is the row stride in elements;
 *   - leading dimensions of bf16 operands must be multiples of 8 elements (16 bytes) unless noted.
 */
#ifndef SUPIR_HIP_H
#define SUPIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUPIR_OK 0
#define SUPIR_ERR_ARG (-1)   /* null pointer / non-positive size */
#define SUPIR_ERR_SHAPE (-2) /* shape or alignment not supported by the kernel */
#define SUPIR_ERR_HIP (-3)   /* HIP launch error */

/* activation codes */
#define SUPIR_ACT_NONE 0
#define SUPIR_ACT_SILU 1
#define SUPIR_ACT_GEGLU 2 /* x * gelu(gate), W rows interleaved [32 value | 32 gate] per 64; output has N/2 columns.  gelu is a FITTED form
                           * g * sigmoid(g * poly(g^2)), |error| <= 2.5e-5 against erf (below bf16 resolution): the speed default */
#define SUPIR_ACT_GELU 3  /* erf GELU (OpenCLIP text tower MLP) */
#define SUPIR_ACT_QUICKGELU 4 /* x * sigmoid(1.702 x) (OpenAI CLIP text tower MLP) */
#define SUPIR_ACT_GEGLU_ERF 5 /* SUPIR_ACT_GEGLU with the reference's own arithmetic: F.gelu = 0.5 g (1 + erf(g / sqrt 2)),
                               * sgm/modules/attention.py:89-91.  Same layouts, same tiles; the choice is THIS ARGUMENT of each launch
                               * (every GEGLU-capable tile honours it) -- the library has no switch that changes arithmetic.
                               * Host mirror: SUPIR_EXACT_GELU=1 / supir_amd.ops.EXACT_GELU routes every GEGLU launch here. */

/* output modes */
#define SUPIR_OUT_BF16 0
#define SUPIR_OUT_F32 1
#define SUPIR_OUT_BF16_T 2 /* transposed per batch: C[b][n][t], ldc = padded token count */

/* ABI version; bumped on any signature change. */
int supir_abi_version(void);
/* Static string naming the compiled target ("gfx950"). Host pointer. */
const char* supir_target_arch(void);
/* Static string naming the 16-bit element type this build's "bf16" buffers hold: "bf16" for libsupir_hip.so (the product
 * default) or "f16" for libsupir_hip_f16.so -- the same sources compiled with -DSUPIR_F16 (csrc/common.h): every entry point
 * below, same names and signatures, with IEEE binary16 buffers and v_mfma_*_f16 operands, fp32 accumulation and epilogues
 * unchanged.  It serves the reference's default `diff_dtype: fp16` (options/SUPIR_v0.yaml:5, test.py:67-68) at the
 * reference's own precision.  The host mirror picks the library by the dtype of the operands (supir_amd/_lib.py). Host pointer. */
const char* supir_elem_type(void);

/* Diagnostics for SUPIR_ERR_HIP: the hipError_t of the last failed launch on this thread, and its text. Host only. */
int supir_last_hip_error(void);
const char* supir_hip_error_string(int code);

/* Optional per-launch requests of the GEMM-family entry points (supir_gemm_bf16_ex, supir_gemm_bf16_ln_ex, supir_gemm_bf16_qkv_ex,
 * supir_conv3x3_bf16_ex), passed by pointer (HOST memory, read during the call only; NULL = none):
 *   next_weight / next_weight_bytes: this launch, after its last store, touches the first `next_weight_bytes` (whole 128-byte lines) of
 *     `next_weight` -- the bf16 weight matrix of a LATER launch -- so that it is found in the Infinity Cache / L2 instead of HBM.
 *     Read-only, results unaffected.  (The host mirror knows the launch order of a network call: supir_amd/ops.py WeightPrefetch.)
 *   gn_partials_out: tiles 32..35 / 38..40 only, bf16 row-major output, rows_per_batch % BM == 0, N % GU == 0 (else SUPIR_ERR_SHAPE): the
 *     launch also writes, per batch b, tile row c (BM = 128 or 256 tokens) and GU-channel unit u, the (sum, sum of squares) of the bf16
 *     values it stored: gn_partials_out[((b * (rows_per_batch / BM) + c) * (N / GU) + u) * 2 + {0,1}].  GU = 10 on the 80 / 160-column
 *     tiles 32..35 / 38 (the input of supir_groupnorm_nhwc_parts), GU = 4 on the 128 / 256-column tiles 39 / 40 (the VAE's 4 / 8 / 16-channel
 *     groups; reduced by supir_groupnorm_parts_finalize into the given_mean_var input of supir_groupnorm_nhwc).
 * No reference counterpart (the reference's cuBLAS / cuDNN calls meet every weight cold each step and its GroupNorm always runs its
 * own statistics pass, sgm/modules/diffusionmodules/util.py:258-276). */
typedef struct supir_launch_hints {
    const void* next_weight;
    size_t next_weight_bytes;
    float* gn_partials_out;
} supir_launch_hints;

/* C = alpha * act(A . W^T + bias + rowbias[batch]) + residual          A:[M][lda] bf16, W:[N][K] bf16 (K contiguous)
 * Replaces every nn.Linear / 1x1 nn.Conv2d on the path:
 *   sgm/modules/attention.py:87 (GEGLU.proj) :100,106 (FeedForward) :213-219 (to_q/k/v/out) :587,611 (proj_in/out)
 *   sgm/modules/diffusionmodules/openaimodel.py:289 (emb_layers) :317 (skip_connection) :666-695 (time/label embed)
 *   SUPIR/modules/SUPIR_v0.py:48,87 (zero_conv)   sgm/modules/diffusionmodules/model.py:124 (nin_shortcut) :164-175 (q,k,v,proj_out)
 * K % 64 == 0, N % 4 == 0. bias fp32 [N]. rowbias bf16 [nbatch][ld_rowbias] (row m uses batch m / rows_per_batch).
 * residual bf16 [M][ldr]. tile: -1 auto; 0..6 force a tile of the table in csrc/gemm.hip (128x128, 128x64, 64x128, 64x64,
 * 256x128/8 waves, 256x256/8 waves, 256x128/4 waves); bits 3-4 optionally force the LDS ring depth (profiling sweeps);
 * 32..35: the exact-fit tiles of csrc/gemm16.hip (128x80, 128x160, 256x160, 128x80 with 3-deep rings); 37: the 256x320 GEGLU tile of
 * csrc/gemm_big.hip (act = GEGLU, W rows interleaved [16 value | 16 gate], M % 256 == 0, N % 320 == 0); 38: 128x80 with FOUR waves
 * and 78 KB of LDS (two workgroups per CU; round 4); 39 / 40: 256x128 and 256x256 of csrc/gemm16.hip (M % 256 == 0, N % 128 / 256 == 0,
 * ordinary epilogue only: the VAE's 128 / 256 / 512-channel layers); 42: 256x256 on the eight-phase ping-pong schedule (round 5: K >= 128,
 * convolutions additionally OH * OW % 256 == 0; bitwise the results of tile 40; its convolutions walk K chunk-major, see csrc/gemm16.hip);
 * 45: 512x128 on the same schedule (M % 512 == 0, N % 128 == 0, convolutions OH * OW % 512 == 0: the VAE's 128-channel layers at
 * 512^2 / 1024^2 pixels, where a 256-row tile cannot be fed from L2 fast enough).  Forced tiles that do not fit the shape return SUPIR_ERR_SHAPE. */
int supir_gemm_bf16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, const float* bias,
                    const void* rowbias, int ld_rowbias, int rows_per_batch, const void* residual, int ldr, int act,
                    int out_mode, float alpha, int tile, void* stream);
/* the same launch with its optional requests as an argument (hints may be NULL) */
int supir_gemm_bf16_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, const float* bias,
                       const void* rowbias, int ld_rowbias, int rows_per_batch, const void* residual, int ldr, int act,
                       int out_mode, float alpha, int tile, const supir_launch_hints* hints, void* stream);

/* supir_gemm_bf16 with LayerNorm folded in (BasicTransformerBlock, sgm/modules/attention.py:465-486: every nn.LayerNorm there
 * feeds an nn.Linear).  Two halves:
 *   producer  rowstats_out != NULL: besides C (bf16), writes per row m and per column tile s the (sum, sum of squares)
 *             of the bf16 values it stored: rowstats_out[(m*rs_ld + s)*2 + {0,1}], s < ceil(N/BN) of the tile used; rs_ld even;
 *   consumer  ln_stats != NULL: A is the UN-normalised token stream x [M][K]; with W' = gamma (.) W (folded by the caller),
 *             ln_colsum[n] = sum_k W'[n][k] and bias' = bias + W.beta, the result is
 *             rstd[m] * (x[m].W'[n] - mean[m]*ln_colsum[n]) + bias'[n] == LayerNorm(x)[m] . W[n] + bias[n],
 *             mean / rstd taken from the first ln_slots slots of ln_stats (row stride ln_ld), eps = ln_eps; or, with
 *             ln_slots = 0, read directly from ln_stats = [M][2] (mean, rstd) produced by supir_rowstats_finalize.
 * No LayerNorm kernel, no normalised copy of x in HBM. */
int supir_gemm_bf16_ln(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, const float* bias,
                       const void* residual, int ldr, int act, int out_mode, int rows_per_batch, float alpha, int tile,
                       float* rowstats_out, int rs_ld, const float* ln_stats, int ln_ld, int ln_slots,
                       const float* ln_colsum, float ln_eps, void* stream);
int supir_gemm_bf16_ln_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, const float* bias,
                          const void* residual, int ldr, int act, int out_mode, int rows_per_batch, float alpha, int tile,
                          float* rowstats_out, int rs_ld, const float* ln_stats, int ln_ld, int ln_slots,
                          const float* ln_colsum, float ln_eps, const supir_launch_hints* hints, void* stream);

/* Fused q | k | v projection of a self-attention layer (sgm/modules/attention.py:213-219, 241-249: to_q, to_k, to_v on the
 * same LayerNorm'ed tokens) in ONE launch: W = [Wq; Wk; Wv] ([N][K], N = 3 * inner).  Columns [0, n_split) (q | k) are written
 * like supir_gemm_bf16_ln's bf16 output into Cqk [M][ldc]; columns [n_split, N) (v) are written TRANSPOSED per batch into
 * Cvt [M / rows_per_batch][N - n_split][ldc_vt] -- the V^T layout supir_flash_attn_d64 reads.  LayerNorm fold as in
 * supir_gemm_bf16_ln (ln_stats may be NULL).  256 x 160 tile of csrc/gemm16.hip: M % 256 == 0, N % 160 == 0, n_split % 160 == 0,
 * K % 64 == 0, K >= 128, rows_per_batch % 4 == 0, ldc % 8 == 0, ldc_vt % 4 == 0; anything else -> SUPIR_ERR_SHAPE (the caller
 * then issues the two separate projections). */
int supir_gemm_bf16_qkv(const void* A, const void* W, void* Cqk, void* Cvt, int M, int N, int n_split, int K, int lda, int ldc,
                        int ldc_vt, int rows_per_batch, const float* bias, const float* ln_stats, int ln_ld, int ln_slots,
                        const float* ln_colsum, float ln_eps, void* stream);
/* (hints->gn_partials_out must be NULL: the fused launch emits no GroupNorm partials) */
int supir_gemm_bf16_qkv_ex(const void* A, const void* W, void* Cqk, void* Cvt, int M, int N, int n_split, int K, int lda, int ldc,
                           int ldc_vt, int rows_per_batch, const float* bias, const float* ln_stats, int ln_ld, int ln_slots,
                           const float* ln_colsum, float ln_eps, const supir_launch_hints* hints, void* stream);

/* Reduce a producer's row-statistic partials [M][ld][2] to (mean, rstd) [M][2] over `dim` elements per row (fixed order).
 * Consumers then pass ln_slots = 0 and ln_stats = that [M][2] array: two floats per row instead of `slots` partials. */
int supir_rowstats_finalize(const float* partials, float* mean_rstd, int M, int ld, int slots, int dim, float eps,
                            void* stream);

/* Which tile (index into the table of csrc/gemm.hip) tile=-1 selects for an (M, N, act) problem: lets a profiler name the
 * kernel instantiation a launch used. Host-only, no GPU access. */
int supir_gemm_tile_for(int M, int N, int act);

/* 3x3 convolution as implicit GEMM on NHWC bf16.  X:[B][H][W][ldx], W:[Cout][3][3][Cin] bf16, Y:[B][OH][OW][ldy].
 * Replaces nn.Conv2d(k=3) at openaimodel.py:127 (Upsample.conv, upsample=1 folds F.interpolate(nearest,2x) :145)
 *   :196 (Downsample.op, stride 2 pad 1) :263,300 (ResBlock in/out conv, rowbias = emb_layers(emb) :338-355,
 *   residual = skip :356); SUPIR_v0.py:79-83 (ZeroSFT mlp_shared / zero_mul / zero_add);
 *   model.py:60,77 (VAE Upsample / Downsample: stride 2, pad_t = pad_l = 0 gives F.pad(0,1,0,1) :81-86) :108-115.
 * Cin % 64 == 0. Padding is implied by (pad_t, pad_l) and OH/OW: taps outside the (virtual) input read zero. */
int supir_conv3x3_bf16(const void* X, const void* W, void* Y, int B, int H, int Wd, int Cin, int ldx, int Cout,
                       int ldy, int OH, int OW, int stride, int pad_t, int pad_l, int upsample, const float* bias,
                       const void* rowbias, int ld_rowbias, const void* residual, int ldr, int act, int out_mode,
                       float alpha, int tile, void* stream);
int supir_conv3x3_bf16_ex(const void* X, const void* W, void* Y, int B, int H, int Wd, int Cin, int ldx, int Cout,
                          int ldy, int OH, int OW, int stride, int pad_t, int pad_l, int upsample, const float* bias,
                          const void* rowbias, int ld_rowbias, const void* residual, int ldr, int act, int out_mode,
                          float alpha, int tile, const supir_launch_hints* hints, void* stream);

/* softmax(Q K^T * scale) V for head dim 64.  Q:[B][Tq][ldq], K:[B][Tk][ldk] with head h at columns h*64..h*64+63;
 * Vt:[B][H*64][ldvt] is V transposed per batch (SUPIR_OUT_BF16_T output of the to_v projection), ldvt >= roundup(Tk,64),
 * padding finite; O:[B][Tq][ldo] (ldo % 8 == 0 and a 16-byte aligned O take the row-contiguous 16-byte store path; ldo % 4 is accepted).
 * Replaces xformers.ops.memory_efficient_attention / F.scaled_dot_product_attention at
 *   sgm/modules/attention.py:273-277, 357-359 and SUPIR/modules/SUPIR_v0.py:146. */
int supir_flash_attn_d64(const void* Q, const void* K, const void* Vt, void* O, int B, int H, int Tq, int Tk, int ldq,
                         int ldk, int ldvt, int ldo, float scale, void* stream);

/* supir_flash_attn_d64 with options.  flags bit 0: causal mask (key j visible to query i only for j <= i; Tq == Tk) -- the
 * text towers of the conditioner (CLIP-L: transformers CLIPTextModel; OpenCLIP bigG: nn.MultiheadAttention with attn_mask;
 * call sites sgm/modules/encoders/modules.py:487-489, 560-603). */
int supir_flash_attn_d64_ex(const void* Q, const void* K, const void* Vt, void* O, int B, int H, int Tq, int Tk, int ldq,
                            int ldk, int ldvt, int ldo, float scale, int flags, void* stream);

/* The to_q projection AND the attention of a cross-attention layer whose keys are the (few, cached) text tokens, in one launch:
 *     q = X . Wq^T (+ bias)   [with the LayerNorm fold of supir_gemm_bf16_ln when ln_stats is given: X is then the LayerNorm INPUT,
 *                              Wq the gamma-folded weight, bias the beta . Wq^T term, ln_colsum[n] = sum_k Wq[n][k]]
 *     O = softmax(q K^T * scale) V      per head of 64 channels
 * X:[B*T][ldx], Wq:[H*64][C], bias / ln_colsum:[H*64] fp32 (16-byte aligned), K:[B][Tk][ldk] and Vt:[B][H*64][ldvt] as in
 * supir_flash_attn_d64, O:[B*T][ldo].  T % 128 == 0, C % 64 == 0, C >= 192, Tk <= 128, all leading dimensions multiples of 8, all
 * pointers 16-byte aligned; anything else returns SUPIR_ERR_SHAPE and the caller issues the projection and the attention separately.
 * ln_stats / ln_ld / ln_slots as in supir_gemm_bf16_ln (ln_slots <= 64; 0 = finalized (mean, rstd) pairs).
 * q never exists in memory; it is rounded to bf16 once (after the fold, with the softmax scale and log2 e folded in), so results
 * differ from the two-launch path by that one rounding.  hints (may be NULL): next_weight only (parity tests: tests/test_kernels_gpu.py::test_xattn_q_*).
 * Replaces sgm/modules/attention.py:241-249 (to_q) + :273-277 / :357-359 for context = the text embeddings. */
int supir_xattn_q_d64(const void* X, const void* Wq, const float* bias, const void* K, const void* Vt, void* O, int B, int H, int T, int Tk,
                      int C, int ldx, int ldk, int ldvt, int ldo, const float* ln_stats, int ln_ld, int ln_slots, const float* ln_colsum,
                      float ln_eps, float scale, const supir_launch_hints* hints, void* stream);

/* softmax(Q K^T * scale) V for ONE head of dimension 512 without materialising the score matrix: the VAE mid-block attention
 * (sgm/modules/diffusionmodules/model.py:177-192 AttnBlock == :228-256 MemoryEfficientAttnBlock; SUPIR/utils/tilevae.py:276,335).
 * Q:[B][Tq][ldq], K:[B][Tk][ldk] (512 contiguous channels per token), Vt:[B][512][ldvt] = V transposed per batch
 * (SUPIR_OUT_BF16_T output of the v projection), ldvt >= roundup(Tk, 32), padding finite; O:[B][Tq][ldo].  Any Tq / Tk >= 1.
 * ldq, ldk, ldvt multiples of 8, ldo of 4, ldq / ldk / ldo >= 512. */
int supir_flash_attn_d512(const void* Q, const void* K, const void* Vt, void* O, int B, int Tq, int Tk, int ldq, int ldk, int ldvt,
                          int ldo, float scale, void* stream);

/* The same attention with the KEYS split over `splits` sets of workgroups and a fixed-order merge (reproducible; the result differs from
 * supir_flash_attn_d512's by fp32 rounding of the merge only).  A launch of supir_flash_attn_d512 has B * ceil(Tq / 128) workgroups --
 * 128 for the mid block of a 1024^2 image (16 384 tokens), 32 for a 512^2 image or a tiled-VAE tile (SUPIR/utils/tilevae.py:276,335) --
 * on a 256-CU part; split s attends to its own range of 32-key tiles with its own exact row maxima and leaves a normalised fp32
 * partial output plus (maximum, sum) per query row in `workspace`, a second kernel merges them.
 *   splits <= 0: chosen by the library (the count <= 16 with >= 128 keys per split that minimises rounds-of-256-workgroups x keys per
 *                split + the merge; 1 below 512 keys);
 *   splits >= 1: that many (clamped to 16 and to the number of key tiles).
 * supir_flash_attn_d512_workspace(B, Tq, Tk, splits) = bytes the call needs for the same arguments (0 when it resolves to one split: the
 * call is then supir_flash_attn_d512 and `workspace` may be NULL).  workspace: device memory, 16-byte aligned, contents undefined
 * before and after.  A workspace smaller than required, or NULL when one is required, returns SUPIR_ERR_ARG. */
size_t supir_flash_attn_d512_workspace(int B, int Tq, int Tk, int splits);
int supir_flash_attn_d512_split(const void* Q, const void* K, const void* Vt, void* O, int B, int Tq, int Tk, int ldq, int ldk, int ldvt,
                                int ldo, float scale, int splits, void* workspace, size_t workspace_bytes, void* stream);

/* P[r][:] = softmax(S[r][:] * scale): fp32 scores -> bf16 probabilities (VAE mid-block single-head attention,
 * sgm/modules/diffusionmodules/model.py:177-192, 228-256; the score matrix itself comes from supir_gemm_bf16).
 * Columns [T, Tpad) (K padding of the following P.V GEMM) are written as zeros. */
int supir_softmax_rows(const float* S, void* P, int rows, int T, int Tpad, long ld_s, long ld_p, float scale,
                       void* stream);

/* GroupNorm(32 groups) over NHWC bf16 with fp32 statistics, optional SiLU, optional channel concat of two sources
 * (channels [0,C1) from x1, [C1,C) from x2), optional ZeroSFT modulation out = GN(x)*(mod_g+1)+mod_b and
 * control_scale lerp against the raw concat (x1raw / x2raw = the tensors before zero_conv; NULL -> x1 / x2 themselves;
 * with the same leading dimensions ld1 / ld2).
 * workspace: B*1024*64 floats.  given_mean_var: NULL (use this tensor's statistics) or externally pooled ones.
 * Replaces GroupNorm32 (sgm/modules/diffusionmodules/util.py:258-276), Normalize (attention.py:122-125,
 * model.py:48-51), nonlinearity/SiLU (model.py:44-46, openaimodel.py:261,296) and ZeroSFT.forward's tail
 * (SUPIR/modules/SUPIR_v0.py:110-113). */
int supir_groupnorm_nhwc(const void* x1, const void* x2, const void* x1raw, const void* x2raw, int B, int HW, int C, int C1, int ld1,
                         int ld2, const float* gamma, const float* beta, float eps, int act, const void* mod_g,
                         const void* mod_b, int ldm, float control_scale, void* out, int ldo, float* workspace,
                         size_t workspace_bytes, const float* given_mean_var, void* stream);

/* GroupNorm with the statistics supplied by the PRODUCER of its input(s) instead of a statistics pass over the tensor:
 *   supir_launch_hints.gn_partials_out of the producing *_ex launch: the supir_gemm_bf16_ex / supir_gemm_bf16_ln_ex /
 *     supir_conv3x3_bf16_ex launch (tiles 32..35 and 39 / 40 only, bf16 row-major output, rows_per_batch % BM == 0, N % unit == 0
 *     with unit = 10 channels for the 80 / 160-column tiles and 4 for the 128 / 256-column ones; anything else -> SUPIR_ERR_SHAPE)
 *     also writes, per batch b, tile row c (BM = 128 or 256 tokens) and channel unit u, the (sum, sum of squares) of the bf16
 *     values it stored: buf[((b * (rows_per_batch / BM) + c) * (N / unit) + u) * 2 + {0,1}];
 *   supir_groupnorm_nhwc_parts(..., part1, nchunk1, part2, nchunk2, stream) -- supir_groupnorm_nhwc with those buffers for x1
 *     (and x2 when C1 < C; each source with its own chunk count) in place of workspace / given_mean_var: one launch, no
 *     statistics pass.  Needs (C / 32) % 10 == 0 and C1 % 10 == 0 (every GroupNorm32 of the UNet / control net qualifies).
 * Replaces the statistics half of GroupNorm32 (sgm/modules/diffusionmodules/util.py:258-276) wherever the input comes straight
 * out of a convolution or linear layer (openaimodel.py:295-308 out_layers, :260-264 in_layers, attention.py:583-586 norm). */
int supir_groupnorm_nhwc_parts(const void* x1, const void* x2, const void* x1raw, const void* x2raw, int B, int HW, int C, int C1,
                               int ld1, int ld2, const float* gamma, const float* beta, float eps, int act, const void* mod_g,
                               const void* mod_b, int ldm, float control_scale, void* out, int ldo, const float* part1, int nchunk1,
                               const float* part2, int nchunk2, void* stream);

/* Producer partials in `unit`-channel units (part[((b * nchunk + c) * (C / unit) + u) * 2 + {0,1}], (C / 32) % unit == 0) -> per batch
 * and group (mean, biased variance), fp32 [B][32][2] = the given_mean_var input of supir_groupnorm_nhwc.  HW = pixels per batch element.
 * Used where a producer leaves thousands of tile rows behind (VAE feature maps at 1024^2: 4096 rows of 256 pixels): one small launch
 * instead of every workgroup of the normalisation re-reducing all of them.  fp64 accumulation, fixed order.  Replaces the statistics
 * pass of Normalize (sgm/modules/diffusionmodules/model.py:48-51) over the outputs of ResnetBlock / Upsample / Downsample convolutions
 * (model.py:55-148). */
int supir_groupnorm_parts_finalize(const float* part, int B, int nchunk, int C, int unit, int HW, float* mean_var_out, void* stream);

/* Statistics half of GroupNorm alone: sums_out[b][g] = (sum, sum of squares) over group g of batch b, fp32 [B][32][2].
 * With given_mean_var ([B][32][2] = mean, biased variance) supir_groupnorm_nhwc skips its own statistics pass and
 * normalises with the supplied ones.  Together they are the tiled VAE's cross-tile GroupNorm
 * (SUPIR/utils/tilevae.py:511-553 get_var_mean / custom_group_norm, :599-648 GroupNormParam). */
int supir_groupnorm_stats(const void* x1, const void* x2, int B, int HW, int C, int C1, int ld1, int ld2, float* sums_out,
                          float* workspace, size_t workspace_bytes, void* stream);

/* LayerNorm over the last dim of token-major bf16 [rows][ld]; gamma/beta fp32 [C]. (attention.py:437-439,465-486) */
int supir_layernorm(const void* x, void* y, const float* gamma, const float* beta, int rows, int C, int ldx, int ldy,
                    float eps, void* stream);

/* 3x3 s1 p1 conv, fp32 NCHW [B][Cin<=8][H][W] -> bf16 NHWC [B][H][W][ldo] (+ optional bf16 NHWC addend).
 * w: fp32 [Cout][Cin][3][3] (reference layout). openaimodel.py:704, SUPIR_v0.py:325,482,531; model.py:512,646. */
int supir_conv3x3_smallcin(const float* x, const float* w, const float* bias, const void* add, void* out, int B,
                           int Cin, int H, int W, int Cout, int ld_add, int ldo, void* stream);

/* 3x3 s1 p1 conv, bf16 NHWC [B][H][W][ldx] -> fp32 NCHW [B][Cout in {3,4,8}][H][W]. w: bf16 [9][Cout][Cin].
 * openaimodel.py:951 (UNet out), model.py:563 (encoder conv_out), :694 (decoder conv_out). */
int supir_conv3x3_smallcout(const void* x, const void* w, const float* bias, float* out, int B, int Cin, int H, int W,
                            int Cout, int ldx, void* stream);

/* 1x1 conv on fp32 NCHW with <= 8 channels: out = W . (in_scale * x) + bias.
 * quant_conv / post_quant_conv, sgm/models/autoencoder.py:297-298,308,314 (in_scale folds 1/scale_factor). */
int supir_pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                         long HW, float in_scale, void* stream);

/* One level of the colour-fix wavelet decomposition on fp32 planes [planes][H][W] (planes = N*3):
 *   low = depthwise 3x3 blur of img, kernel [[1,2,1],[2,4,2],[1,2,1]]/16, dilation `radius`, replicate padding;
 *   high = (first ? 0 : high) + (img - low).
 * img / low / high must be three distinct buffers; high may be NULL when only the low band is wanted (the style image).  Replaces the F.conv2d(F.pad(..., 'replicate'), groups=3, dilation=radius)
 * of wavelet_blur / wavelet_decomposition, SUPIR/utils/colorfix.py:73-107 (called from SUPIR_model.py:129-131). */
int supir_wavelet_level(const float* img, float* low, float* high, int planes, int H, int W, int radius, int first,
                        void* stream);

/* One pass (vertical != 0: along y, else along x) of Pillow's 8-bit separable resampler -- what PIL.Image.resize(size, BICUBIC)
 * does inside PIL2Tensor (SUPIR/util.py:60-83; arithmetic: third-party Pillow, src/libImaging/Resample.c).  src / dst_u8 are HWC
 * uint8 with `channels` interleaved channels.  bounds [out][2] = (first input index, tap count), coeffs [out][ksize] = 22-bit
 * fixed-point weights (host-built, supir_amd/utils/imageio.py); out = clip8((2^21 + sum pixel * coeff) >> 22): bit-exact.
 * dst_f32 (optional, with lut[256]) receives lut[out] in CHW order -- the fp32 [-1, 1] tensor PIL2Tensor returns. */
int supir_resample_u8(const void* src, void* dst_u8, float* dst_f32, const float* lut, const int* bounds, const int* coeffs, int ksize,
                      int in_h, int in_w, int out_h, int out_w, int channels, int vertical, void* stream);

/* Tensor2PIL (SUPIR/util.py:86-94): F.interpolate(x[C][H][W] fp32, size=(OH, OW), mode='bicubic') (A = -0.75, align_corners
 * False, replicated border) -> out_f32 [C][OH][OW] and / or out_u8 [OH][OW][C] = uint8(clip(v * 127.5 + 127.5, 0, 255)). */
int supir_bicubic_f32(const float* src, void* out_u8, float* out_f32, int C, int H, int W, int OH, int OW, void* stream);

/* The elementwise halves of ONE restoration-guided EDM sampler step on fp32 latents, around the network call
 * (RestoreEDMSampler.sampler_step, sgm/modules/diffusionmodules/sampling.py:548-570, with LinearCFG guiders.py:44-74,
 * NoDynamicThresholding sampling_utils.py:7-9, the EpsScaling denoiser wrapper denoiser.py:66-73 + denoiser_scaling.py:16-22,
 * to_d / euler_step sampling_utils.py:39-40 + sampling.py:82-83).  Every sigma-derived factor is uniform over the batch
 * (s_in * sigmas[i]) and passed as a host scalar, computed in fp32 in the reference's operation order.  n = elements of one
 * copy of the latent batch; reps = 2: classifier-free-guidance doubling, uncond half first (n % 4 == 0); reps = 1: no guider.
 *   pre :  x_hat = x + (eps * s_noise) * noise_mul      noise_mul = sqrt(sigma_hat^2 - sigma^2); eps NULL -> x_hat = x
 *          net_in[r*n + i] = x_hat[i] * c_in, r < reps   (x_hat NULL: not stored)
 *   post:  den_r = net_out[r*n + i] * c_out + x_hat[i] * c_skip;  den = den_0 + cfg_scale * (den_1 - den_0)  (reps 2) | den_0
 *          den -= (den - x_center[i]) * restore_mul       (x_center NULL: skipped)
 *          x_next[i] = x_hat[i] + dt * ((x_hat[i] - den) / sigma_hat)          dt = sigma_next - sigma_hat
 * All pointers 16-byte aligned. */
int supir_edm_step_pre(const float* x, const float* eps, float s_noise, float noise_mul, float c_in, float* x_hat, float* net_in,
                       long n, int reps, void* stream);
int supir_edm_step_post(const float* net_out, const float* x_hat, const float* x_center, float c_out, float c_skip, float cfg_scale,
                        float restore_mul, float sigma_hat, float dt, float* x_next, long n, int reps, void* stream);

/* The tiled sampler's step edges (TiledRestoreEDMSampler.__call__, sgm/modules/diffusionmodules/sampling.py:600-660): per step the
 * latent canvas x [b][C][Hc][Wc] (fp32) is cut into overlapping T x T tiles (`_sliding_windows`, :752-766), each tile takes a
 * sampler_step, and `x_next[:, :, hi:he, wi:we] += _x * tile_weights` (:654-657) blends them.  One network call serves k tiles
 * stacked along the batch axis (tile j, sample i at row j*b + i); tile_hw = HOST array of k (hi, wi) origins, k <= 64.
 *   supir_edm_step_pre_tiles: supir_edm_step_pre reading x / eps (both canvases; eps may be NULL) through the k tile windows:
 *       x_hat [k*b][C][T][T], net_in [reps][k*b][C][T][T].  Same arithmetic per element as the crop + supir_edm_step_pre.
 *   supir_tile_blend: canvas[tile_j] += tiles[j] * weights for j = 0..k-1 in ONE launch, per element in tile order, evaluated as
 *       torch does for `fp32 += fp32 * fp64` (weights [T][T] are float64 there, sampling.py:733-750): (float)((double)acc + (double)t * w)
 *       with separately rounded multiply and add -- bitwise the k sequential slice-adds of the reference. */
int supir_edm_step_pre_tiles(const float* x, const float* eps, float s_noise, float noise_mul, float c_in, float* x_hat, float* net_in,
                             const int* tile_hw, int k, int b, int C, int Hc, int Wc, int T, int reps, void* stream);
int supir_tile_blend(const float* tiles, const double* weights, float* canvas, const int* tile_hw, int k, int b, int C, int Hc, int Wc,
                     int T, void* stream);

/* Touch one dword per 128-byte line of [p, p+bytes) (a weight matrix) so that it is in flight through the memory-side
 * cache before the kernel that consumes it starts; launched a few ops ahead on a separate stream. `sink`: any 4 writable
 * device bytes (never written in practice). No reference counterpart: the reference re-reads fp32 weights through
 * autocast casts every step (sgm/modules/diffusionmodules/wrappers.py:87). */
int supir_prefetch(const void* p, size_t bytes, void* sink, void* stream);

/* Split-K form of supir_conv3x3_bf16 for convolutions whose tile grid is a fraction of the machine and whose K = 9 * Cin is long -- the
 * `mlp_shared` convolutions of ZeroSFT (SUPIR/modules/SUPIR_v0.py:72-75, 100: label_nc -> 128 channels: 64 tiles of 64 x 64 at 32 x 32
 * with 180 K steps each).  The grid becomes ksplit x the tile grid; split s accumulates K steps [s, s + 1) * K / ksplit (with ksplit = 9:
 * one filter tap each) and writes its fp32 partial at partials[s][M][Cout] (M = B * OH * OW); supir_splitk_finalize then sums the partials
 * in split order (deterministic), adds the bias, applies the activation (0 none, 1 SiLU) and stores bf16.  tile 0..3 (csrc/gemm.hip);
 * (9 * Cin) % (64 * ksplit) == 0, Cin % 64 == 0, Cout % 4 == 0. */
int supir_conv3x3_bf16_splitk(const void* X, const void* W, float* partials, int B, int H, int Wd, int Cin, int ldx, int Cout, int OH,
                              int OW, int stride, int pad_t, int pad_l, int upsample, int ksplit, int tile, void* stream);
int supir_splitk_finalize(const float* partials, int ksplit, int M, int N, const float* bias, int act, void* out, int ldo, void* stream);

#ifdef SUPIR_EXPERIMENTAL
/* ---- Grouped launches: n (1 or 2) independent problems of identical shape in ONE kernel launch ------------------------------------
 * EXPERIMENTAL (round 3; declared only under -DSUPIR_EXPERIMENTAL, exported by the library either way): correct and tested (bitwise equal to the single launches, tests/test_grouped_gpu.py), but NOT used by the
 * product's default path -- on the 1024^2 step two free-running chains of single launches measured faster than grouped launches
 * (every grouped launch is a join of the two chains: docs/roundlog.md section 3).  The entry points and struct layouts may change.
 * SUPIR runs two networks of identical architecture on independent data inside every sampling step: GLVControl (the control
 * branch, SUPIR/modules/SUPIR_v0.py:499-540) and the encoder half of LightGLVUNet (:600-625) -- the same ResBlock /
 * SpatialTransformer stack, layer for layer the same shapes, different weights and inputs.  With the CFG-doubled batch of one
 * 1024^2 image each layer is M = 2048 tokens: half a machine of tiles large enough to be fed from L2.  A grouped launch runs the
 * two layers as one grid (problem q on XCDs [4 q, 4 q + 4), its operands in those four L2s); the host mirror records both
 * branches and issues the pairs (supir_amd/ops.py paired_run).  Arguments that fix the grid (`shape`) are shared; pointers,
 * strides and optional operands are per problem.  n = 1 is exactly the corresponding single entry point.  Per problem the
 * explicit `prefetch` / `gn_partials_out` fields play the role of supir_launch_hints.
 * The structs are HOST memory, read during the call only. */
typedef struct supir_gemm_problem {
    const void* A;            /* [M][lda] bf16 (conv: NHWC input [B][H][W][lda]) */
    const void* W;            /* [N][K] bf16 */
    void* C;                  /* bf16 [M][ldc] (out_mode 2: [batch][N][ldc]); fused q|k|v: the q|k part [M][ldc] */
    void* C2;                 /* fused q|k|v only: V^T [batch][N - n_split][ldc2] */
    const float* bias;        /* [N] or NULL */
    const void* rowbias;      /* bf16 [batch][ld_rowbias] or NULL */
    const void* residual;     /* bf16 [M][ldr] or NULL */
    float* rowstats_out;      /* LayerNorm-fold producer output or NULL (see supir_gemm_bf16_ln) */
    const float* ln_stats;    /* LayerNorm-fold consumer input or NULL */
    const float* ln_colsum;
    float* gn_partials_out;   /* GroupNorm unit partials or NULL (see supir_launch_hints.gn_partials_out) */
    const void* prefetch;     /* weight matrix of a later launch to touch on the way out, or NULL (see supir_launch_hints.next_weight) */
    size_t prefetch_bytes;
    int lda, ldc, ldc2, ldr, ld_rowbias, rs_ld, ln_ld, ln_slots;
} supir_gemm_problem;

#define SUPIR_GROUP_GEMM 0    /* supir_gemm_bf16 / supir_gemm_bf16_ln */
#define SUPIR_GROUP_CONV3X3 1 /* supir_conv3x3_bf16 (M = B*OH*OW, N = Cout, K = 9*Cin are derived) */
#define SUPIR_GROUP_QKV 2     /* supir_gemm_bf16_qkv */

typedef struct supir_gemm_shape {
    int kind;                 /* SUPIR_GROUP_* */
    int tile;                 /* 33, 34, 35 (csrc/gemm16.hip) or 37 (csrc/gemm_big.hip, GEGLU); ignored for SUPIR_GROUP_QKV */
    int M, N, K;              /* GEMM / q|k|v */
    int rows_per_batch, act, out_mode, n_split;
    float alpha, ln_eps;
    int B, H, W, Cin, Cout, OH, OW, stride, pad_t, pad_l, upsample;   /* conv3x3 */
} supir_gemm_shape;

/* Only the exact-fit tiles have a grouped form; a shape / tile they do not cover returns SUPIR_ERR_SHAPE and the caller issues the
 * problems one by one. */
int supir_gemm_grouped(const supir_gemm_shape* shape, const supir_gemm_problem* problems, int n, void* stream);

typedef struct supir_attn_problem {
    const void* Q;
    const void* K;
    const void* Vt;
    void* O;
    int Tk, ldq, ldk, ldvt, ldo, flags;   /* flags as in supir_flash_attn_d64_ex */
} supir_attn_problem;
/* supir_flash_attn_d64(_ex) for n problems sharing (B, H, Tq, scale); the key count, strides and flags are per problem. */
int supir_flash_attn_d64_grouped(const supir_attn_problem* problems, int n, int B, int H, int Tq, float scale, void* stream);

typedef struct supir_gn_problem {
    const void* x1;
    const void* x2;
    const void* x1raw;
    const void* x2raw;
    const float* gamma;
    const float* beta;
    const void* mod_g;
    const void* mod_b;
    void* out;
    const float* part1;       /* producer partials (supir_groupnorm_nhwc_parts) or NULL -> own statistics through `workspace` */
    const float* part2;
    float* workspace;         /* >= B*1024*64 floats when part1 == NULL */
    int C1, ld1, ld2, ldm, ldo, nchunk1, nchunk2;
    float control_scale;
} supir_gn_problem;
/* supir_groupnorm_nhwc / supir_groupnorm_nhwc_parts for n problems sharing (B, HW, C, eps, act); every problem must take its
 * statistics the same way (all from producer partials, or all from their own statistics pass). */
int supir_groupnorm_grouped(const supir_gn_problem* problems, int n, int B, int HW, int C, float eps, int act, void* stream);
#endif /* SUPIR_EXPERIMENTAL */

#ifdef __cplusplus
}
#endif
#endif /* SUPIR_HIP_H */
