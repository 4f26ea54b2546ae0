/* C ABI of libsupir_hip_f32.so: the fp32 service of SUPIR's restoration hot path on MI355X (gfx950).
 *
 * What it replaces.  The reference computes a `--diff_dtype fp32` / `--ae_dtype fp32` request in plain fp32
 * (/root/reference/test.py:66-67; sgm/modules/diffusionmodules/wrappers.py:87: torch.autocast disables itself for float32;
 * SUPIR/models/SUPIR_model.py:13,41-69: fp32 is the constructor default of both).  libsupir_hip.so / libsupir_hip_f16.so
 * (include/supir_hip.h) hold 16-bit MFMA kernels only; this third library serves such a request in true fp32 on the exact-fp32
 * matrix instruction of gfx950 (v_mfma_f32_16x16x4_f32: fp32 operands, fp32 accumulation, bitwise an fmaf chain; 157 TFLOP/s
 * dense, 1/16 of the bf16 rate).  It is a CORRECTNESS path: one general tile kernel instead of the tuned families of the 16-bit
 * libraries, activations in fp32 NHWC, weights in the reference's own fp32 values.  The host mirror (supir_amd/ops_f32.py) builds every
 * operator of the path from the launches below; attention is batched GEMM -> row softmax -> batched GEMM with the fp32 score matrix
 * materialised in HBM (1.3 GB for the largest SDXL self-attention at a 1024^2 image; 288 GB per GPU).
 *
 * Conventions: as include/supir_hip.h (device pointers unless said otherwise, `stream` = hipStream_t, every call returns SUPIR_OK or a
 * negative SUPIR_ERR_* code from that header, nothing is allocated or synchronised).  Every buffer holds IEEE binary32.
 */
#ifndef SUPIR_HIP_F32_H
#define SUPIR_HIP_F32_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* The ABI version of include/supir_hip.h this library was built beside, and "f32". Host only. */
int supir_abi_version(void);
const char* supir_target_arch(void);
const char* supir_elem_type(void);
int supir_last_hip_error(void);
const char* supir_hip_error_string(int code);

#define SUPIR_F32_GEMM 0
#define SUPIR_F32_CONV3X3 1

/* One (optionally batched) fp32 GEMM or implicit-GEMM 3x3 convolution with the fused epilogue of the 16-bit kernels:
 *   C = alpha * act(A . W^T + bias + rowbias[batch of the row]) + residual
 * kind = SUPIR_F32_GEMM:    A [M][lda], W [N][ldw] (nn.Linear / 1x1-conv weight as stored), any M, N, K >= 1.
 * kind = SUPIR_F32_CONV3X3: A = NHWC input [B][H][W][lda] (Cin channels used), W [N = Cout][3][3][Cin] (ldw = 9 Cin), M = B * OH * OW,
 *                           K = 9 Cin; stride 1 / 2, top / left padding pad_t / pad_l, taps outside the (virtual) input read zero;
 *                           upsample = 1: nearest-2x of the input folded into the gather (virtual input 2H x 2W).  Any Cin.
 *   (replaces: nn.Linear / 1x1 conv -- sgm/modules/attention.py:84-110,222-285, openaimodel.py:287-293, model.py:124,164-175;
 *    3x3 conv, Upsample, Downsample -- openaimodel.py:108-210,260-321, model.py:55-148, SUPIR/modules/SUPIR_v0.py:62-113)
 * act: SUPIR_ACT_NONE / _SILU / _GELU / _QUICKGELU (GEGLU is supir_f32_geglu on the projection's output).
 * out_mode: SUPIR_OUT_BF16 (0) = row-major C[m][ldc]; SUPIR_OUT_BF16_T (2) = transposed per batch, C[b][n][t] with b = m / rows_per_batch,
 *           t = m % rows_per_batch, ldc = padded token count (padding columns are not written).  (The names are the 16-bit header's.)
 * rowbias [nbatch][ld_rowbias]: row m uses batch m / rows_per_batch (rows_per_batch > 0 whenever rowbias is given).
 * Batching: the launch covers nz0 * nz1 independent problems of the same shape; problem (z0, z1) reads A + z0 a_s0 + z1 a_s1, W + z0 w_s0 +
 *   z1 w_s1 and writes C + z0 c_s0 + z1 c_s1 (element strides; bias / rowbias / residual are not batched and must be NULL when nz0 nz1 > 1).
 *   nz0 = nz1 = 1 and zero strides: a single problem.  (Attention: z0 = head, z1 = batch element.)
 * Host struct, read during the call only. */
typedef struct supir_f32_gemm_desc {
    const float* A; const float* W; float* C;
    const float* bias; const float* rowbias; const float* residual;
    int kind, M, N, K;
    int lda, ldw, ldc, ldr, ld_rowbias, rows_per_batch;
    int act, out_mode;
    float alpha;
    int nz0, nz1;
    long a_s0, a_s1, w_s0, w_s1, c_s0, c_s1;
    int B, H, Wd, Cin, OH, OW, stride, pad_t, pad_l, upsample;
} supir_f32_gemm_desc;
int supir_f32_gemm(const supir_f32_gemm_desc* d, void* stream);

/* out[m][j] = value * gelu_erf(gate) of a GEGLU projection's output proj [M][N2] (sgm/modules/attention.py:84-92), j < N2 / 2.
 * block = 0: the reference layout (value = columns [0, N2/2), gate = [N2/2, N2)); block = 16 / 32: the interleave the 16-bit kernels'
 * weights use (supir_amd/weights.py interleave_geglu: `block` value columns, then `block` gate columns, repeated). */
int supir_f32_geglu(const float* proj, float* out, int M, int N2, int ldp, int ldo, int block, void* stream);

/* P[r][:T] = softmax(S[r][:T] * scale), P[r][T:Tpad] = 0 (model.py:177-192, attention.py:254-285). In place (P == S) allowed.
 * causal_tq > 0: the rows are blocks of causal_tq queries (rows % causal_tq == 0) and query i of a block sees keys j <= i only -- the text
 * towers' causal mask (sgm/modules/encoders/modules.py:445-609: transformers' CLIPTextModel, open_clip's attn_mask); 0: no mask. */
int supir_f32_softmax_rows(const float* S, float* P, long rows, int T, int Tpad, long ld_s, long ld_p, float scale, int causal_tq,
                           void* stream);

/* GroupNorm(32) over NHWC fp32: the argument list and semantics of supir_groupnorm_nhwc (include/supir_hip.h) -- optional SiLU (act = 1),
 * optional channel concat of two sources (C1 channels from x1, C - C1 from x2; a group may straddle the seam), optional ZeroSFT modulation
 * out = GN(x) * (mod_g + 1) + mod_b and control_scale lerp against the raw concat (x1raw / x2raw, NULL -> x1 / x2).  Statistics in fp64.
 * workspace: (B * 32 * 64 * 2 + B * 32 * 2) doubles (contents undefined before and after).
 * given_mean_var: NULL (this tensor's own statistics) or externally pooled (mean, biased variance) per batch element and group, fp32
 * [B][32][2] -- the tiled VAE's pooled GroupNorm (SUPIR/utils/tilevae.py:524-553, 610-640). */
int supir_f32_groupnorm(const float* x1, const float* x2, const float* x1raw, const float* x2raw, int B, int HW, int C, int C1, int ld1,
                        int ld2, const float* gamma, const float* beta, float eps, int act, const float* mod_g, const float* mod_b,
                        int ldm, float control_scale, float* out, int ldo, double* workspace, size_t workspace_bytes,
                        const float* given_mean_var, void* stream);

/* (sum, sum of squares) per batch element and group of an NHWC fp32 tensor, fp64 [B][32][2]: the per-tile statistics the tiled VAE pools
 * (supir_groupnorm_stats of the 16-bit header, in fp64).  workspace as supir_f32_groupnorm. */
int supir_f32_groupnorm_stats(const float* x, int B, int HW, int C, int ld, double* sums, double* workspace, size_t workspace_bytes,
                              void* stream);

/* LayerNorm over the last dimension (attention.py:376-486 norm1..3; two-pass mean / variance in fp32). y may alias x. */
int supir_f32_layernorm(const float* x, float* y, const float* gamma, const float* beta, int rows, int C, int ldx, int ldy, float eps,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif
