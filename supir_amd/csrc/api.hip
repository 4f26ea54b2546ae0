// extern "C" surface of libsupir_hip.so (declared in include/supir_hip.h): argument validation + packing only.
#include "kernels.h"
#include <stdlib.h>
#include "../../include/supir_hip.h"

static thread_local int g_last_hip_error = 0;
int supir_note_hip_status(hipError_t e) {
    if (e == hipSuccess) return SUPIR_OK;
    g_last_hip_error = (int)e;
    return SUPIR_ERR_HIP;
}

// Per-launch requests (next-weight prefetch, GroupNorm partials) arrive as an ARGUMENT of the *_ex entry points
// (supir_launch_hints, may be NULL); the entry points without that argument carry none.  The library holds no request state.
static int apply_hints(GemmArgs& a, const supir_launch_hints* h) {
    if (!h) return SUPIR_OK;
    if (h->next_weight_bytes && !h->next_weight) return SUPIR_ERR_ARG;
    const size_t lines = h->next_weight_bytes / 128;
    a.pf_ptr = (const char*)h->next_weight;
    a.pf_lines = lines > 0x7fffffffu ? 0x7fffffffu : (unsigned)lines;
    a.gn_part_out = h->gn_partials_out;
    return SUPIR_OK;
}

// The activation code of the ABI -> the kernels' (act, fast_gelu): SUPIR_ACT_GEGLU is the fitted GELU (|error| <= 2.5e-5), SUPIR_ACT_GEGLU_ERF
// the reference's erf (sgm/modules/attention.py:89-91).  An ARGUMENT of every launch: no process state selects arithmetic.
static void set_act(GemmArgs& a, int act) {
    a.act = act == SUPIR_ACT_GEGLU_ERF ? SUPIR_ACT_GEGLU : act;
    a.fast_gelu = act == SUPIR_ACT_GEGLU ? 1 : 0;
}

#ifdef SUPIR_TOOLS
// libsupir_hip_tools.so only (-DSUPIR_TOOLS; A/B measurements and the tests that hold kernel variants against each other): process-global
// switches that select kernel VARIANTS.  The product libraries do not contain this state or the symbol: supir_debug_knob_value is the
// constant 0 there (kernels.h).
static int g_debug_knobs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
int supir_debug_knob_value(int which) { return (which >= 0 && which < 8) ? g_debug_knobs[which] : 0; }
#endif

extern "C" {

#ifdef SUPIR_TOOLS
int supir_debug_knob(int which, int value) {
    if (which < 0 || which >= 8) return SUPIR_ERR_ARG;
    g_debug_knobs[which] = value;
    return SUPIR_OK;
}
#endif

int supir_last_hip_error(void) { return g_last_hip_error; }
const char* supir_hip_error_string(int code) { return hipGetErrorString((hipError_t)code); }

int supir_abi_version(void) { return 2; }
const char* supir_target_arch(void) { return "gfx950"; }
const char* supir_elem_type(void) { return SUPIR_ELEM_NAME; }
int supir_gemm_tile_for(int M, int N, int act) { return supir_gemm_select_tile(M, N, act == SUPIR_ACT_GEGLU_ERF ? SUPIR_ACT_GEGLU : act, -1); }

int supir_gemm_bf16_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, const float* bias,
                       const void* rowbias, int ld_rowbias, int rows_per_batch, const void* residual, int ldr, int act,
                       int out_mode, float alpha, int tile, const supir_launch_hints* hints, void* stream) {
    if (!A || !W || !C) return SUPIR_ERR_ARG;
    if (act < 0 || act > SUPIR_ACT_GEGLU_ERF || out_mode < 0 || out_mode > 2 || tile > 45 || tile == 36 || tile == 41 || tile == 43 || tile == 44) return SUPIR_ERR_ARG;
    if ((rowbias || out_mode == 2) && rows_per_batch <= 0) return SUPIR_ERR_ARG;
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.Wt = (const bf16_t*)W; a.C = C;
    a.bias = bias; a.rowbias = (const bf16_t*)rowbias; a.res = (const bf16_t*)residual;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr; a.ld_rb = ld_rowbias;
    a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    set_act(a, act); a.out_mode = out_mode; a.alpha = alpha;
    if (out_mode != 2 && (ldc % 4 != 0 || (residual && ldr % 4 != 0) || (rowbias && ld_rowbias % 4 != 0)))
        return SUPIR_ERR_SHAPE;
    if (const int rc = apply_hints(a, hints)) return rc;
    return supir_gemm_launch(a, false, (hipStream_t)stream, tile);
}

int supir_gemm_bf16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, const float* bias,
                    const void* rowbias, int ld_rowbias, int rows_per_batch, const void* residual, int ldr, int act,
                    int out_mode, float alpha, int tile, void* stream) {
    return supir_gemm_bf16_ex(A, W, C, M, N, K, lda, ldc, bias, rowbias, ld_rowbias, rows_per_batch, residual, ldr, act, out_mode, alpha,
                              tile, nullptr, stream);
}

int supir_gemm_bf16_ln_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, const float* bias,
                          const void* residual, int ldr, int act, int out_mode, int rows_per_batch, float alpha, int tile,
                          float* rowstats_out, int rs_ld, const float* ln_stats, int ln_ld, int ln_slots,
                          const float* ln_colsum, float ln_eps, const supir_launch_hints* hints, void* stream) {
    if (!A || !W || !C) return SUPIR_ERR_ARG;
    if (act < 0 || act > SUPIR_ACT_GEGLU_ERF || out_mode < 0 || out_mode > 2 || tile > 45 || tile == 36 || tile == 41 || tile == 43 || tile == 44) return SUPIR_ERR_ARG;
    if (out_mode == 2 && rows_per_batch <= 0) return SUPIR_ERR_ARG;
    if (ln_stats && (!ln_colsum || ln_slots < 0 || (ln_slots > 0 && (ln_ld < ln_slots || (ln_ld & 1))))) return SUPIR_ERR_ARG;
    if (rowstats_out && (out_mode != 0 || act == SUPIR_ACT_GEGLU || act == SUPIR_ACT_GEGLU_ERF || rs_ld <= 0)) return SUPIR_ERR_ARG;
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.Wt = (const bf16_t*)W; a.C = C;
    a.bias = bias; a.res = (const bf16_t*)residual;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr;
    a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    set_act(a, act); a.out_mode = out_mode; a.alpha = alpha;
    a.rowstats_out = rowstats_out; a.rs_ld = rs_ld;
    a.ln_stats = ln_stats; a.ln_ld = ln_ld; a.ln_slots = ln_slots; a.ln_colsum = ln_colsum; a.ln_eps = ln_eps;
    if (out_mode != 2 && (ldc % 4 != 0 || (residual && ldr % 4 != 0))) return SUPIR_ERR_SHAPE;
    if (rowstats_out) {  // the slot index is tile_n * waves_n + wave_n: the caller's rs_ld must cover the tile actually used
        const int sel = supir_gemm_select_tile(M, N, a.act, tile < 0 ? -1 : (tile & 7));
        const int bn = (tile == 32 || tile == 35 || tile == 38) ? 80 : (tile == 33 || tile == 34) ? 160 : (tile == 39 || tile == 45) ? 128 : (tile == 40 || tile == 42) ? 256 : (sel == 1 || sel == 3) ? 64 : (sel == 5 ? 256 : 128);
        if ((N + bn - 1) / bn > rs_ld || (rs_ld & 1)) return SUPIR_ERR_ARG;
    }
    if (const int rc = apply_hints(a, hints)) return rc;
    return supir_gemm_launch(a, false, (hipStream_t)stream, tile);
}

int supir_gemm_bf16_ln(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, const float* bias,
                       const void* residual, int ldr, int act, int out_mode, int rows_per_batch, float alpha, int tile,
                       float* rowstats_out, int rs_ld, const float* ln_stats, int ln_ld, int ln_slots,
                       const float* ln_colsum, float ln_eps, void* stream) {
    return supir_gemm_bf16_ln_ex(A, W, C, M, N, K, lda, ldc, bias, residual, ldr, act, out_mode, rows_per_batch, alpha, tile, rowstats_out,
                                 rs_ld, ln_stats, ln_ld, ln_slots, ln_colsum, ln_eps, nullptr, stream);
}

int supir_gemm_bf16_qkv_ex(const void* A, const void* W, void* Cqk, void* Cvt, int M, int N, int n_split, int K, int lda, int ldc,
                           int ldc_vt, int rows_per_batch, const float* bias, const float* ln_stats, int ln_ld, int ln_slots,
                           const float* ln_colsum, float ln_eps, const supir_launch_hints* hints, void* stream) {
    if (!A || !W || !Cqk || !Cvt || rows_per_batch <= 0) return SUPIR_ERR_ARG;
    if (ln_stats && (!ln_colsum || ln_slots < 0 || (ln_slots > 0 && (ln_ld < ln_slots || (ln_ld & 1))))) return SUPIR_ERR_ARG;
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.Wt = (const bf16_t*)W; a.C = Cqk; a.C2 = Cvt;
    a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldc2 = ldc_vt; a.n_split = n_split;
    a.rows_per_batch = rows_per_batch;
    a.alpha = 1.0f;
    a.ln_stats = ln_stats; a.ln_ld = ln_ld; a.ln_slots = ln_slots; a.ln_colsum = ln_colsum; a.ln_eps = ln_eps;
    if (const int rc = apply_hints(a, hints)) return rc;
    if (a.gn_part_out) return SUPIR_ERR_ARG;   // the fused q|k|v launch emits no GroupNorm partials
    return supir_gemm16_qkv_launch(a, (hipStream_t)stream);
}

int supir_gemm_bf16_qkv(const void* A, const void* W, void* Cqk, void* Cvt, int M, int N, int n_split, int K, int lda, int ldc,
                        int ldc_vt, int rows_per_batch, const float* bias, const float* ln_stats, int ln_ld, int ln_slots,
                        const float* ln_colsum, float ln_eps, void* stream) {
    return supir_gemm_bf16_qkv_ex(A, W, Cqk, Cvt, M, N, n_split, K, lda, ldc, ldc_vt, rows_per_batch, bias, ln_stats, ln_ld, ln_slots,
                                  ln_colsum, ln_eps, nullptr, stream);
}

int supir_rowstats_finalize(const float* partials, float* mean_rstd, int M, int ld, int slots, int dim, float eps,
                            void* stream) {
    if (!partials || !mean_rstd) return SUPIR_ERR_ARG;
    return supir_rowstats_finalize_launch(partials, mean_rstd, M, ld, slots, dim, eps, (hipStream_t)stream);
}

int supir_conv3x3_bf16_ex(const void* X, const void* W, void* Y, int B, int H, int Wd, int Cin, int ldx, int Cout,
                          int ldy, int OH, int OW, int stride, int pad_t, int pad_l, int upsample, const float* bias,
                          const void* rowbias, int ld_rowbias, const void* residual, int ldr, int act, int out_mode,
                          float alpha, int tile, const supir_launch_hints* hints, void* stream) {
    if (!X || !W || !Y) return SUPIR_ERR_ARG;
    if (B <= 0 || H <= 0 || Wd <= 0 || OH <= 0 || OW <= 0) return SUPIR_ERR_ARG;
    if (act < 0 || act > 1 || out_mode < 0 || out_mode > 1 || (tile > 35 && (tile < 38 || tile > 40) && tile != 42 && tile != 45 && (tile < 48 || tile > 51))) return SUPIR_ERR_ARG;
    if (stride != 1 && stride != 2) return SUPIR_ERR_SHAPE;
    if (upsample && stride != 1) return SUPIR_ERR_SHAPE;
    if (ldy % 4 != 0 || (residual && ldr % 4 != 0) || (rowbias && ld_rowbias % 4 != 0)) return SUPIR_ERR_SHAPE;
    GemmArgs a{};
    a.A = (const bf16_t*)X; a.Wt = (const bf16_t*)W; a.C = Y;
    a.bias = bias; a.rowbias = (const bf16_t*)rowbias; a.res = (const bf16_t*)residual;
    a.M = B * OH * OW; a.N = Cout; a.K = 9 * Cin; a.lda = ldx; a.ldc = ldy; a.ldr = ldr; a.ld_rb = ld_rowbias;
    a.rows_per_batch = OH * OW;
    a.H = H; a.W = Wd; a.Cin = Cin; a.OH = OH; a.OW = OW; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
    a.up = upsample ? 1 : 0;
    a.act = act; a.out_mode = out_mode; a.alpha = alpha;
    if (const int rc = apply_hints(a, hints)) return rc;
    return supir_gemm_launch(a, true, (hipStream_t)stream, tile);
}

int supir_conv3x3_bf16(const void* X, const void* W, void* Y, int B, int H, int Wd, int Cin, int ldx, int Cout,
                       int ldy, int OH, int OW, int stride, int pad_t, int pad_l, int upsample, const float* bias,
                       const void* rowbias, int ld_rowbias, const void* residual, int ldr, int act, int out_mode,
                       float alpha, int tile, void* stream) {
    return supir_conv3x3_bf16_ex(X, W, Y, B, H, Wd, Cin, ldx, Cout, ldy, OH, OW, stride, pad_t, pad_l, upsample, bias, rowbias, ld_rowbias,
                                 residual, ldr, act, out_mode, alpha, tile, nullptr, stream);
}

int supir_conv3x3_bf16_splitk(const void* X, const void* W, float* partials, int B, int H, int Wd, int Cin, int ldx, int Cout, int OH,
                              int OW, int stride, int pad_t, int pad_l, int upsample, int ksplit, int tile, void* stream) {
    if (!X || !W || !partials) return SUPIR_ERR_ARG;
    if (B <= 0 || H <= 0 || Wd <= 0 || OH <= 0 || OW <= 0 || ksplit < 2 || tile < 0 || tile > 3) return SUPIR_ERR_ARG;
    if ((stride != 1 && stride != 2) || (upsample && stride != 1) || Cout % 4) return SUPIR_ERR_SHAPE;
    GemmArgs a{};
    a.A = (const bf16_t*)X; a.Wt = (const bf16_t*)W; a.C = partials;
    a.M = B * OH * OW; a.N = Cout; a.K = 9 * Cin; a.lda = ldx; a.ldc = Cout;
    a.rows_per_batch = OH * OW;
    a.H = H; a.W = Wd; a.Cin = Cin; a.OH = OH; a.OW = OW; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
    a.up = upsample ? 1 : 0;
    a.out_mode = 1; a.alpha = 1.0f; a.ksplit = ksplit;
    return supir_gemm_launch(a, true, (hipStream_t)stream, tile);
}

int supir_splitk_finalize(const float* partials, int ksplit, int M, int N, const float* bias, int act, void* out, int ldo, void* stream) {
    if (!partials || !out) return SUPIR_ERR_ARG;
    return supir_splitk_finalize_launch(partials, ksplit, M, N, N, bias, act, (bf16_t*)out, ldo, (hipStream_t)stream);
}

int supir_flash_attn_d64_ex(const void* Q, const void* K, const void* Vt, void* O, int B, int H, int Tq, int Tk, int ldq,
                            int ldk, int ldvt, int ldo, float scale, int flags, void* stream) {
    if (!Q || !K || !Vt || !O || (flags & ~1)) return SUPIR_ERR_ARG;
    if ((flags & 1) && Tq != Tk) return SUPIR_ERR_SHAPE;
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.Vt = (const bf16_t*)Vt; a.O = (bf16_t*)O;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.causal = flags & 1;
    return supir_attn_launch(a, (hipStream_t)stream);
}

int supir_flash_attn_d64(const void* Q, const void* K, const void* Vt, void* O, int B, int H, int Tq, int Tk, int ldq,
                         int ldk, int ldvt, int ldo, float scale, void* stream) {
    if (!Q || !K || !Vt || !O) return SUPIR_ERR_ARG;
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.Vt = (const bf16_t*)Vt; a.O = (bf16_t*)O;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.scale_log2e = scale * 1.4426950408889634f;
    return supir_attn_launch(a, (hipStream_t)stream);
}

int supir_xattn_q_d64(const void* X, const void* Wq, const float* bias, const void* K, const void* Vt, void* O, int B, int H, int T, int Tk,
                      int C, int ldx, int ldk, int ldvt, int ldo, const float* ln_stats, int ln_ld, int ln_slots, const float* ln_colsum,
                      float ln_eps, float scale, const supir_launch_hints* hints, void* stream) {
    if (!X || !Wq || !K || !Vt || !O) return SUPIR_ERR_ARG;
    if (ln_stats && !ln_colsum) return SUPIR_ERR_ARG;
    if (hints && (hints->gn_partials_out || (hints->next_weight_bytes && !hints->next_weight))) return SUPIR_ERR_ARG;
    XattnArgs a{};
    a.X = (const bf16_t*)X; a.Wq = (const bf16_t*)Wq; a.bias = bias; a.K = (const bf16_t*)K; a.Vt = (const bf16_t*)Vt; a.O = (bf16_t*)O;
    a.B = B; a.H = H; a.T = T; a.Tk = Tk; a.C = C; a.ldx = ldx; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.ln_stats = ln_stats; a.ln_ld = ln_ld; a.ln_slots = ln_slots; a.ln_colsum = ln_colsum; a.ln_eps = ln_eps;
    a.scale_log2e = scale * 1.4426950408889634f;
    if (hints && hints->next_weight) {
        const size_t lines = hints->next_weight_bytes / 128;
        a.pf_ptr = (const char*)hints->next_weight;
        a.pf_lines = lines > 0x7fffffffu ? 0x7fffffffu : (unsigned)lines;
    }
    return supir_xattn_q_launch(a, (hipStream_t)stream);
}

int supir_flash_attn_d512(const void* Q, const void* K, const void* Vt, void* O, int B, int Tq, int Tk, int ldq, int ldk, int ldvt,
                          int ldo, float scale, void* stream) {
    if (!Q || !K || !Vt || !O) return SUPIR_ERR_ARG;
    return supir_attn_d512_launch((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, B, Tq, Tk, ldq, ldk, ldvt, ldo,
                                  scale, 1, nullptr, 0, (hipStream_t)stream);
}

size_t supir_flash_attn_d512_workspace(int B, int Tq, int Tk, int splits) { return supir_attn_d512_workspace_bytes(B, Tq, Tk, splits); }

int supir_flash_attn_d512_split(const void* Q, const void* K, const void* Vt, void* O, int B, int Tq, int Tk, int ldq, int ldk, int ldvt,
                                int ldo, float scale, int splits, void* workspace, size_t workspace_bytes, void* stream) {
    if (!Q || !K || !Vt || !O) return SUPIR_ERR_ARG;
    if (!workspace && supir_attn_d512_workspace_bytes(B, Tq, Tk, splits) != 0) return SUPIR_ERR_ARG;
    return supir_attn_d512_launch((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, B, Tq, Tk, ldq, ldk, ldvt, ldo,
                                  scale, splits, workspace, workspace_bytes, (hipStream_t)stream);
}

int supir_softmax_rows(const float* S, void* P, int rows, int T, int Tpad, long ld_s, long ld_p, float scale,
                       void* stream) {
    if (!S || !P) return SUPIR_ERR_ARG;
    return supir_softmax_rows_launch(S, (bf16_t*)P, rows, T, Tpad, ld_s, ld_p, scale, (hipStream_t)stream);
}

int supir_groupnorm_stats(const void* x1, const void* x2, int B, int HW, int C, int C1, int ld1, int ld2, float* sums_out,
                          float* workspace, size_t workspace_bytes, void* stream) {
    if (!x1 || !sums_out || !workspace) return SUPIR_ERR_ARG;
    if (C1 <= 0 || C1 > C) return SUPIR_ERR_ARG;
    if (workspace_bytes < (size_t)B * 1024 * 64 * sizeof(float)) return SUPIR_ERR_ARG;
    GnArgs a{};
    a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.partial = workspace;
    a.B = B; a.HW = HW; a.C = C; a.C1 = C1; a.ld1 = ld1; a.ld2 = ld2;
    return supir_groupnorm_stats_launch(a, sums_out, (hipStream_t)stream);
}

int supir_groupnorm_parts_finalize(const float* part, int B, int nchunk, int C, int unit, int HW, float* mean_var_out, void* stream) {
    if (!part || !mean_var_out) return SUPIR_ERR_ARG;
    return supir_groupnorm_parts_finalize_launch(part, B, nchunk, C, unit, HW, mean_var_out, (hipStream_t)stream);
}

int supir_groupnorm_nhwc(const void* x1, const void* x2, const void* x1raw, const void* x2raw, int B, int HW, int C, int C1, int ld1,
                         int ld2, const float* gamma, const float* beta, float eps, int act, const void* mod_g,
                         const void* mod_b, int ldm, float control_scale, void* out, int ldo, float* workspace,
                         size_t workspace_bytes, const float* given_mean_var, void* stream) {
    if (!x1 || !gamma || !beta || !out || !workspace) return SUPIR_ERR_ARG;
    if (C1 <= 0 || C1 > C) return SUPIR_ERR_ARG;
    if (workspace_bytes < (size_t)B * 1024 * 64 * sizeof(float)) return SUPIR_ERR_ARG;
    GnArgs a{};
    a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.x1raw = (const bf16_t*)x1raw; a.x2raw = (const bf16_t*)x2raw;
    a.partial = workspace; a.gamma = gamma; a.beta = beta; a.given = given_mean_var;
    a.mod_g = (const bf16_t*)mod_g; a.mod_b = (const bf16_t*)mod_b; a.out = (bf16_t*)out;
    a.B = B; a.HW = HW; a.C = C; a.C1 = C1; a.ld1 = ld1; a.ld2 = ld2; a.ldm = ldm; a.ldo = ldo;
    a.act = act; a.eps = eps; a.cscale = control_scale;
    return supir_groupnorm_launch(a, (hipStream_t)stream);
}

int supir_groupnorm_nhwc_parts(const void* x1, const void* x2, const void* x1raw, const void* x2raw, int B, int HW, int C, int C1,
                               int ld1, int ld2, const float* gamma, const float* beta, float eps, int act, const void* mod_g,
                               const void* mod_b, int ldm, float control_scale, void* out, int ldo, const float* part1, int nchunk1,
                               const float* part2, int nchunk2, void* stream) {
    if (!x1 || !gamma || !beta || !out || !part1) return SUPIR_ERR_ARG;
    if (C1 <= 0 || C1 > C) return SUPIR_ERR_ARG;
    GnArgs a{};
    a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.x1raw = (const bf16_t*)x1raw; a.x2raw = (const bf16_t*)x2raw;
    a.gamma = gamma; a.beta = beta;
    a.part_u1 = part1; a.nch1 = nchunk1; a.part_u2 = part2; a.nch2 = nchunk2;
    a.mod_g = (const bf16_t*)mod_g; a.mod_b = (const bf16_t*)mod_b; a.out = (bf16_t*)out;
    a.B = B; a.HW = HW; a.C = C; a.C1 = C1; a.ld1 = ld1; a.ld2 = ld2; a.ldm = ldm; a.ldo = ldo;
    a.act = act; a.eps = eps; a.cscale = control_scale;
    return supir_groupnorm_launch(a, (hipStream_t)stream);
}

int supir_layernorm(const void* x, void* y, const float* gamma, const float* beta, int rows, int C, int ldx, int ldy,
                    float eps, void* stream) {
    if (!x || !y || !gamma || !beta) return SUPIR_ERR_ARG;
    return supir_layernorm_launch((const bf16_t*)x, (bf16_t*)y, gamma, beta, rows, C, ldx, ldy, eps, (hipStream_t)stream);
}

int supir_conv3x3_smallcin(const float* x, const float* w, const float* bias, const void* add, void* out, int B,
                           int Cin, int H, int W, int Cout, int ld_add, int ldo, void* stream) {
    if (!x || !w || !out) return SUPIR_ERR_ARG;
    return supir_conv3x3_smallcin_launch(x, w, bias, (const bf16_t*)add, (bf16_t*)out, B, Cin, H, W, Cout, ld_add, ldo,
                                         (hipStream_t)stream);
}

int supir_conv3x3_smallcout(const void* x, const void* w, const float* bias, float* out, int B, int Cin, int H, int W,
                            int Cout, int ldx, void* stream) {
    if (!x || !w || !out) return SUPIR_ERR_ARG;
    return supir_conv3x3_smallcout_launch((const bf16_t*)x, (const bf16_t*)w, bias, out, B, Cin, H, W, Cout, ldx,
                                          (hipStream_t)stream);
}

int supir_prefetch(const void* p, size_t bytes, void* sink, void* stream) {
    if (!p) return SUPIR_ERR_ARG;
    return supir_prefetch_launch(p, bytes, sink, (hipStream_t)stream);
}

int supir_pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                         long HW, float in_scale, void* stream) {
    if (!x || !w || !out) return SUPIR_ERR_ARG;
    return supir_pointwise_nchw_launch(x, w, bias, out, B, Cin, Cout, HW, in_scale, (hipStream_t)stream);
}

int supir_wavelet_level(const float* img, float* low, float* high, int planes, int H, int W, int radius, int first,
                        void* stream) {
    if (!img || !low) return SUPIR_ERR_ARG;   // high may be NULL: low band only
    return supir_wavelet_level_launch(img, low, high, planes, H, W, radius, first, (hipStream_t)stream);
}

int supir_resample_u8(const void* src, void* dst_u8, float* dst_f32, const float* lut, const int* bounds, const int* coeffs, int ksize,
                      int in_h, int in_w, int out_h, int out_w, int channels, int vertical, void* stream) {
    if (!src || !bounds || !coeffs) return SUPIR_ERR_ARG;
    return supir_resample_u8_launch((const uint8_t*)src, (uint8_t*)dst_u8, dst_f32, lut, bounds, coeffs, ksize, in_h, in_w, out_h, out_w,
                                    channels, vertical, (hipStream_t)stream);
}

int supir_bicubic_f32(const float* src, void* out_u8, float* out_f32, int C, int H, int W, int OH, int OW, void* stream) {
    if (!src) return SUPIR_ERR_ARG;
    return supir_bicubic_f32_launch(src, (uint8_t*)out_u8, out_f32, C, H, W, OH, OW, (hipStream_t)stream);
}

int supir_edm_step_pre(const float* x, const float* eps, float s_noise, float noise_mul, float c_in, float* x_hat, float* net_in,
                       long n, int reps, void* stream) {
    if (!x || !net_in) return SUPIR_ERR_ARG;   // eps NULL: no churn on this step; x_hat NULL: the caller keeps using x
    return supir_edm_pre_launch(x, eps, s_noise, noise_mul, c_in, x_hat, net_in, n, reps, (hipStream_t)stream);
}

int supir_edm_step_post(const float* net_out, const float* x_hat, const float* x_center, float c_out, float c_skip, float cfg_scale,
                        float restore_mul, float sigma_hat, float dt, float* x_next, long n, int reps, void* stream) {
    if (!net_out || !x_hat || !x_next) return SUPIR_ERR_ARG;   // x_center NULL: no restoration guidance on this step
    return supir_edm_post_launch(net_out, x_hat, x_center, c_out, c_skip, cfg_scale, restore_mul, sigma_hat, dt, x_next, n, reps,
                                 (hipStream_t)stream);
}

static int pack_tiles(SupirTileList& tl, const int* tile_hw, int k) {
    if (!tile_hw || k <= 0 || k > SUPIR_MAX_TILES) return SUPIR_ERR_ARG;
    tl.n = k;
    for (int j = 0; j < k; ++j) {
        tl.hi[j] = tile_hw[2 * j];
        tl.wi[j] = tile_hw[2 * j + 1];
    }
    return SUPIR_OK;
}

int supir_edm_step_pre_tiles(const float* x, const float* eps, float s_noise, float noise_mul, float c_in, float* x_hat, float* net_in,
                             const int* tile_hw, int k, int b, int C, int Hc, int Wc, int T, int reps, void* stream) {
    if (!x || !x_hat || !net_in) return SUPIR_ERR_ARG;
    SupirTileList tl;
    if (const int rc = pack_tiles(tl, tile_hw, k)) return rc;
    return supir_edm_pre_tiles_launch(x, eps, s_noise, noise_mul, c_in, x_hat, net_in, tl, b, C, Hc, Wc, T, reps, (hipStream_t)stream);
}

int supir_tile_blend(const float* tiles, const double* weights, float* canvas, const int* tile_hw, int k, int b, int C, int Hc, int Wc,
                     int T, void* stream) {
    if (!tiles || !weights || !canvas) return SUPIR_ERR_ARG;
    SupirTileList tl;
    if (const int rc = pack_tiles(tl, tile_hw, k)) return rc;
    return supir_tile_blend_launch(tiles, weights, canvas, tl, b, C, Hc, Wc, T, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------- grouped launches
static void pack_common(GemmArgs& a, const supir_gemm_shape& sh, const supir_gemm_problem& q) {
    a.A = (const bf16_t*)q.A; a.Wt = (const bf16_t*)q.W; a.C = q.C; a.C2 = q.C2;
    a.bias = q.bias; a.rowbias = (const bf16_t*)q.rowbias; a.res = (const bf16_t*)q.residual;
    a.lda = q.lda; a.ldc = q.ldc; a.ldc2 = q.ldc2; a.ldr = q.ldr; a.ld_rb = q.ld_rowbias;
    set_act(a, sh.act); a.out_mode = sh.out_mode; a.alpha = sh.alpha; a.n_split = sh.n_split;
    a.rowstats_out = q.rowstats_out; a.rs_ld = q.rs_ld;
    a.ln_stats = q.ln_stats; a.ln_ld = q.ln_ld; a.ln_slots = q.ln_slots; a.ln_colsum = q.ln_colsum; a.ln_eps = sh.ln_eps;
    a.gn_part_out = q.gn_partials_out;
    const size_t lines = q.prefetch ? q.prefetch_bytes / 128 : 0;
    a.pf_ptr = (const char*)q.prefetch;
    a.pf_lines = lines > 0x7fffffffu ? 0x7fffffffu : (unsigned)lines;
}

int supir_gemm_grouped(const supir_gemm_shape* shape, const supir_gemm_problem* problems, int n, void* stream) {
    if (!shape || !problems || n < 1 || n > 2) return SUPIR_ERR_ARG;
    const supir_gemm_shape& sh = *shape;
    if (sh.kind < 0 || sh.kind > 2 || sh.act < 0 || sh.act > SUPIR_ACT_GEGLU_ERF || sh.out_mode < 0 || sh.out_mode > 2) return SUPIR_ERR_ARG;
    GemmArgs a[2] = {};
    for (int i = 0; i < n; ++i) {
        const supir_gemm_problem& q = problems[i];
        if (!q.A || !q.W || !q.C) return SUPIR_ERR_ARG;
        if (q.ln_stats && (!q.ln_colsum || q.ln_slots < 0 || (q.ln_slots > 0 && (q.ln_ld < q.ln_slots || (q.ln_ld & 1))))) return SUPIR_ERR_ARG;
        if (q.prefetch_bytes && !q.prefetch) return SUPIR_ERR_ARG;
        pack_common(a[i], sh, q);
        if (sh.kind == SUPIR_GROUP_CONV3X3) {
            if (sh.B <= 0 || sh.H <= 0 || sh.W <= 0 || sh.OH <= 0 || sh.OW <= 0 || sh.act > 1 || sh.out_mode != 0) return SUPIR_ERR_ARG;
            if ((sh.stride != 1 && sh.stride != 2) || (sh.upsample && sh.stride != 1)) return SUPIR_ERR_SHAPE;
            a[i].M = sh.B * sh.OH * sh.OW; a[i].N = sh.Cout; a[i].K = 9 * sh.Cin;
            a[i].rows_per_batch = sh.OH * sh.OW;
            a[i].H = sh.H; a[i].W = sh.W; a[i].Cin = sh.Cin; a[i].OH = sh.OH; a[i].OW = sh.OW; a[i].stride = sh.stride;
            a[i].pad_t = sh.pad_t; a[i].pad_l = sh.pad_l; a[i].up = sh.upsample ? 1 : 0;
        } else {
            a[i].M = sh.M; a[i].N = sh.N; a[i].K = sh.K;
            if ((q.rowbias || sh.out_mode == 2 || sh.kind == SUPIR_GROUP_QKV) && sh.rows_per_batch <= 0) return SUPIR_ERR_ARG;
            a[i].rows_per_batch = sh.rows_per_batch > 0 ? sh.rows_per_batch : sh.M;
            if (q.rowstats_out && (sh.out_mode != 0 || sh.act == SUPIR_ACT_GEGLU || sh.act == SUPIR_ACT_GEGLU_ERF || q.rs_ld <= 0 || (q.rs_ld & 1))) return SUPIR_ERR_ARG;
        }
        if (a[i].M <= 0 || a[i].N <= 0 || a[i].K <= 0) return SUPIR_ERR_ARG;
    }
    if (sh.kind == SUPIR_GROUP_QKV) {
        for (int i = 0; i < n; ++i)
            if (!a[i].C2) return SUPIR_ERR_ARG;
        return supir_gemm16_qkv_launch_n(a, n, (hipStream_t)stream);
    }
    if (sh.tile == 37) {
        if (sh.kind != SUPIR_GROUP_GEMM) return SUPIR_ERR_SHAPE;
        for (int i = 0; i < n; ++i)
            if (a[i].gn_part_out) return SUPIR_ERR_SHAPE;
        return supir_gemm_big_launch_n(a, n, (hipStream_t)stream);
    }
    if (sh.tile < 32 || sh.tile > 35) return SUPIR_ERR_SHAPE;
    for (int i = 0; i < n; ++i) {
        if (a[i].rowstats_out) {   // the producer's slot index is the tile column: the caller's rs_ld must cover the tile used
            const int bn = (sh.tile == 32 || sh.tile == 35) ? 80 : 160;
            if ((a[i].N + bn - 1) / bn > a[i].rs_ld) return SUPIR_ERR_ARG;
        }
    }
    return supir_gemm16_launch_n(a, n, (hipStream_t)stream, sh.tile, sh.kind == SUPIR_GROUP_CONV3X3);
}

int supir_flash_attn_d64_grouped(const supir_attn_problem* problems, int n, int B, int H, int Tq, float scale, void* stream) {
    if (!problems || n < 1 || n > 2) return SUPIR_ERR_ARG;
    AttnArgs a[2] = {};
    for (int i = 0; i < n; ++i) {
        const supir_attn_problem& q = problems[i];
        if (!q.Q || !q.K || !q.Vt || !q.O || (q.flags & ~1)) return SUPIR_ERR_ARG;
        if ((q.flags & 1) && Tq != q.Tk) return SUPIR_ERR_SHAPE;
        a[i].Q = (const bf16_t*)q.Q; a[i].K = (const bf16_t*)q.K; a[i].Vt = (const bf16_t*)q.Vt; a[i].O = (bf16_t*)q.O;
        a[i].B = B; a[i].H = H; a[i].Tq = Tq; a[i].Tk = q.Tk; a[i].ldq = q.ldq; a[i].ldk = q.ldk; a[i].ldvt = q.ldvt; a[i].ldo = q.ldo;
        a[i].scale_log2e = scale * 1.4426950408889634f;
        a[i].causal = q.flags & 1;
    }
    return supir_attn_launch_n(a, n, (hipStream_t)stream);
}

int supir_groupnorm_grouped(const supir_gn_problem* problems, int n, int B, int HW, int C, float eps, int act, void* stream) {
    if (!problems || n < 1 || n > 2) return SUPIR_ERR_ARG;
    GnArgs a[2] = {};
    for (int i = 0; i < n; ++i) {
        const supir_gn_problem& q = problems[i];
        if (!q.x1 || !q.gamma || !q.beta || !q.out || (!q.part1 && !q.workspace)) return SUPIR_ERR_ARG;
        if (q.C1 <= 0 || q.C1 > C) return SUPIR_ERR_ARG;
        a[i].x1 = (const bf16_t*)q.x1; a[i].x2 = (const bf16_t*)q.x2; a[i].x1raw = (const bf16_t*)q.x1raw; a[i].x2raw = (const bf16_t*)q.x2raw;
        a[i].gamma = q.gamma; a[i].beta = q.beta; a[i].partial = q.workspace;
        a[i].part_u1 = q.part1; a[i].nch1 = q.nchunk1; a[i].part_u2 = q.part2; a[i].nch2 = q.nchunk2;
        a[i].mod_g = (const bf16_t*)q.mod_g; a[i].mod_b = (const bf16_t*)q.mod_b; a[i].out = (bf16_t*)q.out;
        a[i].B = B; a[i].HW = HW; a[i].C = C; a[i].C1 = q.C1; a[i].ld1 = q.ld1; a[i].ld2 = q.ld2; a[i].ldm = q.ldm; a[i].ldo = q.ldo;
        a[i].act = act; a[i].eps = eps; a[i].cscale = q.control_scale;
    }
    return supir_groupnorm_launch_n(a, n, (hipStream_t)stream);
}

}  // extern "C"
