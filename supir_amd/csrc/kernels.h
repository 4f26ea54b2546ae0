// Argument blocks and host-side launchers shared between the kernel translation units and the C-ABI (api.hip).
#pragma once
#include "common.h"

struct GemmArgs {
    const bf16_t* A;        // plain: [M][lda]; conv: NHWC input [B][H][W][lda]
    const bf16_t* Wt;       // [N][K]
    void* C;
    const float* bias;      // [N] or null
    const bf16_t* rowbias;  // [batch][ld_rb] or null: added to every row of that batch (time embedding)
    const bf16_t* res;      // [M][ldr] or null: residual
    int M, N, K;
    int lda, ldc, ldr, ld_rb;
    int rows_per_batch;     // rows (tokens) per batch element
    int H, W, Cin, OH, OW, stride, pad_t, pad_l, up;  // conv geometry (virtual input = 2H x 2W when up)
    int act;                // 0 none, 1 SiLU, 2 GEGLU (value/gate interleaved per 32 columns)
    int korder;             // convolutions on the eight-phase tile: 1 = K loop as (channel chunk, tap), 0 = (tap, channel chunk) (set by the launcher)
    int fast_gelu;          // GEGLU epilogues: 1 = the fitted GELU (gelu_fast_f, |error| <= 2.5e-5; SUPIR_ACT_GEGLU), 0 = erf (SUPIR_ACT_GEGLU_ERF); set
                            // by csrc/api.hip from the caller's activation code and honoured by EVERY GEGLU tile (gemm.hip, gemm16.hip,
                            // gemm_big.hip), so a layer's arithmetic does not depend on the tile that ran it
    int out_mode;           // 0 bf16 [M][ldc], 1 fp32 [M][ldc], 2 bf16 transposed [batch][N][ldc]
    float alpha;            // result = alpha * act(acc + bias + rowbias) + residual
    int order;              // tile order inside an XCD's id range: 0 = tile_m fastest (W panel shared), 1 = tile_n fastest
    int gm, gn;             // 2-D XCD grid (gm*gn == 8): XCD x owns tile rows chunk x/gn, tile columns chunk x%gn; 0 = 1-D ranges
    // ---- LayerNorm folding (transformer blocks): y = LN(x).W^T == rstd*(x.W'^T - mean*colsum) + b'  with W' = gamma (.) W
    float* rowstats_out;    // producer: [M][rs_ld][2] per-row (sum, sum of squares) of the bf16 output, one slot per wave column
    int rs_ld;              //           slots allocated per row (>= tiles_n * waves_n)
    const float* ln_stats;  // consumer: the producer's rowstats for THIS GEMM's A rows, [M][ln_ld][2]
    int ln_ld, ln_slots;    //           row stride (slots) and number of valid slots to sum
    const float* ln_colsum; // consumer: [N] sum_k W'[n][k]
    float ln_eps;
    // ---- next-weight prefetch (supir_set_next_prefetch): 128-byte lines of a LATER launch's weight matrix that this launch
    // touches on its way out, so they are in the Infinity Cache / L2 instead of HBM when that launch streams them
    const char* pf_ptr;
    unsigned pf_lines;
    // ---- fused q|k|v projection (supir_gemm_bf16_qkv): columns >= n_split go, transposed per batch, to C2 [batch][N - n_split][ldc2]
    void* C2;
    int ldc2, n_split;
    // ---- GroupNorm statistics from the producer (supir_set_next_gn_partials): per (batch, tile row of BM tokens, 10-channel unit) the
    // (sum, sum of squares) of the bf16 values this launch stores: gn_part_out[((b * (rows_per_batch / BM) + chunk) * (N / 10) + unit) * 2]
    float* gn_part_out;
    // ---- split-K (supir_conv3x3_bf16_splitk): ksplit > 1 -> the grid is ksplit x the tile grid; workgroup (tile, s) accumulates K
    // steps [s, s + 1) * K / ksplit only and stores its fp32 partial at C + s * M * ldc (out_mode 1, no epilogue terms); a second
    // launch (supir_splitk_finalize) sums the partials in a fixed order and applies bias / activation.  For the convolutions whose
    // tile grid is a fraction of the machine and whose K is long (ZeroSFT mlp_shared: 1280 -> 128 at 32 x 32 = 64 tiles x 180 K steps)
    int ksplit;
};

// Grouped launch: NP independent problems of IDENTICAL shape in one grid (supir_gemm_grouped & co).  Block b runs on XCD b % 8, so
// problem q owns the 8 / NP XCDs [q * 8 / NP, (q + 1) * 8 / NP): its operands stay in those L2s and the tile map inside them is the
// single-problem one on a smaller XCD grid.  The problem index is wave-uniform (a scalar offset into the kernel-argument segment).
template <int NP>
struct GemmArgsN {
    GemmArgs p[NP];
};

struct AttnArgs {
    const bf16_t* Q;
    const bf16_t* K;
    const bf16_t* Vt;
    bf16_t* O;
    int B, H, Tq, Tk;
    int ldq, ldk, ldvt, ldo;
    float scale_log2e;  // softmax scale * log2(e)
    int causal;         // 1: key j is visible to query i only for j <= i (text towers of the conditioner); needs Tq == Tk
};

template <int NP>
struct AttnArgsN {
    AttnArgs p[NP];
};

// xattn.hip: to_q projection (with the LayerNorm fold of supir_gemm_bf16_ln) + attention over a short cached key axis, one launch
struct XattnArgs {
    const bf16_t* X;        // [B * T][ldx] tokens (the LayerNorm INPUT when ln_stats is set)
    const bf16_t* Wq;       // [H * 64][C]  (gamma-folded when ln_stats is set)
    const float* bias;      // [H * 64] or null (the folded beta . W^T term)
    const bf16_t* K;        // [B][Tk][ldk], head h at columns h * 64
    const bf16_t* Vt;       // [B * H][64][ldvt], zero beyond Tk
    bf16_t* O;              // [B * T][ldo], head h at columns h * 64
    int B, H, T, Tk, C;
    int ldx, ldk, ldvt, ldo;
    const float* ln_stats;  // as GemmArgs: [M][ln_ld][2] partial (sum, sum of squares) slots, or [M][2] (mean, rstd) when ln_slots == 0
    int ln_ld, ln_slots;
    const float* ln_colsum; // [H * 64]
    float ln_eps;
    float scale_log2e;
    const char* pf_ptr;     // next-weight prefetch, as GemmArgs
    unsigned pf_lines;
    int gm, gn;             // 2-D XCD grid (launcher): XCD x owns token-block chunk x / gn and head chunk x % gn; gm = 0 -> 1-D ranges
};

struct GnArgs {
    const bf16_t* x1;   // [B][HW][ld1], channels [0,C1)
    const bf16_t* x2;   // [B][HW][ld2], channels [C1,C)  (null when C1 == C)  -- the ZeroSFT / skip concat
    const bf16_t* x1raw;  // ZeroSFT lerp only: raw counterpart of x1 (null -> x1 itself)
    const bf16_t* x2raw;  // ZeroSFT lerp only: un-projected skip (h before + zero_conv(c)); null -> x2 itself
    float* partial;     // [B][nchunk][32][2]  (sum, sumsq)
    const float* given; // optional [B][32][2] (mean, var): normalise with these instead of this tensor's own statistics
    const float* part_u1;  // optional producer partials of x1: [B][nch1][C1 / 10][2] (see GemmArgs::gn_part_out); with C1 < C also part_u2
    const float* part_u2;  //          ... of x2: [B][nch2][(C - C1) / 10][2].  When set, no statistics launch is made
    int nch1, nch2;
    const float* gamma; // [C]
    const float* beta;  // [C]
    const bf16_t* mod_g;  // [B][HW][ldm] ZeroSFT gamma map or null
    const bf16_t* mod_b;  // [B][HW][ldm] ZeroSFT beta map
    bf16_t* out;        // [B][HW][ldo]
    int B, HW, C, C1;
    int ld1, ld2, ldm, ldo;
    int nchunk, rows_per_chunk;
    int act;            // 0 none, 1 SiLU
    float eps;
    float cscale;       // ZeroSFT control_scale (1 -> no lerp)
};

template <int NP>
struct GnArgsN {
    GnArgs p[NP];
};

int supir_gemm_select_tile(int M, int N, int act, int force_tile);
int supir_splitk_finalize_launch(const float* part, int ksplit, int M, int N, int ld_part, const float* bias, int act, bf16_t* out, int ldo,
                                 hipStream_t st);
int supir_groupnorm_parts_finalize_launch(const float* part, int B, int nchunk, int C, int unit, int HW, float* mean_var, hipStream_t st);
int supir_rowstats_finalize_launch(const float* part, float* out, int M, int ld, int slots, int dim, float eps, hipStream_t st);
int supir_gemm_launch(const GemmArgs& a, bool conv, hipStream_t st, int force_tile);
// nx = XCDs this problem's tiles are spread over (8; 4 for each problem of a two-problem grouped launch)
void supir_choose_xcd_grid(GemmArgs& a, int tiles_m, int tiles_n, double a_bytes, double w_bytes, int resweep, int wgs_per_cu, int nx = 8);
// gemm16.hip: tiles 32 (128 x 80) / 33 (128 x 160), v_mfma_f32_16x16x32_bf16, two K groups per workgroup
bool supir_gemm16_supported(const GemmArgs& a, int tile, bool conv);
int supir_gemm16_launch(const GemmArgs& a, hipStream_t st, int tile, bool conv);
int supir_gemm16_qkv_launch(const GemmArgs& a, hipStream_t st);
// grouped forms: n (1 or 2) problems of identical shape, one launch (tiles 33 / 34 / 35 and the fused q|k|v tile)
int supir_gemm16_launch_n(const GemmArgs* a, int n, hipStream_t st, int tile, bool conv);
int supir_gemm16_qkv_launch_n(const GemmArgs* a, int n, hipStream_t st);
// gemm_big.hip: tile 37 = 256 x 320, activation operand global -> VGPR, GEGLU epilogue (16-row value / gate interleave)
// measurement knobs (NOT part of the C ABI: undeclared in include/supir_hip.h, used by tools/ A/B scripts only).
// knob 0: 1 = the exact-erf GELU in EVERY GEGLU epilogue (tiles 34 and 37) instead of the fitted one (the product default)
// knob 1: wave arrangement of the 256 x 160 tile (csrc/gemm16.hip): 0 = product policy, 1 = always 4 x 2, 2 = always 8 x 1
// knob 2: GroupNorm apply with n row batches per workgroup instead of ~32 KB per workgroup (measured: no gain; csrc/norm.hip)
// knob 3: flash attention d64: 0 = product policy, 1 = the round-3 kernel, 2 = always eight waves, 3 = always four waves (round-4 form)
// knob 4: fused q|k|v tile: 0 = product policy, 1 = 256 x 160, 2 = 256 x 128
// knob 6: K order of tile 42's convolutions: 0 = (channel chunk, tap) (product), 1 = (tap, channel chunk) as every other tile
// knob 5: xattn_q workgroup order: 0 = 2-D XCD grid (product), 1 = 1-D ranges with the heads fastest (round 4)
#ifdef SUPIR_TOOLS
int supir_debug_knob_value(int which);      // libsupir_hip_tools.so: process-global variant switches (csrc/api.hip)
#else
static inline int supir_debug_knob_value(int) { return 0; }   // product libraries: no switch state exists; every variant test folds away
#endif
bool supir_gemm_big_supported(const GemmArgs& a);
int supir_gemm_big_launch(const GemmArgs& a, hipStream_t st);
int supir_gemm_big_launch_n(const GemmArgs* a, int n, hipStream_t st);
int supir_attn_launch(const AttnArgs& a, hipStream_t st);
int supir_attn_launch_n(const AttnArgs* a, int n, hipStream_t st);
bool supir_xattn_q_supported(const XattnArgs& a);
int supir_xattn_q_launch(const XattnArgs& a, hipStream_t st);
// attention_d512.hip: one head of dimension 512 (VAE mid block), 32-key tiles, 32 x 512 output tile per wave
// (splits / workspace: the key-split form, see attention_d512.hip; workspace == nullptr -> one pass over all keys per workgroup)
int supir_attn_d512_launch(const bf16_t* Q, const bf16_t* K, const bf16_t* Vt, bf16_t* O, int B, int Tq, int Tk, int ldq, int ldk,
                           int ldvt, int ldo, float scale, int splits, void* workspace, size_t workspace_bytes, hipStream_t st);
size_t supir_attn_d512_workspace_bytes(int B, int Tq, int Tk, int splits);
int supir_softmax_rows_launch(const float* S, bf16_t* P, int rows, int T, int Tpad, long lds_, long ldp, float scale,
                              hipStream_t st);
int supir_groupnorm_launch(GnArgs a, hipStream_t st);
int supir_groupnorm_launch_n(const GnArgs* a, int n, hipStream_t st);   // statistics from the producers (part_u1) or given only
int supir_groupnorm_stats_launch(GnArgs a, float* sums_out, hipStream_t st);
int supir_layernorm_launch(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, int rows, int C, int ldx,
                           int ldy, float eps, hipStream_t st);
int supir_conv3x3_smallcin_launch(const float* x, const float* w, const float* bias, const bf16_t* add, bf16_t* out, int B,
                                  int Cin, int H, int W, int Cout, int ld_add, int ldo, hipStream_t st);
int supir_conv3x3_smallcout_launch(const bf16_t* x, const bf16_t* w, const float* bias, float* out, int B, int Cin, int H,
                                   int W, int Cout, int ldx, hipStream_t st);
int supir_pointwise_nchw_launch(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                                long HW, float in_scale, hipStream_t st);
int supir_wavelet_level_launch(const float* img, float* low, float* high, int planes, int H, int W, int radius, int first,
                               hipStream_t st);
// sampler.hip: the elementwise halves of one Restore-EDM sampler step around the network call (fp32 latents, host-side scalars)
int supir_edm_pre_launch(const float* x, const float* eps, float s_noise, float noise_mul, float c_in, float* x_hat, float* net_in,
                         long n, int reps, hipStream_t st);
int supir_edm_post_launch(const float* net_out, const float* x_hat, const float* x_center, float c_out, float c_skip, float cfg,
                          float restore_mul, float sigma_hat, float dt, float* x_next, long n, int reps, hipStream_t st);
// tiled sampler: origins of the (<= SUPIR_MAX_TILES) tiles one network call stacks; passed to the kernels by value
#define SUPIR_MAX_TILES 64
struct SupirTileList {
    int n;
    int hi[SUPIR_MAX_TILES];
    int wi[SUPIR_MAX_TILES];
};
int supir_edm_pre_tiles_launch(const float* x, const float* eps, float s_noise, float noise_mul, float c_in, float* x_hat, float* net_in,
                               const SupirTileList& tl, int b, int C, int Hc, int Wc, int T, int reps, hipStream_t st);
int supir_tile_blend_launch(const float* tiles, const double* w, float* canvas, const SupirTileList& tl, int b, int C, int Hc, int Wc, int T,
                            hipStream_t st);
int supir_prefetch_launch(const void* p, size_t bytes, void* sink, hipStream_t st);
int supir_resample_u8_launch(const uint8_t* src, uint8_t* dst_u8, float* dst_f32, const float* lut, const int* bounds, const int* kk,
                             int ksize, int in_h, int in_w, int out_h, int out_w, int ch, int vertical, hipStream_t st);
int supir_bicubic_f32_launch(const float* src, uint8_t* out_u8, float* out_f32, int C, int H, int W, int OH, int OW, hipStream_t st);
