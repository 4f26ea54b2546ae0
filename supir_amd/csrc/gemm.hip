// bf16 MFMA GEMM / implicit-GEMM conv3x3 for gfx950 (MI355X).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )        (every nn.Linear / 1x1 conv / 3x3 conv of the path)
//
// Replaces the cuBLAS / cuDNN calls issued by the reference's nn.Linear / nn.Conv2d modules
// (reference: sgm/modules/attention.py:87,100,106,213-219,587,611; sgm/modules/diffusionmodules/
// openaimodel.py:127,196,263,300; SUPIR/modules/SUPIR_v0.py:48,79-87; sgm/modules/diffusionmodules/model.py:60-124).
//
// Design (CDNA4):
//  * activations are NHWC bf16 ("tokens x channels"), weights are [N][K] bf16 with K contiguous, so both MFMA
//    operands are K-contiguous rows and are staged with the same code;
//  * global -> LDS by `global_load_lds_dwordx4` (no VGPR round trip). The LDS image is lane-linear, so the
//    bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the ds_read_b128 side:
//    16-B chunk c of row r lives at chunk c ^ ((r>>1)&7)  (conflict-free for the 32x32x16 fragment read);
//  * 3x3 convolution is the same kernel with an im2col gather in the A loader: K runs (ky,kx,cin) and a
//    64-wide K step never straddles a tap because Cin % 64 == 0; the gather address of each row is computed once
//    per tap; halo / padding lanes read a zero page; stride 2, asymmetric padding (VAE Downsample) and
//    nearest-2x upsampling are folded into the gather;
//  * v_mfma_f32_32x32x16_bf16, 4 (or 8) waves, operands swapped (a = W rows, b = A rows) so that every lane
//    ends up with 4 consecutive output channels of one token; fused epilogue (bias, per-batch time-embedding add,
//    residual add, SiLU, GEGLU, scale, LayerNorm fold / row statistics), bf16 output staged through LDS so the
//    global stores are 16 B per lane and row-contiguous;
//  * S-deep LDS ring (2 by default), ONE raw s_barrier per K step, counted vmcnt for S > 2; the next tile's
//    global->LDS instructions are interleaved with the MFMA groups of the current one;
//  * XCD-aware workgroup -> tile mapping (8 private L2s); one-shot prefetch of a later launch's weights on the way out.
// tools/probes/gemm_timeline.hip builds this file with -DSUPIR_GEMM_TIMELINE for a per-phase s_memtime breakdown.
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

#ifndef SUPIR_DEFAULT_STAGES_CODE
#define SUPIR_DEFAULT_STAGES_CODE 1  /* 2-deep ring: measured best (deeper rings cost a resident workgroup per CU) */
#endif


__device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];

#ifdef SUPIR_GEMM_TIMELINE
// tools/probes/gemm_timeline.hip only: per-wave s_memtime breakdown of the main loop (never defined in the product build)
__device__ unsigned long long* g_tl_buf;
extern "C" void supir_tl_set(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &p, sizeof(p)); }
#define TL_NOW() __builtin_amdgcn_s_memtime()
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// waves per SIMD the register allocator must leave room for: as many workgroups as the LDS ring lets a CU hold
constexpr int gemm_min_waves(int BM, int BN, int WM, int WN, int S, int KS) {
    int blocks = 163840 / (KS * S * (BM + BN) * 128 + 256);
    if (blocks < 1) blocks = 1;
    int w = blocks * WM * WN * KS / 4;
    return w < 1 ? 1 : (w > 6 ? 6 : w);
}

// KS > 1 (experimental, tile 7 only; NOT selected by any heuristic or autotune list until it has been validated and measured on
// hardware): KS groups of WM x WN waves share one output tile and take alternate K steps from their own LDS rings; the partial
// accumulators are exchanged through LDS after the loop (group g keeps row block i == g) and each group runs the epilogue of its
// row block.  Meant for the M = 2048 shapes: 160 tiles of 128x128 on 256 CUs leave one 4-wave workgroup per CU, i.e. one wave per
// SIMD with nothing to overlap LDS / load-issue latency with; two groups double that and halve the serial K loop.
template <int BM, int BN, int WM, int WN, int S, bool CONV, bool TRANS, int KS = 1>
__global__ __launch_bounds__(64 * WM * WN * KS, gemm_min_waves(BM, BN, WM, WN, S, KS)) void gemm_bf16_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 64 * WM * WN;          // threads of one K group: WM x WN waves, each owning a (BM/WM) x (BN/WN) sub-tile
    constexpr int RPL = NT / 8;               // tile rows staged per load instruction of the workgroup
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_LOADS = BM / RPL, B_LOADS = BN / RPL, LOADS = A_LOADS + B_LOADS;
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    // epilogue scratch carved out of the (then idle) ring: row statistics < 16 KB, bias / column sums at 16 KB, C staging at 20 KB
    static_assert(20480 + WM * WN * KS * 32 * (WTN * 2 + 16) <= 2 * KS * STAGE, "epilogue scratch must fit the smallest LDS ring");
    static_assert(KS == 1 || (KS == 2 && MI == 2 && !CONV), "split-K groups: plain / transposed GEMM, one 32-row block per group");
    static_assert(KS == 1 || WM * WN * NI * 16 * 64 * 4 * KS <= S * STAGE, "split-K exchange buffer must fit ring 0");
    static_assert(WM * BM * 0 + WN * BM * 8 <= 16384 && 16384 + 2 * BN * 4 <= 20480, "epilogue scratch regions overlap");

    const int kg = KS > 1 ? (int)threadIdx.x / NT : 0;                 // K group of this wave (wave-uniform)
    const int tid = KS > 1 ? (int)threadIdx.x - kg * NT : (int)threadIdx.x;   // thread / wave index INSIDE the group
    const int lane = tid & 63, wave = tid >> 6;
    const int bwave = KS > 1 ? (int)(threadIdx.x >> 6) : wave;        // wave index inside the workgroup
    const int ring_off = KS > 1 ? kg * (S * STAGE) : 0;                // this group's LDS ring
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    // split-K: block = split * tiles + tile (GemmArgs::ksplit); bid = the tile part
    const int ks_split = p.ksplit > 1 ? (int)blockIdx.x / (tiles_m * tiles_n) : 0;
    const int bid = p.ksplit > 1 ? (int)blockIdx.x - ks_split * tiles_m * tiles_n : (int)blockIdx.x;
    // Workgroup -> tile: block b runs on XCD b % 8, and every XCD has a private 4 MB L2.  The host picks the partition of
    // the tile grid over the 8 XCDs that minimises what the L2s have to pull in (each XCD touching an operand panel
    // fetches its own copy): a gm x gn grid of XCD regions (each needing 1/gm of A and 1/gn of W), or, when the tile
    // counts do not divide, contiguous 1-D id ranges; `order` = which tile index runs fastest inside a region.
    int tile_m, tile_n;
    if (p.gm > 0) {
        const int xcd = bid & 7, idx = bid >> 3;
        const int rm = tiles_m / p.gm, rn = tiles_n / p.gn;
        const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
        int lm, ln;
        if (p.order == 0) { ln = idx / rm; lm = idx - ln * rm; }
        else { lm = idx / rn; ln = idx - lm * rn; }
        tile_m = xm * rm + lm;
        tile_n = xn * rn + ln;
    } else {
        const int id = xcd_remap(bid, tiles_m * tiles_n);
        if (p.order == 0) { tile_n = id / tiles_m; tile_m = id - tile_n * tiles_m; }
        else { tile_m = id / tiles_n; tile_n = id - tile_m * tiles_n; }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- loader state: slot s = j*256 + tid -> row j*32 + (tid>>3), physical chunk tid&7 ----
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7);  // logical 16-B chunk this lane fetches
    const bf16_t* a_ptr[A_LOADS];
    int a_iy0[A_LOADS], a_ix0[A_LOADS];  // conv: oy*stride - pad_t, ox*stride - pad_l
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
        int m = m0 + j * RPL + lrow;
        m = m < p.M ? m : p.M - 1;
        if constexpr (CONV) {
            const int b = m / p.rows_per_batch;
            const int r = m - b * p.rows_per_batch;
            const int oy = r / p.OW, ox = r - oy * p.OW;
            a_iy0[j] = oy * p.stride - p.pad_t;
            a_ix0[j] = ox * p.stride - p.pad_l;
            a_ptr[j] = p.A + (size_t)b * p.H * p.W * p.lda + lchunk * 8;
        } else {
            a_ptr[j] = p.A + (size_t)m * p.lda + lchunk * 8;
        }
    }
    const bf16_t* b_ptr[B_LOADS];
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) {
        int n = n0 + j * RPL + lrow;
        n = n < p.N ? n : p.N - 1;
        b_ptr[j] = p.Wt + (size_t)n * p.K + lchunk * 8;
    }
    const int VH = p.up ? 2 * p.H : p.H, VW = p.up ? 2 * p.W : p.W;
    const bf16_t* zero = (const bf16_t*)g_zero_page;

    const int nk_all = (p.K >> 6) / KS;                                     // K steps of this group without split-K
    const int nk = p.ksplit > 1 ? nk_all / p.ksplit : nk_all;               // ... of this workgroup (the host guarantees divisibility)
    const int k_begin = ks_split * nk * 64;                                 // (split-K only with KS == 1)
    int ld_k0 = (KS > 1 ? kg * 64 : 0) + k_begin, ld_cin0 = 0, ld_ky = 0, ld_kx = 0;  // position of the NEXT tile to stage
    if constexpr (CONV) {
        if (k_begin > 0) {
            const int tap0 = k_begin / p.Cin;
            ld_cin0 = k_begin - tap0 * p.Cin;
            ld_ky = tap0 / 3;
            ld_kx = tap0 - ld_ky * 3;
        }
    }
    // conv: the tap (ky, kx) only changes every Cin/64 K steps, so the gather address of each A row (or "padding": null ->
    // zero page) is computed once per tap, not once per load: the im2col arithmetic was ~800 of the 1800-1950 cycles a K step
    // of the 128x128 conv tile took (s_memtime; the plain GEMM's K step is ~1050)
    const bf16_t* a_tap[A_LOADS];
    auto conv_set_tap = [&]() {
        if constexpr (CONV) {
#pragma unroll
            for (int j = 0; j < A_LOADS; ++j) {
                int iy = a_iy0[j] + ld_ky, ix = a_ix0[j] + ld_kx;
                const bool ok = (unsigned)iy < (unsigned)VH && (unsigned)ix < (unsigned)VW;
                if (p.up) { iy >>= 1; ix >>= 1; }
                a_tap[j] = ok ? a_ptr[j] + ((size_t)iy * p.W + ix) * p.lda : nullptr;
            }
        }
    };
    conv_set_tap();
    // one global->LDS instruction (q < A_LOADS: A rows, else W rows) of the tile at the current stage position
    auto stage_one = [&](int buf, int q) {
        char* sA = smem + ring_off + buf * STAGE;
        char* sB = sA + A_BYTES;
        if (q < A_LOADS) {
            const int j = q;
            const bf16_t* src;
            if constexpr (CONV) src = a_tap[j] ? a_tap[j] + ld_cin0 : zero;
            else src = a_ptr[j] + ld_k0;
            glds16(src, sA + (j * NT + wave * 64) * 16);
        } else {
            const int j = q - A_LOADS;
            glds16(b_ptr[j] + ld_k0, sB + (j * NT + wave * 64) * 16);
        }
    };
    auto stage_advance = [&]() {
        ld_k0 += 64 * KS;
        if constexpr (CONV) {
            ld_cin0 += 64;
            if (ld_cin0 == p.Cin) {
                ld_cin0 = 0;
                if (++ld_kx == 3) { ld_kx = 0; ++ld_ky; }
                conv_set_tap();
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LOADS; ++q) stage_one(buf, q);
        stage_advance();
    };

    // epilogue vectors of this tile's BN columns: fetched now (one element per thread), parked in LDS after the main loop
    float pre_bias = 0.f, pre_cs = 0.f;
    if constexpr (!TRANS) {
        if (tid < BN) {
            const int n = n0 + tid < p.N ? n0 + tid : p.N - 1;
            if (p.bias) pre_bias = p.bias[n];
            if (p.ln_stats) pre_cs = p.ln_colsum[n];
        }
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: row = base32 + l31, logical chunk 2*ks+half, physical chunk ^ ((row>>1)&7)
    const int sw = (l31 >> 1) & 7;
    const int a_row_off = (wm * WTM + l31) * 128;
    const int b_row_off = (wn * WTN + l31) * 128;

    // S-deep LDS ring, ONE barrier per K step: loads for tile kt+S-1 are issued right after the barrier of step kt (into
    // the buffer step kt-1 just finished reading) and stay in flight across the next S-2 barriers (counted vmcnt).
    // With M = 2048 the grid is only 1-2 workgroups per CU, so latency hiding has to come from this queue depth.
    // Weights are read once per network call and the two models hold ~5 GB of them, so every launch meets its W cold in HBM
    // and pays that latency on each K step (tools/cold_probe.py: +25..40 %).  On its way out each wave touches a slice of a
    // LATER launch's weights: one 4-byte global->LDS load per 128-byte line into a scratch LDS row behind the ring that nothing
    // reads, issued after the last store so nothing in this kernel ever waits on it (the hardware drains it at s_endpgm with
    // the store acks).  Measured on the 1024^2 step under graph replay: -4..5.5 % with the weight of the NEXT launch; issuing
    // before the main loop instead (more lead time, but the K steps' vmcnt waits then cover these loads) gave only -2.6 %,
    // and looking 2 / 3 / 4 / 8 launches ahead gave -4.4 / -2.3 / -2.0 / +0.5 %.
    auto prefetch_next = [&]() {
        const unsigned pf_lines = p.pf_lines;
        if (pf_lines == 0) return;
        const unsigned total_waves = gridDim.x * (NT / 64 * KS), gw = blockIdx.x * (NT / 64 * KS) + bwave;
        const unsigned n_instr = (pf_lines + 63) >> 6;
        for (unsigned i = gw; i < n_instr; i += total_waves) {
            unsigned line = i * 64 + lane;
            line = line < pf_lines ? line : pf_lines - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.pf_ptr + (size_t)line * 128),
                                             (__attribute__((address_space(3))) void*)(smem + KS * S * STAGE), 4, 0, 0);
        }
    };
    // nk (above) = K steps of this group / workgroup (the host only picks KS > 1 when K % (64 * KS) == 0)
#ifdef SUPIR_GEMM_TIMELINE
    unsigned long long tl_wait = 0, tl_bar = 0, tl_issue = 0, tl_comp = 0;
    const unsigned long long tl_t0 = TL_NOW();
#endif
    {
        const int pre = nk < S - 1 ? nk : S - 1;
        for (int s = 0; s < pre; ++s) stage(s);
    }
#ifdef SUPIR_GEMM_TIMELINE
    const unsigned long long tl_loop0 = TL_NOW();
#endif
    for (int kt = 0; kt < nk; ++kt) {
#ifdef SUPIR_GEMM_TIMELINE
        const unsigned long long tl_a = TL_NOW();
#endif
        const int rem = nk - 1 - kt;
        const int inflight = rem < S - 2 ? rem : S - 2;  // younger tiles allowed to stay outstanding
        if constexpr (S >= 4) {
            if (inflight >= 2) wait_vmcnt<2 * LOADS>();
            else if (inflight == 1) wait_vmcnt<LOADS>();
            else wait_vmcnt<0>();
        } else if constexpr (S == 3) {
            if (inflight >= 1) wait_vmcnt<LOADS>();
            else wait_vmcnt<0>();
        } else {
            wait_vmcnt<0>();
        }
#ifdef SUPIR_GEMM_TIMELINE
        const unsigned long long tl_b = TL_NOW();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef SUPIR_GEMM_TIMELINE
        const unsigned long long tl_c = TL_NOW();
#endif
#ifdef SUPIR_GEMM_TIMELINE
        const unsigned long long tl_d = tl_c;
#endif
        // The next tile's global->LDS instructions are spread over the four 16-wide K slices, between the MFMA groups:
        // issued back to back after the barrier they stall on the CU's 64 B/clk vector-memory path for as long as the
        // whole MFMA group takes (s_memtime: 450 cycles of issue stall + 690 of LDS/MFMA per step on a 128x128 tile, one
        // wave per SIMD), i.e. nothing overlapped.  Fragments for slice ks+1 are read while slice ks is in the matrix pipe.
        const bool do_stage = kt + S - 1 < nk;
        const int sbuf = (kt + S - 1) % S;
        const int buf = kt % S;
        const char* sA = smem + ring_off + buf * STAGE;
        const char* sB = sA + A_BYTES;
        constexpr int FB = (MI * NI >= 8) ? 1 : 2;   // fragment double buffering, unless the accumulators already fill the file
        bf16x8 af[FB][MI], bfr[FB][NI];
        auto read_frags = [&](int ks, int slot) {
            const int coff = ((2 * ks + half) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < MI; ++i) af[slot][i] = *(const bf16x8*)(sA + a_row_off + i * 32 * 128 + coff);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[slot][j] = *(const bf16x8*)(sB + b_row_off + j * 32 * 128 + coff);
        };
        if constexpr (FB == 2) read_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (FB == 2) {
                if (ks < 3) read_frags(ks + 1, (ks + 1) & 1);
            } else {
                read_frags(ks, 0);
            }
            if (do_stage) {
#pragma unroll
                for (int q = (ks * LOADS) / 4; q < ((ks + 1) * LOADS) / 4; ++q) stage_one(sbuf, q);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if constexpr (TRANS)
                        acc[i][j] = SUPIR_MFMA_32x32x16(af[ks & (FB - 1)][i], bfr[ks & (FB - 1)][j], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = SUPIR_MFMA_32x32x16(bfr[ks & (FB - 1)][j], af[ks & (FB - 1)][i], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);   // keep the slice order: the scheduler must not regroup loads and MFMAs
        }
        if (do_stage) stage_advance();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SUPIR_GEMM_TIMELINE
        {
            const unsigned long long tl_e = TL_NOW();
            tl_wait += tl_b - tl_a;
            tl_bar += tl_c - tl_b;
            tl_issue += tl_d - tl_c;
            tl_comp += tl_e - tl_d;
        }
#endif
    }
#ifdef SUPIR_GEMM_TIMELINE
    const unsigned long long tl_loop1 = TL_NOW();
#endif

    // ------------------------------------------------------------------ epilogue
    // LayerNorm folding: per-row mean / rstd of the A operand, reduced from the producer GEMM's per-wave-column partial
    // sums (fixed order: reproducible).  Lane l31 of fragment row-block i owns row m0 + wm*WTM + i*32 + l31.
    float ln_mean[MI], ln_rstd[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        ln_mean[i] = 0.f;
        ln_rstd[i] = 1.f;
    }
    if (p.ln_stats) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int m = m0 + wm * WTM + i * 32 + l31;
            m = m < p.M ? m : p.M - 1;
            if (p.ln_slots == 0) {   // finalised statistics: [M][2] = (mean, rstd) from supir_rowstats_finalize
                const float* st2 = p.ln_stats + (size_t)m * 2;
                ln_mean[i] = st2[0];
                ln_rstd[i] = st2[1];
                continue;
            }
            const float* st = p.ln_stats + (size_t)m * p.ln_ld * 2;   // ln_ld is even: rows are 16-byte aligned
            float sm = 0.f, sq = 0.f;
            int sl = 0;
#pragma unroll 2
            for (; sl + 4 <= p.ln_slots; sl += 4) {   // 2 independent 16-byte loads per trip
                const f32x4 u = *(const f32x4*)(st + 2 * sl), v = *(const f32x4*)(st + 2 * sl + 4);
                sm += (u[0] + u[2]) + (v[0] + v[2]);
                sq += (u[1] + u[3]) + (v[1] + v[3]);
            }
            for (; sl < p.ln_slots; ++sl) {
                sm += st[2 * sl];
                sq += st[2 * sl + 1];
            }
            const float inv = 1.0f / (float)p.K;
            const float mean = sm * inv;
            float var = sq * inv - mean * mean;
            var = var > 0.f ? var : 0.f;
            ln_mean[i] = mean;
            ln_rstd[i] = rsqrtf(var + p.ln_eps);
        }
    }
    __syncthreads();   // LDS is reused below: every wave must be done with its last fragment reads
    if constexpr (KS == 2) {
        // Exchange of the K-partial accumulators (reduce-scatter over the two groups): group g keeps row block i == g and adds
        // the other group's partial of it.  Layout [sender][wave][j][r][lane] fp32: lane-contiguous, conflict-free; 2 x 32 KB in
        // ring 0, which is idle now.  acc[] is indexed with compile-time constants in both (wave-uniform) branches.
        float* xch = (float*)smem;
        const int w_off = wave * (NI * 16 * 64) + lane;
        float* mine = xch + kg * (WM * WN * NI * 16 * 64) + w_off;
        const float* theirs = xch + (1 - kg) * (WM * WN * NI * 16 * 64) + w_off;
        if (kg == 0) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(j * 16 + r) * 64] = acc[1][j][r];
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(j * 16 + r) * 64] = acc[0][j][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][j][r] += theirs[(j * 16 + r) * 64];
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[1][j][r] += theirs[(j * 16 + r) * 64];
        }
        __syncthreads();   // the epilogue scratch below overlays the exchange buffer
    }
    float* s_bias = (float*)(smem + 16384);   // [BN] bias, [BN] LN column sums (the row-statistics scratch lives below 16 KB)
    float* s_cs = s_bias + BN;
    if constexpr (!TRANS) {
        if (tid < BN) {
            s_bias[tid] = pre_bias;
            s_cs[tid] = pre_cs;
        }
        __syncthreads();
    }
    if constexpr (TRANS) {
        // D[i = token][j = channel]: lane owns channel l31, tokens (r&3)+8*(r>>2)+4*half -> 4 consecutive tokens
        bf16_t* Cb = (bf16_t*)p.C;
        float t_bias[NI], t_cs[NI];   // loaded once, ahead of every dependent use (the epilogue is latency-bound otherwise)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WTN + j * 32 + l31;
            const int nc = n < p.N ? n : p.N - 1;
            t_bias[j] = p.bias ? p.bias[nc] : 0.f;
            t_cs[j] = p.ln_stats ? p.ln_colsum[nc] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if constexpr (KS > 1) {
                    if (i != kg) continue;   // the other K group finishes this row block
                }
                const int n = n0 + wn * WTN + j * 32 + l31;
                const bool n_ok = n < p.N;
                const float bz = t_bias[j], cs = t_cs[j];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float val[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = acc[i][j][rg * 4 + e];
                        if (p.ln_stats) {  // stats of token 8*rg + 4*half + e live in the lane with that l31
                            const int src = 8 * rg + 4 * half + e;
                            const float mu = __shfl(ln_mean[i], src, 64), rs = __shfl(ln_rstd[i], src, 64);
                            a = rs * (a - mu * cs);
                        }
                        val[e] = p.alpha * (a + bz);
                    }
                    const int m = m0 + wm * WTM + i * 32 + 8 * rg + 4 * half;
                    if (!n_ok || m >= p.M) continue;
                    const int b = m / p.rows_per_batch, t = m - b * p.rows_per_batch;
                    if (m + 3 < p.M && t + 3 < p.rows_per_batch && ((t | p.ldc) & 3) == 0) {
                        u16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = f2bf(val[e]);
                        supir_store8(Cb + ((size_t)b * p.N + n) * p.ldc + t, __builtin_bit_cast(u32x2, o));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int me = m + e;
                            if (me < p.M) {
                                const int be = me / p.rows_per_batch, te = me - be * p.rows_per_batch;
                                ((u16*)Cb)[((size_t)be * p.N + n) * p.ldc + te] = f2bf(val[e]);
                            }
                        }
                    }
                }
            }
        prefetch_next();
        return;
    } else {
        // D[i = channel][j = token]: lane owns token l31, channels (r&3)+8*(r>>2)+4*half -> 4 consecutive channels.
        //
        // Measured with s_memtime (tools/probes/gemm_timeline.hip) the first form of this epilogue -- per fragment:
        // conditional loads, conditional math, store -- took 40-50 % of a K = 1280 launch: ~7000 instructions of branchy,
        // fully unrolled code streamed once through the instruction cache, with a dependent global load in every
        // iteration.  Now: bias / LN column sums come from LDS (fetched before the main loop), row bias and residual
        // are fetched as one batch per 32-row block, the arithmetic is branch-free (absent terms are exact zeros /
        // mean 0 / rstd 1), and the three run-time switches that remain (SiLU, output type, store path) select one of a
        // few compact compile-time variants so a launch only executes the code it needs.
        if (p.act == 2) {
            // GEGLU: fragment pair (value, gate) = (j even, j odd) inside the wave's 64 columns
            if constexpr (NI == 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if constexpr (KS > 1) {
                        if (i != kg) continue;   // the other K group finishes this row block
                    }
                    const int m = m0 + wm * WTM + i * 32 + l31;
                    const bool m_ok = m < p.M;
                    const float mu = ln_mean[i], rs = ln_rstd[i];
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int nl = wn * 64 + 8 * rg + 4 * half;  // column inside the tile (each wave owns 64 columns);
                        // value rows nl.., gate rows nl+32.. (interleaved weights)
                        const f32x4 bv = *(const f32x4*)(s_bias + nl), bg = *(const f32x4*)(s_bias + nl + 32);
                        const f32x4 cv = *(const f32x4*)(s_cs + nl), cg = *(const f32x4*)(s_cs + nl + 32);
                        float r[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = rs * (acc[i][0][rg * 4 + e] - mu * cv[e]) + bv[e];
                            const float g = rs * (acc[i][1][rg * 4 + e] - mu * cg[e]) + bg[e];
                            r[e] = v * (p.fast_gelu ? gelu_fast_f(g) : gelu_f(g));
                        }
                        u32x2 o = {f2bf_pk(r[0], r[1]), f2bf_pk(r[2], r[3])};
                        const int no = (n0 >> 1) + wn * 32 + 8 * rg + 4 * half;
                        if (m_ok && n0 + nl + 32 < p.N) supir_store8((bf16_t*)p.C + (size_t)m * p.ldc + no, o);
                    }
                }
            }
            prefetch_next();
            return;
        }
        // bf16 output can go through a wave-private LDS block (row stride padded by 16 B) so that the global stores are
        // 16 B per lane and row-contiguous instead of 32 rows x 16 bytes per store instruction
        constexpr int C_RS = WTN * 2 + 16;
        char* c_stage = smem + 20480 + bwave * (32 * C_RS);
        auto epilogue = [&](auto silu_c, auto mode_c) {
            constexpr bool SILU = decltype(silu_c)::value;
            constexpr int MODE = decltype(mode_c)::value;   // 0: bf16 direct, 1: fp32 direct, 2: bf16 via LDS
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (KS > 1) {
                    if (i != kg) continue;   // the other K group finishes this row block
                }
                const int m = m0 + wm * WTM + i * 32 + l31;
                const bool m_ok = m < p.M;
                const int mc = m_ok ? m : p.M - 1;
                const float mu = ln_mean[i], rs = ln_rstd[i];
                u32x2 e_rb[NI][4], e_res[NI][4];
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) e_rb[j][rg] = e_res[j][rg] = u32x2{0u, 0u};
                if (p.rowbias) {
                    const bf16_t* rbp = p.rowbias + (size_t)(mc / p.rows_per_batch) * p.ld_rb;
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int n = n0 + wn * WTN + j * 32 + 8 * rg + 4 * half;
                            e_rb[j][rg] = *(const u32x2*)(rbp + (n < p.N ? n : 0));
                        }
                }
                if (p.res) {
                    const bf16_t* rp = p.res + (size_t)mc * p.ldr;
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int n = n0 + wn * WTN + j * 32 + 8 * rg + 4 * half;
                            e_res[j][rg] = *(const u32x2*)(rp + (n < p.N ? n : 0));
                        }
                }
                float rsum = 0.f, rsq = 0.f;   // row statistics of what this wave writes (for the NEXT LayerNorm)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int nl = wn * WTN + j * 32 + 8 * rg + 4 * half;
                        const int n = n0 + nl;
                        const bool ok = m_ok && n < p.N;   // N % 4 == 0: a lane's 4 channels are valid together
                        const f32x4 cs = *(const f32x4*)(s_cs + nl), bz = *(const f32x4*)(s_bias + nl);
                        const u32x2 rb = e_rb[j][rg], rr = e_res[j][rg];
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = rs * (acc[i][j][rg * 4 + e] - mu * cs[e]) + bz[e];
                            t += (e & 1) ? bfhi2f(rb[e >> 1]) : bflo2f(rb[e >> 1]);
                            if constexpr (SILU) t = p.act == 1 ? silu_f(t) : (p.act == 3 ? gelu_f(t) : quick_gelu_f(t));
                            t *= p.alpha;
                            t += (e & 1) ? bfhi2f(rr[e >> 1]) : bflo2f(rr[e >> 1]);
                            v[e] = t;
                        }
                        if constexpr (MODE == 1) {
                            if (ok) *(f32x4*)((float*)p.C + (size_t)ks_split * p.M * p.ldc + (size_t)m * p.ldc + n) = v;
                        } else {
                            const u32x2 o = {f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3])};
                            const float r0 = bflo2f(o[0]), r1 = bfhi2f(o[0]), r2 = bflo2f(o[1]), r3 = bfhi2f(o[1]);
                            const float ps = (r0 + r1) + (r2 + r3), pq = (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
                            rsum += ok ? ps : 0.f;
                            rsq += ok ? pq : 0.f;
                            if constexpr (MODE == 2) {
                                *(u32x2*)(c_stage + l31 * C_RS + (j * 32 + 8 * rg + 4 * half) * 2) = o;
                            } else {
                                if (ok) supir_store8((bf16_t*)p.C + (size_t)m * p.ldc + n, o);
                            }
                        }
                    }
                if constexpr (MODE == 2) {
                    // the wave's 32 x WTN block, now row-major in LDS: CPR lanes cover one row contiguously
                    constexpr int CPR = WTN / 8, RPI = 64 / CPR;
#pragma unroll
                    for (int rr = 0; rr < 32 / RPI; ++rr) {
                        const int row = rr * RPI + lane / CPR, ch = lane % CPR;
                        const int m2 = m0 + wm * WTM + i * 32 + row, n2 = n0 + wn * WTN + ch * 8;
                        const f32x4 piece = *(const f32x4*)(c_stage + row * C_RS + ch * 16);
                        if (m2 < p.M && n2 < p.N) supir_store16((bf16_t*)p.C + (size_t)m2 * p.ldc + n2, piece);
                    }
                }
                if (p.rowstats_out) {
                    rsum += __shfl_xor(rsum, 32, 64);
                    rsq += __shfl_xor(rsq, 32, 64);
                    if (half == 0) {   // per wave column -> LDS [WN][BM][2]
                        float* red = (float*)smem + ((size_t)wn * BM + wm * WTM + i * 32 + l31) * 2;
                        red[0] = rsum;
                        red[1] = rsq;
                    }
                }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        using M0_ = std::integral_constant<int, 0>;
        using M1_ = std::integral_constant<int, 1>;
        using M2_ = std::integral_constant<int, 2>;
        const bool via_lds = (p.ldc & 7) == 0 && (p.N & 7) == 0 && (((size_t)p.C) & 15) == 0;
        if (p.out_mode == 1) {
            if (p.act != 0) epilogue(T_{}, M1_{});
            else epilogue(F_{}, M1_{});
        } else if (via_lds) {
            if (p.act != 0) epilogue(T_{}, M2_{});
            else epilogue(F_{}, M2_{});
        } else {
            if (p.act != 0) epilogue(T_{}, M0_{});
            else epilogue(F_{}, M0_{});
        }
        if (p.rowstats_out) {
            // one slot per tile column: the WN wave columns are combined here in a fixed order (reproducible)
            __syncthreads();
            for (int r = tid; r < (KS > 1 && kg > 0 ? 0 : BM); r += NT) {   // one K group writes the tile's slots
                const int m = m0 + r;
                if (m >= p.M) continue;
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) {
                    sm += ((const float*)smem)[((size_t)w * BM + r) * 2];
                    sq += ((const float*)smem)[((size_t)w * BM + r) * 2 + 1];
                }
                float* dst = p.rowstats_out + ((size_t)m * p.rs_ld + tile_n) * 2;
                dst[0] = sm;
                dst[1] = sq;
            }
        }
        prefetch_next();
    }
#ifdef SUPIR_GEMM_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g_tl_buf && lane == 0) {
        const unsigned long long tl_end = TL_NOW();
        unsigned long long* o = g_tl_buf + ((size_t)blockIdx.x * (NT / 64) + wave) * 8;
        o[0] = tl_t0;
        o[1] = tl_loop0 - tl_t0;
        o[2] = tl_wait;
        o[3] = tl_bar;
        o[4] = tl_issue;
        o[5] = tl_comp;
        o[6] = tl_end - tl_loop1;
        o[7] = tl_end - tl_t0;
    }
#endif
}

static bool xcd_grid_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SUPIR_XCD_GRID");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// Choose the XCD partition for a tiles_m x tiles_n grid (see the kernel comment): a gm x gn grid of XCD regions and the tile
// order inside a region, minimising the bytes the eight private L2s have to pull in over the fabric.  Model (checked against
// rocprofv3 FETCH_SIZE, profiles/r02/pmc_summary.json): an XCD runs `slots` = 32 CUs x wgs_per_cu workgroups at a time; within one
// round the co-resident workgroups stream their operand panels through the L2 together, so each panel is fetched once.  What is
// re-fetched: (a) an operand that is swept more than once -- the implicit-GEMM input is read once per tap (resweep = 9) -- unless
// the XCD's share of it stays resident (<= ~3 MB of the 4 MB L2); (b) with several rounds per XCD, the operand that every round
// needs again, unless it fits beside the other round's band.  The first version of this model charged gn*A + gm*W only: it put the
// 1280-channel 32x32 convolutions on a (1, 8) grid whose 5.2 MB activation map does not stay resident -- 389 MB fetched for 35 MB
// of operands (11x), 4.5 TB/s of fabric traffic in an "MFMA-bound" kernel.
void supir_choose_xcd_grid(GemmArgs& a, int tiles_m, int tiles_n, double a_bytes, double w_bytes, int resweep, int wgs_per_cu, int nx) {
    a.gm = a.gn = 0;
    if (!xcd_grid_enabled() || ((tiles_m * tiles_n) % nx)) return;
    static int model = -1;
    if (model < 0) {
        const char* e = getenv("SUPIR_XCD_MODEL");   // 0: the round-1 cost model (kept for A/B runs)
        model = (e && e[0] == '0') ? 0 : 1;
    }
    const double c_bytes = 2.0 * (double)a.M * a.N / (double)nx;   // each XCD also write-allocates its share of the output
    const double RES = 3.0e6, BOTH = 3.5e6;
    const int slots = 32 * (wgs_per_cu < 1 ? 1 : wgs_per_cu);
    double best = 0.0;
    for (int gm = nx; gm >= 1; gm >>= 1) {
        const int gn = nx / gm;
        if (tiles_m % gm || tiles_n % gn) continue;
        if (model == 0) {
            double cost = gn * a_bytes + gm * w_bytes;
            if (a_bytes / gm + w_bytes / gn + c_bytes > 3.3e6) cost *= 1.5;
            if (a.gm == 0 || cost < best) {
                best = cost;
                a.gm = gm;
                a.gn = gn;
                a.order = (a_bytes / gm > w_bytes / gn) ? 1 : 0;
            }
            continue;
        }
        const double ax = a_bytes / gm, wx = w_bytes / gn;
        const int n_wg = (tiles_m / gm) * (tiles_n / gn);
        const int rounds = (n_wg + slots - 1) / slots;
        for (int order = 0; order < 2; ++order) {
            double at, wt;
            if (rounds == 1) {
                at = ax <= RES ? ax : resweep * ax;
                wt = wx;
            } else if (order == 1) {   // tile_n fastest: a round = a band of tile rows x all tile columns of the region
                const double band_a = ax / rounds;
                at = band_a <= RES ? ax : resweep * ax;
                wt = (wx + band_a <= BOTH) ? wx : rounds * wx;
            } else {                   // tile_m fastest: a round = all tile rows x a band of tile columns
                const double band_w = wx / rounds;
                wt = wx;
                at = (ax + band_w <= BOTH) ? ax : rounds * (ax <= RES ? ax : resweep * ax);
            }
            const double cost = at + wt;
            if (a.gm == 0 || cost < best) {
                best = cost;
                a.gm = gm;
                a.gn = gn;
                a.order = order;
            }
            if (rounds == 1) break;   // order is irrelevant inside a single round: keep 0
        }
    }
}

template <int BM, int BN, int WM, int WN, int S, bool CONV, bool TRANS, int KS = 1>
static int launch_gemm(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    {
        const double a_bytes = CONV ? 2.0 * (double)a.M * (a.up ? 0.25 : (double)(a.stride * a.stride)) * a.Cin
                                    : 2.0 * (double)a.M * a.K;
        constexpr int lds = KS * S * (BM + BN) * 128 + 256;
        supir_choose_xcd_grid(a, (a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a_bytes, 2.0 * (double)a.N * a.K, CONV ? 9 : 1, 163840 / lds);
    }
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * (a.ksplit > 1 ? a.ksplit : 1);
    constexpr int smem = KS * S * (BM + BN) * 128 + 256;   // ring(s) + the prefetch scratch row
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, S, CONV, TRANS, KS>;
    static bool attr_set = false;
    if (!attr_set) {
        if (supir_note_hip_status(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) != SUPIR_OK) return SUPIR_ERR_HIP;
        attr_set = true;
    }
    SUPIR_LAUNCH(kern, dim3(tiles), dim3(64 * WM * WN * KS), smem, st, a);
    return SUPIR_LAUNCH_STATUS();
}

// Tile table (index -> block tile, wave grid, per-wave tile):
//   0: 128x128 2x2 (64x64)   1: 128x64 2x2 (64x32)   2: 64x128 2x2 (32x64)   3: 64x64 2x2 (32x32)
//   4: 256x128 4x2 (64x64, 512 threads)   5: 256x256 2x4 (128x64, 512 threads)   6: 256x128 2x2 (128x64)
//   7: 128x128 2x2 x 2 K groups (512 threads; experimental -- see the kernel comment; never chosen automatically)
// Heuristic default: biggest of tiles 0-3 that still yields >= ~1 wave of workgroups over 256 CUs; the Python layer
// autotunes over the whole table per problem shape.
int supir_gemm_select_tile(int M, int N, int act, int force_tile) {
    auto tiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    int sel = force_tile;
    if (act == 2 && (sel == 1 || sel == 3)) sel = -1;  // GEGLU needs 64 columns per wave (value + gate fragment pair)
    if (sel < 0 && act == 2) sel = (tiles(128, 128) >= 200) ? 0 : 2;
    if (sel < 0) {
        if (tiles(128, 128) >= 240) sel = 0;
        else if (N % 128 != 0 && tiles(128, 64) >= 200) sel = 1;
        else if (tiles(128, 64) >= 240) sel = 1;
        else if (tiles(64, 128) >= 240) sel = 2;
        else sel = 3;
    }
    return sel;
}

// `force_tile`: -1 auto; otherwise bits 0-2 = tile index (table above), bits 3-4 = LDS ring depth override
// (0 default, 1 -> 2 stages, 2 -> 3, 3 -> 4) -- the override exists for tools/bench_kernels.py's sweeps.
template <bool CONV, bool TRANS>
static int dispatch_gemm(const GemmArgs& a, hipStream_t st, int force_tile) {
    const int sel = supir_gemm_select_tile(a.M, a.N, a.act, force_tile < 0 ? -1 : (force_tile & 7));
    int stages = force_tile < 0 ? 0 : ((force_tile >> 3) & 3);
    if (stages == 0) stages = SUPIR_DEFAULT_STAGES_CODE;
    stages += 1;
    if ((a.K >> 6) < 3 && stages > 2) stages = 2;
#define SUPIR_GEMM_CASE(BM_, BN_, WM_, WN_)                                        \
    switch (stages) {                                                              \
        case 2: return launch_gemm<BM_, BN_, WM_, WN_, 2, CONV, TRANS>(a, st);     \
        default: return launch_gemm<BM_, BN_, WM_, WN_, 3, CONV, TRANS>(a, st);    \
    }
    switch (sel) {
        case 0: SUPIR_GEMM_CASE(128, 128, 2, 2)
        case 1: SUPIR_GEMM_CASE(128, 64, 2, 2)
        case 2: SUPIR_GEMM_CASE(64, 128, 2, 2)
        case 3: SUPIR_GEMM_CASE(64, 64, 2, 2)
        case 4: return launch_gemm<256, 128, 4, 2, 2, CONV, TRANS>(a, st);
        case 5: return launch_gemm<256, 256, 2, 4, 2, CONV, TRANS>(a, st);
        case 6: return launch_gemm<256, 128, 2, 2, 2, CONV, TRANS>(a, st);
        default:
            // tile 7 (experimental, explicit request only): 128x128 with two K groups; plain / transposed GEMM with K % 128 == 0,
            // else tile 0
            if constexpr (!CONV) {
                if (a.K % 128 == 0 && a.act != 2) return launch_gemm<128, 128, 2, 2, 2, false, TRANS, 2>(a, st);
            }
            SUPIR_GEMM_CASE(128, 128, 2, 2)
    }
#undef SUPIR_GEMM_CASE
}

// split-K finalize: out[m][n] = act(sum_s part[s][m][n] + bias[n]) in bf16; the partials are summed in split order (deterministic)
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ part, int ksplit, int M, int N, int ld_part,
                                                              const float* __restrict__ bias, int act, bf16_t* __restrict__ out, int ldo) {
    const int nv = N >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * nv) return;
    const int m = (int)(idx / nv), n = (int)(idx - (long)m * nv) * 4;
    f32x4 acc = *(const f32x4*)(part + (size_t)m * ld_part + n);
    for (int s = 1; s < ksplit; ++s) {
        const f32x4 v = *(const f32x4*)(part + ((size_t)s * M + m) * ld_part + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = acc[e] + (bias ? bias[n + e] : 0.f);
        if (act == 1) t = silu_f(t);
        y[e] = t;
    }
    const u32x2 o = {f2bf_pk(y[0], y[1]), f2bf_pk(y[2], y[3])};
    supir_store8(out + (size_t)m * ldo + n, o);
}

int supir_splitk_finalize_launch(const float* part, int ksplit, int M, int N, int ld_part, const float* bias, int act, bf16_t* out, int ldo,
                                 hipStream_t st) {
    if (ksplit < 1 || M <= 0 || N <= 0 || N % 4 || ld_part % 4 || ldo % 4 || act < 0 || act > 1) return SUPIR_ERR_SHAPE;
    const long total = (long)M * (N / 4);
    SUPIR_LAUNCH(splitk_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, ksplit, M, N, ld_part, bias, act, out, ldo);
    return SUPIR_LAUNCH_STATUS();
}

// (sum, sum of squares) partials [M][ld][2] -> (mean, rstd) [M][2]; one thread per row, fixed summation order
__global__ __launch_bounds__(256) void rowstats_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int M,
                                                                int ld, int slots, int dim, float eps) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* st = part + (size_t)m * ld * 2;
    float sm = 0.f, sq = 0.f;
    int sl = 0;
    for (; sl + 4 <= slots; sl += 4) {
        const f32x4 u = *(const f32x4*)(st + 2 * sl), v = *(const f32x4*)(st + 2 * sl + 4);
        sm += (u[0] + u[2]) + (v[0] + v[2]);
        sq += (u[1] + u[3]) + (v[1] + v[3]);
    }
    for (; sl < slots; ++sl) {
        sm += st[2 * sl];
        sq += st[2 * sl + 1];
    }
    const float inv = 1.0f / (float)dim;
    const float mean = sm * inv;
    float var = sq * inv - mean * mean;
    var = var > 0.f ? var : 0.f;
    out[(size_t)m * 2] = mean;
    out[(size_t)m * 2 + 1] = rsqrtf(var + eps);
}

int supir_rowstats_finalize_launch(const float* part, float* out, int M, int ld, int slots, int dim, float eps, hipStream_t st) {
    if (M <= 0 || slots <= 0 || ld < slots || (ld & 1) || dim <= 0) return SUPIR_ERR_SHAPE;
    SUPIR_LAUNCH(rowstats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, st, part, out, M, ld, slots, dim, eps);
    return SUPIR_LAUNCH_STATUS();
}

int supir_gemm_launch(const GemmArgs& a_in, bool conv, hipStream_t st, int force_tile) {
    GemmArgs a = a_in;
    {
        // bytes each operand costs in L2 fills under the two orders (8 XCDs, each with a private L2)
        const double a_bytes = conv ? 2.0 * (double)a.M * (a.up ? 0.25 : (double)(a.stride * a.stride)) * a.Cin : 2.0 * (double)a.M * a.K;
        const double w_bytes = 2.0 * (double)a.N * a.K;
        const int tm = (a.M + 127) / 128, tn = (a.N + 127) / 128;
        const double cost_m_fast = a_bytes * (tn < 8 ? tn : 8) + w_bytes;
        const double cost_n_fast = w_bytes * (tm < 8 ? tm : 8) + a_bytes;
        a.order = cost_n_fast < cost_m_fast ? 1 : 0;
    }
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return SUPIR_ERR_ARG;
    if (a.ksplit > 1) {   // split-K: fp32 partials from the gemm.hip tiles 0..3 only, no epilogue terms, whole K steps per split
        if (force_tile < 0 || force_tile > 3 || a.out_mode != 1 || a.bias || a.res || a.rowbias || a.act != 0 || a.alpha != 1.0f ||
            a.rowstats_out || a.ln_stats || a.gn_part_out || a.K % (64 * a.ksplit) != 0)
            return SUPIR_ERR_SHAPE;
    }
    if (a.gn_part_out && ((force_tile < 32 || force_tile > 35) && (force_tile < 38 || force_tile > 40) && force_tile != 42 && force_tile != 45 && (force_tile < 48 || force_tile > 51))) return SUPIR_ERR_SHAPE;   // GroupNorm partials: gemm16 epilogues only
    if (force_tile == 37) return conv ? SUPIR_ERR_SHAPE : supir_gemm_big_launch(a, st);   // 256 x 320 GEGLU tile (gemm_big.hip)
    if (force_tile >= 32) {   // the 16x16x32-MFMA, 256-workgroup tiles (gemm16.hip): exact shapes only
        return supir_gemm16_launch(a, st, force_tile, conv);
    }
    if (a.K % 64 != 0 || a.N % 4 != 0 || a.lda % 8 != 0) return SUPIR_ERR_SHAPE;
    if (conv && (a.Cin % 64 != 0 || a.K != 9 * a.Cin)) return SUPIR_ERR_SHAPE;
    if (a.act == 2 && (a.N % 128 != 0 || a.out_mode != 0 || a.res || a.rowbias)) return SUPIR_ERR_SHAPE;
    if (a.out_mode == 2) {
        if (a.act != 0 || a.res || a.rowbias) return SUPIR_ERR_SHAPE;
        return conv ? SUPIR_ERR_SHAPE : dispatch_gemm<false, true>(a, st, force_tile);
    }
    return conv ? dispatch_gemm<true, false>(a, st, force_tile) : dispatch_gemm<false, false>(a, st, force_tile);
}
