// bf16 MFMA GEMM for the M = 2048-token shapes of the 32x32-latent transformer stacks (tiles 32 / 33 of the tile table).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )      same contract as gemm.hip, for M % 128 == 0, N % BN == 0, K % 128 == 0
//
// Why a second kernel: 70 % of the GEMM launches of a 1024^2 step have M = 2048 and N = 1280 (to_q / to_out / ff.net.2 /
// proj_in / proj_out of the 1280-wide SpatialTransformers, sgm/modules/attention.py:100-106,213-219,587,611).  On the
// 32x32x16 tiles of gemm.hip that is 160 tiles of 128x128 (96 of 256 CUs idle) or 640 tiles of 64x64 (2.5 waves of workgroups,
// twice the L1 traffic per FLOP); measured hot: 17.4 / 14.6 us = 385-460 TFLOP/s whatever the tile.  What bounds a CU here
// is its global->LDS fill rate (64 B/clk): a tile of area a needs (BM + BN) * 128 B per K step, so the right tile is the
// squarest one that gives EXACTLY one workgroup per CU: 2048 x 1280 / 256 = 128 x 80.  80 = 5 x 16, hence
// v_mfma_f32_16x16x32_bf16 (same FLOP rate as the 32x32x16 form).  N = 2560 (fused to_q|to_k) -> 128 x 160, also 256 tiles.
//
//  * 512 threads = two K groups of four waves: group g takes K steps g, g+2, ... from its own 2-deep LDS ring, so every SIMD
//    holds two waves (one per group) whose load issue / LDS reads / MFMAs interleave; the partial accumulators are
//    reduce-scattered through LDS after the loop (group g finishes token-fragment rows [g*MI/2, (g+1)*MI/2)) and both groups
//    run the epilogue on their half;
//  * global -> LDS by global_load_lds_dwordx4, 8 rows (1 KB) per wave instruction, source-side XOR swizzle
//    (chunk ^= (row >> 1) & 7), undone by the ds_read_b128 fragment reads: conflict-free for the 16-row x 4-chunk
//    fragment of the 16x16x32 MFMA as well (checked per ds_read_b128 lane group);
//  * operands swapped (a = W rows, b = A rows): a lane ends with 4 consecutive channels of one token; the fused epilogue
//    is the one of gemm.hip (bias, LayerNorm fold / row statistics, row bias, residual, SiLU, alpha; bf16 output staged
//    through LDS for 16-byte row-contiguous stores); the transposed (V^T) variant swaps the operands back so that a lane
//    holds 4 consecutive tokens of one channel;
//  * exact shapes only (the dispatcher falls back to gemm.hip otherwise): no bounds checks anywhere in the kernel.
#include "kernels.h"
#include <type_traits>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int BN, int WM, int WN, bool TRANS>
__global__ __launch_bounds__(512, 2) void gemm16_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, KS = 2, S = 2, NTG = 256;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES, RING = S * STAGE_BYTES;
    constexpr int A_Q = BM / 32;                       // 8-row chunks of A per wave and K step (chunk id = wave + 4 q)
    constexpr int B_CH = BN / 8, B_Q = (B_CH + 3) / 4;   // chunks of W per tile / per wave (the last q may be partial)
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16, MIH = MI / 2;
    static_assert(WM * WN == 4 && MI >= 2 && (MI & 1) == 0 && BN % 16 == 0 && WTN % 16 == 0, "tile / wave grid");
    // epilogue LDS map (the rings are idle by then): [0, XCH) K-group exchange, then per-wave C staging, bias / column sums,
    // row-statistics scratch
    constexpr int XCH_HALF = 4 * MIH * NI * 4 * 64 * 4;          // bytes one group sends
    constexpr int XCH = 2 * XCH_HALF;
    constexpr int C_RS = WTN * 2 + 16;                         // staged row stride (bytes): 16-byte pad against bank conflicts
    constexpr int C_STAGE = 16 * C_RS;                         // one 16-token block per wave
    constexpr int OFF_CST = XCH, OFF_BIAS = OFF_CST + 8 * C_STAGE, OFF_RED = OFF_BIAS + 2 * BN * 4;
    static_assert(OFF_RED + WN * BM * 8 <= KS * RING, "epilogue scratch must fit the LDS rings");

    // wave-uniform ids as scalars (readfirstlane): LDS destinations / branches on them stay on the scalar unit
    const int bwave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // wave inside the workgroup, 0..7
    const int kg = bwave >> 2;                         // K group
    const int wave = bwave & 3;                        // wave inside the group
    const int tid = (int)threadIdx.x & 255;
    const int lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int quad = lane >> 4, l15 = lane & 15;
    char* ring = smem + kg * RING;

    const int tiles_m = p.M / BM, tiles_n = p.N / BN;
    int tile_m, tile_n;
    if (p.gm > 0) {   // 2-D XCD grid, see gemm.hip
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int rm = tiles_m / p.gm, rn = tiles_n / p.gn;
        const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
        int lm, ln;
        if (p.order == 0) { ln = idx / rm; lm = idx - ln * rm; }
        else { lm = idx / rn; ln = idx - lm * rn; }
        tile_m = xm * rm + lm;
        tile_n = xn * rn + ln;
    } else {
        const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
        if (p.order == 0) { tile_n = id / tiles_m; tile_m = id - tile_n * tiles_m; }
        else { tile_m = id / tiles_n; tile_n = id - tile_m * tiles_n; }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- loader: wave w stages the 8-row chunks w, w+4, w+8, ... of A and of W; lane -> row lane>>3, physical 16-B chunk lane&7.
    // chunk ids of one wave all have the parity of w, so the swizzle ((row >> 1) & 7 with row = 8*chunk + lane>>3) is fixed per lane
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const bf16_t* a_src = p.A + (size_t)(m0 + wave * 8 + lrow) * p.lda + lchunk * 8 + kg * 64;
    const bf16_t* b_src = p.Wt + (size_t)(n0 + wave * 8 + lrow) * p.K + lchunk * 8 + kg * 64;
    const size_t a_qstride = (size_t)32 * p.lda, b_qstride = (size_t)32 * p.K;
    // one global->LDS instruction: q < A_Q -> A chunk wave + 4q, else W chunk wave + 4(q - A_Q) (skipped when beyond the tile)
    auto stage_one = [&](int buf, int q) {
        char* sA = ring + buf * STAGE_BYTES;
        if (q < A_Q) {
            glds16(a_src + q * a_qstride, sA + (wave + 4 * q) * 1024);
        } else {
            const int qb = q - A_Q;
            if ((B_CH % 4 == 0) || qb < B_Q - 1 || wave < (B_CH & 3))
                glds16(b_src + qb * b_qstride, sA + A_BYTES + (wave + 4 * qb) * 1024);
        }
    };
    constexpr int LOADS = A_Q + B_Q;
    auto stage_advance = [&]() {
        a_src += 64 * KS;
        b_src += 64 * KS;
    };

    // epilogue vectors of this tile's BN columns, fetched now (one element per thread of group 0), parked in LDS after the loop
    float pre_bias = 0.f, pre_cs = 0.f;
    if constexpr (!TRANS) {
        if (kg == 0 && tid < BN) {
            if (p.bias) pre_bias = p.bias[n0 + tid];
            if (p.ln_stats) pre_cs = p.ln_colsum[n0 + tid];
        }
    }

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment reads: row = base16 + l15, logical chunk 4*kk + quad, physical chunk ^ ((row >> 1) & 7) (bases are multiples of 16)
    const int sw = (l15 >> 1) & 7;
    const int a_row_off = (wm * WTM + l15) * 128;
    const int b_row_off = A_BYTES + (wn * WTN + l15) * 128;

    const int nk = (p.K >> 6) / KS;   // K steps of this group
#pragma unroll
    for (int q = 0; q < LOADS; ++q) stage_one(0, q);
    stage_advance();

    // LayerNorm folding: mean / rstd of the token rows this wave finishes after the K-group exchange (token fragments
    // i = kg*MIH + h).  Fetched and reduced HERE, under the first tile's load latency, not in the epilogue (a chain of dependent
    // L2 / fabric round trips there).  The four lanes that share a token (one per quad) take a contiguous quarter of the
    // producer's slots each -- all loads issued before the first use -- and combine by two xor shuffles (fixed order).
    float ln_mean[MIH], ln_rstd[MIH];
#pragma unroll
    for (int h = 0; h < MIH; ++h) {
        ln_mean[h] = 0.f;
        ln_rstd[h] = 1.f;
    }
    if (p.ln_stats) {
        const int spq = (p.ln_slots + 3) >> 2;            // slots per quad (<= 8: dispatcher)
        const int s_lo = quad * spq;
#pragma unroll
        for (int h = 0; h < MIH; ++h) {
            const int m = m0 + wm * WTM + (kg * MIH + h) * 16 + l15;
            if (p.ln_slots == 0) {
                const float* st2 = p.ln_stats + (size_t)m * 2;
                ln_mean[h] = st2[0];
                ln_rstd[h] = st2[1];
                continue;
            }
            const supir_f32x2* st = (const supir_f32x2*)(p.ln_stats + (size_t)m * p.ln_ld * 2);
            supir_f32x2 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                   // clamped index + zero weight instead of a branch around the load
                const int sl = s_lo + (e < spq ? e : spq - 1);   // beyond this quad's share: repeat its last slot (cache hit)
                v[e] = st[sl < p.ln_slots ? sl : p.ln_slots - 1];
            }
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = e < spq && s_lo + e < p.ln_slots;
                sm += ok ? v[e][0] : 0.f;
                sq += ok ? v[e][1] : 0.f;
            }
            sm += __shfl_xor(sm, 16, 64);
            sq += __shfl_xor(sq, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float inv = 1.0f / (float)p.K;
            const float mean = sm * inv;
            float var = sq * inv - mean * mean;
            var = var > 0.f ? var : 0.f;
            ln_mean[h] = mean;
            ln_rstd[h] = rsqrtf(var + p.ln_eps);
        }
    }

    // one K step; STAGE: also issue the next tile's global->LDS loads, spread over the two 32-wide K slices
    auto kstep = [&](int kt, auto stage_c) {
        constexpr bool STAGE = decltype(stage_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int sbuf = (kt + 1) & 1;
        const char* sT = ring + (kt & 1) * STAGE_BYTES;
        bf16x8 af[2][MI], bfr[2][NI];
        auto read_frags = [&](int kk, int slot) {
            const int coff = ((4 * kk + quad) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < MI; ++i) af[slot][i] = *(const bf16x8*)(sT + a_row_off + i * 16 * 128 + coff);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[slot][j] = *(const bf16x8*)(sT + b_row_off + j * 16 * 128 + coff);
        };
        read_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 0) read_frags(1, 1);
            if constexpr (STAGE) {
#pragma unroll
                for (int q = (kk * LOADS) / 2; q < ((kk + 1) * LOADS) / 2; ++q) stage_one(sbuf, q);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if constexpr (TRANS)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (STAGE) stage_advance();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int kt = 0; kt < nk - 1; ++kt) kstep(kt, std::true_type{});
    kstep(nk - 1, std::false_type{});

    // ------------------------------------------------------------------ epilogue
    __syncthreads();   // every wave is done with its last fragment reads: the rings are scratch from here on
    {
        // reduce-scatter of the K partials: group g keeps token fragments [g*MIH, (g+1)*MIH) and adds the other group's partial.
        // layout [sender][wave][h][j][r][lane] fp32: lane-contiguous, conflict-free
        float* xch = (float*)smem;
        const int w_off = wave * (MIH * NI * 4 * 64) + lane;
        float* mine = xch + kg * (XCH_HALF / 4) + w_off;
        const float* theirs = xch + (1 - kg) * (XCH_HALF / 4) + w_off;
        if (kg == 0) {
#pragma unroll
            for (int h = 0; h < MIH; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((h * NI + j) * 4 + r) * 64] = acc[MIH + h][j][r];
        } else {
#pragma unroll
            for (int h = 0; h < MIH; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((h * NI + j) * 4 + r) * 64] = acc[h][j][r];
        }
        if constexpr (!TRANS) {
            if (kg == 0 && tid < BN) {
                ((float*)(smem + OFF_BIAS))[tid] = pre_bias;
                ((float*)(smem + OFF_BIAS))[BN + tid] = pre_cs;
            }
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int h = 0; h < MIH; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[h][j][r] += theirs[((h * NI + j) * 4 + r) * 64];
        } else {
#pragma unroll
            for (int h = 0; h < MIH; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[MIH + h][j][r] += theirs[((h * NI + j) * 4 + r) * 64];
        }
    }
    // from here on the wave's data is fin[h][j] = acc[kg*MIH + h][j] (compile-time indices in both wave-uniform branches)
    auto prefetch_next = [&]() {
        const unsigned pf_lines = p.pf_lines;
        if (pf_lines == 0) return;
        const unsigned total_waves = gridDim.x * 8, gw = blockIdx.x * 8 + bwave;
        const unsigned n_instr = (pf_lines + 63) >> 6;
        for (unsigned i = gw; i < n_instr; i += total_waves) {
            unsigned line = i * 64 + lane;
            line = line < pf_lines ? line : pf_lines - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.pf_ptr + (size_t)line * 128),
                                             (__attribute__((address_space(3))) void*)(smem + KS * RING), 4, 0, 0);
        }
    };

    if constexpr (TRANS) {
        // D[i = token][j = channel]: lane owns channel l15 of fragment j, tokens 4*quad + r of fragment i -> 4 consecutive tokens
        bf16_t* Cb = (bf16_t*)p.C;
        auto run = [&](auto kg_c) {
            constexpr int KG = decltype(kg_c)::value;
#pragma unroll
            for (int h = 0; h < MIH; ++h) {
                const int mb = m0 + wm * WTM + (KG * MIH + h) * 16 + 4 * quad;      // first of this lane's 4 tokens
                const int b = mb / p.rows_per_batch, t = mb - b * p.rows_per_batch;   // rows_per_batch % 4 == 0 (dispatcher)
                float mu[4], rs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // statistics of token 4*quad + r live in the lanes with that l15
                    mu[r] = __shfl(ln_mean[h], 4 * quad + r, 64);
                    rs[r] = __shfl(ln_rstd[h], 4 * quad + r, 64);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = n0 + wn * WTN + j * 16 + l15;
                    const float bz = p.bias ? p.bias[n] : 0.f;
                    const float cs = p.ln_stats ? p.ln_colsum[n] : 0.f;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = p.alpha * (rs[r] * (acc[KG * MIH + h][j][r] - mu[r] * cs) + bz);
                    const u32x2 o = {f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3])};
                    *(u32x2*)(Cb + ((size_t)b * p.N + n) * p.ldc + t) = o;
                }
            }
        };
        if (kg == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
        prefetch_next();
        return;
    } else {
        const float* s_bias = (const float*)(smem + OFF_BIAS);
        const float* s_cs = s_bias + BN;
        char* c_stage = smem + OFF_CST + bwave * C_STAGE;
        auto run = [&](auto kg_c, auto silu_c) {
            constexpr int KG = decltype(kg_c)::value;
            constexpr bool SILU = decltype(silu_c)::value;
#pragma unroll
            for (int h = 0; h < MIH; ++h) {
                const int mrow = wm * WTM + (KG * MIH + h) * 16;    // first token of this 16-token block inside the tile
                const int m = m0 + mrow + l15;
                const float mu = ln_mean[h], rs = ln_rstd[h];
                u32x2 e_rb[NI], e_res[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) e_rb[j] = e_res[j] = u32x2{0u, 0u};
                if (p.rowbias) {
                    const bf16_t* rbp = p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rb + n0 + wn * WTN + 4 * quad;
#pragma unroll
                    for (int j = 0; j < NI; ++j) e_rb[j] = *(const u32x2*)(rbp + j * 16);
                }
                if (p.res) {
                    const bf16_t* rp = p.res + (size_t)m * p.ldr + n0 + wn * WTN + 4 * quad;
#pragma unroll
                    for (int j = 0; j < NI; ++j) e_res[j] = *(const u32x2*)(rp + j * 16);
                }
                float rsum = 0.f, rsq = 0.f;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int nl = wn * WTN + j * 16 + 4 * quad;
                    const f32x4 cs = *(const f32x4*)(s_cs + nl), bz = *(const f32x4*)(s_bias + nl);
                    const u32x2 rb = e_rb[j], rr = e_res[j];
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = rs * (acc[KG * MIH + h][j][e] - mu * cs[e]) + bz[e];
                        t += (e & 1) ? bfhi2f(rb[e >> 1]) : bflo2f(rb[e >> 1]);
                        if constexpr (SILU) t = silu_f(t);
                        t *= p.alpha;
                        t += (e & 1) ? bfhi2f(rr[e >> 1]) : bflo2f(rr[e >> 1]);
                        v[e] = t;
                    }
                    const u32x2 o = {f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3])};
                    const float r0 = bflo2f(o[0]), r1 = bfhi2f(o[0]), r2 = bflo2f(o[1]), r3 = bfhi2f(o[1]);
                    rsum += (r0 + r1) + (r2 + r3);
                    rsq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
                    *(u32x2*)(c_stage + l15 * C_RS + (j * 16 + 4 * quad) * 2) = o;
                }
                // the wave's 16 x WTN block is row-major in LDS now: 16-byte pieces, row-contiguous global stores
                constexpr int PPR = WTN / 8, PIECES = 16 * PPR;
#pragma unroll
                for (int rr = 0; rr < (PIECES + 63) / 64; ++rr) {
                    const int id = rr * 64 + lane;
                    if (PIECES % 64 == 0 || id < PIECES) {
                        const int row = id / PPR, ch = id - row * PPR;
                        const f32x4 piece = *(const f32x4*)(c_stage + row * C_RS + ch * 16);
                        *(f32x4*)((bf16_t*)p.C + (size_t)(m0 + mrow + row) * p.ldc + n0 + wn * WTN + ch * 8) = piece;
                    }
                }
                if (p.rowstats_out) {
                    rsum += __shfl_xor(rsum, 16, 64);
                    rsq += __shfl_xor(rsq, 16, 64);
                    rsum += __shfl_xor(rsum, 32, 64);
                    rsq += __shfl_xor(rsq, 32, 64);
                    if (quad == 0) {
                        if constexpr (WN == 1) {   // the wave covers the tile's whole column range: this IS the tile's slot
                            float* dst = p.rowstats_out + ((size_t)m * p.rs_ld + tile_n) * 2;
                            dst[0] = rsum;
                            dst[1] = rsq;
                        } else {
                            float* red = (float*)(smem + OFF_RED) + ((size_t)wn * BM + mrow + l15) * 2;
                            red[0] = rsum;
                            red[1] = rsq;
                        }
                    }
                }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        using K0_ = std::integral_constant<int, 0>;
        using K1_ = std::integral_constant<int, 1>;
        if (kg == 0) {
            if (p.act == 1) run(K0_{}, T_{});
            else run(K0_{}, F_{});
        } else {
            if (p.act == 1) run(K1_{}, T_{});
            else run(K1_{}, F_{});
        }
        if constexpr (WN > 1) {
            if (p.rowstats_out) {   // combine the wave columns in a fixed order: one slot per tile column
                __syncthreads();
                for (int r = (int)threadIdx.x; r < BM; r += 512) {
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) {
                        sm += ((const float*)(smem + OFF_RED))[((size_t)w * BM + r) * 2];
                        sq += ((const float*)(smem + OFF_RED))[((size_t)w * BM + r) * 2 + 1];
                    }
                    float* dst = p.rowstats_out + ((size_t)(m0 + r) * p.rs_ld + tile_n) * 2;
                    dst[0] = sm;
                    dst[1] = sq;
                }
            }
        }
        prefetch_next();
    }
}

template <int BN, int WM, int WN, bool TRANS>
static int launch_gemm16(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    supir_choose_xcd_grid(a, a.M / 128, a.N / BN, 2.0 * (double)a.M * a.K, 2.0 * (double)a.N * a.K);
    constexpr int smem = 2 * 2 * (128 + BN) * 128 + 256;   // two rings of two stages + the prefetch scratch row
    auto kern = gemm16_kernel<BN, WM, WN, TRANS>;
    static bool attr_set = false;
    if (!attr_set) {
        if (supir_note_hip_status(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) != SUPIR_OK) return SUPIR_ERR_HIP;
        attr_set = true;
    }
    SUPIR_LAUNCH(kern, dim3((a.M / 128) * (a.N / BN)), dim3(512), smem, st, a);
    return SUPIR_LAUNCH_STATUS();
}

// tiles 32 (128 x 80) and 33 (128 x 160); returns SUPIR_ERR_SHAPE when the problem is not an exact fit (the caller falls back)
bool supir_gemm16_supported(const GemmArgs& a, int tile) {
    const int bn = tile == 32 ? 80 : 160;
    if (tile != 32 && tile != 33) return false;
    if (a.M % 128 || a.N % bn || a.K % 128 || a.lda % 8) return false;
    if (a.act == 2 || a.out_mode == 1 || a.ln_slots > 32) return false;
    if (a.out_mode == 2) return a.rows_per_batch % 4 == 0 && a.ldc % 4 == 0 && !a.res && !a.rowbias && a.act == 0;
    if (a.ldc % 8 || (((size_t)a.C) & 15)) return false;
    if ((a.res && a.ldr % 4) || (a.rowbias && a.ld_rb % 4)) return false;
    return true;
}

int supir_gemm16_launch(const GemmArgs& a, hipStream_t st, int tile) {
    if (!supir_gemm16_supported(a, tile)) return SUPIR_ERR_SHAPE;
    if (a.out_mode == 2) return tile == 32 ? launch_gemm16<80, 4, 1, true>(a, st) : launch_gemm16<160, 2, 2, true>(a, st);
    return tile == 32 ? launch_gemm16<80, 4, 1, false>(a, st) : launch_gemm16<160, 2, 2, false>(a, st);
}
