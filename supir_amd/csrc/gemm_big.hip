// GEGLU projection GEMM on a 256 x 320 tile with 64 x 160 per wave, for gfx950 (MI355X).
//
// The projection GEGLU.proj (sgm/modules/attention.py:84-91: Linear(dim, 2*inner) -> value * gelu(gate)) of the 1280-wide
// transformer blocks is the single largest item of a 1024^2 step: (M, N, K) = (2048, 10240, 1280), 53.7 GFLOP, 90 launches.
// On the 256 x 160 tile of gemm16.hip it runs at 0.35 of the MFMA peak and what bounds it is LDS bandwidth, not the matrix
// pipe: a wave tile of TM x TN reads (TM + TN) x 32 B of fragments per 16-wide K slice for TM x TN / 32 MFMA cycles, and four
// SIMDs share one 128 B/clk LDS, so TM TN / (TM + TN) must exceed 32 before the matrix pipe can be the limit (plus ~30 % for the
// global->LDS fill, which uses the same port).  Eight waves of 32 x 160 (tile 34) sit at 26.7: 976 LDS cycles per 640 MFMA
// cycles.  Here: tile 256 x 320 = (2048 / 256) x (10240 / 320) = exactly 256 workgroups, eight waves as 4 (M) x 2 (N), each
// 64 tokens x 160 channels = 2 x 5 blocks of v_mfma_f32_32x32x16_bf16 (160 accumulator registers, two waves per SIMD):
// intensity 45.7, 592 LDS cycles per 640 MFMA cycles.  Both operands go through a 2-deep global_load_lds ring (72 KB per
// 64-wide K step), swizzled as in gemm.hip; fragment reads run two MFMA pairs ahead of their use.
// (Measured and rejected: loading the activation fragments global -> VGPR directly, bypassing LDS.  A fragment wants one token
// row per lane, so every wave-level load touches 32 cache lines for 32 bytes each and each line is fetched four times per K
// step: 72.9 us against 57.0 us for tile 34 -- the texture-address path and the L2 saturate long before the matrix pipe.)
// Epilogue = the GEGLU epilogue of gemm16.hip (LayerNorm fold, bias, value * gelu(gate) with the 16-row value / gate
// interleave of weights.interleave_geglu(block=16), bf16 rows staged through LDS for 16-byte stores).
#include "kernels.h"
#include <type_traits>

#ifdef SUPIR_G16_TIMELINE
// tools/probes/g16_timeline.py only (never defined in the product build): per-wave s_memtime stamps of the kernel's phases
__device__ unsigned long long* big_tl_buf;
extern "C" void supir_big_tl_set(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(big_tl_buf), &p, sizeof(p)); }
#define BIG_TL(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define BIG_TL(var)
#endif

namespace {

template <int N>
__device__ __forceinline__ void big_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// gelu_f of common.h on two values at once: the epilogue is a pure-VALU phase (no MFMA beside it), where v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 retire two lanes' worth per issue slot.  Same formula and constants.
__device__ __forceinline__ f32x2 gelu2(f32x2 g) {
    const f32x2 x = g * 0.70710678118654752f;
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 d = ax * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 pl = t * 1.061405429f + -1.453152027f;
    pl = pl * t + 1.421413741f;
    pl = pl * t + -0.284496736f;
    pl = pl * t + 0.254829592f;
    const f32x2 e = {__expf(-ax[0] * ax[0]), __expf(-ax[1] * ax[1])};
    const f32x2 y = 1.0f - pl * t * e;
    const f32x2 ys = {copysignf(y[0], x[0]), copysignf(y[1], x[1])};
    return 0.5f * g * (1.0f + ys);
}

// gelu_fast_f of common.h on two values at once (round 4): x * sigmoid(x * poly(x^2)), 5 packed + 3 x 2 scalar instructions per pair
// instead of 12 packed + 4 x 2 scalar
__device__ __forceinline__ f32x2 gelu2_fast(f32x2 g) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(g[0], -7.0f, 7.0f), __builtin_amdgcn_fmed3f(g[1], -7.0f, 7.0f)};
    const f32x2 u = xc * xc;
    f32x2 pl = u * 1.01426305e-03f + -1.06775724e-01f;
    pl = pl * u + -2.30112133f;
    const f32x2 a = xc * pl;
    const f32x2 d = {1.0f + __builtin_amdgcn_exp2f(a[0]), 1.0f + __builtin_amdgcn_exp2f(a[1])};
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return g * r;
}

constexpr int BM = 256, BN = 320, S = 2;
constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;   // one K step (64 wide): A tile, then W tile
constexpr int A_LOADS = BM * 8 / 512, W_LOADS = BN * 8 / 512;   // global->LDS instructions per thread per K step (4 + 5)
constexpr int LOADS = A_LOADS + W_LOADS;
constexpr int C_RS = 80 * 2 + 16;                  // staged output row: 80 bf16 + 16 B pad

// NP = 2: two problems of identical shape in one grid, problem q on XCDs [4 q, 4 q + 4) (see gemm16.hip)
// FASTGELU: gelu2_fast (SUPIR_ACT_GEGLU, the fitted GELU) or gelu2 (SUPIR_ACT_GEGLU_ERF, the reference's erf) in the epilogue: the caller's
// activation code selects the instantiation (GemmArgs::fast_gelu)
template <int NP, bool FASTGELU = true>
__global__ __launch_bounds__(512, 2) void geglu_big_kernel(const GemmArgsN<NP> pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NX = 8 / NP;
    const int prob = NP == 1 ? 0 : (int)(blockIdx.x & 7) / NX;
    const GemmArgs& p = pp.p[prob];
    const int vxcd = (int)blockIdx.x & (NX - 1), vidx = (int)blockIdx.x >> 3;
    BIG_TL(tl_start);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int tiles_m = p.M / BM, tiles_n = p.N / BN;
    int tile_m, tile_n;   // workgroup -> tile: see gemm.hip (XCD-aware partition chosen on the host)
    if (p.gm > 0) {
        const int xcd = vxcd, idx = vidx;
        const int rm = tiles_m / p.gm, rn = tiles_n / p.gn;
        const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
        int lm, ln;
        if (p.order == 0) { ln = idx / rm; lm = idx - ln * rm; }
        else { lm = idx / rn; ln = idx - lm * rn; }
        tile_m = xm * rm + lm;
        tile_n = xn * rn + ln;
    } else {
        const int id = NP == 1 ? xcd_remap(blockIdx.x, tiles_m * tiles_n) : vxcd * (tiles_m * tiles_n / NX) + vidx;
        if (p.order == 0) { tile_n = id / tiles_m; tile_m = id - tile_n * tiles_m; }
        else { tile_m = id / tiles_n; tile_n = id - tile_m * tiles_n; }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- loader: slot s = j*512 + tid -> tile row j*64 + (tid>>3), physical chunk tid&7 holds logical chunk ^ ((row>>1)&7).
    // Source = wave-uniform base (SGPR pair, advanced per K step) + a per-lane byte offset that never changes.
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7);
    const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda);
    const char* w_base = (const char*)(p.Wt + (size_t)n0 * p.K);
    unsigned a_off[A_LOADS], w_off[W_LOADS];
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) a_off[j] = (unsigned)((j * 64 + lrow) * p.lda * 2 + lchunk * 16);
#pragma unroll
    for (int j = 0; j < W_LOADS; ++j) w_off[j] = (unsigned)((j * 64 + lrow) * p.K * 2 + lchunk * 16);
    auto stage_one = [&](int kt, int soff, int q) {   // q < A_LOADS: A rows, else W rows, of K step kt into ring slot soff
        if (q < A_LOADS) glds16(a_base + (size_t)kt * 128 + a_off[q], smem + soff + (q * 512 + wave * 64) * 16);
        else glds16(w_base + (size_t)kt * 128 + w_off[q - A_LOADS], smem + soff + A_BYTES + ((q - A_LOADS) * 512 + wave * 64) * 16);
    };
    // ---- fragment addresses: A row wm*64 + i*32 + l31, W row wn*160 + j*32 + l31; K slice ks = logical chunk 4*half + ks of the
    // 128-byte row for both operands (any permutation of k inside the step is fine as long as both agree)
    const int sw = (l31 >> 1) & 7;
    int a_frag[4], w_frag[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int c = ((4 * half + ks) ^ sw) * 16;
        a_frag[ks] = (wm * 64 + l31) * 128 + c;
        w_frag[ks] = A_BYTES + (wn * 160 + l31) * 128 + c;
    }

    // epilogue vectors of this tile's 320 columns and the LayerNorm statistics of this lane's two tokens: fetched now, used
    // after the main loop
    float pre_bias = 0.f, pre_cs = 0.f;
    if (tid < BN) {
        if (p.bias) pre_bias = p.bias[n0 + tid];
        if (p.ln_stats) pre_cs = p.ln_colsum[n0 + tid];
    }
    float ln_mean[2] = {0.f, 0.f}, ln_rstd[2] = {1.f, 1.f};
    if (p.ln_stats) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + i * 32 + l31;
            if (p.ln_slots == 0) {   // finalised statistics: [M][2] = (mean, rstd)
                ln_mean[i] = p.ln_stats[(size_t)m * 2];
                ln_rstd[i] = p.ln_stats[(size_t)m * 2 + 1];
                continue;
            }
            const float* st = p.ln_stats + (size_t)m * p.ln_ld * 2;   // partial (sum, sum of squares) per producer tile column
            float sm = 0.f, sq = 0.f;
            for (int sl = half; sl < p.ln_slots; sl += 2) {            // the two lanes of a token split the slots
                sm += st[2 * sl];
                sq += st[2 * sl + 1];
            }
            sm += __shfl_xor(sm, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float inv = 1.0f / (float)p.K;
            const float mean = sm * inv;
            float var = sq * inv - mean * mean;
            var = var > 0.f ? var : 0.f;
            ln_mean[i] = mean;
            ln_rstd[i] = rsqrtf(var + p.ln_eps);
        }
    }

    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K >> 6;
#pragma unroll
    for (int q = 0; q < LOADS; ++q) stage_one(0, 0, q);

    BIG_TL(tl_loop0);
    int boff = 0;   // ring slot of the tile being read
    // One K step.  The loads of step kt+1 are issued after the barrier of step kt into the other slot (free: every wave has
    // finished step kt-1), one per tick over the first nine ticks, and awaited at the top of step kt+1.  The step is 20 ticks q = 5 ks + j:
    // tick q runs the two MFMAs of W block (ks, j) and issues the ds_read of block q + 2 (three fragment buffers in rotation,
    // the activation fragments of slice ks + 1 ride along at j = 2), so a fragment read has two ticks = 128 MFMA cycles of lead.
    auto kstep = [&](const int kt, auto more_c) {
        constexpr bool MORE = decltype(more_c)::value;
        big_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int soff = boff == 0 ? STAGE : 0;
        const char* sT = smem + boff;
        bf16x8 wf[3], af[2][2];
        auto read_w = [&](int q) { return *(const bf16x8*)(sT + w_frag[q / 5] + (q % 5) * 32 * 128); };
        auto read_a = [&](int ks, int slot) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[slot][i] = *(const bf16x8*)(sT + a_frag[ks] + i * 32 * 128);
        };
        read_a(0, 0);
        wf[0] = read_w(0);
        wf[1] = read_w(1);
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            const int ks = q / 5, j = q % 5;
            if (q + 2 < 20) wf[(q + 2) % 3] = read_w(q + 2);
            if (j == 2 && ks < 3) read_a(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = SUPIR_MFMA_32x32x16(wf[q % 3], af[ks & 1][i], acc[i][j], 0, 0, 0);
            if constexpr (MORE) {
                // 9 loads on ticks 0 .. 8: as early as the slot is free (they are awaited at the top of the next step, so the last
                // one needs its whole latency inside this step), but not back to back (that stalls on the vector-memory path)
                if (q < LOADS) stage_one(kt + 1, soff, q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        boff = soff;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    for (int kt = 0; kt < nk - 1; ++kt) kstep(kt, T_{});
    kstep(nk - 1, F_{});

    // ------------------------------------------------------------------ epilogue
#ifdef SUPIR_G16_TIMELINE
    asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[1][4][15]));
#endif
    BIG_TL(tl_loop1);
    __syncthreads();   // the ring is reused: every wave is done with its last fragment reads
    float* s_bias = (float*)smem;   // [BN] bias, [BN] LayerNorm column sums, then one staging block per wave
    float* s_cs = s_bias + BN;
    if (tid < BN) {
        s_bias[tid] = pre_bias;
        s_cs[tid] = pre_cs;
    }
    __syncthreads();
    char* c_stage = smem + 4096 + wave * (32 * C_RS);
    // D[channel][token]: lane owns token l31 and, per 32-row block of W' (= 16 value rows, then their 16 gate rows), the value
    // channels 4*half + 8*rg + 0..3 in registers 4*rg + e and their gates in registers 8 + 4*rg + e
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float mu = ln_mean[i], rs = ln_rstd[i];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                const int nl = wn * 160 + j * 32 + 8 * rg + 4 * half;   // value rows nl.., gate rows nl + 16..
                const f32x4 bv = *(const f32x4*)(s_bias + nl), bg = *(const f32x4*)(s_bias + nl + 16);
                const f32x4 cv = *(const f32x4*)(s_cs + nl), cg = *(const f32x4*)(s_cs + nl + 16);
                f32x2 r[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x2 av = {acc[i][j][4 * rg + 2 * e], acc[i][j][4 * rg + 2 * e + 1]};
                    const f32x2 ag = {acc[i][j][8 + 4 * rg + 2 * e], acc[i][j][8 + 4 * rg + 2 * e + 1]};
                    const f32x2 cv2 = {cv[2 * e], cv[2 * e + 1]}, cg2 = {cg[2 * e], cg[2 * e + 1]};
                    const f32x2 bv2 = {bv[2 * e], bv[2 * e + 1]}, bg2 = {bg[2 * e], bg[2 * e + 1]};
                    const f32x2 v = rs * (av - mu * cv2) + bv2;
                    const f32x2 g = rs * (ag - mu * cg2) + bg2;
                    r[e] = v * (FASTGELU ? gelu2_fast(g) : gelu2(g));
                }
                const u32x2 o = {f2bf_pk(r[0][0], r[0][1]), f2bf_pk(r[1][0], r[1][1])};
                *(u32x2*)(c_stage + l31 * C_RS + (j * 16 + 8 * rg + 4 * half) * 2) = o;
            }
        }
        // the wave's 32 tokens x 80 output channels, row-major in LDS: 10 lanes x 16 B cover one row
        const int row_base = m0 + wm * 64 + i * 32, col_base = (n0 >> 1) + wn * 80;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
            const int row = rr * 6 + lane / 10, ch = lane % 10;
            if (lane < 60 && row < 32) {
                const f32x4 piece = *(const f32x4*)(c_stage + row * C_RS + ch * 16);
                supir_store16((bf16_t*)p.C + (size_t)(row_base + row) * p.ldc + col_base + ch * 8, piece);
            }
        }
    }
#ifdef SUPIR_G16_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (big_tl_buf && lane == 0) {
        const unsigned long long tl_end = __builtin_amdgcn_s_memtime();
        unsigned long long* o = big_tl_buf + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = tl_start;
        o[1] = tl_loop0 - tl_start;
        o[2] = tl_loop1 - tl_loop0;
        o[3] = 0;
        o[4] = tl_end - tl_loop1;
        o[5] = tl_end - tl_start;
        o[6] = tl_end;
    }
#endif
    // next-weight prefetch (supir_set_next_prefetch): see gemm.hip
    if (p.pf_lines) {
        const unsigned total_waves = gridDim.x / NP * 8, gw = (unsigned)(vidx * NX + vxcd) * 8 + wave;   // per problem
        const unsigned n_instr = (p.pf_lines + 63) >> 6;
        for (unsigned i = gw; i < n_instr; i += total_waves) {
            unsigned line = i * 64 + lane;
            line = line < p.pf_lines ? line : p.pf_lines - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.pf_ptr + (size_t)line * 128),
                                             (__attribute__((address_space(3))) void*)(smem + S * STAGE), 4, 0, 0);
        }
    }
}

}  // namespace

// tile 37: exact fits of the 256 x 320 tile, GEGLU epilogue only (weights interleaved per 16 rows)
bool supir_gemm_big_supported(const GemmArgs& a) {
    if (a.act != 2 || a.out_mode != 0 || a.res || a.rowbias || a.rowstats_out || a.alpha != 1.0f) return false;
    if (a.M % BM || a.N % BN || a.K % 64 || (a.K >> 6) < 2) return false;
    if (a.lda % 8 || a.ldc % 8 || (((size_t)a.A) & 15) || (((size_t)a.C) & 15) || (((size_t)a.Wt) & 15)) return false;
    if (a.ln_stats && (a.ln_slots < 0 || a.ln_slots > 64)) return false;
    return true;
}

template <int NP>
static int launch_big(const GemmArgs* a_in, hipStream_t st) {
    GemmArgsN<NP> pp;
    for (int q = 0; q < NP; ++q) pp.p[q] = a_in[q];
    GemmArgs& a = pp.p[0];
    const int tiles = (a.M / BM) * (a.N / BN);
    if (NP > 1 && tiles % (8 / NP)) return SUPIR_ERR_SHAPE;
    supir_choose_xcd_grid(a, a.M / BM, a.N / BN, 2.0 * (double)a.M * a.K, 2.0 * (double)a.N * a.K, 1, 1, 8 / NP);
    for (int q = 1; q < NP; ++q) {
        pp.p[q].gm = a.gm;
        pp.p[q].gn = a.gn;
        pp.p[q].order = a.order;
    }
    constexpr int smem = S * STAGE + 256;   // the W ring + the prefetch scratch row; the epilogue reuses the ring
    static_assert(4096 + 8 * 32 * C_RS <= S * STAGE && 2 * BN * 4 <= 4096, "epilogue scratch must fit the ring");
    static bool attr_set = false;
    if (!attr_set) {
        if (supir_note_hip_status(hipFuncSetAttribute((const void*)geglu_big_kernel<NP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) != SUPIR_OK)
            return SUPIR_ERR_HIP;
        if (supir_note_hip_status(hipFuncSetAttribute((const void*)geglu_big_kernel<NP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) != SUPIR_OK)
            return SUPIR_ERR_HIP;
        attr_set = true;
    }
    if (!a.fast_gelu) {   // SUPIR_ACT_GEGLU_ERF: the reference's erf
        SUPIR_LAUNCH((geglu_big_kernel<NP, false>), dim3(NP * tiles), dim3(512), smem, st, pp);
    } else {
        SUPIR_LAUNCH((geglu_big_kernel<NP, true>), dim3(NP * tiles), dim3(512), smem, st, pp);
    }
    return SUPIR_LAUNCH_STATUS();
}

int supir_gemm_big_launch(const GemmArgs& a_in, hipStream_t st) {
    if (!supir_gemm_big_supported(a_in)) return SUPIR_ERR_SHAPE;
    return launch_big<1>(&a_in, st);
}

int supir_gemm_big_launch_n(const GemmArgs* a, int n, hipStream_t st) {
    if (n == 1) return supir_gemm_big_launch(a[0], st);
    if (n != 2 || !supir_gemm_big_supported(a[0]) || !supir_gemm_big_supported(a[1])) return SUPIR_ERR_SHAPE;
    const GemmArgs &x = a[0], &y = a[1];
    if (x.M != y.M || x.N != y.N || x.K != y.K) return SUPIR_ERR_SHAPE;   // everything else is read per problem
    return launch_big<2>(a, st);
}
