// Flash attention forward, head dim 64, bf16 in / fp32 online softmax / bf16 out, for gfx950 (MI355X).
//
// Replaces xformers.ops.memory_efficient_attention / F.scaled_dot_product_attention at
// sgm/modules/attention.py:273-277,357-359 (CrossAttention / MemoryEfficientCrossAttention) and the
// ZeroCrossAttn call at SUPIR/modules/SUPIR_v0.py:146.  softmax(Q K^T / sqrt(64)) V, no mask, no dropout.
//
// Layout (chosen so no transpose kernel is ever needed):
//   Q  [B][Tq][ldq]  head h at columns h*64..h*64+63   (exactly what the to_q GEMM writes: 'b n (h d)')
//   K  [B][Tk][ldk]  same
//   Vt [B][H*64][ldvt]  V TRANSPOSED per batch (row = h*64+d, column = key), written by the to_v GEMM's
//                       transposed epilogue; ldvt >= round_up(Tk,64) and the padding is finite (zero)
//   O  [B][Tq][ldo]
// One workgroup = 4 (or 2) waves = 128 (64) query rows of one (batch, head); K / Vt tiles of 64 keys are staged through LDS
// by global_load_lds (double buffered, counted vmcnt) and shared by the 4 waves.
// MFMA 32x32x16 with swapped operands: S^T = K.Q^T puts one query row per lane (softmax needs one shfl_xor 32),
// and P feeds the PV MFMA straight from those registers: the key order inside each 16-wide k block is
// arranged (K rows read with bits 2/3 of the row index swapped) so that it is one 16-byte chunk of a Vt row: no
// cross-lane exchange, conflict-free ds_read_b128 on both operands.
#include "kernels.h"
#include <stdlib.h>


#ifdef SUPIR_ATTN_TIMELINE
// tools/probes/attn_timeline.hip only: per-wave s_memtime breakdown of the KV loop (never defined in the product build)
__device__ unsigned long long* g_atl_buf;
extern "C" void supir_atl_set(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_atl_buf), &p, sizeof(p)); }
#define ATL(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define ATL(var)
#endif

template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int S, int NW>
__global__ __launch_bounds__(64 * NW) void attn_d64_kernel(const AttnArgs p) {
    // S-deep ring of (K tile, V^T tile) pairs, 16 KB each (2 measured best: deeper rings bought nothing, the kernel is not
    // load-latency bound).  NW waves = 32*NW query rows per workgroup: 4 normally, 2 when the launch would otherwise put
    // fewer than two 4-wave workgroups on a CU (320 workgroups on 256 CUs run as two rounds at 62 % occupancy).
    __shared__ __attribute__((aligned(16))) char smem[S * 16384];
    constexpr int NT = 64 * NW, QB = 32 * NW, LPT = 512 / NT;   // threads, queries per workgroup, loads per thread per 8 KB tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    const int nqb = (p.Tq + QB - 1) / QB;
    const int id = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    const int bh = id / nqb, qb = id - bh * nqb;  // consecutive ids (same XCD) share K/V of one (b,h)
    const int b = bh / p.H, h = bh - b * p.H;

    const bf16_t* Kb = p.K + (size_t)b * p.Tk * p.ldk + h * 64;
    const bf16_t* Vb = p.Vt + ((size_t)b * p.H + h) * 64 * p.ldvt;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane -> query l31, d = 16*ks + 8*half .. +7
    int q = qb * QB + wave * 32 + l31;
    const bool q_ok = q < p.Tq;
    const int qc = q_ok ? q : p.Tq - 1;
    const bf16_t* Qp = p.Q + ((size_t)b * p.Tq + qc) * p.ldq + h * 64 + 8 * half;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Qp + 16 * ks);

    // ---- loader: slot s = j*NT+tid -> row j*(NT/8) + (tid>>3), physical chunk tid&7, logical chunk swizzled
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7);
    const int nt = (p.Tk + 63) >> 6;
    constexpr int NL = 2 * LPT;   // global->LDS instructions per thread per (K, V^T) tile pair: K rows first, then V^T rows
    auto stage_one = [&](int t, int buf, int q) {
        char* sK = smem + buf * 16384;
        char* sV = sK + 8192;
        if (q < LPT) {
            const int j = q;
            int key = t * 64 + j * (NT / 8) + lrow;
            key = key < p.Tk ? key : p.Tk - 1;
            glds16(Kb + (size_t)key * p.ldk + lchunk * 8, sK + (j * NT + wave * 64) * 16);
        } else {
            const int j = q - LPT;
            const int d = j * (NT / 8) + lrow;
            glds16(Vb + (size_t)d * p.ldvt + t * 64 + lchunk * 8, sV + (j * NT + wave * 64) * 16);
        }
    };
    auto stage = [&](int t, int buf) {
#pragma unroll
        for (int q = 0; q < NL; ++q) stage_one(t, buf, q);
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int sw = (l31 >> 1) & 7;
    const int row_off = l31 * 128;
    // K fragment rows are read PERMUTED: MFMA row i of S^T holds key swap_bits(2,3)(i).  With that, the 8 P values a lane
    // owns per 16-key block (rows 4*half + (r&3) + 8*(r>>2)) are the 8 CONSECUTIVE keys 16*kb + 8*half + 0..7, i.e. exactly
    // one 16-byte chunk of a V^T row: the PV operand is a conflict-free ds_read_b128 and no lane exchange is needed.
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int krow_off = krow * 128;
    const int ksw = (krow >> 1) & 7;

#ifdef SUPIR_ATTN_TIMELINE
    unsigned long long tl_sync = 0, tl_issue = 0, tl_qk = 0, tl_sm = 0, tl_pv = 0;
#endif
    ATL(tl_t0);
    {
        const int pre = nt < S - 1 ? nt : S - 1;
        for (int t = 0; t < pre; ++t) stage(t, t);
    }
    for (int t = 0; t < nt; ++t) {
        ATL(tl_a);
        const int buf = t % S;
        const int rem = nt - 1 - t;
        const int inflight = rem < S - 2 ? rem : S - 2;   // younger tiles allowed to stay outstanding (4 loads each)
        if constexpr (S >= 4) {
            if (inflight >= 2) attn_wait_vmcnt<4 * LPT>();
            else if (inflight == 1) attn_wait_vmcnt<2 * LPT>();
            else attn_wait_vmcnt<0>();
        } else if constexpr (S == 3) {
            if (inflight >= 1) attn_wait_vmcnt<2 * LPT>();
            else attn_wait_vmcnt<0>();
        } else {
            attn_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();   // tile t visible to all waves; all waves are done with tile t-1's buffer
        asm volatile("" ::: "memory");
        ATL(tl_b);
        // The next tile pair's global->LDS instructions are spread over the 8 MFMA groups of this iteration (4 in Q.K^T,
        // 4 in P.V): issued back to back they stall ~550 cycles on the CU's vector-memory path with the matrix pipe idle
        // (s_memtime, tools/probes/attn_timeline.hip).  The two independent accumulators of each product alternate, so
        // consecutive MFMAs never wait on each other's result.
        const bool do_stage = t + S - 1 < nt;
        const int st_t = t + S - 1, st_buf = (t + S - 1) % S;
        ATL(tl_c);
        const char* sK = smem + buf * 16384;
        const char* sV = sK + 8192;

        // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]
        f32x16 s[2];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kf][r] = 0.f;
        bf16x8 kfr[2][2];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) kfr[0][kf] = *(const bf16x8*)(sK + kf * 4096 + krow_off + (((0 + half) ^ ksw) * 16));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int kf = 0; kf < 2; ++kf)
                    kfr[(ks + 1) & 1][kf] = *(const bf16x8*)(sK + kf * 4096 + krow_off + (((2 * (ks + 1) + half) ^ ksw) * 16));
            }
            if (do_stage) {
#pragma unroll
                for (int q = (ks * NL) / 8; q < ((ks + 1) * NL) / 8; ++q) stage_one(st_t, st_buf, q);
            }
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) s[kf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks & 1][kf], qf[ks], s[kf], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef SUPIR_ATTN_TIMELINE
        asm volatile("s_nop 0" ::"v"(s[0][0]), "v"(s[1][0]));   // force the QK results before the timestamp
#endif
        ATL(tl_d);
        // lane holds query l31, keys t*64 + kf*32 + 16*(r>>3) + 8*half + (r&7)
        if ((t == nt - 1 && (p.Tk & 63)) || p.causal) {
            asm volatile("" ::: "memory");   // keep this a (wave-uniform) branch: if-converted it costs 32 v_cndmask per tile
            const int klim = p.causal ? (q < p.Tk - 1 ? q : p.Tk - 1) : p.Tk - 1;   // last visible key of this lane's query
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + kf * 32 + 16 * (r >> 3) + 8 * half + (r & 7);
                    if (key > klim) s[kf][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kf][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        // the running maximum settles after the first tiles: rescale O and l only when some row's maximum moved
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
            asm volatile("" ::: "memory");
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2e);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        m_run = m_new;
        const float mb = m_new * p.scale_log2e;
        float psum = 0.f;
        bf16x8 pf[4];  // pf[kf*2+kb]: keys kf*32 + 16*kb + 8*half + 0..7
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[kf][r] * p.scale_log2e - mb);
                psum += pv;
                pf[kf * 2 + (r >> 3)][r & 7] = (bf16_t)pv;
            }
        l_run += psum;

#ifdef SUPIR_ATTN_TIMELINE
        asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[1][15]), "v"(pf[3]));
#endif
        ATL(tl_e);
        // ---- O^T[d][q] += sum_key Vt[d][key] P[q][key]
        // 16 keys 16*kb4 .. +15 = logical chunks 2*kb4, 2*kb4+1 of the V^T row; lane half h takes chunk 2*kb4+h
        bf16x8 vfr[2][2];
#pragma unroll
        for (int df = 0; df < 2; ++df) vfr[0][df] = *(const bf16x8*)(sV + df * 4096 + row_off + (((0 + half) ^ sw) * 16));
#pragma unroll
        for (int kb4 = 0; kb4 < 4; ++kb4) {
            if (kb4 < 3) {
#pragma unroll
                for (int df = 0; df < 2; ++df)
                    vfr[(kb4 + 1) & 1][df] = *(const bf16x8*)(sV + df * 4096 + row_off + (((2 * (kb4 + 1) + half) ^ sw) * 16));
            }
            if (do_stage) {
#pragma unroll
                for (int q = ((kb4 + 4) * NL) / 8; q < ((kb4 + 5) * NL) / 8; ++q) stage_one(st_t, st_buf, q);
            }
#pragma unroll
            for (int df = 0; df < 2; ++df) o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[kb4 & 1][df], pf[kb4], o[df], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SUPIR_ATTN_TIMELINE
        asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[1][15]));
        {
            ATL(tl_f);
            tl_sync += tl_b - tl_a;
            tl_issue += tl_c - tl_b;
            tl_qk += tl_d - tl_c;
            tl_sm += tl_e - tl_d;
            tl_pv += tl_f - tl_e;
        }
#endif
    }
    ATL(tl_loop1);

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        bf16_t* Op = p.O + ((size_t)b * p.Tq + q) * p.ldo + h * 64;
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                u16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[df][rg * 4 + e] * inv);
                *(u16x4*)(Op + df * 32 + 8 * rg + 4 * half) = ov;
            }
    }
#ifdef SUPIR_ATTN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g_atl_buf && lane == 0) {
        ATL(tl_end);
        unsigned long long* ob = g_atl_buf + ((size_t)blockIdx.x * NW + wave) * 8;
        ob[0] = tl_t0;
        ob[1] = tl_sync;
        ob[2] = tl_issue;
        ob[3] = tl_qk;
        ob[4] = tl_sm;
        ob[5] = tl_pv;
        ob[6] = tl_end - tl_loop1;
        ob[7] = tl_end - tl_t0;
    }
#endif
}

int supir_attn_launch(const AttnArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tk <= 0) return SUPIR_ERR_ARG;
    if ((a.ldq | a.ldk | a.ldvt) % 8 != 0 || a.ldo % 4 != 0) return SUPIR_ERR_SHAPE;
    if (a.ldvt < ((a.Tk + 63) / 64) * 64) return SUPIR_ERR_SHAPE;
    static int force_nw = -1;   // SUPIR_ATTN_NW=2|4 pins the workgroup size (tools/attn_probe.py)
    if (force_nw < 0) {
        const char* e = getenv("SUPIR_ATTN_NW");
        force_nw = e ? atoi(e) : 0;
    }
    const int blocks4 = ((a.Tq + 127) / 128) * a.H * a.B;
    const bool small = force_nw == 2;   // measured: the 2-wave form is never faster (the kernel is VALU-bound, not occupancy-bound)
    if (small) SUPIR_LAUNCH((attn_d64_kernel<2, 2>), dim3(((a.Tq + 63) / 64) * a.H * a.B), dim3(128), 0, st, a);
    else SUPIR_LAUNCH((attn_d64_kernel<2, 4>), dim3(blocks4), dim3(256), 0, st, a);
    return SUPIR_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// Row softmax for the VAE mid-block single-head attention (head dim 512, reference:
// sgm/modules/diffusionmodules/model.py:177-192, 228-256). With 288 GB of HBM the [T][T] score matrix is simply
// materialised in fp32 by the GEMM kernel (out_mode 1); this kernel turns each fp32 row into bf16 probabilities.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, bf16_t* __restrict__ P,
                                                            int T, int Tpad, long lds_, long ldp, float scale) {
    // columns [0, T) are the keys; [T, Tpad) is K-padding of the following P.V GEMM and is written as exact zeros
    __shared__ float red[8];
    const float* s = S + (size_t)blockIdx.x * lds_;
    bf16_t* pr = P + (size_t)blockIdx.x * ldp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid * 4; i < T; i += 1024) {
        const f32x4 v = *(const f32x4*)(s + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i + e < T) mx = fmaxf(mx, v[e]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid * 4; i < T; i += 1024) {
        const f32x4 v = *(const f32x4*)(s + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i + e < T) sum += __expf((v[e] - mx) * scale);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int i = tid * 4; i < Tpad; i += 1024) {
        const f32x4 v = *(const f32x4*)(s + i);
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (i + e < T) ? f2bf(__expf((v[e] - mx) * scale) * inv) : (u16)0;
        *(u16x4*)(pr + i) = o;
    }
}

int supir_softmax_rows_launch(const float* S, bf16_t* P, int rows, int T, int Tpad, long lds_, long ldp, float scale,
                              hipStream_t st) {
    if (rows <= 0 || T <= 0 || Tpad < T || Tpad % 4 != 0 || lds_ % 4 != 0 || ldp % 4 != 0 || lds_ < Tpad || ldp < Tpad)
        return SUPIR_ERR_SHAPE;
    SUPIR_LAUNCH(softmax_rows_kernel, dim3(rows), dim3(256), 0, st, S, P, T, Tpad, lds_, ldp, scale);
    return SUPIR_LAUNCH_STATUS();
}
