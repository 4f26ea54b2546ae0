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
// One workgroup = 4 waves = 128 query rows of one (batch, head); K / Vt tiles of 64 keys are staged through LDS
// by global_load_lds (double buffered, counted vmcnt) and shared by the 4 waves.
// MFMA 32x32x16 with swapped operands: S^T = K.Q^T puts one query row per lane (softmax needs one shfl_xor 32),
// and P feeds the PV MFMA straight from those registers: the key order inside each 16-wide k block is
// arranged (K rows read with bits 2/3 of the row index swapped) so that it is one 16-byte chunk of a Vt row: no
// cross-lane exchange, conflict-free ds_read_b128 on both operands.
#include "kernels.h"
#include <stdlib.h>


template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int S>
__global__ __launch_bounds__(256) void attn_d64_kernel(const AttnArgs p) {
    // S-deep ring of (K tile, V^T tile) pairs, 16 KB each.  A (batch, head, 128-query) workgroup is alone or nearly alone
    // on its CU (320-640 workgroups per launch), so the latency of the next tiles has to be hidden by queue depth.
    __shared__ __attribute__((aligned(16))) char smem[S * 16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    const int nqb = (p.Tq + 127) >> 7;
    const int id = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    const int bh = id / nqb, qb = id - bh * nqb;  // consecutive ids (same XCD) share K/V of one (b,h)
    const int b = bh / p.H, h = bh - b * p.H;

    const bf16_t* Kb = p.K + (size_t)b * p.Tk * p.ldk + h * 64;
    const bf16_t* Vb = p.Vt + ((size_t)b * p.H + h) * 64 * p.ldvt;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane -> query l31, d = 16*ks + 8*half .. +7
    int q = qb * 128 + wave * 32 + l31;
    const bool q_ok = q < p.Tq;
    const int qc = q_ok ? q : p.Tq - 1;
    const bf16_t* Qp = p.Q + ((size_t)b * p.Tq + qc) * p.ldq + h * 64 + 8 * half;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Qp + 16 * ks);

    // ---- loader: slot s = j*256+tid -> row j*32 + (tid>>3), physical chunk tid&7, logical chunk swizzled
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7);
    const int nt = (p.Tk + 63) >> 6;
    auto stage = [&](int t, int buf) {
        char* sK = smem + buf * 16384;
        char* sV = sK + 8192;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int key = t * 64 + j * 32 + lrow;
            key = key < p.Tk ? key : p.Tk - 1;
            glds16(Kb + (size_t)key * p.ldk + lchunk * 8, sK + (j * 256 + wave * 64) * 16);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = j * 32 + lrow;
            glds16(Vb + (size_t)d * p.ldvt + t * 64 + lchunk * 8, sV + (j * 256 + wave * 64) * 16);
        }
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

    {
        const int pre = nt < S - 1 ? nt : S - 1;
        for (int t = 0; t < pre; ++t) stage(t, t);
    }
    for (int t = 0; t < nt; ++t) {
        const int buf = t % S;
        const int rem = nt - 1 - t;
        const int inflight = rem < S - 2 ? rem : S - 2;   // younger tiles allowed to stay outstanding (4 loads each)
        if constexpr (S >= 4) {
            if (inflight >= 2) attn_wait_vmcnt<8>();
            else if (inflight == 1) attn_wait_vmcnt<4>();
            else attn_wait_vmcnt<0>();
        } else if constexpr (S == 3) {
            if (inflight >= 1) attn_wait_vmcnt<4>();
            else attn_wait_vmcnt<0>();
        } else {
            attn_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();   // tile t visible to all waves; all waves are done with tile t-1's buffer
        asm volatile("" ::: "memory");
        if (t + S - 1 < nt) stage(t + S - 1, (t + S - 1) % S);
        const char* sK = smem + buf * 16384;
        const char* sV = sK + 8192;

        // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]
        f32x16 s[2];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kf][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kfrag = *(const bf16x8*)(sK + kf * 4096 + krow_off + (((2 * ks + half) ^ ksw) * 16));
                s[kf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag, qf[ks], s[kf], 0, 0, 0);
            }
        }
        // lane holds query l31, keys t*64 + kf*32 + 16*(r>>3) + 8*half + (r&7)
        if (t == nt - 1 && (p.Tk & 63)) {
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + kf * 32 + 16 * (r >> 3) + 8 * half + (r & 7);
                    if (key >= p.Tk) s[kf][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kf][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2e);
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
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

        // ---- O^T[d][q] += sum_key Vt[d][key] P[q][key]
#pragma unroll
        for (int df = 0; df < 2; ++df) {
            const char* vrow = sV + df * 4096 + row_off;
#pragma unroll
            for (int kb4 = 0; kb4 < 4; ++kb4) {
                // 16 keys 16*kb4 .. +15 = logical chunks 2*kb4, 2*kb4+1 of the V^T row; lane half h takes chunk 2*kb4+h
                const bf16x8 vf = *(const bf16x8*)(vrow + (((2 * kb4 + half) ^ sw) * 16));
                o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb4], o[df], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

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
}

int supir_attn_launch(const AttnArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tk <= 0) return SUPIR_ERR_ARG;
    if ((a.ldq | a.ldk | a.ldvt) % 8 != 0 || a.ldo % 4 != 0) return SUPIR_ERR_SHAPE;
    if (a.ldvt < ((a.Tk + 63) / 64) * 64) return SUPIR_ERR_SHAPE;
    const int nqb = (a.Tq + 127) / 128;
    static int ring = -1;
    if (ring < 0) {
        const char* e = getenv("SUPIR_ATTN_RING");
        ring = e ? atoi(e) : 3;
    }
    if (ring == 2) SUPIR_LAUNCH(attn_d64_kernel<2>, dim3(nqb * a.H * a.B), dim3(256), 0, st, a);
    else if (ring == 4) SUPIR_LAUNCH(attn_d64_kernel<4>, dim3(nqb * a.H * a.B), dim3(256), 0, st, a);
    else SUPIR_LAUNCH(attn_d64_kernel<3>, dim3(nqb * a.H * a.B), dim3(256), 0, st, a);
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
