// Fused "to_q projection + text cross-attention" for head dim 64 and a SHORT key axis (Tk <= 128: the 77 text tokens), gfx950.
//
// Replaces, in one launch, the pair the reference runs for every cross-attention of a BasicTransformerBlock
// (sgm/modules/attention.py:241-249 to_q, :273-277 / :357-359 the attention itself; K / V of the text context are projected once per
// prompt and cached by the host mirror):
//     q = LayerNorm(x) . Wq^T          (2048, 1280, 1280) GEMM: 13-20 us
//     o = softmax(q k^T / 8) v         104 launches per 1024^2 step of ~9-12 us each that are launch + one cold round trip: two 64-key tiles
// A workgroup = 4 waves = 128 token rows of ONE head: it computes its own 128 x 64 slice of q (K loop over C through a 3-deep
// global_load_lds ring, v_mfma_f32_32x32x16 with swapped operands so that D[channel][token] puts one token per lane), keeps it in
// registers as the B operand of S^T = K . Q^T -- the accumulator's channel order inside a 16-wide k block is
// [0-3, 8-11 | 4-7, 12-15] over the two lane halves, so the K fragments are read with the same permutation (two 8-byte LDS reads
// instead of one 16-byte read): no cross-lane exchange, no q round trip through HBM -- and runs the attention over the <= 2 key tiles,
// which arrive in the ring buffers the last two K steps no longer need.  LayerNorm fold as in supir_gemm_bf16_ln
// (q = rstd (x . W'^T - mean colsum) + b'), softmax scale folded into q.  Grid = (T / 128) x H x B workgroups of 72 KB LDS (two per CU);
// the heads of one row block are neighbours on one XCD (its L2 keeps the 128 x C token tile they all read).
#include "kernels.h"

namespace {

template <int N>
__device__ __forceinline__ void xq_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float xq_xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

typedef uint32_t xq_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t xq_u32x4 __attribute__((ext_vector_type(4)));

constexpr int XQ_STAGE = 24576;   // one K step: token tile 128 x 64 (16 KB), then weight tile 64 x 64 (8 KB)
constexpr int XQ_S = 3;
constexpr int XQ_LOADS = 6;       // global->LDS instructions per wave and K step (4 token chunks + 2 weight chunks of 8 rows)
constexpr int XQ_KV_LOADS = 4;    // per wave and key tile (K 64 x 64 + V^T 64 x 64 = 16 KB)

__global__ __launch_bounds__(256, 2) void xattn_q_kernel(const XattnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[XQ_S * XQ_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int nqb = p.T >> 7;
    int h, rq;
    if (p.gm > 0) {
        // 2-D XCD grid (round 5): block b runs on XCD b % 8 with a private L2; XCD (xm, xn) owns the token blocks of chunk xm and the heads
        // of chunk xn, so it pulls (blocks / gm) token tiles and (H / gn) heads' weight rows instead of 2 token tiles and ALL of W_q
        const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
        const int hpx = p.H / p.gn, tpx = (nqb * p.B) / p.gm;
        const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
        const int lt = idx / hpx, lh = idx - lt * hpx;   // heads fastest inside the XCD's region
        h = xn * hpx + lh;
        rq = xm * tpx + lt;
    } else {
        const int id = xcd_remap(blockIdx.x, nqb * p.H * p.B);
        h = id % p.H;                                 // heads fastest
        rq = id / p.H;
    }
    const int qb = rq % nqb, b = rq / nqb;
    const size_t m0 = (size_t)b * p.T + (size_t)qb * 128;

    // ---- GEMM loader: wave w stages the 8-row chunks w, w + 4, ... of the token tile and of the head's weight rows (same scheme and
    // source-side XOR swizzle as gemm16.hip: physical 16-byte chunk c of a 128-byte row holds logical chunk c ^ ((row >> 1) & 7))
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const bf16_t* x_src = p.X + (m0 + wave * 8 + lrow) * p.ldx + lchunk * 8;
    const bf16_t* w_src = p.Wq + ((size_t)h * 64 + wave * 8 + lrow) * p.C + lchunk * 8;
    const size_t x_q = (size_t)32 * p.ldx, w_q = (size_t)32 * p.C;
    auto stage = [&](int buf, int kt) {
        char* s = smem + buf * XQ_STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(x_src + q * x_q + kt * 64, s + (wave + 4 * q) * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(w_src + q * w_q + kt * 64, s + 16384 + (wave + 4 * q) * 1024);
    };
    // ---- key-tile loader (attention.hip's): tile t -> a ring buffer: K rows at +0, V^T rows at +8192, 128 bytes per row, swizzled
    const int nt = (p.Tk + 63) >> 6;
    const char* Kb = (const char*)(p.K + (size_t)b * p.Tk * p.ldk + h * 64);
    const char* Vb = (const char*)(p.Vt + ((size_t)b * p.H + h) * 64 * p.ldvt);
    const int a_lrow = tid >> 3, a_lchunk = (tid & 7) ^ ((tid >> 4) & 7);
    auto stage_kv = [&](int buf, int t) {
        char* sK = smem + buf * XQ_STAGE;
        char* sV = sK + 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i * 32 + a_lrow;
            int kr = t * 64 + r;
            kr = kr < p.Tk ? kr : p.Tk - 1;      // rows past the last key: any finite row (masked below)
            glds16(Kb + (size_t)kr * p.ldk * 2 + a_lchunk * 16, sK + (i * 256 + wave * 64) * 16);
            glds16(Vb + (size_t)r * p.ldvt * 2 + (size_t)t * 128 + a_lchunk * 16, sV + (i * 256 + wave * 64) * 16);
        }
    };

    // ---- per-token LayerNorm statistics and per-channel fold vectors, fetched under the first tiles' latency
    const size_t m = m0 + wave * 32 + l31;
    float mean = 0.f, rstd = 1.f;
    if (p.ln_stats) {
        if (p.ln_slots == 0) {
            mean = p.ln_stats[m * 2];
            rstd = p.ln_stats[m * 2 + 1];
        } else {
            const float* st = p.ln_stats + m * p.ln_ld * 2;
            float sm = 0.f, sq = 0.f;
            for (int s = 0; s < p.ln_slots; ++s) {
                sm += st[2 * s];
                sq += st[2 * s + 1];
            }
            const float inv = 1.0f / (float)p.C;
            mean = sm * inv;
            float var = sq * inv - mean * mean;
            var = var > 0.f ? var : 0.f;
            rstd = rsqrtf(var + p.ln_eps);
        }
    }
    // this lane's channels of accumulator register r of channel block cb: h*64 + 32 cb + 8 (r / 4) + 4 half + (r % 4)
    f32x4 cs4[2][4], b4[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = h * 64 + 32 * cb + 8 * g + 4 * half;
            cs4[cb][g] = p.ln_stats ? *(const f32x4*)(p.ln_colsum + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            b4[cb][g] = p.bias ? *(const f32x4*)(p.bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // ---- q slice: D[channel][token] = W_h . X^T
    const int nk = p.C >> 6;     // >= 3 (dispatcher)
    stage(0, 0);
    stage(1, 1);
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    const int sw = (l31 >> 1) & 7;
    const int x_row_off = (wave * 32 + l31) * 128;
    const int w_row_off = 16384 + l31 * 128;
    int buf = 0;
    auto kstep = [&]() {
        const char* sT = smem + buf * XQ_STAGE;
        bf16x8 xf[4], wf[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int coff = ((2 * kk + half) ^ sw) * 16;
            xf[kk] = *(const bf16x8*)(sT + x_row_off + coff);
            wf[kk][0] = *(const bf16x8*)(sT + w_row_off + coff);
            wf[kk][1] = *(const bf16x8*)(sT + w_row_off + 32 * 128 + coff);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[0] = SUPIR_MFMA_32x32x16(wf[kk][0], xf[kk], acc[0], 0, 0, 0);
            acc[1] = SUPIR_MFMA_32x32x16(wf[kk][1], xf[kk], acc[1], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        buf = buf + 1 == XQ_S ? 0 : buf + 1;
    };
    // K steps 0 .. nk-3: tile kt landed (kt+1 may still be in flight), every wave is done with the buffer tile kt+2 goes to
    for (int kt = 0; kt + 2 < nk; ++kt) {
        xq_wait_vmcnt<XQ_LOADS>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int sbuf = buf + 2 >= XQ_S ? buf + 2 - XQ_S : buf + 2;
        stage(sbuf, kt + 2);
        kstep();
    }
    // K step nk-2: its free buffer takes key tile 0; K step nk-1: key tile 1 (when there is one)
    const int kv0 = buf + 2 >= XQ_S ? buf + 2 - XQ_S : buf + 2;
    xq_wait_vmcnt<XQ_LOADS>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage_kv(kv0, 0);
    kstep();
    const int kv1 = buf + 2 >= XQ_S ? buf + 2 - XQ_S : buf + 2;
    xq_wait_vmcnt<XQ_KV_LOADS>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nt > 1) stage_kv(kv1, 1);
    kstep();

    // ---- q -> bf16 B-operand fragments of S^T = K . Q^T: k block i = 2 cb + pp holds accumulator registers 8 pp .. 8 pp + 7
    const float c = p.scale_log2e;
    bf16x8 qf[4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * pp + j;
                const float v = rstd * (acc[cb][r] - mean * cs4[cb][r >> 2][r & 3]) + b4[cb][r >> 2][r & 3];
                qf[2 * cb + pp][j] = (bf16_t)(v * c);
            }

    xq_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // next-weight prefetch (supir_launch_hints): lines of a LATER launch's weight matrix pulled towards L2 / Infinity Cache; the 4 bytes
    // per lane land in the ring buffer the last K step used (free from here on; the output staging below stays under +18 KB)
    if (p.pf_lines) {
        const unsigned total_waves = gridDim.x * 4, gw = blockIdx.x * 4 + wave, n_instr = (p.pf_lines + 63) >> 6;
        char* dump = smem + (buf == 0 ? XQ_S - 1 : buf - 1) * XQ_STAGE + 20480 + wave * 256;
        for (unsigned i = gw; i < n_instr; i += total_waves) {
            unsigned line = i * 64 + lane;
            line = line < p.pf_lines ? line : p.pf_lines - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.pf_ptr + (size_t)line * 128),
                                             (__attribute__((address_space(3))) void*)dump, 4, 0, 0);
        }
    }

    // ---- attention over the nt (1 or 2) key tiles, one 32-key half tile at a time (attention.hip's layouts: K rows read with bits
    // 2 / 3 of the row index swapped so that the 8 P values a lane owns per 16-key block are 8 consecutive keys = one 16-byte chunk
    // of a V^T row)
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int ksw = (krow >> 1) & 7, vsw = (l31 >> 1) & 7;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    for (int t = 0; t < nt; ++t) {
        const char* sK = smem + (t == 0 ? kv0 : kv1) * XQ_STAGE;
        const char* sV = sK + 8192;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int k0 = t * 64 + hf * 32;
            if (k0 >= p.Tk) break;      // wave-uniform
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* kr = sK + hf * 4096 + krow * 128 + 8 * half;
                const xq_u32x2 lo = *(const xq_u32x2*)(kr + (((2 * i) ^ ksw) * 16));
                const xq_u32x2 hi = *(const xq_u32x2*)(kr + (((2 * i + 1) ^ ksw) * 16));
                const xq_u32x4 kk4 = {lo[0], lo[1], hi[0], hi[1]};
                s = SUPIR_MFMA_32x32x16(__builtin_bit_cast(bf16x8, kk4), qf[i], s, 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (k0 + 16 * (r >> 3) + 8 * half + (r & 7) >= p.Tk) s[r] = -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            mx = xq_xhalf_max(mx);           // finite: key k0 is valid for every query
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[0][r] *= alpha;
                o[1][r] *= alpha;
            }
            bf16x8 pf[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[r] - m_new);
                l_run += pv;
                pf[r >> 3][r & 7] = (bf16_t)pv;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int df = 0; df < 2; ++df) {
                    const bf16x8 vf = *(const bf16x8*)(sV + df * 4096 + l31 * 128 + (((2 * (2 * hf + kb) + half) ^ vsw) * 16));
                    o[df] = SUPIR_MFMA_32x32x16(vf, pf[kb], o[df], 0, 0, 0);
                }
            m_run = m_new;
        }
    }

    // ---- output: this wave's 32 x 64 block through LDS, whole 128-byte rows out (as attention.hip's round-4 epilogue)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    constexpr int ORS = 144;
    char* o_stage = smem + wave * (32 * ORS);
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            u16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[df][rg * 4 + e] * inv);
            *(u16x4*)(o_stage + l31 * ORS + (df * 32 + 8 * rg + 4 * half) * 2) = ov;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* Ob = p.O + (m0 + wave * 32) * p.ldo + h * 64;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const f32x4 piece = *(const f32x4*)(o_stage + row * ORS + ch * 16);
        supir_store16(Ob + (size_t)row * p.ldo + ch * 8, piece);
    }
}

}  // namespace

bool supir_xattn_q_supported(const XattnArgs& a) {
    if (a.B <= 0 || a.H <= 0 || a.T <= 0 || a.Tk <= 0 || a.C <= 0) return false;
    if (a.T % 128 || a.C % 64 || (a.C >> 6) < 3 || a.Tk > 128) return false;
    if (a.ldx % 8 || a.ldk % 8 || a.ldvt % 8 || a.ldo % 8 || a.ldvt < ((a.Tk + 63) / 64) * 64) return false;
    if ((((size_t)a.X) | ((size_t)a.Wq) | ((size_t)a.K) | ((size_t)a.Vt) | ((size_t)a.O)) & 15) return false;
    if (a.ln_stats && (!a.ln_colsum || a.ln_slots < 0 || a.ln_slots > 64 || (a.ln_slots > 0 && a.ln_ld < a.ln_slots))) return false;
    if (a.ln_colsum && (((size_t)a.ln_colsum) & 15)) return false;
    if (a.bias && (((size_t)a.bias) & 15)) return false;
    return true;
}

int supir_xattn_q_launch(const XattnArgs& a_in, hipStream_t st) {
    if (!supir_xattn_q_supported(a_in)) return SUPIR_ERR_SHAPE;
    XattnArgs a = a_in;
    // XCD partition: every XCD's L2 pulls its own copy of what its workgroups read.  1-D ranges with the heads fastest give an XCD
    // (blocks / 8) token tiles and every head's weight rows -- at (B 2, H 20, T 1024, C 1280) 0.65 MB of tokens and ALL 3.3 MB of W_q,
    // 35.8 MB fetched for 9.3 MB of operands (profiles/pmc_traffic.json, round 4).  A gm x gn grid of XCDs (token-block chunks x head
    // chunks) cuts that to (blocks / gm) x 128 x C + (H / gn) x 64 x C elements per XCD; pick the cheapest grid that divides both counts
    // (knob 5 = 1, tools: keep the 1-D ranges)
    a.gm = a.gn = 0;
    const int blocks = (a.T / 128) * a.B;
    if (supir_debug_knob_value(5) != 1) {
        double best = 0.0;
        for (int gm = 8; gm >= 1; gm >>= 1) {
            const int gn = 8 / gm;
            if (blocks % gm || a.H % gn) continue;
            const double cost = (double)(blocks / gm) * 128.0 + (double)(a.H / gn) * 64.0;     // x C x 2 bytes
            if (a.gm == 0 || cost < best) {
                best = cost;
                a.gm = gm;
                a.gn = gn;
            }
        }
    }
    SUPIR_LAUNCH(xattn_q_kernel, dim3((a.T / 128) * a.H * a.B), dim3(256), 0, st, a);
    return SUPIR_LAUNCH_STATUS();
}
