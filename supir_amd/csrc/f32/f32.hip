// libsupir_hip_f32.so: the fp32 service (include/supir_hip_f32.h).  One general tile kernel on the exact-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32 (fp32 operands and accumulation, bitwise an fmaf chain in k order) for every GEMM / 3x3 convolution /
// attention product of the path, plus GroupNorm, LayerNorm, row softmax and the GEGLU gate.  A correctness path for
// `--diff_dtype fp32` / `--ae_dtype fp32` requests (reference: sgm/modules/diffusionmodules/wrappers.py:87,
// SUPIR/models/SUPIR_model.py:41-69): general shapes, guarded edges, no tuning tables.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/supir_hip.h"
#include "../../../include/supir_hip_f32.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

static thread_local int g_last_hip_error = 0;
static int note_status(hipError_t e) {
    if (e == hipSuccess) return SUPIR_OK;
    g_last_hip_error = (int)e;
    return SUPIR_ERR_HIP;
}
#define F32_LAUNCH(...)                  \
    do {                                 \
        (void)hipGetLastError();         \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)
#define F32_STATUS() note_status(hipGetLastError())

extern "C" int supir_abi_version(void) { return 2; }
extern "C" const char* supir_target_arch(void) { return "gfx950"; }
extern "C" const char* supir_elem_type(void) { return "f32"; }
extern "C" int supir_last_hip_error(void) { return g_last_hip_error; }
extern "C" const char* supir_hip_error_string(int code) { return hipGetErrorString((hipError_t)code); }

// ------------------------------------------------------------------------------------------------------------------- GEMM / conv
// Tile BM x BN, K step 16, 256 threads = 4 waves as 2 x 2, each wave (BM/2) x (BN/2) in 16 x 16 fragments.  Global -> registers -> LDS
// with the next K step's loads in flight over the current step's MFMAs (two LDS buffers, one barrier per step).  Every load is guarded
// (rows beyond M / N and columns beyond K read zero), so any shape runs; 16-byte loads where base, leading dimension and K (Cin) allow.
struct GemmP {
    supir_f32_gemm_desc d;
    int vec_a, vec_w;
};

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == SUPIR_ACT_SILU) return v / (1.f + expf(-v));
    if (act == SUPIR_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    if (act == SUPIR_ACT_QUICKGELU) return v / (1.f + expf(-1.702f * v));
    return v;
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void f32_gemm_kernel(const GemmP p) {
    constexpr int BK = 16, LD = BK + 4;
    constexpr int QA = BM * BK / 4 / 256, QW = BN * BK / 4 / 256;   // 16-byte quads per thread and K step
    constexpr int MI = BM / 32, NI = BN / 32;                        // fragments per wave
    __shared__ float As[2][BM][LD];
    __shared__ float Ws[2][BN][LD];
    const supir_f32_gemm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;   // M tiles on grid.x (limit 2^31 - 1: the 3-channel conv_out of a 2048^2 image is 64 K+ row tiles)
    const int z0 = blockIdx.z % d.nz0, z1 = blockIdx.z / d.nz0;
    const float* A = d.A + (size_t)z0 * d.a_s0 + (size_t)z1 * d.a_s1;
    const float* W = d.W + (size_t)z0 * d.w_s0 + (size_t)z1 * d.w_s1;
    float* C = d.C + (size_t)z0 * d.c_s0 + (size_t)z1 * d.c_s1;
    const bool conv = d.kind == SUPIR_F32_CONV3X3;
    const int VH = d.upsample ? 2 * d.H : d.H, VW = d.upsample ? 2 * d.Wd : d.Wd;

    // per-thread rows of the A / W tiles (fixed over the K loop)
    const float* a_row[QA];
    int a_ok[QA], a_iy0[QA], a_ix0[QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int row = (tid + q * 256) >> 2, m = m0 + row;
        a_ok[q] = m < d.M;
        a_iy0[q] = a_ix0[q] = 0;
        if (conv) {
            const int mm = a_ok[q] ? m : 0;
            const int b = mm / (d.OH * d.OW), r = mm - b * (d.OH * d.OW);
            const int oy = r / d.OW, ox = r - oy * d.OW;
            a_iy0[q] = oy * d.stride - d.pad_t;
            a_ix0[q] = ox * d.stride - d.pad_l;
            a_row[q] = A + (size_t)b * d.H * d.Wd * d.lda;
        } else {
            a_row[q] = A + (size_t)(a_ok[q] ? m : 0) * d.lda;
        }
    }
    const float* w_row[QW];
    int w_ok[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        const int row = (tid + q * 256) >> 2, n = n0 + row;
        w_ok[q] = n < d.N;
        w_row[q] = W + (size_t)(w_ok[q] ? n : 0) * d.ldw;
    }
    const int kq = (tid & 3) * 4;

    auto load_a1 = [&](int q, int k) -> float {      // one element of the A tile: row of quad q, column k
        if (!a_ok[q] || k >= d.K) return 0.f;
        if (!conv) return a_row[q][k];
        const int tap = k / d.Cin, cin = k - tap * d.Cin, ky = tap / 3, kx = tap - ky * 3;
        const int iy = a_iy0[q] + ky, ix = a_ix0[q] + kx;
        if ((unsigned)iy >= (unsigned)VH || (unsigned)ix >= (unsigned)VW) return 0.f;
        return a_row[q][((size_t)(iy >> d.upsample) * d.Wd + (ix >> d.upsample)) * d.lda + cin];
    };
    auto load_a = [&](int q, int k) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p.vec_a) {                                // K % 4 == 0 (conv: Cin % 4 == 0): the quad is inside K and inside one tap
            if (!a_ok[q] || k >= d.K) return v;
            if (!conv) return *(const f32x4*)(a_row[q] + k);
            const int tap = k / d.Cin, cin = k - tap * d.Cin, ky = tap / 3, kx = tap - ky * 3;
            const int iy = a_iy0[q] + ky, ix = a_ix0[q] + kx;
            if ((unsigned)iy >= (unsigned)VH || (unsigned)ix >= (unsigned)VW) return v;
            return *(const f32x4*)(a_row[q] + ((size_t)(iy >> d.upsample) * d.Wd + (ix >> d.upsample)) * d.lda + cin);
        }
        v[0] = load_a1(q, k); v[1] = load_a1(q, k + 1); v[2] = load_a1(q, k + 2); v[3] = load_a1(q, k + 3);
        return v;
    };
    auto load_w = [&](int q, int k) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!w_ok[q] || k >= d.K) return v;
        if (p.vec_w) return *(const f32x4*)(w_row[q] + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k + e < d.K) ? w_row[q][k + e] : 0.f;
        return v;
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[QA], rw[QW];
    const int nk = (d.K + BK - 1) / BK;
#pragma unroll
    for (int q = 0; q < QA; ++q) ra[q] = load_a(q, kq);
#pragma unroll
    for (int q = 0; q < QW; ++q) rw[q] = load_w(q, kq);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int q = 0; q < QA; ++q) *(f32x4*)&As[buf][(tid + q * 256) >> 2][kq] = ra[q];
#pragma unroll
        for (int q = 0; q < QW; ++q) *(f32x4*)&Ws[buf][(tid + q * 256) >> 2][kq] = rw[q];
        __syncthreads();     // the other buffer was last read before the previous iteration's barrier: one barrier per step is enough
        if (kt + 1 < nk) {
#pragma unroll
            for (int q = 0; q < QA; ++q) ra[q] = load_a(q, (kt + 1) * BK + kq);
#pragma unroll
            for (int q = 0; q < QW; ++q) rw[q] = load_w(q, (kt + 1) * BK + kq);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = As[buf][wm * (BM / 2) + i * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = Ws[buf][wn * (BN / 2) + j * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wf[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: fragment (i, j), register r holds C[m][n] with m = ... + (lane >> 4) * 4 + r, n = ... + (lane & 15)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
            if (m >= d.M) continue;
            const int b = d.rows_per_batch > 0 ? m / d.rows_per_batch : 0;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
                if (n >= d.N) continue;
                float v = acc[i][j][r];
                if (d.bias) v += d.bias[n];
                if (d.rowbias) v += d.rowbias[(size_t)b * d.ld_rowbias + n];
                v = act_apply(v, d.act) * d.alpha;
                if (d.residual) v += d.residual[(size_t)m * d.ldr + n];
                if (d.out_mode == SUPIR_OUT_BF16_T) C[((size_t)b * d.N + n) * d.ldc + (m - b * d.rows_per_batch)] = v;
                else C[(size_t)m * d.ldc + n] = v;
            }
        }
}

extern "C" int supir_f32_gemm(const supir_f32_gemm_desc* dp, void* stream) {
    if (!dp || !dp->A || !dp->W || !dp->C) return SUPIR_ERR_ARG;
    GemmP p;
    p.d = *dp;
    supir_f32_gemm_desc& d = p.d;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.nz0 <= 0 || d.nz1 <= 0) return SUPIR_ERR_ARG;
    if (d.kind != SUPIR_F32_GEMM && d.kind != SUPIR_F32_CONV3X3) return SUPIR_ERR_ARG;
    if (d.act != SUPIR_ACT_NONE && d.act != SUPIR_ACT_SILU && d.act != SUPIR_ACT_GELU && d.act != SUPIR_ACT_QUICKGELU) return SUPIR_ERR_ARG;
    if (d.out_mode != SUPIR_OUT_BF16 && d.out_mode != SUPIR_OUT_BF16_T) return SUPIR_ERR_ARG;
    if ((d.rowbias || d.out_mode == SUPIR_OUT_BF16_T) && d.rows_per_batch <= 0) return SUPIR_ERR_ARG;
    if ((size_t)d.nz0 * d.nz1 > 1 && (d.bias || d.rowbias || d.residual)) return SUPIR_ERR_ARG;
    if (d.kind == SUPIR_F32_CONV3X3) {
        if (d.B <= 0 || d.H <= 0 || d.Wd <= 0 || d.Cin <= 0 || d.OH <= 0 || d.OW <= 0 || (d.stride != 1 && d.stride != 2)) return SUPIR_ERR_ARG;
        if (d.K != 9 * d.Cin || (long)d.M != (long)d.B * d.OH * d.OW || d.lda < d.Cin) return SUPIR_ERR_SHAPE;
        d.upsample = d.upsample ? 1 : 0;
        p.vec_a = d.Cin % 4 == 0 && d.lda % 4 == 0 && ((uintptr_t)d.A % 16) == 0 && d.a_s0 % 4 == 0 && d.a_s1 % 4 == 0;
    } else {
        if (d.lda < d.K) return SUPIR_ERR_SHAPE;
        p.vec_a = d.K % 4 == 0 && d.lda % 4 == 0 && ((uintptr_t)d.A % 16) == 0 && d.a_s0 % 4 == 0 && d.a_s1 % 4 == 0;
    }
    if (d.ldw < d.K) return SUPIR_ERR_SHAPE;
    p.vec_w = d.K % 4 == 0 && d.ldw % 4 == 0 && ((uintptr_t)d.W % 16) == 0 && d.w_s0 % 4 == 0 && d.w_s1 % 4 == 0;
    const long nz = (long)d.nz0 * d.nz1;
    if (nz > 65535) return SUPIR_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (d.M >= 1024 && d.N >= 96) {
        dim3 grid((d.M + 127) / 128, (d.N + 127) / 128, (unsigned)nz);
        if (grid.y > 65535) return SUPIR_ERR_SHAPE;
        F32_LAUNCH((f32_gemm_kernel<128, 128>), grid, dim3(256), 0, s, p);
    } else {
        dim3 grid((d.M + 63) / 64, (d.N + 63) / 64, (unsigned)nz);
        if (grid.y > 65535) return SUPIR_ERR_SHAPE;
        F32_LAUNCH((f32_gemm_kernel<64, 64>), grid, dim3(256), 0, s, p);
    }
    return F32_STATUS();
}

// ------------------------------------------------------------------------------------------------------------------- GEGLU gate
__global__ void f32_geglu_kernel(const float* __restrict__ proj, float* __restrict__ out, int M, int N2, int ldp, int ldo, int block) {
    const int half = N2 / 2;
    const size_t total = (size_t)M * half;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / half), j = (int)(e - (size_t)m * half);
        int cv, cg;
        if (block == 0) { cv = j; cg = half + j; }
        else { cv = (j / block) * 2 * block + j % block; cg = cv + block; }
        const float v = proj[(size_t)m * ldp + cv], g = proj[(size_t)m * ldp + cg];
        out[(size_t)m * ldo + j] = v * (0.5f * g * (1.f + erff(g * 0.70710678118654752f)));
    }
}

extern "C" int supir_f32_geglu(const float* proj, float* out, int M, int N2, int ldp, int ldo, int block, void* stream) {
    if (!proj || !out || M <= 0 || N2 <= 0 || N2 % 2) return SUPIR_ERR_ARG;
    if (block != 0 && (block < 0 || (N2 / 2) % block)) return SUPIR_ERR_SHAPE;
    const size_t total = (size_t)M * (N2 / 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    F32_LAUNCH(f32_geglu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, proj, out, M, N2, ldp, ldo, block);
    return F32_STATUS();
}

// ------------------------------------------------------------------------------------------------------------------- reductions
__device__ __forceinline__ float block_sum(float v, float* sh) {      // 256 threads; result on every thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------------------------------------- softmax
__global__ __launch_bounds__(256) void f32_softmax_kernel(const float* __restrict__ S, float* __restrict__ P, int T, int Tpad, long ld_s,
                                                          long ld_p, float scale, int causal_tq, long row0) {
    __shared__ float sh[4];
    const float* s = S + (size_t)blockIdx.x * ld_s;
    float* o = P + (size_t)blockIdx.x * ld_p;
    if (causal_tq > 0) T = min(T, (int)((row0 + blockIdx.x) % causal_tq) + 1);   // query i of its block sees keys j <= i; the rest become zeros
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256) mx = fmaxf(mx, s[t] * scale);
    mx = block_max(mx, sh);
    float sum = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) sum += expf(s[t] * scale - mx);
    sum = block_sum(sum, sh);
    const float inv = 1.f / sum;
    for (int t = threadIdx.x; t < Tpad; t += 256) o[t] = t < T ? expf(s[t] * scale - mx) * inv : 0.f;
}

extern "C" int supir_f32_softmax_rows(const float* S, float* P, long rows, int T, int Tpad, long ld_s, long ld_p, float scale, int causal_tq,
                                      void* stream) {
    if (!S || !P || rows <= 0 || T <= 0 || Tpad < T || ld_s < T || ld_p < Tpad || causal_tq < 0) return SUPIR_ERR_ARG;
    if (causal_tq > 0 && rows % causal_tq) return SUPIR_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    for (long r0 = 0; r0 < rows; r0 += 1 << 30) {       // grid.x limit
        const long n = rows - r0 < (1L << 30) ? rows - r0 : (1L << 30);
        F32_LAUNCH(f32_softmax_kernel, dim3((unsigned)n), dim3(256), 0, s, S + (size_t)r0 * ld_s, P + (size_t)r0 * ld_p, T, Tpad, ld_s, ld_p, scale,
                   causal_tq, r0);
        const int rc = F32_STATUS();
        if (rc != SUPIR_OK) return rc;
    }
    return SUPIR_OK;
}

// ------------------------------------------------------------------------------------------------------------------- LayerNorm
__global__ __launch_bounds__(256) void f32_layernorm_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int C, int ldx, int ldy, float eps) {
    __shared__ float sh[4];
    const float* xr = x + (size_t)blockIdx.x * ldx;
    float* yr = y + (size_t)blockIdx.x * ldy;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += xr[c];
    const float mean = block_sum(s, sh) / (float)C;
    float q = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float dlt = xr[c] - mean; q += dlt * dlt; }
    const float rstd = rsqrtf(block_sum(q, sh) / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) yr[c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

extern "C" int supir_f32_layernorm(const float* x, float* y, const float* gamma, const float* beta, int rows, int C, int ldx, int ldy, float eps,
                                   void* stream) {
    if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || ldx < C || ldy < C) return SUPIR_ERR_ARG;
    F32_LAUNCH(f32_layernorm_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta, C, ldx, ldy, eps);
    return F32_STATUS();
}

// ------------------------------------------------------------------------------------------------------------------- GroupNorm
// statistics: grid (B * 32 groups, NS row slices): fp64 (sum, sum of squares) of one slice -> workspace; finalize: (mean, rstd) per group
// into the first B * 32 * 2 doubles ... kept apart from the partials (second half of the workspace); apply: one thread per element.
#define GN_SLICES 64
struct GnP {
    const float *x1, *x2, *x1raw, *x2raw, *gamma, *beta, *mod_g, *mod_b;
    const float* given;      // externally pooled (mean, biased variance) per (batch, group), or NULL
    float* out;
    double* ws;
    double* sums_out;        // supir_f32_groupnorm_stats: (sum, sum of squares) per (batch, group) instead of (mean, rstd)
    int B, HW, C, C1, ld1, ld2, ldm, ldo, act, ns;
    float eps, control_scale;
};

__device__ __forceinline__ float gn_src(const float* x1, const float* x2, int ld1, int ld2, int C1, size_t row, int c) {
    return c < C1 ? x1[row * ld1 + c] : x2[row * ld2 + (c - C1)];
}

__global__ __launch_bounds__(256) void f32_gn_stats_kernel(const GnP p) {
    __shared__ double sh[4];
    const int bg = blockIdx.x, b = bg >> 5, g = bg & 31, cg = p.C / 32;
    const int rows_per = (p.HW + p.ns - 1) / p.ns;
    const int r0 = blockIdx.y * rows_per, r1 = min(p.HW, r0 + rows_per);
    double s = 0.0, q = 0.0;
    const long total = (long)(r1 - r0) * cg;
    for (long e = threadIdx.x; e < total; e += 256) {
        const int r = r0 + (int)(e / cg), c = g * cg + (int)(e % cg);
        const double v = (double)gn_src(p.x1, p.x2, p.ld1, p.ld2, p.C1, (size_t)b * p.HW + r, c);
        s += v;
        q += v * v;
    }
    s = block_sum_d(s, sh);
    q = block_sum_d(q, sh);
    if (threadIdx.x == 0) {
        double* o = p.ws + ((size_t)bg * GN_SLICES + blockIdx.y) * 2;
        o[0] = s;
        o[1] = q;
    }
}

__global__ void f32_gn_finalize_kernel(const GnP p) {
    const int bg = blockIdx.x * blockDim.x + threadIdx.x;
    if (bg >= p.B * 32) return;
    double mean, var;
    if (p.given) {
        mean = (double)p.given[bg * 2];
        var = (double)p.given[bg * 2 + 1];
    } else {
        double s = 0.0, q = 0.0;
        for (int i = 0; i < p.ns; ++i) {
            s += p.ws[((size_t)bg * GN_SLICES + i) * 2];
            q += p.ws[((size_t)bg * GN_SLICES + i) * 2 + 1];
        }
        if (p.sums_out) {
            p.sums_out[bg * 2] = s;
            p.sums_out[bg * 2 + 1] = q;
            return;
        }
        const double n = (double)p.HW * (p.C / 32);
        mean = s / n;
        var = q / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
    }
    double* mr = p.ws + (size_t)p.B * 32 * GN_SLICES * 2 + (size_t)bg * 2;
    mr[0] = mean;
    mr[1] = 1.0 / sqrt(var + (double)p.eps);
}

__global__ __launch_bounds__(256) void f32_gn_apply_kernel(const GnP p) {
    const size_t total = (size_t)p.B * p.HW * p.C;
    const int cg = p.C / 32;
    const double* mr = p.ws + (size_t)p.B * 32 * GN_SLICES * 2;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / p.C;
        const int c = (int)(e - row * p.C), b = (int)(row / p.HW), g = c / cg;
        const float mean = (float)mr[(b * 32 + g) * 2], rstd = (float)mr[(b * 32 + g) * 2 + 1];
        const float x = gn_src(p.x1, p.x2, p.ld1, p.ld2, p.C1, row, c);
        float v = (x - mean) * rstd * p.gamma[c] + p.beta[c];
        if (p.mod_g) v = v * (p.mod_g[row * p.ldm + c] + 1.f) + p.mod_b[row * p.ldm + c];
        if (p.act == 1) v = v / (1.f + expf(-v));
        if (p.control_scale != 1.f) {
            const float raw = gn_src(p.x1raw ? p.x1raw : p.x1, p.x2raw ? p.x2raw : p.x2, p.ld1, p.ld2, p.C1, row, c);
            v = v * p.control_scale + raw * (1.f - p.control_scale);
        }
        p.out[row * p.ldo + c] = v;
    }
}

extern "C" int supir_f32_groupnorm(const float* x1, const float* x2, const float* x1raw, const float* x2raw, int B, int HW, int C, int C1, int ld1,
                                   int ld2, const float* gamma, const float* beta, float eps, int act, const float* mod_g, const float* mod_b,
                                   int ldm, float control_scale, float* out, int ldo, double* workspace, size_t workspace_bytes,
                                   const float* given_mean_var, void* stream) {
    if (!x1 || !gamma || !beta || !out || !workspace || B <= 0 || HW <= 0 || C <= 0) return SUPIR_ERR_ARG;
    if (C % 32 || C1 <= 0 || C1 > C || (C1 < C && !x2) || ld1 < C1 || (x2 && ld2 < C - C1) || ldo < C) return SUPIR_ERR_SHAPE;
    if ((mod_g == nullptr) != (mod_b == nullptr) || (mod_g && ldm < C) || (act != 0 && act != 1)) return SUPIR_ERR_ARG;
    if (workspace_bytes < ((size_t)B * 32 * GN_SLICES * 2 + (size_t)B * 32 * 2) * sizeof(double)) return SUPIR_ERR_ARG;
    GnP p;
    p.x1 = x1; p.x2 = x2; p.x1raw = x1raw; p.x2raw = x2raw; p.gamma = gamma; p.beta = beta; p.mod_g = mod_g; p.mod_b = mod_b;
    p.out = out; p.ws = workspace; p.given = given_mean_var; p.sums_out = nullptr;
    p.B = B; p.HW = HW; p.C = C; p.C1 = C1; p.ld1 = ld1; p.ld2 = ld2; p.ldm = ldm; p.ldo = ldo; p.act = act;
    p.eps = eps; p.control_scale = control_scale;
    const long per_group = (long)HW * (C / 32);
    p.ns = (int)((per_group + 16383) / 16384);
    p.ns = p.ns < 1 ? 1 : p.ns > GN_SLICES ? GN_SLICES : p.ns;
    p.ns = p.ns > HW ? HW : p.ns;
    hipStream_t s = (hipStream_t)stream;
    int rc = SUPIR_OK;
    if (!given_mean_var) {
        F32_LAUNCH(f32_gn_stats_kernel, dim3(B * 32, p.ns), dim3(256), 0, s, p);
        rc = F32_STATUS();
        if (rc != SUPIR_OK) return rc;
    }
    F32_LAUNCH(f32_gn_finalize_kernel, dim3((B * 32 + 63) / 64), dim3(64), 0, s, p);
    rc = F32_STATUS();
    if (rc != SUPIR_OK) return rc;
    const size_t total = (size_t)B * HW * C;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 262144 ? 262144 : (total + 255) / 256);
    F32_LAUNCH(f32_gn_apply_kernel, dim3(blocks), dim3(256), 0, s, p);
    return F32_STATUS();
}

extern "C" int supir_f32_groupnorm_stats(const float* x, int B, int HW, int C, int ld, double* sums, double* workspace, size_t workspace_bytes,
                                         void* stream) {
    if (!x || !sums || !workspace || B <= 0 || HW <= 0 || C <= 0) return SUPIR_ERR_ARG;
    if (C % 32 || ld < C) return SUPIR_ERR_SHAPE;
    if (workspace_bytes < ((size_t)B * 32 * GN_SLICES * 2 + (size_t)B * 32 * 2) * sizeof(double)) return SUPIR_ERR_ARG;
    GnP p = {};
    p.x1 = x; p.ws = workspace; p.sums_out = sums;
    p.B = B; p.HW = HW; p.C = C; p.C1 = C; p.ld1 = ld;
    const long per_group = (long)HW * (C / 32);
    p.ns = (int)((per_group + 16383) / 16384);
    p.ns = p.ns < 1 ? 1 : p.ns > GN_SLICES ? GN_SLICES : p.ns;
    p.ns = p.ns > HW ? HW : p.ns;
    hipStream_t s = (hipStream_t)stream;
    F32_LAUNCH(f32_gn_stats_kernel, dim3(B * 32, p.ns), dim3(256), 0, s, p);
    const int rc = F32_STATUS();
    if (rc != SUPIR_OK) return rc;
    F32_LAUNCH(f32_gn_finalize_kernel, dim3((B * 32 + 63) / 64), dim3(64), 0, s, p);
    return F32_STATUS();
}
