// Elementwise halves of one restoration-guided EDM sampler step on fp32 latents, fused around the network call.
//
// Reference: RestoreEDMSampler.sampler_step (sgm/modules/diffusionmodules/sampling.py:548-570) with the pieces it calls --
// LinearCFG.prepare_inputs / __call__ + NoDynamicThresholding (guiders.py:44-74, sampling_utils.py:7-9), the EpsScaling
// denoiser wrapper DiscreteDenoiserWithControl.__call__ (denoiser.py:66-73, denoiser_scaling.py:16-22), to_d / euler_step
// (sampling_utils.py:39-40, sampling.py:82-83).  In torch these are ~45 single-purpose launches per step on a 64 K-element
// latent (add, mul, cat, pow, ...): launch-bound, ~0.3 ms per step next to a 29 ms network call.  Every sigma-derived factor
// is the same for all samples of a batch (s_in * sigmas[i]) and is known on the host (the schedule lives there), so the two
// kernels take them as scalars, computed on the host in fp32 with the reference's operation order.
//
//   pre :  x_hat = x + (eps * s_noise) * sqrt(sigma_hat^2 - sigma^2)           (churn noise; eps == NULL -> x_hat = x)
//          net_in[r] = x_hat * c_in  for r < reps                              (CFG batch doubling: [x_hat; x_hat])
//   post:  den_r = net_out[r] * c_out + x_hat * c_skip                         (EpsScaling: c_skip = 1, c_out = -sigma)
//          den   = den_0 + cfg * (den_1 - den_0)     (reps == 2: uncond first)  |  den_0 (reps == 1)
//          den  -= (den - x_center) * restore_mul     (x_center != NULL: restoration guidance, (sigma / sigma_max)^restore_cfg)
//          x_next = x_hat + dt * ((x_hat - den) / sigma_hat)                    (dt = sigma_next - sigma_hat)
// fp32 throughout, IEEE division, same association as the reference expressions; n = elements of ONE copy of the latent batch.
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void edm_pre_kernel(const float* __restrict__ x, const float* __restrict__ eps, float s_noise,
                                                       float noise_mul, float c_in, float* __restrict__ x_hat,
                                                       float* __restrict__ net_in, long n, int reps) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        f32x4 v = *(const f32x4*)(x + i);
        if (eps) {
            const f32x4 e = *(const f32x4*)(eps + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] + (e[k] * s_noise) * noise_mul;
        }
        if (x_hat) *(f32x4*)(x_hat + i) = v;
        f32x4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = v[k] * c_in;
        for (int r = 0; r < reps; ++r) *(f32x4*)(net_in + (long)r * n + i) = w;
    } else {
        for (long j = i; j < n; ++j) {
            float v = x[j];
            if (eps) v = v + (eps[j] * s_noise) * noise_mul;
            if (x_hat) x_hat[j] = v;
            for (int r = 0; r < reps; ++r) net_in[(long)r * n + j] = v * c_in;
        }
    }
}

__device__ __forceinline__ float edm_post_one(float n0, float n1, float xh, float xc, bool has_center, int reps, float c_out,
                                               float c_skip, float cfg, float restore_mul, float sigma_hat, float dt) {
    float den = n0 * c_out + xh * c_skip;
    if (reps == 2) {
        const float den1 = n1 * c_out + xh * c_skip;
        den = den + cfg * (den1 - den);
    }
    if (has_center) den = den - (den - xc) * restore_mul;
    const float d = __fdiv_rn(xh - den, sigma_hat);
    return xh + dt * d;
}

__global__ __launch_bounds__(256) void edm_post_kernel(const float* __restrict__ net_out, const float* __restrict__ x_hat,
                                                        const float* __restrict__ x_center, float c_out, float c_skip, float cfg,
                                                        float restore_mul, float sigma_hat, float dt, float* __restrict__ x_next,
                                                        long n, int reps) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const bool hc = x_center != nullptr;
    if (i + 4 <= n) {
        const f32x4 n0 = *(const f32x4*)(net_out + i);
        const f32x4 n1 = reps == 2 ? *(const f32x4*)(net_out + n + i) : n0;
        const f32x4 xh = *(const f32x4*)(x_hat + i);
        const f32x4 xc = hc ? *(const f32x4*)(x_center + i) : xh;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = edm_post_one(n0[k], n1[k], xh[k], xc[k], hc, reps, c_out, c_skip, cfg, restore_mul, sigma_hat, dt);
        *(f32x4*)(x_next + i) = o;
    } else {
        for (long j = i; j < n; ++j)
            x_next[j] = edm_post_one(net_out[j], reps == 2 ? net_out[n + j] : 0.f, x_hat[j], hc ? x_center[j] : 0.f, hc, reps, c_out,
                                     c_skip, cfg, restore_mul, sigma_hat, dt);
    }
}

}  // namespace

int supir_edm_pre_launch(const float* x, const float* eps, float s_noise, float noise_mul, float c_in, float* x_hat, float* net_in,
                         long n, int reps, hipStream_t st) {
    if (n <= 0 || reps < 1 || reps > 2) return SUPIR_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)eps | (uintptr_t)x_hat | (uintptr_t)net_in) & 15 || (reps == 2 && (n & 3))) return SUPIR_ERR_SHAPE;
    const long nb = (n + 1023) / 1024;
    if (nb > 0x7fffffffL) return SUPIR_ERR_SHAPE;
    SUPIR_LAUNCH(edm_pre_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, eps, s_noise, noise_mul, c_in, x_hat, net_in, n, reps);
    return SUPIR_LAUNCH_STATUS();
}

int supir_edm_post_launch(const float* net_out, const float* x_hat, const float* x_center, float c_out, float c_skip, float cfg,
                          float restore_mul, float sigma_hat, float dt, float* x_next, long n, int reps, hipStream_t st) {
    if (n <= 0 || reps < 1 || reps > 2 || !(sigma_hat > 0.f)) return SUPIR_ERR_ARG;
    if (((uintptr_t)net_out | (uintptr_t)x_hat | (uintptr_t)x_center | (uintptr_t)x_next) & 15 || (reps == 2 && (n & 3)))
        return SUPIR_ERR_SHAPE;
    const long nb = (n + 1023) / 1024;
    if (nb > 0x7fffffffL) return SUPIR_ERR_SHAPE;
    SUPIR_LAUNCH(edm_post_kernel, dim3((unsigned)nb), dim3(256), 0, st, net_out, x_hat, x_center, c_out, c_skip, cfg, restore_mul,
                 sigma_hat, dt, x_next, n, reps);
    return SUPIR_LAUNCH_STATUS();
}
