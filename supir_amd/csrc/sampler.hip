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
//
// Tiled sampler (TiledRestoreEDMSampler.__call__, sampling.py:600-660): every step cuts the latent canvas [b][C][Hc][Wc] into k
// overlapping T x T tiles, runs sampler_step on each and blends `x_next[tile] += out * w; count[tile] += w` (sampling.py:654-657).
//   pre_tiles : `pre` with the crop folded in -- reads x / eps at the tile windows of the canvas, writes the STACKED tiles
//               [k*b][C][T][T] (tile j, sample bi at row j*b + bi: torch.cat of the crops along dim 0); same arithmetic per element;
//   blend     : x_next[tile_j] += out_j * w for the k tiles of one network call, one launch.  Tiles overlap, so a thread owns one
//               canvas element of the group's bounding box and adds the tiles that cover it IN TILE ORDER: the fp32 result is
//               bitwise the k sequential slice-adds.  The reference's weights are float64 (gaussian_weights builds them with
//               numpy, sampling.py:733-750), so torch evaluates `fp32 += fp32 * fp64` in float64 and rounds to fp32 once per add:
//               (float)((double)acc + (double)out * w), mul and add rounded separately (no fma) -- reproduced here.
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

__global__ __launch_bounds__(256) void edm_pre_tiles_kernel(const float* __restrict__ x, const float* __restrict__ eps, float s_noise,
                                                             float noise_mul, float c_in, float* __restrict__ x_hat,
                                                             float* __restrict__ net_in, SupirTileList tl, int b, int C, int Hc, int Wc,
                                                             int T, int reps) {
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= T * T) return;
    const int y = idx / T, xx = idx - y * T;
    const int bc = (int)blockIdx.y, j = (int)blockIdx.z;   // (sample, channel) plane; tile
    const int bi = bc / C;
    const long src = ((long)bc * Hc + tl.hi[j] + y) * Wc + tl.wi[j] + xx;
    const long dst = (((long)j * b * C + bc) * T + y) * T + xx;
    (void)bi;
    float v = x[src];
    if (eps) v = v + (eps[src] * s_noise) * noise_mul;
    x_hat[dst] = v;
    const float w = v * c_in;
    const long n = (long)tl.n * b * C * T * T;
    for (int r = 0; r < reps; ++r) net_in[(long)r * n + dst] = w;
}

__global__ __launch_bounds__(256) void tile_blend_kernel(const float* __restrict__ tiles, const double* __restrict__ w,
                                                          float* __restrict__ canvas, SupirTileList tl, int b, int C, int Hc, int Wc,
                                                          int T, int y0, int x0, int bh, int bw) {
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= bh * bw) return;
    const int yy = idx / bw, Y = y0 + yy, X = x0 + (idx - yy * bw);
    const int bc = (int)blockIdx.y;
    float* dst = canvas + ((long)bc * Hc + Y) * Wc + X;
    float acc = *dst;
    bool any = false;
    for (int j = 0; j < tl.n; ++j) {
        const int ty = Y - tl.hi[j], tx = X - tl.wi[j];
        if ((unsigned)ty < (unsigned)T && (unsigned)tx < (unsigned)T) {
            const double prod = __dmul_rn((double)tiles[(((long)j * b * C + bc) * T + ty) * T + tx], w[ty * T + tx]);
            acc = (float)__dadd_rn((double)acc, prod);
            any = true;
        }
    }
    if (any) *dst = acc;
}

}  // namespace

int supir_edm_pre_tiles_launch(const float* x, const float* eps, float s_noise, float noise_mul, float c_in, float* x_hat, float* net_in,
                               const SupirTileList& tl, int b, int C, int Hc, int Wc, int T, int reps, hipStream_t st) {
    if (tl.n <= 0 || tl.n > SUPIR_MAX_TILES || b <= 0 || C <= 0 || T <= 0 || reps < 1 || reps > 2) return SUPIR_ERR_ARG;
    for (int j = 0; j < tl.n; ++j)
        if (tl.hi[j] < 0 || tl.wi[j] < 0 || tl.hi[j] + T > Hc || tl.wi[j] + T > Wc) return SUPIR_ERR_SHAPE;
    if ((long)b * C > 65535) return SUPIR_ERR_SHAPE;
    SUPIR_LAUNCH(edm_pre_tiles_kernel, dim3((unsigned)((T * T + 255) / 256), (unsigned)(b * C), (unsigned)tl.n), dim3(256), 0, st, x, eps,
                 s_noise, noise_mul, c_in, x_hat, net_in, tl, b, C, Hc, Wc, T, reps);
    return SUPIR_LAUNCH_STATUS();
}

int supir_tile_blend_launch(const float* tiles, const double* w, float* canvas, const SupirTileList& tl, int b, int C, int Hc, int Wc, int T,
                            hipStream_t st) {
    if (tl.n <= 0 || tl.n > SUPIR_MAX_TILES || b <= 0 || C <= 0 || T <= 0) return SUPIR_ERR_ARG;
    int y0 = Hc, x0 = Wc, y1 = 0, x1 = 0;
    for (int j = 0; j < tl.n; ++j) {
        if (tl.hi[j] < 0 || tl.wi[j] < 0 || tl.hi[j] + T > Hc || tl.wi[j] + T > Wc) return SUPIR_ERR_SHAPE;
        y0 = tl.hi[j] < y0 ? tl.hi[j] : y0;
        x0 = tl.wi[j] < x0 ? tl.wi[j] : x0;
        y1 = tl.hi[j] + T > y1 ? tl.hi[j] + T : y1;
        x1 = tl.wi[j] + T > x1 ? tl.wi[j] + T : x1;
    }
    if ((long)b * C > 65535) return SUPIR_ERR_SHAPE;
    const int bh = y1 - y0, bw = x1 - x0;
    SUPIR_LAUNCH(tile_blend_kernel, dim3((unsigned)(((long)bh * bw + 255) / 256), (unsigned)(b * C)), dim3(256), 0, st, tiles, w, canvas, tl,
                 b, C, Hc, Wc, T, y0, x0, bh, bw);
    return SUPIR_LAUNCH_STATUS();
}

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
