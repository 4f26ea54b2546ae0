// GroupNorm(32) (+SiLU, +ZeroSFT modulation) and LayerNorm on NHWC / token-major bf16, for gfx950.
//
// Replaces ATen group_norm / layer_norm / silu called from
//   GroupNorm32  sgm/modules/diffusionmodules/util.py:258-276 (eps 1e-5; ResBlock, ZeroSFT, ZeroCrossAttn, UNet.out)
//   Normalize    sgm/modules/attention.py:122-125 and sgm/modules/diffusionmodules/model.py:48-51 (eps 1e-6)
//   nn.LayerNorm sgm/modules/attention.py:437-439
//   ZeroSFT      SUPIR/modules/SUPIR_v0.py:91-113  (GN(cat[h_ori,h]) * (gamma+1) + beta, lerp by control_scale)
// These are HBM-bound kernels: 16-byte loads/stores, fp32 statistics, two launches per GroupNorm
// (partial sums per row-chunk, then normalise+activate); the second read is served by L2 / Infinity Cache.
#include "kernels.h"
#include <stdlib.h>


__device__ __forceinline__ const bf16_t* gn_src(const GnArgs& p, int b, int row, int c) {
    return (c < p.C1) ? p.x1 + ((size_t)b * p.HW + row) * p.ld1 + c
                      : p.x2 + ((size_t)b * p.HW + row) * p.ld2 + (c - p.C1);
}

// NP > 1: blockIdx.z selects one of NP independent GroupNorm problems of identical geometry (supir_groupnorm_grouped)
template <int NP>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgsN<NP> pp) {
    const GnArgs& p = pp.p[NP == 1 ? 0 : blockIdx.z];
    // per-channel partial sums go through LDS and are reduced in a FIXED order (no atomics): results are bitwise
    // reproducible run to run, which the parity tests and hipGraph-vs-eager checks rely on.
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cv = p.C >> 3, cpg = p.C >> 5;
    const int cvb = cv < 256 ? cv : 256;
    const int TY = 256 / cvb;
    float* ssum = (float*)smem_raw;        // [TY][C]
    float* ssq = ssum + TY * p.C;          // [TY][C]
    const int vx = tid % cvb, ty = tid / cvb;
    const int row0 = chunk * p.rows_per_chunk;
    const int row1 = min(p.HW, row0 + p.rows_per_chunk);
    if (ty < TY) {
        for (int v = vx; v < cv; v += cvb) {
            float s[8], q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
            const int c = v * 8;
            int row = row0 + ty;
            for (; row + 3 * TY < row1; row += 4 * TY) {
                u16x8 xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = *(const u16x8*)gn_src(p, b, row + u * TY, c);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = bf2f(xv[u][e]);
                        s[e] += f;
                        q[e] += f * f;
                    }
            }
            for (; row < row1; row += TY) {
                const u16x8 xv = *(const u16x8*)gn_src(p, b, row, c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf2f(xv[e]);
                    s[e] += f;
                    q[e] += f * f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ssum[ty * p.C + c + e] = s[e];
                ssq[ty * p.C + c + e] = q[e];
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int g = tid >> 1;
        const float* src = (tid & 1) ? ssq : ssum;
        float a = 0.f;
        for (int t = 0; t < TY; ++t)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) a += src[t * p.C + c];
        p.partial[((size_t)b * p.nchunk + chunk) * 64 + tid] = a;
    }
}

// Apply pass.  Thread layout: lane column v = tid % cv owns the 8 channels [8 v, 8 v + 8) of every row it touches (cv = C / 8
// <= 512), row slot ty = tid / cv of TY; a workgroup of cv x TY threads walks its row chunk TY rows at a time.  Because a thread's
// channels never change, the per-channel scale / shift live in 16 REGISTERS (the first version kept a [C] x 2 table in LDS and read
// it at a stride of 8 floats per lane: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.82, profiles/r02/pmc_summary_final_build.json),
// and the loads of the first row batch -- x, gamma / beta, the ZeroSFT maps -- are issued BEFORE the statistics are reduced: the
// launch is a chain of dependent round trips (partials -> statistics -> rows -> store, ~8.7 us whatever the row loop does,
// profiles/r02/groupnorm_more_loads_in_flight_experiment.log) and this overlaps the two longest of them.
// MOD = the ZeroSFT form (modulation maps, optional lerp): its extra row operands live in registers only in that instantiation -- the
// plain form must stay small (<= 64 VGPRs: eight waves per SIMD), a streaming kernel lives on the loads the resident waves keep in
// flight (a first version with one 150-register body ran the 16384 x 320 map in 46.8 us instead of 31).
template <int NP, bool MOD, bool SILU>
__global__ __launch_bounds__(512) void gn_apply_kernel(const GnArgsN<NP> pp, int nchunk_apply, int rows_per_chunk_apply, int TY) {
    const GnArgs& p = pp.p[NP == 1 ? 0 : blockIdx.z];
    __shared__ float s_mean[32], s_rstd[32];
    __shared__ double s_part[4][64];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cv = p.C >> 3, cpg = p.C >> 5;
    const int v = tid % cv, ty = tid / cv, c = v * 8;
    const int row0 = chunk * rows_per_chunk_apply;
    const int row1 = min(p.HW, row0 + rows_per_chunk_apply);
    const bool lerp = MOD && p.cscale != 1.0f;
    constexpr int U = 2;   // rows in flight per thread (4 cost 83 VGPRs = 5 waves per SIMD instead of 8)
    // ---- loads that do not depend on the statistics: this thread's affine parameters and its first U rows
    const f32x4 g0 = *(const f32x4*)(p.gamma + c), g1 = *(const f32x4*)(p.gamma + c + 4);
    const f32x4 be0 = *(const f32x4*)(p.beta + c), be1 = *(const f32x4*)(p.beta + c + 4);
    constexpr int UM = MOD ? U : 1;
    u16x8 xv[U], gv[UM], bv[UM], rv[UM];
    auto load_rows = [&](int row) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = row + u * TY;
            if (r < row1) {
                xv[u] = *(const u16x8*)gn_src(p, b, r, c);
                if constexpr (MOD) {
                    const size_t mo = ((size_t)b * p.HW + r) * p.ldm + c;
                    gv[u] = *(const u16x8*)(p.mod_g + mo);
                    bv[u] = *(const u16x8*)(p.mod_b + mo);
                    if (lerp) {   // h_raw = cat[h_ori, h] (h BEFORE the zero_conv projection)
                        rv[u] = xv[u];
                        if (c < p.C1) {
                            if (p.x1raw) rv[u] = *(const u16x8*)(p.x1raw + ((size_t)b * p.HW + r) * p.ld1 + c);
                        } else if (p.x2raw) {
                            rv[u] = *(const u16x8*)(p.x2raw + ((size_t)b * p.HW + r) * p.ld2 + (c - p.C1));
                        }
                    }
                }
            }
        }
    };
    load_rows(row0 + ty);
    // ---- statistics of this batch element: (sum, sum of squares) per group, reduced in a fixed order (reproducible)
    if (tid < 256) {
        const int sv = tid & 63, sub = tid >> 6;
        if (p.part_u1) {
            // left behind by the producer GEMM / conv epilogues (GemmArgs::gn_part_out): per tile row and 10-channel unit of each source
            // tensor.  Group g = units [g * upg, (g + 1) * upg) of the concatenation; every group width on the path is a multiple of 10
            const int g = sv >> 1, st = sv & 1;
            const int upg = cpg / 10, U1 = p.C1 / 10, U2 = (p.C - p.C1) / 10;
            double a = 0.0;
            for (int uu = g * upg; uu < (g + 1) * upg; ++uu) {
                const bool first = uu < U1;
                const float* src = first ? p.part_u1 : p.part_u2;
                const int nch = first ? p.nch1 : p.nch2, Un = first ? U1 : U2, u = first ? uu : uu - U1;
                for (int k = sub; k < nch; k += 4) a += (double)src[(((size_t)b * nch + k) * Un + u) * 2 + st];
            }
            s_part[sub][sv] = a;
        } else if (!p.given) {
            double a = 0.0;
            for (int k = sub; k < p.nchunk; k += 4) a += (double)p.partial[((size_t)b * p.nchunk + k) * 64 + sv];
            s_part[sub][sv] = a;
        }
    }
    __syncthreads();
    if (p.given) {
        // externally pooled statistics (tiled VAE: SUPIR/utils/tilevae.py:524-553 custom_group_norm)
        if (tid < 32) {
            s_mean[tid] = p.given[((size_t)b * 32 + tid) * 2];
            s_rstd[tid] = rsqrtf(p.given[((size_t)b * 32 + tid) * 2 + 1] + p.eps);
        }
    } else if (tid < 32) {
        const double s = s_part[0][2 * tid] + s_part[1][2 * tid] + s_part[2][2 * tid] + s_part[3][2 * tid];
        const double q = s_part[0][2 * tid + 1] + s_part[1][2 * tid + 1] + s_part[2][2 * tid + 1] + s_part[3][2 * tid + 1];
        const double n = (double)p.HW * (double)cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    float sa[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / cpg;
        const float a = s_rstd[g] * (e < 4 ? g0[e] : g1[e - 4]);
        sa[e] = a;
        sb[e] = (e < 4 ? be0[e] : be1[e - 4]) - s_mean[g] * a;
    }
    if (ty >= TY) return;   // (never: the block is exactly cv x TY threads)
    for (int row = row0 + ty; row < row1; row += U * TY) {
        u16x8 ov[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = bf2f(xv[u][e]) * sa[e] + sb[e];
                if constexpr (SILU) y = silu_f(y);
                if constexpr (MOD) {
                    y = y * (bf2f(gv[u][e]) + 1.0f) + bf2f(bv[u][e]);
                    if (lerp) y = y * p.cscale + bf2f(rv[u][e]) * (1.0f - p.cscale);
                }
                ov[u][e] = f2bf(y);
            }
        }
        const int next = row + U * TY;
        const int here = row;
        if (next < row1) load_rows(next);   // the next batch is in flight while this one is stored
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = here + u * TY;
            if (r < row1) supir_store16(p.out + ((size_t)b * p.HW + r) * p.ldo + c, __builtin_bit_cast(f32x4, ov[u]));
        }
    }
}

// v1 of the apply pass (round 1-2: [C] x 2 scale / shift table in LDS, flat item loop), kept selectable (SUPIR_GN_APPLY=v1) for in-process A/B
template <int NP>
__global__ __launch_bounds__(256) void gn_apply_kernel_v1(const GnArgsN<NP> pp, int nchunk_apply, int rows_per_chunk_apply) {
    const GnArgs& p = pp.p[NP == 1 ? 0 : blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sA = (float*)smem_raw;  // [C] scale
    float* sB = sA + p.C;          // [C] shift
    __shared__ float s_mean[32], s_rstd[32];
    __shared__ double s_part[4][64];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cv = p.C >> 3, cpg = p.C >> 5;
    if (p.part_u1) {
        // statistics left behind by the producer GEMM / conv epilogues (GemmArgs::gn_part_out): (sum, sum of squares) per tile row
        // and 10-channel unit of each source tensor.  Group g = units [g * upg, (g + 1) * upg) of the concatenation; every group
        // width on the path is a multiple of 10 (dispatcher).  Same fixed order on every workgroup -> reproducible.
        const int v = tid & 63, sub = tid >> 6, g = v >> 1, st = v & 1;
        const int upg = cpg / 10, U1 = p.C1 / 10, U2 = (p.C - p.C1) / 10;
        double a = 0.0;
        for (int uu = g * upg; uu < (g + 1) * upg; ++uu) {
            const bool first = uu < U1;
            const float* src = first ? p.part_u1 : p.part_u2;
            const int nch = first ? p.nch1 : p.nch2, U = first ? U1 : U2, u = first ? uu : uu - U1;
            for (int k = sub; k < nch; k += 4) a += (double)src[(((size_t)b * nch + k) * U + u) * 2 + st];
        }
        s_part[sub][v] = a;
    } else if (!p.given) {
        // all 256 threads reduce the per-chunk partials (fixed order -> reproducible): value v = tid&63, chunks sub, sub+4, ...
        const int v = tid & 63, sub = tid >> 6;
        double a = 0.0;
        for (int k = sub; k < p.nchunk; k += 4) a += (double)p.partial[((size_t)b * p.nchunk + k) * 64 + v];
        s_part[sub][v] = a;
    }
    __syncthreads();
    if (p.given) {
        // externally pooled statistics (tiled VAE: SUPIR/utils/tilevae.py:524-553 custom_group_norm)
        if (tid < 32) {
            s_mean[tid] = p.given[((size_t)b * 32 + tid) * 2];
            s_rstd[tid] = rsqrtf(p.given[((size_t)b * 32 + tid) * 2 + 1] + p.eps);
        }
    } else if (tid < 32) {
        const double s = s_part[0][2 * tid] + s_part[1][2 * tid] + s_part[2][2 * tid] + s_part[3][2 * tid];
        const double q = s_part[0][2 * tid + 1] + s_part[1][2 * tid + 1] + s_part[2][2 * tid + 1] + s_part[3][2 * tid + 1];
        const double n = (double)p.HW * (double)cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        const int g = c / cpg;
        const float a = s_rstd[g] * p.gamma[c];
        sA[c] = a;
        sB[c] = p.beta[c] - s_mean[g] * a;
    }
    __syncthreads();
    const int row0 = chunk * rows_per_chunk_apply;
    const int row1 = min(p.HW, row0 + rows_per_chunk_apply);
    const int items = (row1 - row0) * cv;
    const bool lerp = p.cscale != 1.0f;
    for (int idx = tid; idx < items; idx += 256) {
        const int r = idx / cv, v = idx - r * cv;
        const int row = row0 + r, c = v * 8;
        const u16x8 xv = *(const u16x8*)gn_src(p, b, row, c);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = bf2f(xv[e]) * sA[c + e] + sB[c + e];
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = silu_f(y[e]);
        }
        if (p.mod_g) {
            const size_t mo = ((size_t)b * p.HW + row) * p.ldm + c;
            const u16x8 gv = *(const u16x8*)(p.mod_g + mo);
            const u16x8 bv = *(const u16x8*)(p.mod_b + mo);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = y[e] * (bf2f(gv[e]) + 1.0f) + bf2f(bv[e]);
            if (lerp) {
                u16x8 rv = xv;  // h_raw = cat[h_ori, h] (h BEFORE the zero_conv projection)
                if (c < p.C1) {
                    if (p.x1raw) rv = *(const u16x8*)(p.x1raw + ((size_t)b * p.HW + row) * p.ld1 + c);
                } else if (p.x2raw) {
                    rv = *(const u16x8*)(p.x2raw + ((size_t)b * p.HW + row) * p.ld2 + (c - p.C1));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = y[e] * p.cscale + bf2f(rv[e]) * (1.0f - p.cscale);
            }
        }
        u16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = f2bf(y[e]);
        supir_store16(p.out + ((size_t)b * p.HW + row) * p.ldo + c, __builtin_bit_cast(f32x4, ov));
    }
}

static int gn_prepare(GnArgs& a) {
    if (a.B <= 0 || a.HW <= 0 || a.C <= 0 || a.C % 32 != 0) return SUPIR_ERR_SHAPE;
    if (a.C % 8 != 0 || a.C1 % 8 != 0 || a.ld1 % 8 != 0 || a.ldo % 8 != 0) return SUPIR_ERR_SHAPE;
    if (a.C1 < a.C && (!a.x2 || a.ld2 % 8 != 0)) return SUPIR_ERR_ARG;
    if (a.mod_g && (!a.mod_b || a.ldm % 8 != 0)) return SUPIR_ERR_ARG;
    // (splitting small maps finer -- 8 rows per chunk, >= 2 workgroups per CU -- was measured neutral: 2.88 vs 3.09 ms per step)
    a.nchunk = a.HW / 64;
    if (a.nchunk < 1) a.nchunk = 1;
    if (a.nchunk > 128 && (long)a.HW * a.C <= (8L << 20)) a.nchunk = 128;  // UNet-sized maps: fewer partials to re-reduce
    if (a.nchunk > 1024) a.nchunk = 1024;   // workspace contract: B * 1024 * 64 floats
    a.rows_per_chunk = (a.HW + a.nchunk - 1) / a.nchunk;
    if (a.part_u1) {
        // producer-side statistics: no statistics launch.  Needs 10-channel units that tile every group and both sources.
        if (a.given || (a.C / 32) % 10 || a.C1 % 10 || a.nch1 <= 0 || (a.C1 < a.C && (!a.part_u2 || a.nch2 <= 0))) return SUPIR_ERR_ARG;
    } else if (!a.given) {
        const int cv = a.C / 8, cvb = cv < 256 ? cv : 256, TY = 256 / cvb;
        if ((size_t)TY * a.C * 2 * sizeof(float) > 64 * 1024) return SUPIR_ERR_SHAPE;
    }
    return SUPIR_OK;
}

template <int NP>
static int gn_launch(const GnArgs* a_in, hipStream_t st) {
    GnArgsN<NP> pp;
    for (int q = 0; q < NP; ++q) {
        pp.p[q] = a_in[q];
        const int rc = gn_prepare(pp.p[q]);
        if (rc != SUPIR_OK) return rc;
    }
    const GnArgs& a = pp.p[0];
    for (int q = 1; q < NP; ++q) {   // one grid for all problems: same geometry and the same way of getting the statistics
        const GnArgs& b = pp.p[q];
        if (b.B != a.B || b.HW != a.HW || b.C != a.C || (b.part_u1 == nullptr) != (a.part_u1 == nullptr) || (b.given == nullptr) != (a.given == nullptr))
            return SUPIR_ERR_SHAPE;
    }
    if (!a.part_u1 && !a.given) {
        const int cv = a.C / 8, cvb = cv < 256 ? cv : 256, TY = 256 / cvb;
        const size_t smem_stats = (size_t)TY * a.C * 2 * sizeof(float);
        SUPIR_LAUNCH(gn_stats_kernel<NP>, dim3(a.nchunk, a.B, NP), dim3(256), smem_stats, st, pp);
    }
    // cv x TY threads: every thread owns one 8-channel column; TY row slots so that the block has >= 256 threads where cv allows
    const int cv = a.C / 8;
    if (cv > 512) return SUPIR_ERR_SHAPE;
    int ty = (256 + cv - 1) / cv;
    if (ty > 512 / cv) ty = 512 / cv;
    if (ty < 1) ty = 1;
    // apply: ~64 KB of bf16 per workgroup iteration, at least 2 waves of workgroups when there is enough work
    long rows_target = (long)(32 * 1024) / (2L * a.C);
    if (rows_target < 8) rows_target = 8;
    // knob 2 (tools only): ONE row batch per workgroup (2 rows in flight per thread x TY row slots): a workgroup is a single
    // load -> compute -> store round trip instead of 3-4 dependent ones, at the price of more workgroups re-reducing the statistics
    if (supir_debug_knob_value(2)) rows_target = 2L * ty * supir_debug_knob_value(2);
    int nca = (int)((a.HW + rows_target - 1) / rows_target);
    if (nca > 2048 && !supir_debug_knob_value(2)) nca = 2048;
    if (nca > 65535) nca = 65535;
    const int rpc = (a.HW + nca - 1) / nca;
    nca = (a.HW + rpc - 1) / rpc;
    for (int q = 1; q < NP; ++q)
        if ((pp.p[q].mod_g == nullptr) != (a.mod_g == nullptr)) return SUPIR_ERR_SHAPE;
    const char* gn_env = getenv("SUPIR_GN_APPLY");   // read per launch: tools flip it between the variants of an in-process A/B
    if (gn_env && gn_env[0] == 'v' && gn_env[1] == '1') {
        SUPIR_LAUNCH(gn_apply_kernel_v1<NP>, dim3(nca, a.B, NP), dim3(256), (size_t)a.C * 2 * sizeof(float), st, pp, nca, rpc);
        return SUPIR_LAUNCH_STATUS();
    }
    for (int q = 1; q < NP; ++q)
        if (pp.p[q].act != a.act) return SUPIR_ERR_SHAPE;
    const dim3 grid(nca, a.B, NP), block(cv * ty);
    if (a.mod_g) {
        if (a.act == 1) SUPIR_LAUNCH((gn_apply_kernel<NP, true, true>), grid, block, 0, st, pp, nca, rpc, ty);
        else SUPIR_LAUNCH((gn_apply_kernel<NP, true, false>), grid, block, 0, st, pp, nca, rpc, ty);
    } else {
        if (a.act == 1) SUPIR_LAUNCH((gn_apply_kernel<NP, false, true>), grid, block, 0, st, pp, nca, rpc, ty);
        else SUPIR_LAUNCH((gn_apply_kernel<NP, false, false>), grid, block, 0, st, pp, nca, rpc, ty);
    }
    return SUPIR_LAUNCH_STATUS();
}

int supir_groupnorm_launch(GnArgs a, hipStream_t st) { return gn_launch<1>(&a, st); }

int supir_groupnorm_launch_n(const GnArgs* a, int n, hipStream_t st) {
    if (n == 1) return gn_launch<1>(a, st);
    if (n != 2) return SUPIR_ERR_SHAPE;
    return gn_launch<2>(a, st);
}

// per-(batch, group) sum / sum of squares of one tensor: [B][32][2] fp32 (chunk partials reduced in fp64, fixed order)
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ sums,
                                                         int nchunk) {
    const int b = blockIdx.x, v = threadIdx.x;
    double a = 0.0;
    for (int k = 0; k < nchunk; ++k) a += (double)partial[((size_t)b * nchunk + k) * 64 + v];
    sums[(size_t)b * 64 + v] = (float)a;
}

int supir_groupnorm_stats_launch(GnArgs a, float* sums_out, hipStream_t st) {
    if (a.B <= 0 || a.HW <= 0 || a.C <= 0 || a.C % 32 != 0 || a.C1 % 8 != 0 || a.ld1 % 8 != 0) return SUPIR_ERR_SHAPE;
    if (a.C1 < a.C && (!a.x2 || a.ld2 % 8 != 0)) return SUPIR_ERR_ARG;
    a.nchunk = a.HW / 64;
    if (a.nchunk < 1) a.nchunk = 1;
    if (a.nchunk > 1024) a.nchunk = 1024;
    a.rows_per_chunk = (a.HW + a.nchunk - 1) / a.nchunk;
    const int cv = a.C / 8, cvb = cv < 256 ? cv : 256, TY = 256 / cvb;
    const size_t smem_stats = (size_t)TY * a.C * 2 * sizeof(float);
    if (smem_stats > 64 * 1024) return SUPIR_ERR_SHAPE;
    GnArgsN<1> pp;
    pp.p[0] = a;
    SUPIR_LAUNCH(gn_stats_kernel<1>, dim3(a.nchunk, a.B), dim3(256), smem_stats, st, pp);
    SUPIR_LAUNCH(gn_finalize_kernel, dim3(a.B), dim3(64), 0, st, a.partial, sums_out, a.nchunk);
    return SUPIR_LAUNCH_STATUS();
}

// Producer partials in `unit`-channel units ([B][nchunk][C / unit][2], GemmArgs::gn_part_out) -> per-(batch, group) (mean, biased
// variance) [B][32][2]: the `given` input of the apply pass.  For maps whose producers leave THOUSANDS of tile rows behind (the VAE at
// 1024^2: 4096 rows of 256 pixels) -- there the apply kernel's own prologue, which has every workgroup re-reduce all partials, would read
// more than the tensor itself.  One workgroup per (group, batch); fp64 accumulation in a fixed order (reproducible).
__global__ __launch_bounds__(256) void gn_parts_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean_var, int nchunk,
                                                                int U, int upg, double inv_n) {
    __shared__ double s_s[256], s_q[256];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int k = tid; k < nchunk; k += 256) {
        const float* e = part + (((size_t)b * nchunk + k) * U + (size_t)g * upg) * 2;
        for (int u = 0; u < upg; ++u) {
            s += (double)e[2 * u];
            q += (double)e[2 * u + 1];
        }
    }
    s_s[tid] = s;
    s_q[tid] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) {
            s_s[tid] += s_s[tid + w];
            s_q[tid] += s_q[tid + w];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double mean = s_s[0] * inv_n;
        double var = s_q[0] * inv_n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mean_var[((size_t)b * 32 + g) * 2] = (float)mean;
        mean_var[((size_t)b * 32 + g) * 2 + 1] = (float)var;
    }
}

int supir_groupnorm_parts_finalize_launch(const float* part, int B, int nchunk, int C, int unit, int HW, float* mean_var, hipStream_t st) {
    if (B <= 0 || nchunk <= 0 || HW <= 0 || C <= 0 || C % 32 || unit <= 0 || (C / 32) % unit) return SUPIR_ERR_SHAPE;
    const int cpg = C / 32;
    SUPIR_LAUNCH(gn_parts_finalize_kernel, dim3(32, B), dim3(256), 0, st, part, mean_var, nchunk, C / unit, cpg / unit,
                 1.0 / ((double)HW * (double)cpg));
    return SUPIR_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over the channel dim of token-major bf16 [rows][ld]; one wave per row, row kept in registers.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int C, int ldx, int ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int cv = C >> 3;
    const bf16_t* xr = x + (size_t)row * ldx;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < cv) {
            const u16x8 xv = *(const u16x8*)(xr + vi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = bf2f(xv[e]);
                s += v[i][e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < cv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < cv) {
            const f32x4 g0 = *(const f32x4*)(gamma + vi * 8), g1 = *(const f32x4*)(gamma + vi * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(beta + vi * 8), b1 = *(const f32x4*)(beta + vi * 8 + 4);
            u16x8 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ov[e] = f2bf((v[i][e] - mean) * rstd * g0[e] + b0[e]);
                ov[4 + e] = f2bf((v[i][4 + e] - mean) * rstd * g1[e] + b1[e]);
            }
            supir_store16(yr + vi * 8, __builtin_bit_cast(f32x4, ov));
        }
    }
}

int supir_layernorm_launch(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, int rows, int C,
                           int ldx, int ldy, float eps, hipStream_t st) {
    if (rows <= 0 || C <= 0 || C % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || C > 4096) return SUPIR_ERR_SHAPE;
    const int nv = (C / 8 + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
    switch (nv) {
        case 1: SUPIR_LAUNCH(layernorm_kernel<1>, grid, block, 0, st, x, y, gamma, beta, rows, C, ldx, ldy, eps); break;
        case 2: SUPIR_LAUNCH(layernorm_kernel<2>, grid, block, 0, st, x, y, gamma, beta, rows, C, ldx, ldy, eps); break;
        case 3: SUPIR_LAUNCH(layernorm_kernel<3>, grid, block, 0, st, x, y, gamma, beta, rows, C, ldx, ldy, eps); break;
        case 4: SUPIR_LAUNCH(layernorm_kernel<4>, grid, block, 0, st, x, y, gamma, beta, rows, C, ldx, ldy, eps); break;
        default: SUPIR_LAUNCH(layernorm_kernel<8>, grid, block, 0, st, x, y, gamma, beta, rows, C, ldx, ldy, eps); break;
    }
    return SUPIR_LAUNCH_STATUS();
}
