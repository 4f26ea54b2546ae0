// Per-CU L2 -> {LDS via global_load_lds, VGPR via global_load_dwordx4} streaming bandwidth probe (gfx950).
// Each workgroup re-reads its own region (64 KB or 1 MB, L2 resident) ITERS times.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ src, uint4* __restrict__ sink, int region_vec, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const uint4* base = src + (size_t)blockIdx.x * region_vec;
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (int v = 0; v < region_vec; v += 256 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint4* p = base + v + u * 256 + tid;
                if (MODE == 0) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                        (__attribute__((address_space(3))) void*)(smem + ((u * 256 + wave * 64) * 16) % 32768), 16, 0, 0);
                } else {
                    const uint4 x = *p;
                    acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w;
                }
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 0) { __syncthreads(); acc = *(uint4*)(smem + tid * 16); }
    if (acc.x == 0x12345678u) sink[blockIdx.x * 256 + tid] = acc;
}

int main() {
    const int region_bytes_opts[2] = {64 * 1024, 1024 * 1024};
    for (int ro = 0; ro < 2; ++ro) {
        const int region_vec = region_bytes_opts[ro] / 16;
        for (int bpc = 1; bpc <= 4; bpc *= 2) {
            const int blocks = 256 * bpc;
            uint4 *src, *sink;
            CHECK(hipMalloc(&src, (size_t)blocks * region_vec * 16));
            CHECK(hipMalloc(&sink, (size_t)blocks * 256 * 16));
            CHECK(hipMemset(src, 1, (size_t)blocks * region_vec * 16));
            const int iters = ro == 0 ? 64 : 8;
            for (int mode = 0; mode < 2; ++mode) {
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                for (int rep = 0; rep < 2; ++rep) {
                    CHECK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 32768, 0, src, sink, region_vec, iters);
                    else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 32768, 0, src, sink, region_vec, iters);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                }
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)blocks * region_vec * 16 * iters;
                printf("region %4d KB  blocks/CU %d  mode %s : %.1f us  %.2f TB/s  (%.1f GB/s/CU)\n", region_bytes_opts[ro] / 1024, bpc,
                       mode == 0 ? "glds->LDS " : "load->VGPR", ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
            }
            CHECK(hipFree(src)); CHECK(hipFree(sink));
        }
    }
    return 0;
}
