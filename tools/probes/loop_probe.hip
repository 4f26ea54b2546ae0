// What does one iteration of the 128 x 80 / two-K-group loop of csrc/gemm16.hip cost, ingredient by ingredient?  (round 6)
// 256 workgroups x 512 threads (8 waves = 2 per SIMD), 104 KB of LDS per workgroup (one workgroup per CU), per wave and iteration:
//   20 x v_mfma_f32_16x16x32_bf16 (10 accumulators, two passes)                          -- always
//   R : 14 x ds_read_b128 feeding the MFMAs of the SAME iteration (the one-barrier loop's fragment reads)
//   P : the same reads feeding the NEXT iteration's MFMAs (software-pipelined: no wait in front of the MFMAs)
//   G : 7 x global_load_lds_dwordx4 (1 KB per wave instruction, rows of an L2-resident buffer) into an LDS ring, counted vmcnt
//   B : one s_barrier per iteration
// Every combination is timed with s_memtime (cycles per iteration per wave, mean over waves) and with events.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/loop_probe.hip -o tools/probes/loop_probe && tools/probes/loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <bool R, bool P, bool G, bool B, int NLOADS>
__global__ __launch_bounds__(512, 1) void probe(const char* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ out, int iters, size_t src_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int quad = lane >> 4, l15 = lane & 15, sw = (l15 >> 1) & 7;
    // LDS: [0, 26.6 KB) x 3 ring stages of (128 + 80) rows x 128 B per K group (two groups) -- the real layout; reads use the real swizzle
    const int kg = wave >> 2, w4 = wave & 3;
    char* ring = smem + kg * (3 * 26624);
    for (int i = threadIdx.x; i < 104 * 1024 / 4; i += 512) ((float*)smem)[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][2][2], bfr[2][2][5];      // [set][kk][frag]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[s][kk][i] = *(const bf16x8*)(ring + (w4 * 32 + i * 16 + l15) * 128 + (((4 * kk + quad) ^ sw) * 16));
#pragma unroll
            for (int j = 0; j < 5; ++j) bfr[s][kk][j] = *(const bf16x8*)(ring + 16384 + (j * 16 + l15) * 128 + (((4 * kk + quad) ^ sw) * 16));
        }
    // global source: every workgroup streams its own 1 MB window of an L2 / MALL resident buffer, 8 rows x 128 B per instruction
    const char* gsrc = src + ((size_t)blockIdx.x * (1u << 20)) % src_bytes + (size_t)(lane >> 3) * 2560 + (lane & 7) * 16;
    size_t goff = (size_t)wave * 8 * 2560;
    int buf = 0;
    auto reads = [&](int set, int stage) {
        const char* sT = ring + stage * 26624;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[set][kk][i] = *(const bf16x8*)(sT + (w4 * 32 + i * 16 + l15) * 128 + (((4 * kk + quad) ^ sw) * 16));
#pragma unroll
            for (int j = 0; j < 5; ++j) bfr[set][kk][j] = *(const bf16x8*)(sT + 16384 + (j * 16 + l15) * 128 + (((4 * kk + quad) ^ sw) * 16));
        }
    };
    auto loads = [&](int stage, int q0, int q1) {
#pragma unroll
        for (int q = 0; q < NLOADS; ++q)
            if (q >= q0 && q < q1) glds16(gsrc + ((goff + (size_t)q * 64 * 2560) & ((1u << 20) - 1)), ring + stage * 26624 + ((w4 + 4 * q) % 26) * 1024);
    };
    auto mfmas = [&](int set, int kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[set][kk][j], af[set][kk][i], acc[i][j], 0, 0, 0);
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            if constexpr (G) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");      // the previous iteration's loads stay in flight
            if constexpr (B) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            const int cur = P ? par : 0, nxt = P ? par ^ 1 : 0;
            if constexpr (R || P) reads(nxt, buf);
            if constexpr (G) loads(buf == 0 ? 2 : buf - 1, 0, NLOADS / 2);
            mfmas(cur, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (G) loads(buf == 0 ? 2 : buf - 1, NLOADS / 2, NLOADS);
            mfmas(cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
            asm volatile("" ::: "memory");
            buf = buf == 2 ? 0 : buf + 1;
            goff += 64 * NLOADS * 2560;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <bool R, bool P, bool G, bool B, int NLOADS>
static int run(const char* name, const char* src, size_t src_bytes, float* sink, unsigned long long* out, int nwg) {
    const int iters = 200;
    auto k = probe<R, P, G, B, NLOADS>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = 2 * 3 * 26624 + 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), lds, 0, src, sink, out, iters, src_bytes);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), lds, 0, src, sink, out, iters, src_bytes);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nwg * 8);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    unsigned long long mx = 0;
    for (auto v : h) { sum += (double)v; mx = v > mx ? v : mx; }
    const double cyc = sum / h.size() / iters;
    printf("%-58s %7.0f cycles / iteration (max wave %7.0f); event %7.1f us -> %5.2f GHz; MFMA 640 ideal -> %4.0f %%\n", name, cyc, (double)mx / iters, ms * 1e3,
           sum / h.size() / (ms * 1e3) / 1e3, 100.0 * 640.0 / cyc);
    return 0;
}

int main() {
    const size_t src_bytes = 64u << 20;
    char* src;
    float* sink;
    unsigned long long* out;
    CHECK(hipMalloc(&src, src_bytes + (4u << 20)));
    CHECK(hipMemset(src, 0x11, src_bytes + (4u << 20)));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMalloc(&out, 256 * 8 * 8));
    for (int nwg : {256, 32}) {
        printf("---- %d workgroups x 8 waves, 20 MFMA 16x16x32 per wave and iteration (2 waves per SIMD: 640 cycles of matrix pipe per iteration)\n", nwg);
        run<false, false, false, false, 7>("MFMA only", src, src_bytes, sink, out, nwg);
        run<false, false, false, true, 7>("MFMA + barrier", src, src_bytes, sink, out, nwg);
        run<true, false, false, false, 7>("MFMA + 14 ds_read_b128 (same iteration)", src, src_bytes, sink, out, nwg);
        run<true, false, false, true, 7>("MFMA + 14 ds_read_b128 (same iteration) + barrier", src, src_bytes, sink, out, nwg);
        run<false, true, false, false, 7>("MFMA + 14 ds_read_b128 (next iteration's)", src, src_bytes, sink, out, nwg);
        run<false, true, false, true, 7>("MFMA + 14 ds_read_b128 (next iteration's) + barrier", src, src_bytes, sink, out, nwg);
        run<false, false, true, false, 7>("MFMA + 7 global_load_lds", src, src_bytes, sink, out, nwg);
        run<false, false, true, false, 4>("MFMA + 4 global_load_lds", src, src_bytes, sink, out, nwg);
        run<false, false, true, true, 7>("MFMA + 7 global_load_lds + barrier", src, src_bytes, sink, out, nwg);
        run<true, false, true, true, 7>("one-barrier loop: reads (same it.) + 7 loads + barrier", src, src_bytes, sink, out, nwg);
        run<false, true, true, true, 7>("software-pipelined: reads (next it.) + 7 loads + barrier", src, src_bytes, sink, out, nwg);
        run<true, false, true, true, 4>("one-barrier loop with 4 loads (halo form)", src, src_bytes, sink, out, nwg);
        run<false, true, true, true, 4>("software-pipelined with 4 loads (halo form)", src, src_bytes, sink, out, nwg);
    }
    return 0;
}
