// What does the matrix pipe sustain on THIS chip when nothing else runs and the operands are not zeros?  (round 6, late)
// Register-only streams of v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 (no LDS, no loads in the loop), 256 workgroups x 8 waves
// (two per SIMD), independent accumulators so the pipe is issue-bound, run for ~0.3 s so the power management settles; operands either
// ZERO or RANDOM finite bf16 (per-lane hash).  The package runs into its ~1.4 kW limit on random operands (profiles/r06/power_during_bench.json):
// the figure this prints for random data is the practical ceiling a bf16 kernel can be held against, the 2.5 PF nameplate is the zero-operand one.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power_probe.hip -o tools/probes/mfma_power_probe && tools/probes/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// two finite bf16 in [-2, 2) with random mantissas / signs: exponent field 0x3f (values 0.5 .. 2) or 0x3e
__device__ __forceinline__ uint32_t rnd_bf16_pair(uint32_t seed) {
    const uint32_t h = hash32(seed);
    const uint32_t lo = (h & 0x80ffu) | 0x3f00u, hi = ((h >> 16) & 0x80ffu) | 0x3e80u;
    return lo | (hi << 16);
}

template <bool BIG, bool RANDOM>
__global__ __launch_bounds__(512, 2) void probe(float* sink, int iters) {
    const uint32_t t = blockIdx.x * 512 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x4 ra, rb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ra[k] = RANDOM ? rnd_bf16_pair(t * 64 + i * 8 + k) : 0u;
            rb[k] = RANDOM ? rnd_bf16_pair(t * 64 + i * 8 + k + 4 + 0x9e3779b9u) : 0u;
        }
        a[i] = __builtin_bit_cast(bf16x8, ra);
        b[i] = __builtin_bit_cast(bf16x8, rb);
    }
    if constexpr (BIG) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 12345.678f) sink[t] = s;
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + u) & 3], b[i & 3], acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][r];
        if (s == 12345.678f) sink[t] = s;
    }
}

template <bool BIG, bool RANDOM>
static int run(const char* name, float* sink) {
    const int iters = 20000;                       // 16 (32x32) / 32 (16x16) MFMAs per iteration and wave
    const double flop_per_launch = 256.0 * 8 * iters * (BIG ? 16 * 32768.0 : 32 * 16384.0);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<BIG, RANDOM>), dim3(256), dim3(512), 0, 0, sink, iters);   // warm-up
    CHECK(hipDeviceSynchronize());
    const int reps = 40;                           // ~0.2-0.4 s of back-to-back launches: long enough for the power limit to act
    double first = 0, last = 0;
    for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<BIG, RANDOM>), dim3(256), dim3(512), 0, 0, sink, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        (pass == 0 ? first : last) = flop_per_launch * reps / (ms * 1e-3) / 1e12;
    }
    printf("%-34s  %7.0f TFLOP/s (first 40 launches)  %7.0f TFLOP/s (next 40)\n", name, first, last);
    fflush(stdout);
    return 0;
}

int main() {
    float* sink;
    CHECK(hipMalloc(&sink, 256 * 512 * sizeof(float)));
    if (run<true, false>("32x32x16 bf16, zero operands", sink)) return 1;
    if (run<true, true>("32x32x16 bf16, random operands", sink)) return 1;
    if (run<false, false>("16x16x32 bf16, zero operands", sink)) return 1;
    if (run<false, true>("16x16x32 bf16, random operands", sink)) return 1;
    if (run<true, true>("32x32x16 bf16, random (again)", sink)) return 1;
    if (run<false, true>("16x16x32 bf16, random (again)", sink)) return 1;
    return 0;
}
