// Shared device helpers for the SUPIR gfx950 (CDNA4) kernels.
// Everything here is wave64 / MFMA / LDS specific; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));

#include "../../include/supir_hip.h"  // error codes shared with the C ABI

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process (PyTorch): clear whatever
// is pending before our launch so that the status we report afterwards is the status of OUR launch only.
#define SUPIR_LAUNCH(...)            \
    do {                             \
        (void)hipGetLastError();     \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

// status of the launch just issued: SUPIR_OK or SUPIR_ERR_HIP (the hipError_t is kept for supir_last_hip_error())
int supir_note_hip_status(hipError_t e);
#define SUPIR_LAUNCH_STATUS() supir_note_hip_status(hipGetLastError())

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ u16 f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, F.gelu default  (reference: sgm/modules/attention.py:91)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// async global -> LDS copy of 16 B per lane. LDS destination is wave-uniform base + lane*16
// (hardware adds the lane offset); the global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware bijective remap of a linear workgroup id: block b runs on XCD b%8; give every XCD a
// contiguous chunk of the logical id space so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
