// Shared device helpers for the SUPIR gfx950 (CDNA4) kernels.
// Everything here is wave64 / MFMA / LDS specific; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit element type of every activation / weight buffer.  The product library (libsupir_hip.so) is built with bfloat16;
// the SAME sources compiled with -DSUPIR_F16 give libsupir_hip_f16.so, in which every "bf16" buffer of the C ABI holds IEEE
// binary16 instead (the reference's default diff_dtype, options/SUPIR_v0.yaml:5, test.py:67-68): same kernels, same tiles, same
// fp32 accumulation, v_mfma_*_f16 in place of v_mfma_*_bf16.  Only the few helpers below know which of the two it is.
#ifdef SUPIR_F16
typedef _Float16 bf16_t;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x2 __attribute__((ext_vector_type(2)));
#define SUPIR_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SUPIR_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define SUPIR_ELEM_NAME "f16"
#else
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define SUPIR_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SUPIR_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define SUPIR_ELEM_NAME "bf16"
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));

#define SUPIR_EXPERIMENTAL 1          // the library always builds (and exports) the experimental section of the header
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

// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
typedef float supir_f32x2 __attribute__((ext_vector_type(2)));
#ifdef SUPIR_F16
// fp32 -> fp16, round to nearest even (v_cvt_f16_f32; values beyond 65504 become +-inf exactly as torch's .half() does)
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {   // (lo, hi) -> two fp16 in one dword
    const supir_f32x2 v = {lo, hi};
    const bf16x2 b = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (_Float16)f); }
__device__ __forceinline__ float bf2f(u16 h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float bflo2f(uint32_t w) { return bf2f((u16)(w & 0xffffu)); }
__device__ __forceinline__ float bfhi2f(uint32_t w) { return bf2f((u16)(w >> 16)); }
#else
typedef __bf16 supir_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {   // (lo, hi) -> two bf16 in one dword
    const supir_f32x2 v = {lo, hi};
    const supir_bf16x2 b = __builtin_convertvector(v, supir_bf16x2);
    return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ u16 f2bf(float f) { return (u16)(f2bf_pk(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bflo2f(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi2f(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }
#endif

// acc + lo + hi of a packed pair of 16-bit elements AS STORED (v_dot2c_f32_bf16 / v_dot2_f32_f16 against (1, 1)): one instruction for two
// terms of a softmax row sum, and the sum is then the sum of the rounded probabilities the P.V product actually uses
__device__ __forceinline__ float pair_sum_acc(uint32_t packed, float acc) {
#ifdef SUPIR_F16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, 0x3c003c00u), acc, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, 0x3f803f80u), acc, false);
#endif
}

// acc + a.lo * b.lo + a.hi * b.hi over packed pairs of 16-bit elements (v_dot2c_f32_bf16 / v_dot2_f32_f16): two MACs per VALU instruction
__device__ __forceinline__ float dot2_acc(uint32_t a, uint32_t b, float acc) {
#ifdef SUPIR_F16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
#endif
}

// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division: the division expands to ~10 VALU
// instructions per element -- more than the rest of the GroupNorm + SiLU apply pass put together -- and the result is rounded to bf16
// (2^-9) right after.  exp2(+inf) -> rcp(inf) = 0 -> -0 for very negative x, as before.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
// erf GELU (F.gelu default; reference: sgm/modules/attention.py:91).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7,
// below fp32 resolution of the 1 + erf term and far below the bf16 output): a dozen instructions instead of the libm erff call
// in the epilogue of the widest GEMM of every transformer block.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float y = 1.0f - pl * t * __expf(-ax * ax);
    return copysignf(y, x);
}
// QuickGELU x * sigmoid(1.702 x): the activation of OpenAI CLIP text towers (transformers CLIPTextModel "quick_gelu")
__device__ __forceinline__ float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.4554669595930156f)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// The same function as x * sigmoid(x * (c0 + c1 x^2 + c2 x^4)): a minimax fit of Phi(x) = (1 + erf(x / sqrt 2)) / 2 (scipy, [-8, 8]) with
// max |gelu error| 2.5e-5 ABSOLUTE -- below half a bf16 ulp of the result for |result| >= 0.013 and 100x below the bf16 rounding of
// typical activations; 8 VALU + 2 transcendental instructions instead of 15 + 2.  The argument is clamped to [-7, 7] (the fitted
// polynomial turns over beyond |x| = 7.25; sigmoid is saturated to 1 - 2e-11 there).  Coefficients carry the -log2(e) of exp2.
__device__ __forceinline__ float gelu_fast_f(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
    const float u = xc * xc;
    float pl = fmaf(1.01426305e-03f, u, -1.06775724e-01f);   // -log2(e) * (-7.03033579e-04, 7.40112920e-02)
    pl = fmaf(pl, u, -2.30112133f);                          // -log2(e) * 1.59501577
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xc * pl));
}

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

// Cache policy of the big OUTPUT stores (16 / 8 bytes per lane, row-contiguous: GEMM / convolution / attention / GroupNorm / LayerNorm
// epilogues).  A build-time PROBE switch (tools/store_policy_ab.py builds one extra library per policy and swaps it in inside one process);
// the product build is 0 = plain C++ stores.  1 = `sc1` (agent-scope, write-through: the line does not stay dirty in the XCD's L2, so the
// release at the end of the kernel has nothing to write back -- the guide's "boundary" / "publish-large" rows), 2 = nt, 3 = sc0 sc1.
// Measured on the replayed 1024^2 step, interleaved, bitwise-equal outputs (profiles/r06/store_policy_ab.json): plain 27.33 ms, sc1
// 27.57, sc0 sc1 27.45 -- the end-of-kernel write-back of 5-16 MB is not what a boundary costs here; plain stays.  (A first build without
// the s_nop below produced a few corrupt elements per launch, NaN downstream -- and ran the step in 25.1 ms: operands that do not toggle
// let the chip clock ~10 % higher.  Never read a speed-up off a variant whose output has not been compared.)
#ifndef SUPIR_STORE_POLICY
#define SUPIR_STORE_POLICY 0
#endif
// (The stores are inline assembly -- the compiler offers no 16-byte store with a scope bit -- so its hazard recogniser does not see a
// store: a VALU write to the data registers of a > 8-byte store within one wait state is the programmer's to avoid, hence the s_nop.)
__device__ __forceinline__ void supir_store16(void* p, f32x4 v) {
#if SUPIR_STORE_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif SUPIR_STORE_POLICY == 2
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif SUPIR_STORE_POLICY == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
    *(f32x4*)p = v;
#endif
}
__device__ __forceinline__ void supir_store8(void* p, u32x2 v) {
#if SUPIR_STORE_POLICY == 1
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif SUPIR_STORE_POLICY == 2
    asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
#elif SUPIR_STORE_POLICY == 3
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    *(u32x2*)p = v;
#endif
}

// XCD-aware bijective remap of a linear workgroup id: block b runs on XCD b%8; give every XCD a
// contiguous chunk of the logical id space so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
