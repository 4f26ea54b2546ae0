// Issue-cost probe for gfx950: what a wave pays for VALU / transcendental fillers placed beside v_mfma_f32_32x32x16_bf16,
// with one and with two waves per SIMD.  Every variant is a fixed inline-asm block (registers v0..v63 by hand), so the
// compiler cannot reorder anything; cycles are s_memtime deltas of wave 0, cross-checked against event time.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/issue_probe.hip -o tools/probes/issue_probe && tools/probes/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define MA "v_mfma_f32_32x32x16_bf16 v[0:15], v[48:51], v[52:55], v[0:15]\n"
#define MB "v_mfma_f32_32x32x16_bf16 v[16:31], v[48:51], v[52:55], v[16:31]\n"
#define FMA(i) "v_fma_f32 v" #i ", v60, v61, v62\n"
#define EXP(i) "v_exp_f32 v" #i ", v62\n"
#define ADDC(i) "v_add_f32 v63, v63, v" #i "\n"
#define CVT(d, i, j) "v_cvt_pk_bf16_f32 v" #d ", v" #i ", v" #j "\n"
#define PKF(i, j) "v_pk_fma_f32 v[" #i ":" #j "], v[60:61], v[60:61], v[60:61]\n"
#define MAX3(i, j) "v_max3_f32 v59, v59, v" #i ", v" #j "\n"
#define F4A FMA(32) FMA(33) FMA(34) FMA(35)
#define F4B FMA(36) FMA(37) FMA(38) FMA(39)
#define F4C FMA(40) FMA(41) FMA(42) FMA(43)
#define F4D FMA(44) FMA(45) FMA(46) FMA(47)
#define E2A EXP(32) EXP(33)
#define E2B EXP(34) EXP(35)
#define E4A EXP(32) EXP(33) EXP(34) EXP(35)
#define E4B EXP(36) EXP(37) EXP(38) EXP(39)
#define E4C EXP(40) EXP(41) EXP(42) EXP(43)
#define E4D EXP(44) EXP(45) EXP(46) EXP(47)
// the stage-A slot of the pipelined attention kernel: 4 x (fma, exp, dependent row-sum add) + 2 bf16 packs
#define SLOTA FMA(32) EXP(36) FMA(33) EXP(37) FMA(34) EXP(38) FMA(35) EXP(39) ADDC(36) ADDC(37) ADDC(38) ADDC(39) CVT(56, 36, 37) CVT(57, 38, 39)
#define SLOTB FMA(40) EXP(44) FMA(41) EXP(45) FMA(42) EXP(46) FMA(43) EXP(47) ADDC(44) ADDC(45) ADDC(46) ADDC(47) CVT(58, 44, 45) CVT(57, 46, 47)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"

#define VARIANTS(X) \
    X(0, "2 MFMA (two accumulators)", MA MB) \
    X(1, "2 x (MFMA + 4 fma)", MA F4A MB F4B) \
    X(2, "2 x (MFMA + 8 fma)", MA F4A F4B MB F4C F4D) \
    X(3, "2 x (MFMA + 12 fma)", MA F4A F4B F4C MB F4D F4A F4B) \
    X(4, "2 x (MFMA + 16 fma)", MA F4A F4B F4C F4D MB F4A F4B F4C F4D) \
    X(5, "2 x (MFMA + 2 exp)", MA E2A MB E2B) \
    X(6, "2 x (MFMA + 4 exp)", MA E4A MB E4B) \
    X(7, "2 x (MFMA + 8 exp)", MA E4A E4B MB E4C E4D) \
    X(8, "16 fma, no MFMA", F4A F4B F4C F4D) \
    X(9, "16 exp, no MFMA", E4A E4B E4C E4D) \
    X(10, "2 x (MFMA + attention slot: 4 fma 4 exp 4 chained add 2 cvt)", MA SLOTA MB SLOTB) \
    X(11, "2 x (MFMA + 2 pk_fma)", MA PKF(32, 33) PKF(34, 35) MB PKF(36, 37) PKF(38, 39)) \
    X(12, "2 dependent MFMA (one accumulator)", MA MA) \
    X(13, "2 x (dependent MFMA + 8 fma)", MA F4A F4B MA F4C F4D) \
    X(14, "2 x (MFMA + 4 fma 4 exp interleaved)", MA FMA(32) EXP(36) FMA(33) EXP(37) FMA(34) EXP(38) FMA(35) EXP(39) MB FMA(40) EXP(44) FMA(41) EXP(45) FMA(42) EXP(46) FMA(43) EXP(47)) \
    X(15, "2 x (MFMA + 4 chained add)", MA ADDC(36) ADDC(37) ADDC(38) ADDC(39) MB ADDC(44) ADDC(45) ADDC(46) ADDC(47)) \
    X(16, "2 x (MFMA + 4 max3)", MA MAX3(32, 33) MAX3(34, 35) MAX3(36, 37) MAX3(38, 39) MB MAX3(40, 41) MAX3(42, 43) MAX3(44, 45) MAX3(46, 47)) \
    X(17, "16 chained add, no MFMA", ADDC(32) ADDC(33) ADDC(34) ADDC(35) ADDC(36) ADDC(37) ADDC(38) ADDC(39) ADDC(40) ADDC(41) ADDC(42) ADDC(43) ADDC(44) ADDC(45) ADDC(46) ADDC(47)) \
    X(18, "2 x (MFMA + 6 fma)", MA F4A FMA(36) FMA(37) MB F4C FMA(44) FMA(45)) \
    X(19, "2 x (MFMA + 3 exp + 3 fma)", MA EXP(32) FMA(36) EXP(33) FMA(37) EXP(34) FMA(38) MB EXP(40) FMA(44) EXP(41) FMA(45) EXP(42) FMA(46))

template <int V>
__global__ void probe(unsigned long long* out, int iters) {
    asm volatile("v_mov_b32 v60, 1.0\nv_mov_b32 v61, 0.5\nv_mov_b32 v62, 0.25\nv_mov_b32 v63, 0\nv_mov_b32 v59, 0\n"
                 "v_mov_b32 v48, 0\nv_mov_b32 v49, 0\nv_mov_b32 v50, 0\nv_mov_b32 v51, 0\nv_mov_b32 v52, 0\nv_mov_b32 v53, 0\nv_mov_b32 v54, 0\nv_mov_b32 v55, 0\n" ::: CLOB);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define X(id, name, body) if constexpr (V == id) asm volatile(body body body body ::: CLOB);
        VARIANTS(X)
#undef X
    }
    asm volatile("s_nop 7\ns_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int V>
int run(const char* name, unsigned long long* dout, int iters) {
    for (int threads = 256; threads <= 512; threads *= 2) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(probe<V>, dim3(256), dim3(threads), 0, 0, dout, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned long long h[256];
        CHECK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
        double avg = 0;
        for (int i = 0; i < 256; ++i) avg += (double)h[i];
        avg /= 256;
        const double groups = (double)iters * 4;   // a "group" = one body = two MFMA slots (or 16 plain instructions)
        printf("%-66s waves/SIMD %d : %7.1f memtime ticks / group, %7.1f ns / group\n", name, threads / 256, avg / groups, ms * 1e6 / groups);
    }
    return 0;
}

int main() {
    unsigned long long* dout;
    CHECK(hipMalloc(&dout, 256 * sizeof(unsigned long long)));
    const int iters = 20000;
#define X(id, name, body) if (run<id>(name, dout, iters)) return 1;
    VARIANTS(X)
#undef X
    return 0;
}
