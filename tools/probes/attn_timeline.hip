// Per-phase cycle breakdown of the flash-attention KV loop (s_memtime) at the shapes of one 1024^2 step.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSUPIR_ATTN_TIMELINE tools/probes/attn_timeline.hip \
//        supir_amd/csrc/{gemm,attention,norm,edge,api}.hip -o tools/probes/attn_timeline
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../supir_amd/csrc/kernels.h"

extern "C" void supir_atl_set(unsigned long long* p);
int supir_attn_launch(const AttnArgs& a, hipStream_t st);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
    struct Shape { int B, H, Tq, Tk; };
    const Shape shapes[] = {{2, 20, 1024, 1024}, {2, 10, 4096, 4096}, {2, 20, 1024, 77}, {1, 1, 16384, 16384}};
    unsigned long long* tl;
    CK(hipMalloc(&tl, (size_t)(1 << 20) * 8));
    supir_atl_set(tl);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const int C = s.H * 64, Tp = (s.Tk + 63) / 64 * 64;
        bf16_t *Q, *K, *Vt, *O;
        CK(hipMalloc(&Q, (size_t)s.B * s.Tq * C * 2));
        CK(hipMalloc(&K, (size_t)s.B * s.Tk * C * 2));
        CK(hipMalloc(&Vt, (size_t)s.B * C * Tp * 2));
        CK(hipMalloc(&O, (size_t)s.B * s.Tq * C * 2));
        CK(hipMemset(Q, 0x3c, (size_t)s.B * s.Tq * C * 2));
        CK(hipMemset(K, 0x3c, (size_t)s.B * s.Tk * C * 2));
        CK(hipMemset(Vt, 0x3c, (size_t)s.B * C * Tp * 2));
        AttnArgs a = {};
        a.Q = Q; a.K = K; a.Vt = Vt; a.O = O;
        a.B = s.B; a.H = s.H; a.Tq = s.Tq; a.Tk = s.Tk;
        a.ldq = C; a.ldk = C; a.ldvt = Tp; a.ldo = C;
        a.scale_log2e = 0.125f * 1.4426950408889634f;
        for (int i = 0; i < 5; ++i) supir_attn_launch(a, st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) supir_attn_launch(a, st);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const int blocks4 = ((s.Tq + 127) / 128) * s.H * s.B;
        const int nblk = blocks4, nwv = 4;
        const int nw = nblk * nwv, nt = (s.Tk + 63) / 64;
        std::vector<unsigned long long> h((size_t)nw * 8);
        CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
        double sum[8] = {0}, mx_total = 0, mx_loop = 0;
        for (int w = 0; w < nw; ++w) {
            for (int c = 0; c < 8; ++c) sum[c] += (double)h[(size_t)w * 8 + c];
            if ((double)h[(size_t)w * 8 + 7] > mx_total) mx_total = (double)h[(size_t)w * 8 + 7];
            if ((double)h[(size_t)w * 8 + 0] > mx_loop) mx_loop = (double)h[(size_t)w * 8 + 0];
        }
        // sync = vmcnt wait + barrier, A0/A1 = Q.K^T(next half) beside exp(this half), B0/B1 = P.V beside the row maximum + rebase
        printf("B=%d H=%d Tq=%d Tk=%d blocks=%d x %d waves | event %.1f us, TF %.0f | per KV tile (%d): sync %.0f  A0 %.0f  B0 %.0f  A1 %.0f  B1 %.0f "
               "(sum %.0f) | prologue+loop %.0f (max %.0f) epilogue %.0f | total avg %.0f max %.0f ticks\n",
               s.B, s.H, s.Tq, s.Tk, nblk, nwv, ms * 1000 / 20, 4.0 * s.B * s.H * s.Tq * s.Tk * 64 / (ms * 1e-3 / 20) / 1e12, nt,
               sum[1] / nw / nt, sum[2] / nw / nt, sum[3] / nw / nt, sum[4] / nw / nt, sum[5] / nw / nt,
               (sum[1] + sum[2] + sum[3] + sum[4] + sum[5]) / nw / nt, sum[0] / nw, mx_loop, sum[6] / nw, sum[7] / nw, mx_total);
        fflush(stdout);
        CK(hipFree(Q)); CK(hipFree(K)); CK(hipFree(Vt)); CK(hipFree(O));
    }
    return 0;
}
