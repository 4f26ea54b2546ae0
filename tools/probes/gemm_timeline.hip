// Per-phase cycle breakdown of the GEMM main loop (s_memtime), for the shapes of one 1024^2 step.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSUPIR_GEMM_TIMELINE tools/probes/gemm_timeline.hip \
//        supir_amd/csrc/{gemm,attention,norm,edge,api}.hip -o tools/probes/gemm_timeline
// Columns (cycles per wave, averaged over all waves of the launch): prologue | per K step: vmcnt wait, barrier,
// load issue, LDS reads + MFMA | epilogue | total; plus the launch's wall time (first wave start -> last wave end) and event time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../supir_amd/csrc/kernels.h"

extern "C" void supir_tl_set(unsigned long long* p);
int supir_gemm_launch(const GemmArgs& a, bool conv, hipStream_t st, int force_tile);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
    struct Shape { int M, N, K; };
    const Shape shapes[] = {{2048, 1280, 1280}, {2048, 10240, 1280}, {2048, 1280, 5120}, {8192, 640, 640}, {8192, 5120, 640}};
    const int tiles[] = {0, 1, 3};
    const int tb[7][3] = {{128, 128, 4}, {128, 64, 4}, {64, 128, 4}, {64, 64, 4}, {256, 128, 8}, {256, 256, 8}, {256, 128, 4}};
    unsigned long long* tl;
    const size_t tl_n = (size_t)1 << 20;
    CK(hipMalloc(&tl, tl_n * 8));
    supir_tl_set(tl);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        bf16_t *A, *W, *C;
        float* bias;
        CK(hipMalloc(&A, (size_t)s.M * s.K * 2));
        CK(hipMalloc(&W, (size_t)s.N * s.K * 2));
        CK(hipMalloc(&C, (size_t)s.M * s.N * 2));
        CK(hipMalloc(&bias, (size_t)s.N * 4));
        CK(hipMemset(A, 0x3c, (size_t)s.M * s.K * 2));   // bf16 0x3c3c ~ 0.0115
        CK(hipMemset(W, 0x3c, (size_t)s.N * s.K * 2));
        CK(hipMemset(bias, 0, (size_t)s.N * 4));
        for (int tile : tiles)
            for (int stages = 1; stages <= 2; ++stages) {
                GemmArgs a = {};
                a.A = A; a.Wt = W; a.C = C; a.bias = bias;
                a.M = s.M; a.N = s.N; a.K = s.K; a.lda = s.K; a.ldc = s.N; a.rows_per_batch = s.M; a.alpha = 1.f;
                const int ft = tile | (stages << 3);
                for (int i = 0; i < 5; ++i) supir_gemm_launch(a, false, st, ft);
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 20; ++i) supir_gemm_launch(a, false, st, ft);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const int nblk = ((s.M + tb[tile][0] - 1) / tb[tile][0]) * ((s.N + tb[tile][1] - 1) / tb[tile][1]);
                const int nw = nblk * tb[tile][2];
                std::vector<unsigned long long> h((size_t)nw * 8);
                CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
                double sum[8] = {0};
                unsigned long long tmin = ~0ull, tmax = 0;
                for (int w = 0; w < nw; ++w) {
                    for (int c = 1; c < 8; ++c) sum[c] += (double)h[(size_t)w * 8 + c];
                    if (h[(size_t)w * 8] < tmin) tmin = h[(size_t)w * 8];
                    if (h[(size_t)w * 8] + h[(size_t)w * 8 + 7] > tmax) tmax = h[(size_t)w * 8] + h[(size_t)w * 8 + 7];
                }
                const int nk = s.K / 64;
                printf("M=%d N=%d K=%d tile=%d stages=%d blocks=%d | event %.1f us, TF %.0f | wall ticks %llu | per wave: prologue %.0f | "
                       "per K step (%d): wait %.0f  barrier %.0f  issue %.0f  lds+mfma %.0f | epilogue %.0f | total %.0f\n",
                       s.M, s.N, s.K, tile, stages + 1, nblk, ms * 1000 / 20, 2.0 * s.M * s.N * s.K / (ms * 1e-3 / 20) / 1e12,
                       tmax - tmin, sum[1] / nw, nk, sum[2] / nw / nk, sum[3] / nw / nk, sum[4] / nw / nk, sum[5] / nw / nk,
                       sum[6] / nw, sum[7] / nw);
                fflush(stdout);
            }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(bias));
    }
    // implicit-GEMM 3x3 convolutions of the step (stride 1, pad 1): [B][H][W][Cin] -> [B][H][W][Cout]
    struct Conv { int B, H, W, Cin, Cout; };
    const Conv convs[] = {{2, 32, 32, 1280, 1280}, {2, 64, 64, 640, 640}, {2, 128, 128, 320, 320}};
    for (const Conv& c : convs) {
        const int M = c.B * c.H * c.W, K = 9 * c.Cin;
        bf16_t *X, *W, *Y;
        float* bias;
        CK(hipMalloc(&X, (size_t)M * c.Cin * 2));
        CK(hipMalloc(&W, (size_t)c.Cout * K * 2));
        CK(hipMalloc(&Y, (size_t)M * c.Cout * 2));
        CK(hipMalloc(&bias, (size_t)c.Cout * 4));
        CK(hipMemset(X, 0x3c, (size_t)M * c.Cin * 2));
        CK(hipMemset(W, 0x3c, (size_t)c.Cout * K * 2));
        CK(hipMemset(bias, 0, (size_t)c.Cout * 4));
        for (int tile : tiles) {
            GemmArgs a = {};
            a.A = X; a.Wt = W; a.C = Y; a.bias = bias;
            a.M = M; a.N = c.Cout; a.K = K; a.lda = c.Cin; a.ldc = c.Cout; a.rows_per_batch = c.H * c.W; a.alpha = 1.f;
            a.H = c.H; a.W = c.W; a.Cin = c.Cin; a.OH = c.H; a.OW = c.W; a.stride = 1; a.pad_t = 1; a.pad_l = 1;
            const int ft = tile | (1 << 3);
            for (int i = 0; i < 3; ++i) supir_gemm_launch(a, true, st, ft);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 10; ++i) supir_gemm_launch(a, true, st, ft);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const int nblk = ((M + tb[tile][0] - 1) / tb[tile][0]) * ((c.Cout + tb[tile][1] - 1) / tb[tile][1]);
            const int nw = nblk * tb[tile][2];
            std::vector<unsigned long long> h((size_t)nw * 8);
            CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
            double sum[8] = {0};
            for (int w = 0; w < nw; ++w)
                for (int q = 1; q < 8; ++q) sum[q] += (double)h[(size_t)w * 8 + q];
            const int nk = K / 64;
            printf("conv B=%d %dx%d Cin=%d Cout=%d tile=%d blocks=%d | event %.1f us, TF %.0f | per wave: prologue %.0f | per K step (%d): wait %.0f  "
                   "barrier %.0f  issue %.0f  lds+mfma %.0f | epilogue %.0f | total %.0f\n",
                   c.B, c.H, c.W, c.Cin, c.Cout, tile, nblk, ms * 1000 / 10, 2.0 * M * c.Cout * K / (ms * 1e-3 / 10) / 1e12, sum[1] / nw, nk,
                   sum[2] / nw / nk, sum[3] / nw / nk, sum[4] / nw / nk, sum[5] / nw / nk, sum[6] / nw, sum[7] / nw);
            fflush(stdout);
        }
        CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(Y)); CK(hipFree(bias));
    }
    return 0;
}
