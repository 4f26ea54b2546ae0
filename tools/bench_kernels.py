"""Micro-benchmarks of the hot kernels at the production shapes of one CFG-doubled 1024^2 step.
HIP-event timing on the launch stream; prints achieved TFLOP/s (MFMA kernels) or GB/s (HBM-bound kernels)."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import ops
from supir_amd.weights import interleave_geglu

BF = torch.bfloat16
dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


res = []
for (M, N, K, cnt) in [(2048, 1280, 1280, 560), (2048, 10240, 1280, 90), (2048, 1280, 5120, 90), (8192, 640, 640, 100),
                       (8192, 5120, 640, 14), (8192, 640, 2560, 14), (4096, 10240, 1280, 0), (4096, 1280, 1280, 0), (32768, 320, 320, 0)]:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    b = torch.randn(N, device=dev)
    for tile in range(7):
        for st in (1,):
            t = timeit(lambda: ops.gemm(a, w, b, tile=tile | (st << 3)))
            res.append(dict(op="gemm", M=M, N=N, K=K, tile=tile, stages=st + 1, us=t * 1e6, tflops=2.0 * M * N * K / t / 1e12, count=cnt))
            print(res[-1], flush=True)
    # torch (hipBLASLt) reference speed for context only
    t = timeit(lambda: torch.nn.functional.linear(a, w))
    print(dict(op="torch.linear", M=M, N=N, K=K, us=t * 1e6, tflops=2.0 * M * N * K / t / 1e12), flush=True)

for (B, H, W, Cin, Cout, cnt) in [(2, 32, 32, 1280, 1280, 17), (2, 64, 64, 640, 640, 9), (2, 128, 128, 320, 320, 11),
                                  (2, 32, 32, 2560, 1280, 2), (1, 512, 512, 256, 256, 0), (1, 1024, 1024, 128, 128, 0)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device=dev)
    for tile in range(7):
        for st in (1,):
            t = timeit(lambda: ops.conv3x3(x, w, b, tile=tile | (st << 3)), iters=10)
            fl = 2.0 * B * H * W * Cout * 9 * Cin
            res.append(dict(op="conv3x3", B=B, H=H, W=W, Cin=Cin, Cout=Cout, tile=tile, stages=st + 1, us=t * 1e6, tflops=fl / t / 1e12, count=cnt))
            print(res[-1], flush=True)

for (B, H, Tq, Tk, cnt) in [(2, 20, 1024, 1024, 91), (2, 10, 4096, 4096, 15), (2, 20, 1024, 77, 90), (2, 10, 4096, 77, 14)]:
    C = H * 64
    q = torch.randn(B, Tq, C, device=dev).to(BF)
    k = torch.randn(B, Tk, C, device=dev).to(BF)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tp, device=dev, dtype=BF)
    vt[:, :, :Tk] = torch.randn(B, C, Tk, device=dev).to(BF)
    t = timeit(lambda: ops.flash_attn(q, k, vt, B, H, Tq, Tk))
    fl = 4.0 * B * H * Tq * Tk * 64
    res.append(dict(op="attn", B=B, H=H, Tq=Tq, Tk=Tk, us=t * 1e6, tflops=fl / t / 1e12, count=cnt))
    print(res[-1], flush=True)

for (B, HW, C, cnt) in [(2, 16384, 320, 12), (2, 4096, 640, 17), (2, 1024, 1280, 28), (1, 1 << 20, 128, 0)]:
    x = torch.randn(B, HW, C, device=dev).to(BF)
    g = torch.ones(C, device=dev)
    bb = torch.zeros(C, device=dev)
    t = timeit(lambda: ops.groupnorm(x, g, bb, 1e-5, silu=True))
    by = 2.0 * B * HW * C * 2  # algorithmic: read once + write once
    res.append(dict(op="groupnorm_silu", B=B, HW=HW, C=C, us=t * 1e6, gbps=by / t / 1e9, count=cnt))
    print(res[-1], flush=True)

for (rows, C, cnt) in [(2048, 1280, 250), (8192, 640, 62)]:
    x = torch.randn(rows, C, device=dev).to(BF)
    g = torch.ones(C, device=dev)
    bb = torch.zeros(C, device=dev)
    t = timeit(lambda: ops.layernorm(x, g, bb))
    res.append(dict(op="layernorm", rows=rows, C=C, us=t * 1e6, gbps=4.0 * rows * C / t / 1e9, count=cnt))
    print(res[-1], flush=True)

os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_kernels.json", "w"), indent=1)
