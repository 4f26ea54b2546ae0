"""Hot vs cold operands: the microbenchmarks re-use one weight buffer (L2 / Infinity-Cache hot); in the real step every GEMM
streams its own weights from HBM (7.7 GB of bf16 weights per step >> 256 MB MALL). Cycle through enough buffers to be cold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from supir_amd import ops
BF = torch.bfloat16
dev = "cuda"


def timeit(fn, iters):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for (M, N, K, tile) in [(2048, 1280, 1280, 3), (2048, 1280, 1280, 1), (2048, 10240, 1280, 0), (2048, 1280, 5120, 3), (8192, 640, 640, 2)]:
    nbuf = max(4, int(600e6 / (2 * N * K)))
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(nbuf)]
    nab = max(4, int(600e6 / (2 * M * K)))
    as_ = [torch.randn(M, K, device=dev).to(BF) for _ in range(nab)]
    out = torch.empty(M, N, device=dev, dtype=BF)
    fl = 2.0 * M * N * K
    t_hot = timeit(lambda i: ops.gemm(as_[0], ws[0], None, out=out, tile=tile), 200)
    t_cw = timeit(lambda i: ops.gemm(as_[0], ws[i % nbuf], None, out=out, tile=tile), 200)
    t_cold = timeit(lambda i: ops.gemm(as_[i % nab], ws[i % nbuf], None, out=out, tile=tile), 200)
    print(f"M{M} N{N} K{K} tile{tile}: hot {t_hot * 1e6:.1f} us ({fl / t_hot / 1e12:.0f} TF) | cold W {t_cw * 1e6:.1f} us ({fl / t_cw / 1e12:.0f} TF) | "
          f"cold A+W {t_cold * 1e6:.1f} us ({fl / t_cold / 1e12:.0f} TF)", flush=True)
    del ws, as_

# ---- does touching the NEXT GEMM's weights from a second stream (HBM is idle: 7.7 GB / 44 ms = 0.2 TB/s) hide the cold start?
print("prefetch experiment", flush=True)
side = torch.cuda.Stream()
for (M, N, K, tile) in [(2048, 1280, 1280, 3), (2048, 1280, 5120, 3)]:
    nbuf = max(4, int(600e6 / (2 * N * K)))
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(nbuf)]
    a = torch.randn(M, K, device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    junk = torch.zeros(1, device=dev)
    fl = 2.0 * M * N * K

    def step_plain(i):
        ops.gemm(a, ws[i % nbuf], None, out=out, tile=tile)

    def step_pf(i):
        main = torch.cuda.current_stream()
        side.wait_stream(main)                       # throttle: prefetch i+1 starts when GEMM i-1 has finished
        with torch.cuda.stream(side):
            w = ws[(i + 1) % nbuf]
            junk.add_(w.view(-1)[:: 64].float().sum())   # touches every 128-byte line of the next weight matrix
        ops.gemm(a, ws[i % nbuf], None, out=out, tile=tile)

    t0 = timeit(step_plain, 300)
    t1 = timeit(step_pf, 300)
    print(f"M{M} N{N} K{K}: cold W {t0 * 1e6:.1f} us -> with next-weight prefetch on a side stream {t1 * 1e6:.1f} us", flush=True)
    del ws
