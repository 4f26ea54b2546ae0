#!/bin/bash
# gemm16 (tiles 32 / 33): parity, microbenchmark against the 32x32x16 tiles, step-level A/B.  Outputs: gpurun_out/r02_gemm16/
set -u
O=gpurun_out/r02_gemm16
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm16 > $O/pytest.log 2>&1; echo "gemm16 pytest rc=$?" | tee $O/summary.log
tail -3 $O/pytest.log
timeout 120 python - > $O/timing.log 2>&1 <<'PY'
import torch
from supir_amd import ops
BF = torch.bfloat16
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (2048, 1280, 640), (8192, 640, 640), (8192, 640, 2560), (2048, 10240, 1280)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF); b = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(BF)
    for tile in (0, 1, 3, 32, 33, 34, 35):
        if tile >= 32 and (N % {32: 80, 33: 160, 34: 160, 35: 80}[tile] or M % (256 if tile == 34 else 128)): continue
        us = timeit(lambda: ops.gemm(a, w, b, residual=res, tile=tile))
        print(dict(M=M, N=N, K=K, tile=tile, us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1)), flush=True)
from supir_amd.weights import interleave_geglu
for (M, N2, K) in [(2048, 10240, 1280), (8192, 5120, 640)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N2, K, device="cuda") * K ** -0.5).to(BF); b = torch.randn(N2, device="cuda")
    w32, b32 = interleave_geglu(w, b, 32); w16, b16 = interleave_geglu(w, b, 16)
    for tile in (0, 2, 4, 34):
        ww, bb = (w16, b16) if tile == 34 else (w32, b32)
        us = timeit(lambda: ops.gemm(a, ww, bb, act=2, tile=tile))
        print(dict(geglu=True, M=M, N2=N2, K=K, tile=tile, us=round(us, 1), tflops=round(2.0 * M * N2 * K / us / 1e6, 1)), flush=True)
PY
echo "timing rc=$?" | tee -a $O/summary.log
cat $O/timing.log
timeout 600 python tools/step_ab.py ${STEP_VARIANTS:-base gemm16} > $O/step_ab.log 2>&1; echo "step_ab rc=$?" | tee -a $O/summary.log
grep -E "ms/step|rel-L2|^\{" $O/step_ab.log | cut -c1-400
