#!/bin/bash
# First GPU call of the next round (everything that was written after round 1's GPU budget ran out), ~3-4 minutes:
#   gpurun --timeout 420 -- 'bash tools/r02_first_call.sh'
# Outputs under gpurun_out/r02_first/.  Every step has its own timeout so a hang in the experimental kernel cannot eat the call.
set -u
mkdir -p gpurun_out/r02_first
O=gpurun_out/r02_first
# 1. experimental GEMM tile 7 (two K groups per workgroup): parity first, in its own process
SUPIR_TEST_EXPERIMENTAL=1 timeout 90 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k split_k > $O/tile7_pytest.log 2>&1
echo "tile7 pytest rc=$?" | tee -a $O/summary.log
# 2. timing of tile 7 against the current tiles at the two shapes it is meant for (only if parity passed)
if grep -q " passed" $O/tile7_pytest.log && ! grep -q "failed" $O/tile7_pytest.log; then
timeout 90 python - > $O/tile7_timing.log 2>&1 <<'PY'
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
for (M, N, K) in [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 10240, 1280), (8192, 640, 640), (8192, 640, 2560)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF); b = torch.randn(N, device="cuda")
    for tile in (0, 1, 3, 7):
        us = timeit(lambda: ops.gemm(a, w, b, tile=tile))
        print(dict(M=M, N=N, K=K, tile=tile, us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1)), flush=True)
PY
echo "tile7 timing rc=$?" | tee -a $O/summary.log
fi
# 3. wavelet colour-fix kernel: launch time of one 1024^2 decomposition (5 levels)
timeout 60 python - > $O/wavelet_timing.log 2>&1 <<'PY'
import torch
from supir_amd import ops
x = torch.randn(1, 3, 1024, 1024, device="cuda")
for _ in range(3): ops.wavelet_decomposition(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.wavelet_decomposition(x)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(dict(op="wavelet_decomposition 1x3x1024x1024, 5 levels", us=round(us, 1), gbps=round(5 * 16 * 3 * 1024 * 1024 / us / 1e3, 1)))
PY
echo "wavelet timing rc=$?" | tee -a $O/summary.log
# 4. the regular gates
timeout 120 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.log
timeout 150 python bench.py --steps 2 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.log
tail -2 $O/tile7_pytest.log; cat $O/tile7_timing.log 2>/dev/null; cat $O/wavelet_timing.log; tail -2 $O/pytest_gpu.log; head -c 300 $O/bench_n1.json
