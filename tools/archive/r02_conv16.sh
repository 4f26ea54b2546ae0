#!/bin/bash
set -u
O=gpurun_out/r02_conv16
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm16_conv" > $O/pytest.log 2>&1; echo "conv16 pytest rc=$?" | tee $O/summary.log
tail -4 $O/pytest.log
timeout 200 python - > $O/timing.log 2>&1 <<'PY'
import torch
from supir_amd import ops
BF = torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
G = {32: (128, 80, 2), 33: (128, 160, 2), 34: (256, 160, 1), 35: (128, 80, 2)}
for (B, H, W, Cin, Cout) in [(2, 32, 32, 1280, 1280), (2, 32, 32, 2560, 1280), (2, 64, 64, 640, 640), (2, 128, 128, 320, 320), (2, 64, 64, 1280, 1280), (2, 128, 128, 640, 640), (2, 32, 32, 128, 2560), (2, 64, 64, 128, 1280)]:
    x = torch.randn(B, H, W, Cin, device="cuda").to(BF); w = (torch.randn(Cout, 3, 3, Cin, device="cuda") * (9 * Cin) ** -0.5).to(BF); b = torch.randn(Cout, device="cuda")
    M = B * H * W
    for tile in (0, 1, 2, 3, 5, 32, 33, 34, 35):
        if tile >= 32:
            bm, bn, ks = G[tile]
            if M % bm or Cout % bn or Cin % (64 * ks): continue
        us = timeit(lambda: ops.conv3x3(x, w, b, tile=tile))
        print(dict(B=B, H=H, Cin=Cin, Cout=Cout, tile=tile, us=round(us, 1), tflops=round(2.0 * M * Cout * 9 * Cin / us / 1e6, 1)), flush=True)
PY
echo "timing rc=$?" | tee -a $O/summary.log
cat $O/timing.log | grep -v amdgpu.ids
timeout 500 python tools/step_ab.py base gemm16 > $O/step_ab.log 2>&1; echo "step_ab rc=$?" | tee -a $O/summary.log
grep -E "ms/step|rel-L2|^\{" $O/step_ab.log | cut -c1-200
