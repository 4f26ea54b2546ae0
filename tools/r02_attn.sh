#!/bin/bash
set -u
O=gpurun_out/r02_attn
mkdir -p $O
for KVS in 1 2 3; do
SUPIR_ATTN_KVS=$KVS timeout 200 python - > $O/timing_kvs$KVS.log 2>&1 <<'PY'
import torch, os
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
for (B, H, Tq, Tk) in [(2, 20, 1024, 1024), (2, 10, 4096, 4096), (2, 20, 1024, 77), (2, 10, 4096, 77), (2, 20, 1024, 1024)]:
    C = H * 64
    q = torch.randn(B, Tq, C, device="cuda").to(BF); k = torch.randn(B, Tk, C, device="cuda").to(BF)
    Tp = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tp, device="cuda", dtype=BF); vt[:, :, :Tk] = torch.randn(B, C, Tk, device="cuda").to(BF)
    us = timeit(lambda: ops.flash_attn(q, k, vt, B, H, Tq, Tk))
    print(dict(kvs=os.environ["SUPIR_ATTN_KVS"], B=B, H=H, Tq=Tq, Tk=Tk, us=round(us, 1), tflops=round(4.0 * B * H * Tq * Tk * 64 / us / 1e6, 1)), flush=True)
PY
grep -v amdgpu $O/timing_kvs$KVS.log
done
timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "flash_attn or causal" > $O/pytest_auto.log 2>&1; echo "attn pytest(auto) rc=$?"; tail -2 $O/pytest_auto.log
SUPIR_ATTN_KVS=2 timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "flash_attn or causal" > $O/pytest_kvs2.log 2>&1; echo "attn pytest(kvs2) rc=$?"; tail -2 $O/pytest_kvs2.log
SUPIR_ATTN_KVS=3 timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "flash_attn or causal" > $O/pytest_kvs3.log 2>&1; echo "attn pytest(kvs3) rc=$?"; tail -2 $O/pytest_kvs3.log
for KVS in 1 0; do SUPIR_ATTN_KVS=$KVS timeout 300 python tools/step_ab.py gemm16 2>&1 | grep -E "^\{" | sed "s/^/kvs=$KVS /"; done
timeout 700 python -m pytest tests/test_parity_production_gpu.py -m gpu -q -x > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; grep -E "parity\]|passed|failed" $O/pytest_parity.log | cut -c1-400
