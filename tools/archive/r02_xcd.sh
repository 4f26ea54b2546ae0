#!/bin/bash
# A/B of the XCD partition cost model (SUPIR_XCD_MODEL=0 round-1 model, 1 new model): conv / GEMM microbench, step time, FETCH_SIZE
set -u
O=$PWD/gpurun_out/r02_xcd
mkdir -p $O
for MODEL in 0 1 0 1; do
SUPIR_XCD_MODEL=$MODEL timeout 200 python - >> $O/timing_model$MODEL.log 2>&1 <<'PY'
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
for (B, H, W, Cin, Cout, tile) in [(2, 32, 32, 1280, 1280, 35), (2, 32, 32, 2560, 1280, 35), (2, 64, 64, 640, 640, 33), (2, 128, 128, 320, 320, 34), (2, 64, 64, 1280, 1280, 34), (2, 32, 32, 1280, 1280, 3), (2, 64, 64, 1280, 640, 33), (2, 32, 32, 128, 2560, 33)]:
    x = torch.randn(B, H, W, Cin, device="cuda").to(BF); w = (torch.randn(Cout, 3, 3, Cin, device="cuda") * (9 * Cin) ** -0.5).to(BF); b = torch.randn(Cout, device="cuda")
    us = timeit(lambda: ops.conv3x3(x, w, b, tile=tile))
    print(dict(conv=(B, H, Cin, Cout), tile=tile, us=round(us, 1), tflops=round(2.0 * B * H * W * Cout * 9 * Cin / us / 1e6, 1)), flush=True)
for (M, N, K, tile) in [(2048, 1280, 1280, 35), (2048, 1280, 5120, 35), (2048, 2560, 1280, 33), (2048, 10240, 1280, 34), (8192, 640, 2560, 33), (2048, 10240, 1280, 0)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    us = timeit(lambda: ops.gemm(a, w, None, tile=tile))
    print(dict(gemm=(M, N, K), tile=tile, us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1)), flush=True)
PY
SUPIR_XCD_MODEL=$MODEL timeout 300 python tools/step_ab.py gemm16 2>&1 | grep -E "^\{" >> $O/step_model$MODEL.log
done
for MODEL in 0 1; do echo "== model $MODEL"; grep -v amdgpu $O/timing_model$MODEL.log | cut -c1-120; cat $O/step_model$MODEL.log; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py $O/cases.json > $O/run_fetch.log 2>&1
F=$(find $O/raw -name '*counter_collection.csv' | head -1)
(head -1 $F; grep -E "gemm16_kernel|gemm_bf16_kernel|attn_d64|gn_" $F) > $O/pmc_FETCH_SIZE_newmodel.csv
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $O/pmc_fetch_newmodel.json $O/pmc_FETCH_SIZE_newmodel.csv | grep gemm16 | cut -c1-200
rm -rf $O/raw
