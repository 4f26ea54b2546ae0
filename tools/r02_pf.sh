#!/bin/bash
set -u
O=gpurun_out/r02_pf
mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "prefetch" > $O/pytest.log 2>&1; echo "prefetch pytest rc=$?"; tail -2 $O/pytest.log
for X in 0 1 0 1; do SUPIR_XCD_PREFETCH=$X timeout 300 python tools/step_ab.py gemm16 2>&1 | grep -E "^\{" | sed "s/^/xcd_prefetch=$X /"; done | tee $O/step_ab.log
