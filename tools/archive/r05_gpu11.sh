#!/bin/bash
# Round 5, eleventh GPU call: the fp32 service, second pass (pooled GroupNorm statistics / tiled VAE, test.py flow in fp32)
set -u
O=$PWD/gpurun_out/r05_j
mkdir -p $O
timeout 600 python -m pytest tests/test_fp32_gpu.py -q -m gpu -x -s 2>&1 | tail -30 > $O/pytest_fp32.log
echo "pytest fp32 rc=${PIPESTATUS[0]}"; grep -v amdgpu.ids $O/pytest_fp32.log | tail -25
timeout 900 python -m pytest tests/test_testpy_flow_gpu.py -q -m gpu -x -k "fp32" -s 2>&1 | tail -30 > $O/pytest_flow_fp32.log
echo "pytest flow fp32 rc=${PIPESTATUS[0]}"; grep -v amdgpu.ids $O/pytest_flow_fp32.log | tail -15
cp gpurun_out/parity_fp32.json $O/ 2>/dev/null
