#!/bin/bash
# Round 5, tenth GPU call: the fp32 service -- operator tests, network / VAE / config-1 parity against the fp32 oracle
set -u
O=$PWD/gpurun_out/r05_i
mkdir -p $O
timeout 600 python -m pytest tests/test_fp32_gpu.py -q -m gpu -x 2>&1 | tail -30 > $O/pytest_fp32.log
echo "pytest fp32 rc=${PIPESTATUS[0]}"; tail -30 $O/pytest_fp32.log
timeout 900 python -m pytest tests/test_parity_production_gpu.py -q -m gpu -x -k "fp32_service" -s 2>&1 | tail -30 > $O/pytest_cfg1_fp32.log
echo "pytest cfg1 fp32 rc=${PIPESTATUS[0]}"; tail -15 $O/pytest_cfg1_fp32.log
cp gpurun_out/parity_fp32.json $O/ 2>/dev/null
