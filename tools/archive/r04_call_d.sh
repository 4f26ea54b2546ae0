#!/bin/bash
# round 4, third session, call D: head-dim-512 attention with key splits -- parity, timing report, VAE end to end
set -u
O=$PWD/gpurun_out/r04c_d
mkdir -p $O
timeout 600 python -m pytest tests/test_attn_d512_gpu.py tests/test_abi.py -q -x -s > $O/pytest_attn_d512_key_splits.log 2>&1
echo "pytest rc=$?"; grep -E "\[d512\]|AttnBlock|passed|failed|Error|rel-L2" $O/pytest_attn_d512_key_splits.log | cut -c1-300 | tail -14
cp gpurun_out/attn_d512_timing.json $O/ 2>/dev/null
timeout 300 python -m pytest tests/test_model_gpu.py tests/test_torch_ops_vs_hip_gpu.py -q -m gpu -k "vae or d512" -x > $O/pytest_vae.log 2>&1
echo "pytest vae rc=$?"; tail -3 $O/pytest_vae.log
