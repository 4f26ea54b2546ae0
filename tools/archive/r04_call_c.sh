#!/bin/bash
# round 4, third session, call C: VAE producer-statistics test (relative bar), fused q|k|v on the 256 x 128 tile: parity + step A/B
set -u
O=$PWD/gpurun_out/r04c_c
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -k "vae or qkv" -x -s > $O/pytest_vae_qkv.log 2>&1
echo "pytest rc=$?"; grep -E "vae with producer|passed|failed|Error" $O/pytest_vae_qkv.log | cut -c1-400 | tail -8
timeout 500 python tools/step_ab4.py fastgelu qkv160 > $O/step_ab4_qkv_256x128_vs_256x160.log 2>&1
echo "ab rc=$?"; tail -8 $O/step_ab4_qkv_256x128_vs_256x160.log
