#!/bin/bash
# Round 5: the small-Cin boundary convolution on the exact-fp32 matrix instruction -- parity cases, in-process A/B against the VALU form
set -u
O=$PWD/gpurun_out/r05_smallcin
mkdir -p $O
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_torch_ops_vs_hip_gpu.py tests/test_model_gpu.py -q -m gpu -k "smallcin or vae or golden or control or unet or network" 2>&1 | tail -15 > $O/pytest.log
echo "pytest rc=${PIPESTATUS[0]}"; grep -v amdgpu.ids $O/pytest.log | tail -8
timeout 200 python tools/bench_smallcin.py $O/smallcin_ab.json 2>&1 | grep -v amdgpu.ids | cut -c1-400
