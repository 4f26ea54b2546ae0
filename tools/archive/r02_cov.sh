#!/bin/bash
# coverage items of round 2 on the GPU: image I/O edges, text conditioner, causal attention / new activations, wavelet
set -u
O=gpurun_out/r02_cov
mkdir -p $O
timeout 600 python -m pytest tests/test_imageio.py tests/test_conditioner.py tests/test_kernels_gpu.py -m gpu -q -k "imageio or resample or pil2tensor or tensor2pil or conditioner or towers or causal or gelu or wavelet" > $O/pytest.log 2>&1
echo "cov pytest rc=$?" | tee $O/summary.log
grep -E "parity\]|passed|failed|FAILED|Error" $O/pytest.log | tail -20
