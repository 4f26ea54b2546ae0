#!/bin/bash
set -u
O=gpurun_out/r02_qkv
mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "qkv" > $O/pytest.log 2>&1; echo "qkv pytest rc=$?" | tee $O/summary.log
tail -3 $O/pytest.log
timeout 500 python tools/step_ab.py noqkv gemm16 > $O/step_ab.log 2>&1; echo "step_ab rc=$?" | tee -a $O/summary.log
grep -E "ms/step|rel-L2|^\{" $O/step_ab.log | sed 's/tiles {.*} choices/choices/' | cut -c1-300
