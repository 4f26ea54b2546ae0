#!/bin/bash
# GroupNorm statistics from the producer epilogues: kernel parity, model parity (graph == eager, oracle), step timing with / without
mkdir -p gpurun_out/r02_gn; O=gpurun_out/r02_gn
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "groupnorm or gn_ or gemm16 or conv3x3" > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.log | cut -c1-300
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parity_production_gpu.py -q -x -s -m gpu -k "not config2_50 and not config1" > $O/pytest_model.log 2>&1; echo "model rc=$?"; grep "parity\]\|passed\|failed\|Error" $O/pytest_model.log | cut -c1-260 | tail -12
for V in 1 0; do SUPIR_GN_PARTS=$V timeout 400 python tools/step_ab.py gemm16 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200 | sed "s/^/gn_parts=$V /" | tee -a $O/step.log; done
