#!/bin/bash
# tile 37 (256 x 320 GEGLU tile): parity, per-phase timeline against tile 34, step timing with / without it
mkdir -p gpurun_out/r02_big; O=gpurun_out/r02_big
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "gemm_big or gemm16_geglu or gemm_geglu" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/probes/g16_timeline.py geglu 2>&1 | grep -v amdgpu.ids | grep -v "start skew" | tee $O/timeline.log
timeout 400 python tools/step_ab.py gemm16 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200 | tee $O/step.log
grep -o '"(2048, 10240, 1280, 2, 0)": [0-9]*\|"(8192, 5120, 640, 2, 0)": [0-9]*' $O/step.log | sort | uniq -c
SUPIR_GEMM_BIG=0 timeout 400 python tools/step_ab.py gemm16 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 | sed "s/^/nobig /" | tee -a $O/step.log
