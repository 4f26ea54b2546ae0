#!/bin/bash
# flash attention: kernel parity, timing per shape, per-phase timeline, network-level parity, step timing
mkdir -p gpurun_out/r02_attn; O=gpurun_out/r02_attn
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attn or causal" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 120 python tools/attn_probe.py 2>/dev/null | tee $O/probe.log
timeout 120 tools/probes/attn_timeline | tee $O/timeline.log
timeout 400 python -m pytest tests/test_parity_production_gpu.py -q -x -s -m gpu -k "test_full_depth_network_call" 2>&1 | grep "parity\]\|passed\|failed" | tee $O/parity.log
timeout 300 python tools/step_ab.py gemm16 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300 | tee $O/step.log
