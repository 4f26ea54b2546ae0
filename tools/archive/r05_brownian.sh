#!/bin/bash
# Round 5: the virtual Brownian tree on the device + config 5 (DPM++ 2M, 8 / 4 steps) end to end with it as the default noise sampler
set -u
O=$PWD/gpurun_out/r05_brownian
mkdir -p $O
( timeout 100 python -m pytest tests/test_brownian.py tests/test_parity_production_gpu.py -q -m gpu -k "brownian or dpmpp" 2>&1 | tail -6 > $O/pytest.log; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -3 ) &
timeout 170 python tools/bench_configs.py --only-config5 > $O/config5.log 2>&1
echo "config5 rc=$?"; grep -v amdgpu.ids $O/config5.log | tail -4 | cut -c1-600
wait
