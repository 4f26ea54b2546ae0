#!/bin/bash
# Round 5, third GPU call: tile 43 / PH8 qkv / epilogue-prefetch tests, big-tile A/B, timelines, the B = 8 step re-tuned on this box, config 3 sampler
set -u
O=$PWD/gpurun_out/r05_c
mkdir -p $O
timeout 500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm16 or qkv or vae_tiles or groupnorm_statistics" 2>&1 | tail -12 > $O/pytest_kernels.log
echo "pytest kernels rc=${PIPESTATUS[0]}"; tail -6 $O/pytest_kernels.log
timeout 300 python tools/bench_big_tiles.py $O/big_tiles.json --rounds 5 > $O/big_tiles.log 2>&1
echo "big tiles rc=$?"; grep -v amdgpu.ids $O/big_tiles.log | cut -c1-640
timeout 200 python tools/probes/g16_timeline.py large > $O/timeline_large.log 2>&1
echo "timeline rc=$?"; grep -v amdgpu.ids $O/timeline_large.log | grep -v "start skew" | cut -c1-300
SUPIR_TUNE_FILE=none timeout 900 python tools/largeM_profile.py $O/largeM.json --batches 8 --tile-batches 4 --steps 3 --no-generic > $O/largeM.log 2>&1
echo "largeM rc=$?"; grep -v amdgpu.ids $O/largeM.log | cut -c1-300 | tail -24
