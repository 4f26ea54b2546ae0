#!/bin/bash
# Round 5, first GPU call: the new tile edges / tiled fused path / tile 42 tests, the big-tile A/B, the large-M breakdown + tiled sampler sweep.
set -u
O=$PWD/gpurun_out/r05_a
mkdir -p $O
timeout 400 python -m pytest tests/test_sampler_fused_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_sampler.log
echo "pytest sampler rc=${PIPESTATUS[0]}"; tail -5 $O/pytest_sampler.log
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "vae_tiles or prefetch_is_read_only" 2>&1 | tail -25 > $O/pytest_tile42.log
echo "pytest tile42 rc=${PIPESTATUS[0]}"; tail -8 $O/pytest_tile42.log
timeout 300 python tools/bench_big_tiles.py $O/big_tiles.json > $O/big_tiles.log 2>&1
echo "big tiles rc=$?"; grep -v amdgpu.ids $O/big_tiles.log | cut -c1-420
timeout 900 python tools/largeM_profile.py $O/largeM.json --batches 8 --tile-batches 4,7,13 --steps 3 > $O/largeM.log 2>&1
echo "largeM rc=$?"; grep -v amdgpu.ids $O/largeM.log | cut -c1-300 | tail -40
