#!/bin/bash
# Round 5, ninth GPU call: tile 45 (512 x 128 on the eight-phase schedule): parity, A/B against tile 39 on the VAE's 128-channel layers, timeline
set -u
O=$PWD/gpurun_out/r05_h
mkdir -p $O
timeout 500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "vae_tiles" 2>&1 | tail -8 > $O/pytest_kernels.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -4 $O/pytest_kernels.log
timeout 400 python tools/bench_big_tiles.py $O/big_tiles.json --rounds 5 > $O/big_tiles.log 2>&1
echo "big tiles rc=$?"; grep -v amdgpu.ids $O/big_tiles.log | cut -c1-900
timeout 200 python tools/probes/g16_timeline.py large $O/timeline.json > $O/timeline.log 2>&1
echo "timeline rc=$?"; tail -12 $O/timeline.log | cut -c1-400
