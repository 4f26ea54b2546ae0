#!/bin/bash
# Round 5, fourth GPU call: tile 44 / xattn grid tests + A/B; tile 42 A/B after the epilogue revert (spot check)
set -u
O=$PWD/gpurun_out/r05_d
mkdir -p $O
timeout 500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "geglu or xattn or vae_tiles_plain" 2>&1 | tail -12 > $O/pytest_kernels.log
echo "pytest kernels rc=${PIPESTATUS[0]}"; tail -6 $O/pytest_kernels.log
timeout 300 python tools/bench_geglu_xattn.py $O/geglu_xattn.json > $O/geglu_xattn.log 2>&1
echo "geglu/xattn rc=$?"; grep -v amdgpu.ids $O/geglu_xattn.log | cut -c1-500
timeout 300 python tools/bench_big_tiles.py $O/big_tiles.json --rounds 5 > $O/big_tiles.log 2>&1
echo "big tiles rc=$?"; grep -v amdgpu.ids $O/big_tiles.log | cut -c1-420 | head -4
