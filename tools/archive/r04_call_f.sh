#!/bin/bash
# round 4, third session, call F: everything that touches the VAE / head-dim-512 attention after the policy change (flash from 1024 tokens)
set -u
O=$PWD/gpurun_out/r04c_f
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "vae or d512 or tiled or abi or testpy or config1 or smoke" > $O/pytest_vae_tiled_d512.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_vae_tiled_d512.log
