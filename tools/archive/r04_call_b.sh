#!/bin/bash
# round 4, second session, call B: parity of the VAE producer-statistics path, GroupNorm micro-benchmark, VAE-only re-tune of the packaged picks
set -u
O=$PWD/gpurun_out/r04b_b
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu -k "vae" -x > $O/pytest_vae.log 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest_vae.log
timeout 500 python tools/r04_micro_vae_gn.py > $O/micro_vae_groupnorm_and_tail.log 2>&1
echo "micro rc=$?"; tail -14 $O/micro_vae_groupnorm_and_tail.log | cut -c1-600
timeout 500 python tools/make_tune.py --vae-only $O/tune_gfx950.json > $O/make_tune_vae_only.log 2>&1
echo "tune rc=$?"; tail -40 $O/make_tune_vae_only.log
