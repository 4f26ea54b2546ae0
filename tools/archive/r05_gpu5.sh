#!/bin/bash
# Round 5, fifth GPU call: VAE re-tune with tile 42 among the candidates, the new parity tests, the one-rank scale dry run
set -u
O=$PWD/gpurun_out/r05_e
mkdir -p $O
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "geglu or xattn or gemm16_plain or qkv" 2>&1 | tail -8 > $O/pytest_kernels.log
echo "pytest kernels rc=${PIPESTATUS[0]}"; tail -4 $O/pytest_kernels.log
timeout 400 python tools/make_tune.py --vae-only $O/tune_gfx950.json > $O/make_tune_vae_only.log 2>&1
echo "make_tune rc=$?"; grep -v amdgpu.ids $O/make_tune_vae_only.log | tail -45 | cut -c1-200
timeout 900 python -m pytest tests/test_testpy_flow_gpu.py "tests/test_parity_production_gpu.py::test_num_samples_4_vs_four_single_image_runs" -q -m gpu -x -s 2>&1 | grep -v "Warning\|warn" | tail -25 > $O/pytest_new_parity.log
echo "pytest new parity rc=${PIPESTATUS[0]}"; cut -c1-400 $O/pytest_new_parity.log | tail -16
SCALE_DRYRUN_OUT=$O/scale_dryrun_n1 timeout 600 bash tools/scale_dryrun.sh 1 1 1 > $O/scale_dryrun_n1.log 2>&1
echo "scale dryrun rc=$?"; cut -c1-600 $O/scale_dryrun_n1.log | tail -6
