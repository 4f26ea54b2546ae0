#!/bin/bash
# Round 5, eighth GPU call: tile 42's convolutions with the chunk-major K order: tests, A/B against the tap-major order, FETCH_SIZE pass
set -u
O=$PWD/gpurun_out/r05_g
mkdir -p $O
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "vae_tiles" 2>&1 | tail -8 > $O/pytest_kernels.log
echo "pytest rc=${PIPESTATUS[0]}"; tail -4 $O/pytest_kernels.log
timeout 300 python tools/bench_big_tiles.py $O/big_tiles.json --rounds 5 > $O/big_tiles.log 2>&1
echo "big tiles rc=$?"; grep -v amdgpu.ids $O/big_tiles.log | cut -c1-760
cd /tmp && export TMPDIR=/tmp
PMC_TILES=42 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_F -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe_large.py $O/pmc_cases.json > $O/run_F.log 2>&1
echo "pmc rc=$?"; F=$(find $O/raw_F -name '*counter_collection.csv' | head -1); cp $F $O/pmc_FETCH_SIZE.csv; rm -rf $O/raw_F
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $O/pmc_summary.json $O/pmc_FETCH_SIZE.csv | cut -c1-300
