#!/bin/bash
# Round 5, sixth GPU call: the whole GPU test tier + the one-rank scale dry run
set -u
O=$PWD/gpurun_out/r05_f
mkdir -p $O
SCALE_DRYRUN_OUT=$O/scale_dryrun_n1 timeout 600 bash tools/scale_dryrun.sh 1 1 1 > $O/scale_dryrun_n1.log 2>&1
echo "scale dryrun rc=$?"; cut -c1-700 $O/scale_dryrun_n1.log | tail -5
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 2>&1 | grep -v "Warning\|warn(" | tail -40 > $O/pytest_gpu_full_suite.log
echo "pytest full rc=${PIPESTATUS[0]}"; tail -22 $O/pytest_gpu_full_suite.log | cut -c1-300
