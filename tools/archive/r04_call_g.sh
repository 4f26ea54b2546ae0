#!/bin/bash
# round 4, third session, call G: the full GPU suite on the final build (with the 15 slowest tests listed)
set -u
O=$PWD/gpurun_out/r04c_g
mkdir -p $O
timeout 1100 python -m pytest tests -q -m gpu --durations=15 > $O/pytest_gpu_full_suite.log 2>&1
echo "pytest rc=$?"; tail -24 $O/pytest_gpu_full_suite.log | cut -c1-200
