#!/bin/bash
# Round 5: the whole GPU tier + smoke at HEAD (what the driver runs at round end), with the slowest tests listed
set -u
O=$PWD/gpurun_out/r05_suite
mkdir -p $O
START=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu --durations=20 2>&1 | tail -80 > $O/pytest_gpu.log
echo "pytest rc=${PIPESTATUS[0]} seconds=$(( $(date +%s) - START ))"; grep -v amdgpu.ids $O/pytest_gpu.log | tail -34
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
echo "smoke rc=$?"; grep smoke $O/smoke.log
