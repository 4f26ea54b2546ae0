#!/bin/bash
# kernel trace of isolated graph replays of the 1024^2 step on the round-3 build -> per-queue busy / gap / concurrency summary
set -u
O=$PWD/gpurun_out/r03_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/raw -o t -- python $GRAFT_REPO_ROOT/tools/trace_step.py gemm16 > $O/run.log 2>&1
echo "trace rc=$?"
grep -E "replay|ms/step" $O/run.log
F=$(find $O/raw -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/analyze_trace.py $F $O/step_trace_summary.json > $O/step_trace_summary.txt 2>&1
head -60 $O/step_trace_summary.txt
rm -rf $O/raw
