#!/bin/bash
# kernel trace of isolated graph replays of the 1024^2 step -> per-queue busy / gap / concurrency summary (profiles/r02/)
set -u
O=$PWD/gpurun_out/r02_trace
mkdir -p $O
V=${1:-gemm16}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/raw_$V -o t -- python $GRAFT_REPO_ROOT/tools/trace_step.py $V > $O/run_$V.log 2>&1
echo "trace rc=$?"
grep -E "replay|ms/step" $O/run_$V.log
F=$(find $O/raw_$V -name '*kernel_trace.csv' | head -1)
ls -la $F
python $GRAFT_REPO_ROOT/tools/analyze_trace.py $F $O/step_trace_summary_$V.json > $O/step_trace_summary_$V.txt 2>&1
cat $O/step_trace_summary_$V.txt
rm -rf $O/raw_$V
