#!/bin/bash
# Round 2, last GPU-seconds: rocprofv3 --kernel-trace --stats of the bench command on the FINAL build (fused sampler step on), with
# the bench line that profiled run printed; the summary CSV is what profiles/r02/bench_rocprofv3_kernel_stats_final_build.csv holds.
set -u
O=$PWD/gpurun_out/r02_call_c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 230 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --extra-batch 0 > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err
echo "rocprof rc=$?"
F=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $F $O/bench_rocprofv3_kernel_stats.csv; head -14 $O/bench_rocprofv3_kernel_stats.csv | cut -c1-170
rm -rf $O/prof
tail -c 600 $O/bench_under_rocprofv3.json | head -c 600; echo
