#!/bin/bash
# Round-end evidence on one box: GPU test suite (with the long 50-step fp32 parity leg), default bench line, rocprofv3 kernel stats of
# the same bench command, smoke.  Outputs under gpurun_out/r02_final (copied to profiles/r02 by hand).
set -u
O=$PWD/gpurun_out/r02_final
mkdir -p $O
rm -f gpurun_out/parity_r02.json
SUPIR_TEST_LONG=1 timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee $O/summary.log
tail -14 $O/pytest_gpu.log
cp gpurun_out/parity_r02.json $O/parity.json 2>/dev/null
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r02_final/bench_n1.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","ms_per_unet_step","roofline","cpu_baseline","batched"): print(k, j.get(k))
for e in j.get("roofline_by_kernel", []): print({k: e[k] for k in ("kernel","achieved","frac","share_of_step_time","avg_launch_us")})
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --extra-batch 0 > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err
echo "rocprof rc=$?" | tee -a $O/summary.log
F=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $F $O/bench_rocprofv3_kernel_stats.csv; head -12 $O/bench_rocprofv3_kernel_stats.csv | cut -c1-160
rm -rf $O/prof
cd $GRAFT_REPO_ROOT
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
