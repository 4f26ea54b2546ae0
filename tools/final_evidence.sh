#!/bin/bash
# Round-end evidence at HEAD on one box: the GPU tier, smoke(), the driver's bench command with chip power / clocks sampled beside it.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_head.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu_head.log
tail -3 gpurun_out/pytest_gpu_head.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_head.log 2>&1; tail -2 gpurun_out/smoke_head.log
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 2000; echo; sleep 1; done ) > gpurun_out/smi_during_bench.log 2>/dev/null &
SMI=$!
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_head.out 2> gpurun_out/bench_head.err; echo "bench rc $?"
kill $SMI 2>/dev/null
tail -c 2600 gpurun_out/bench_head.out
