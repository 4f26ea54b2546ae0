#!/bin/bash
# round 4, second session, call A: parity of the new gemm16 tiles 39 / 40, VAE convolution micro-benchmark, kernel stats of a 4-images-per-call run
set -u
O=$PWD/gpurun_out/r04b_a
mkdir -p $O
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "vae_tiles" -x > $O/pytest_vae_tiles.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_vae_tiles.log
timeout 400 python tools/r04_micro_vae.py > $O/micro_vae_conv_tiles.log 2>&1
echo "micro rc=$?"; tail -30 $O/micro_vae_conv_tiles.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --images-per-gpu 4 --steps 1 --warmup 1 --edm-steps 10 --no-cpu-baseline --extra-batch 0 --no-kernel-profile --tune file > $O/bench_b8_under_rocprofv3.json 2> $O/bench_b8_under_rocprofv3.err
echo "rocprof rc=$?"; tail -c 300 $O/bench_b8_under_rocprofv3.err
F=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $F $O/bench_b8_rocprofv3_kernel_stats.csv; head -25 $O/bench_b8_rocprofv3_kernel_stats.csv | cut -c1-200
rm -rf $O/prof
