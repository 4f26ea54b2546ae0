#!/bin/bash
# round 4, third session, call I: the other BASELINE configs on the final build (config 3 at 50 steps, tile_batch 4)
set -u
O=$PWD/gpurun_out/r04c_i
mkdir -p $O
timeout 500 python tools/bench_configs.py --tiled-steps 50 --tiled-single --tile-batch 4 > $O/other_configs.log 2>&1
echo "rc=$?"; tail -3 $O/other_configs.log | cut -c1-1500
