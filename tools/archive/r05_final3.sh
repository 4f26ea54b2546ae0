#!/bin/bash
# Round 5, evidence at HEAD with the matrix-instruction small-Cin convolutions: the default bench line (+ its own rocprofv3 sub-step)
set -u
O=$PWD/gpurun_out/r05_head3
mkdir -p $O
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench rc=$?"; tail -c 300 $O/bench_n1.err
cp gpurun_out/tune_used.json $O/tune_used.json 2>/dev/null
cp gpurun_out/bench_replay_rocprofv3_kernel_stats.csv $O/bench_rocprofv3_kernel_stats.csv 2>/dev/null
grep -i "smallcin\|gemm16_kernel<128, 80, 4, 1, 2, 3, false, false, false" $O/bench_rocprofv3_kernel_stats.csv | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_head3/bench_n1.json").read().strip().splitlines()[-1])
print("images/s", d["value"], "ms/image", d["ms_per_step"], "ms/unet step", d["ms_per_unet_step"], "in sampler", d.get("ms_per_unet_step_inside_the_sampler"), "batched", d.get("batched"))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "shapes"})[:900])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"), d["cpu_baseline"].get("kind"))
print("vae tail", d["kernel_breakdown_vae_colorfix"]["wall_ms_eager"], d["kernel_breakdown_vae_colorfix"]["kernels"].get("conv_smallcin"))
print("step smallcin", d["kernel_breakdown_unet_step"].get("conv_smallcin"))
PY
