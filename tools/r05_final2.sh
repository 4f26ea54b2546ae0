#!/bin/bash
# Round 5, evidence at HEAD after the fp32 service went in: the default bench line (+ its own rocprofv3 sub-step) and what the fp32 service costs
set -u
O=$PWD/gpurun_out/r05_head
mkdir -p $O
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench rc=$?"; tail -c 300 $O/bench_n1.err
cp gpurun_out/tune_used.json $O/tune_used.json 2>/dev/null
cp gpurun_out/bench_replay_rocprofv3_kernel_stats.csv $O/bench_rocprofv3_kernel_stats.csv 2>/dev/null
head -6 $O/bench_rocprofv3_kernel_stats.csv | cut -c1-170
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_head/bench_n1.json").read().strip().splitlines()[-1])
print("images/s", d["value"], "ms/image", d["ms_per_step"], "ms/unet step", d["ms_per_unet_step"], "in sampler", d.get("ms_per_unet_step_inside_the_sampler"), "batched", d.get("batched"))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "shapes"}))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"), d["cpu_baseline"].get("kind"))
print("vae tail", d["kernel_breakdown_vae_colorfix"]["wall_ms_eager"])
PY
timeout 600 python tools/fp32_timing.py $O/fp32_timing.json 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900
