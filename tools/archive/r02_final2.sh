#!/bin/bash
# after the GroupNorm-from-producer change: GPU suite (default legs), bench line, smoke
set -u
O=$PWD/gpurun_out/r02_final
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu2.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $O/pytest_gpu2.log | cut -c1-200
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r02_final/bench_n1.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","ms_per_unet_step","end_to_end_tflops_per_gpu"): print(k, j.get(k))
r=j["roofline"]; print({k:v for k,v in r.items() if k not in ("shapes","traffic")})
print(j["cpu_baseline"]["config1_end_to_end_s"], j["cpu_baseline"]["value"], j["batched"])
for k,v in j["kernel_breakdown_unet_step"].items():
    if "groupnorm" in k: print(k, v)
for e in j.get("roofline_by_kernel", []): print({k: e[k] for k in ("kernel","achieved","frac","share_of_step_time","avg_launch_us")})
PY
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
