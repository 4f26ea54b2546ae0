#!/bin/bash
# Round 2, final call of the second session: (1) head-dim-512 flash attention, pipelined version: parity + timing at T = 16384 / 4096;
# (2) the token threshold from which it replaces the materialised form is set FROM THAT MEASUREMENT (only if every parity case passed and
# it is >= 15 % faster) and exported, so that (3) the full GPU suite validates exactly the configuration that ships; (4) BASELINE config 5
# with diff_dtype fp16 on the fp16 build; (5) a reduced bench line.  Legs 4 / 5 run only as far as the GPU-minute budget allows.
set -u
O=$PWD/gpurun_out/r02_call_b
mkdir -p $O
rm -f gpurun_out/attn_d512_timing.json gpurun_out/parity_r02.json gpurun_out/parity_fp16.json
timeout 150 python -m pytest tests/test_attn_d512_gpu.py -q -s > $O/pytest_d512.log 2>&1; D512_RC=$?
echo "d512 rc=$D512_RC"; grep -E "passed|failed|\[d512\]|AttnBlock|Error" $O/pytest_d512.log | tail -12 | cut -c1-200
MIN_TOKENS=$(python - <<PY
import json, sys
rc = $D512_RC
try:
    t = json.load(open("gpurun_out/attn_d512_timing.json"))
except Exception:
    t = {}
thr = 46341
if rc == 0 and t:
    f16k, f4k = t.get("T16384", {}), t.get("T4096", {})
    if f16k and f16k["flash_us"] <= 0.85 * f16k["materialised_us"]:
        thr = 8192
        if f4k and f4k["flash_us"] <= 0.85 * f4k["materialised_us"]:
            thr = 2048
print(thr)
PY
)
echo "SUPIR_FLASH_D512_MIN_TOKENS=$MIN_TOKENS" | tee $O/flash_d512_threshold.txt
export SUPIR_FLASH_D512_MIN_TOKENS=$MIN_TOKENS
cp gpurun_out/attn_d512_timing.json $O/ 2>/dev/null
timeout 430 python -m pytest tests -m gpu -q --ignore=tests/test_attn_d512_gpu.py > $O/pytest_gpu_full.log 2>&1; echo "full suite rc=$?"
tail -6 $O/pytest_gpu_full.log | cut -c1-240
cp gpurun_out/parity_r02.json gpurun_out/parity_fp16.json $O/ 2>/dev/null
timeout 160 python tools/bench_configs.py --only-config5 --fp16 > $O/config5_fp16.log 2>&1; echo "config5 fp16 rc=$?"; tail -1 $O/config5_fp16.log | cut -c1-400
cp gpurun_out/configs_fp16.json $O/ 2>/dev/null
timeout 220 python bench.py --no-cpu-baseline --extra-batch 0 > $O/bench_n1_reduced.json 2> $O/bench_n1_reduced.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r02_call_b/bench_n1_reduced.json").read().strip().splitlines()[-1])
    print({k: j.get(k) for k in ("value", "ms_per_step", "ms_per_unet_step")}, {k: v for k, v in j["roofline"].items() if k not in ("shapes", "traffic")})
except Exception as e:
    print("no bench line:", e)
PY
