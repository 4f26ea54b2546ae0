#!/bin/bash
# Round 4 evidence on the final build, one gpurun call (same recipe as round 3, tools/archive/r03_final.sh):
#  1. the default bench line (kernel picks timed on this box, saved to gpurun_out/tune_used.json);
#  2. rocprofv3 --kernel-trace --stats of the same program REPLAYING those picks (--tune file);
#  3. three --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ) over tools/pmc_probe.py, each in its own run with --kernel-trace only;
#  4. the same bench under `torch.distributed.run --nproc-per-node 1` (backend nccl = RCCL, world size 1: the launch path the driver uses for N > 1).
set -u
O=$PWD/gpurun_out/r04_final
mkdir -p $O
timeout 700 python bench.py --steps 5 --warmup 2 > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench rc=$?"; tail -c 300 $O/bench_n1.err
cp gpurun_out/tune_used.json $O/tune_used.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --extra-batch 0 --tune file --tune-file $O/tune_used.json --save-tune $O/tune_profiled.json > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err
echo "rocprof rc=$?"
F=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $F $O/bench_rocprofv3_kernel_stats.csv; head -12 $O/bench_rocprofv3_kernel_stats.csv | cut -c1-180
rm -rf $O/prof
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/raw_$T -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py $O/pmc_cases.json > $O/run_$T.log 2>&1
  echo "pmc $T rc=$?"
  F=$(find $O/raw_$T -name '*counter_collection.csv' | head -1)
  cp $F $O/pmc_$T.csv 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $O/pmc_summary.json $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/pmc_SQ_WAVE_CYCLES.csv > $O/pmc_summary.txt 2>&1
rm -rf $O/raw_*
for f in $O/pmc_*.csv; do (head -1 $f; grep -E "gemm16_kernel|gemm_bf16_kernel|geglu_big|attn_d64|xattn_q|gn_" $f) > $f.tmp && mv $f.tmp $f; done
MASTER_ADDR=127.0.0.1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --extra-batch 0 --no-kernel-profile > $O/bench_torchrun_nproc1.json 2> $O/bench_torchrun_nproc1.err
echo "torchrun rc=$?"; tail -c 200 $O/bench_torchrun_nproc1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_final/bench_n1.json").read().strip().splitlines()[-1])
print("images/s", d["value"], "ms/image", d["ms_per_step"], "ms/unet step", d["ms_per_unet_step"], "in sampler", d.get("ms_per_unet_step_inside_the_sampler"), "batched", d.get("batched"))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "shapes"}))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"), d["cpu_baseline"].get("kind"), d["cpu_baseline"].get("unet_step_1024px_cfg_doubled_s"))
print("picks", d["kernel_picks"])
p = json.loads(open("gpurun_out/r04_final/bench_under_rocprofv3.json").read().strip().splitlines()[-1])
print("under rocprof: ms/unet step", p["ms_per_unet_step"], "picks", p["kernel_picks"])
try:
    t = json.loads(open("gpurun_out/r04_final/bench_torchrun_nproc1.json").read().strip().splitlines()[-1])
    print("torchrun nproc 1: images/s", t["value"], "ms/unet step", t.get("ms_per_unet_step"))
except Exception as e:
    print("torchrun line missing", e)
PY
tail -30 $O/pmc_summary.txt | cut -c1-260
ls -la $O
