#!/bin/bash
# Round 5 evidence on the final build, one gpurun call:
#  1. the default bench line (kernel picks timed on this box; its own rocprofv3 --kernel-trace --stats sub-step of the same command, same box,
#     is copied to gpurun_out/bench_replay_rocprofv3_kernel_stats.csv by bench.py itself);
#  2. three --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ) over tools/pmc_probe.py (the step's kernels at the CFG batch of one image), each in
#     its own run with --kernel-trace only;
#  3. the other BASELINE configs (1, 5, 3 at 50 steps).
set -u
O=$PWD/gpurun_out/r05_final
mkdir -p $O
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench rc=$?"; tail -c 300 $O/bench_n1.err
cp gpurun_out/tune_used.json $O/tune_used.json 2>/dev/null
cp gpurun_out/bench_replay_rocprofv3_kernel_stats.csv $O/bench_rocprofv3_kernel_stats.csv 2>/dev/null
head -8 $O/bench_rocprofv3_kernel_stats.csv | cut -c1-170
cd /tmp && export TMPDIR=/tmp
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
# tile 42 alone through the large-M probe (the first pass of this round named it 41 and skipped it): four passes incl. GRBM_GUI_ACTIVE
mkdir -p $O/t42
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
  T=$(echo $C | cut -d' ' -f1)
  PMC_TILES=42,40 timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/t42/raw_$T -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe_large.py $O/t42/pmc_cases.json > $O/t42/run_$T.log 2>&1
  echo "pmc t42 $T rc=$?"
  F=$(find $O/t42/raw_$T -name '*counter_collection.csv' | head -1)
  cp $F $O/t42/pmc_$T.csv 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $O/t42/pmc_summary.json $O/t42/pmc_FETCH_SIZE.csv $O/t42/pmc_WRITE_SIZE.csv $O/t42/pmc_SQ_WAVE_CYCLES.csv $O/t42/pmc_GRBM_GUI_ACTIVE.csv > $O/t42/pmc_summary.txt 2>&1
rm -rf $O/t42/raw_*
for f in $O/t42/pmc_*.csv; do (head -1 $f; grep -E "gemm16_kernel" $f) > $f.tmp && mv $f.tmp $f; done
cut -c1-330 $O/t42/pmc_summary.txt | tail -6
timeout 900 python tools/bench_configs.py --tiled-steps 50 --tiled-single --tile-batch 4 > $O/other_configs.log 2>&1
echo "configs rc=$?"; cp gpurun_out/configs.json $O/other_configs.json 2>/dev/null; tail -3 $O/other_configs.log | cut -c1-900
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_final/bench_n1.json").read().strip().splitlines()[-1])
print("images/s", d["value"], "ms/image", d["ms_per_step"], "ms/unet step", d["ms_per_unet_step"], "in sampler", d.get("ms_per_unet_step_inside_the_sampler"), "batched", d.get("batched"))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k != "shapes"}))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"), d["cpu_baseline"].get("kind"))
print("vae tail", d["kernel_breakdown_vae_colorfix"]["wall_ms_eager"], {k: (v["ms"], v.get("tflops")) for k, v in list(d["kernel_breakdown_vae_colorfix"]["kernels"].items())[:6]})
PY
tail -24 $O/pmc_summary.txt | cut -c1-330
ls $O
