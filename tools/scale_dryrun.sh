#!/bin/bash
# Dry run of the N-GPU bench path (BASELINE configs[3]: independent 1024^2 images, one per GPU, ONE RCCL weight broadcast, no collective
# inside a sample) with per-rank reporting -- what the driver's scaling run will exercise, made observable:
#   per rank : construction seconds (rank 0 real, ranks > 0 on META + to_empty), weight fill / broadcast seconds and GB/s, host peak RSS,
#              device peak memory, autotune entries the rank-0 sync changed, seconds of the re-capture image, its own images/s;
#   rank 0   : the bench JSON line (max-over-ranks timing, whole-job images/s).
# Usage: tools/scale_dryrun.sh [NPROC=8] [STEPS=2] [WARMUP=1]      (NPROC=1 runs the same path through a one-rank RCCL group)
set -u
N=${1:-8}; STEPS=${2:-2}; WARMUP=${3:-1}
cd "$(dirname "$0")/.."
OUT=${SCALE_DRYRUN_OUT:-gpurun_out/scale_dryrun_n$N}
mkdir -p "$OUT"
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29541} \
    bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" --rank-report --no-cpu-baseline --extra-batch 0 --no-kernel-profile \
    > "$OUT/bench_line.json" 2> "$OUT/stderr.log"
rc=$?
echo "torchrun rc=$rc"
grep -h "^\[rank-report\]" "$OUT/stderr.log" | sed 's/^\[rank-report\] //' | sort > "$OUT/rank_reports.jsonl"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
reps = [json.loads(l) for l in open(f"{out}/rank_reports.jsonl") if l.strip()]
for r in sorted(reps, key=lambda r: r["rank"]):
    print(f"rank {r['rank']}: construct {r.get('construct_s')} s ({'meta' if r.get('meta_construction') else 'device'}), fill {r.get('fill_s')} s, "
          f"broadcast {r.get('broadcast_s')} s = {r.get('broadcast_GBps')} GB/s of {r.get('broadcast_GB')} GB, warm-up {r.get('warmup_s')} s, "
          f"autotune entries changed {r.get('autotune_entries_changed_on_this_rank')}, re-capture image {r.get('recapture_image_s')} s, "
          f"host RSS {r.get('host_peak_rss_GB')} GB, device peak {r.get('device_mem_peak_GB')} GB, {r.get('images_per_s_this_rank')} images/s; "
          f"process start -> first timed image {r.get('start_to_first_timed_image_s')} s (imports {r.get('imports_s')}, RCCL init + library "
          f"{r.get('process_group_and_library_s')})")
try:
    line = json.loads(open(f"{out}/bench_line.json").read().strip().splitlines()[-1])
    print("whole job:", line["value"], line["unit"], "on", line["n_gpus"], "GPU(s); ms per image (max over ranks)", round(line["ms_per_step"], 1),
          "; process group", line.get("process_group"))
except Exception as e:   # noqa: BLE001
    print("no bench line:", e)
PY
exit $rc
