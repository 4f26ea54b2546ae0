#!/bin/bash
# Round 5, second GPU call: big-tile A/B (all rows), per-phase timelines of the large tiles, --pmc passes over tools/pmc_probe_large.py
set -u
O=$PWD/gpurun_out/r05_b
mkdir -p $O
timeout 300 python tools/bench_big_tiles.py $O/big_tiles.json --rounds 5 > $O/big_tiles.log 2>&1
echo "big tiles rc=$?"; grep -v amdgpu.ids $O/big_tiles.log | cut -c1-520
timeout 200 python tools/probes/g16_timeline.py large > $O/timeline_large.log 2>&1
echo "timeline rc=$?"; grep -v amdgpu.ids $O/timeline_large.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/raw_$T -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe_large.py $O/pmc_cases.json > $O/run_$T.log 2>&1
  echo "pmc $T rc=$?"
  F=$(find $O/raw_$T -name '*counter_collection.csv' | head -1)
  cp $F $O/pmc_$T.csv 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $O/pmc_summary.json $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/pmc_SQ_WAVE_CYCLES.csv $O/pmc_GRBM_GUI_ACTIVE.csv > $O/pmc_summary.txt 2>&1
rm -rf $O/raw_*
for f in $O/pmc_*.csv; do (head -1 $f; grep -E "gemm16_kernel|gemm_bf16_kernel|geglu_big" $f) > $f.tmp && mv $f.tmp $f; done
cut -c1-400 $O/pmc_summary.txt | tail -40
ls -la $O
