#!/bin/bash
# PMC passes (each counter group in its own run, kernel-trace only): FETCH_SIZE, WRITE_SIZE, SQ busy / MFMA busy.
set -u
O=$PWD/gpurun_out/r02_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/raw_$T -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py $O/cases.json > $O/run_$T.log 2>&1
  echo "pmc $T rc=$?"
  F=$(find $O/raw_$T -name '*counter_collection.csv' | head -1)
  cp $F $O/pmc_$T.csv 2>/dev/null
done
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $O/pmc_summary.json $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/pmc_SQ_WAVE_CYCLES.csv
rm -rf $O/raw_*
# keep the CSVs small: only our kernels
for f in $O/pmc_*.csv; do (head -1 $f; grep -E "gemm16_kernel|gemm_bf16_kernel|geglu_big|attn_d64|gn_" $f) > $f.tmp && mv $f.tmp $f; done
ls -la $O
