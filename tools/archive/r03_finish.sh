#!/bin/bash
# Round 3, closing call: refresh the shipped tune file (new candidates: tap-split convolutions), full GPU suite on it, then the evidence
# script (bench line, rocprofv3 kernel stats replaying the bench's picks, PMC passes).
set -u
O=$PWD/gpurun_out/r03_finish
mkdir -p $O
timeout 400 python tools/make_tune.py $O/tune_gfx950.json > $O/make_tune.log 2>&1; tail -1 $O/make_tune.log
cp $O/tune_gfx950.json supir_amd/tune_gfx950.json
timeout 1300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 200 python tools/step_variants.py file retune > $O/step_variants.log 2>&1; tail -3 $O/step_variants.log | cut -c1-300
bash tools/r03_final.sh > $O/final.log 2>&1; grep -E "^images/s|^roofline|^cpu|^picks|^under rocprof|rc=" $O/final.log | cut -c1-700
