#!/bin/bash
# Round 5, twelfth GPU call: fp32 config-1 test alone and after the bf16 tests of its file (oracle with / without MIOpen; image digest),
# the rest of the GPU tier from that file on (the full run stopped there), the new fp32 tests (causal attention, text towers)
set -u
O=$PWD/gpurun_out/r05_k
mkdir -p $O
timeout 600 python -m pytest tests/test_parity_production_gpu.py -q -m gpu -x -s -k "fp32_service" 2>&1 | grep -E "parity|passed|failed|Error" | cut -c1-700 > $O/cfg1_alone.log
echo "alone:"; cat $O/cfg1_alone.log
timeout 900 python -m pytest tests/test_parity_production_gpu.py -q -m gpu -x -s -k "config1" 2>&1 | grep -E "parity|passed|failed|Error" | cut -c1-700 > $O/cfg1_after_bf16.log
echo "after bf16:"; cat $O/cfg1_after_bf16.log
timeout 600 python -m pytest tests/test_fp32_gpu.py tests/test_conditioner.py -q -m gpu -x -s 2>&1 | grep -E "parity|passed|failed|Error|assert" | cut -c1-400 > $O/fp32_tests.log
echo "fp32 tests:"; cat $O/fp32_tests.log
