#!/bin/bash
# round 4, third session, call K: flash attention row sum as v_dot2c of the packed probabilities: parity, micro A/B, step A/B
set -u
O=$PWD/gpurun_out/r04c_k
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_grouped_gpu.py -q -m gpu -k "attn or attention" -x > $O/pytest_attn.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_attn.log
timeout 200 python tools/r04_micro_attn_ds.py > $O/micro_flash_attention_row_sum_dot2_vs_fp32.log 2>&1
echo "micro rc=$?"; grep "^{" $O/micro_flash_attention_row_sum_dot2_vs_fp32.log
timeout 400 python tools/step_ab4.py fastgelu attnfp32sum > $O/step_ab4_attention_row_sum.log 2>&1
echo "ab rc=$?"; tail -5 $O/step_ab4_attention_row_sum.log
