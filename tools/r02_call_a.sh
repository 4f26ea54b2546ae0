#!/bin/bash
# Round 2, validation call for the work of the round's second session: fp16 build of the kernels, flash D=512 VAE attention,
# fused sampler step, thread-local graph capture; then an in-process A/B of the two performance changes.  Every leg under its own
# timeout; the D=512 kernel (new device code with LDS-DMA + barriers) runs in its own process, last among the tests.
set -u
O=$PWD/gpurun_out/r02_call_a
mkdir -p $O
export SUPIR_TEST_FP16=1 SUPIR_TEST_D512=1 SUPIR_TEST_FUSED_STEP=1
timeout 300 python -m pytest tests/test_fp16_gpu.py tests/test_sampler_fused_gpu.py -q -s > $O/pytest_fp16_fused.log 2>&1; echo "fp16+fused rc=$?"
grep -E "passed|failed|parity-fp16|fused step|Error|error" $O/pytest_fp16_fused.log | tail -25 | cut -c1-220
SUPIR_GRAPH_CAPTURE_MODE=thread_local timeout 200 python -m pytest tests/test_model_gpu.py -q -k "graph_replay_matches_eager or graph_survives_new_prompt" > $O/pytest_capture_thread_local.log 2>&1; echo "thread_local capture rc=$?"
tail -2 $O/pytest_capture_thread_local.log | cut -c1-200
timeout 240 python -m pytest tests/test_attn_d512_gpu.py -q -s > $O/pytest_d512.log 2>&1; echo "d512 rc=$?"
grep -E "passed|failed|d512|AttnBlock|Error|error" $O/pytest_d512.log | tail -20 | cut -c1-220
timeout 400 python tools/ab_fused_step_d512.py > $O/ab.log 2>&1; echo "ab rc=$?"
tail -3 $O/ab.log | cut -c1-600
cp gpurun_out/parity_fp16.json gpurun_out/attn_d512_timing.json gpurun_out/ab_fused_step_d512.json $O/ 2>/dev/null
