#!/bin/bash
# Round 2, validation call for the second session's work: fp16 build of the kernels, fused sampler step, thread-local graph
# capture (P1, with the config-1 end-to-end parity test running on the fused step); flash D=512 VAE attention in its own process
# (P2: new device code with LDS-DMA and barriers), with the VAE-vs-reference-golden tests on the flash path; then the in-process
# A/B of the two performance changes and the fp16 build at production scale (P3).  Every leg under its own timeout.
set -u
O=$PWD/gpurun_out/r02_call_a
mkdir -p $O
export SUPIR_TEST_FP16=1 SUPIR_TEST_D512=1 SUPIR_TEST_FUSED_STEP=1
SUPIR_GRAPH_CAPTURE_MODE=thread_local SUPIR_FUSED_EDM_STEP=1 timeout 330 python -m pytest tests/test_fp16_gpu.py tests/test_sampler_fused_gpu.py \
    tests/test_parity_production_gpu.py -q -s -k "not test_parity_production_gpu or batchify_sample_config1" > $O/p1_fp16_fused_config1.log 2>&1
echo "P1 fp16 + fused step + config1 parity rc=$?"
grep -E "passed|failed|parity-fp16|fused step|\[parity\]|Error" $O/p1_fp16_fused_config1.log | tail -25 | cut -c1-260
SUPIR_FLASH_D512=1 timeout 200 python -m pytest tests/test_attn_d512_gpu.py tests/test_model_gpu.py -q -s -k "test_attn_d512_gpu or vae" > $O/p2_d512.log 2>&1
echo "P2 d512 rc=$?"
grep -E "passed|failed|\[d512\]|AttnBlock|Error|golden|vae" $O/p2_d512.log | tail -20 | cut -c1-260
timeout 300 python tools/ab_fused_step_d512.py > $O/p3_ab.log 2>&1; echo "P3 ab rc=$?"
tail -2 $O/p3_ab.log | cut -c1-1200
cp gpurun_out/parity_fp16.json gpurun_out/attn_d512_timing.json gpurun_out/ab_fused_step_d512.json gpurun_out/parity_r02.json $O/ 2>/dev/null
