#!/bin/bash
# round 4, third session, call E: key-split sweep of the head-dim-512 attention; VAE tail with it
set -u
O=$PWD/gpurun_out/r04c_e
mkdir -p $O
timeout 300 python tools/r04_micro_d512.py > $O/micro_attn_d512_key_split_sweep.log 2>&1
echo "sweep rc=$?"; grep "^{" $O/micro_attn_d512_key_split_sweep.log
timeout 400 python tools/r04_micro_vae_gn.py > $O/micro_vae_tail.log 2>&1
echo "vae rc=$?"; tail -1 $O/micro_vae_tail.log | cut -c1-500
