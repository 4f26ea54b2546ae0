"""In-process A/B at IMAGE level (1024^2, 50 EDM steps, bench.py's workload, hipGraph path): seconds per image with a module-level
switch flipped, interleaved (never compare numbers from two calls: box-to-box spread is ~5 %).
  sched     the per-image timestep-embedding schedule (supir_amd.modules.sampling.EMB_SCHEDULE) on (default) / off
Usage: python tools/image_ab.py [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from supir_amd.modules import sampling
from supir_amd.synth import synth_tensor

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
device = torch.device("cuda", 0)
model, _, _ = bench.build_model(device, 0, 1)
model.model.enable_graph(True)
P = 1024
x = synth_tensor("bench.img0.1", (1, 3, P, P), scale=0.5).clamp(-1, 1).to(device)
c = {"crossattn": synth_tensor("bench.c", (1, 77, 2048)).to(device), "vector": synth_tensor("bench.v", (1, 2816)).to(device)}
uc = {"crossattn": synth_tensor("bench.uc", (1, 77, 2048)).to(device), "vector": synth_tensor("bench.uv", (1, 2816)).to(device)}
res, outs = {}, {}
for v in (True, False):          # warm both paths (autotune, graph capture)
    sampling.EMB_SCHEDULE = v
    bench.one_image(model, x, (c, uc), 1234, 50)
for rep in range(reps):
    for v in (True, False):
        sampling.EMB_SCHEDULE = v
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = bench.one_image(model, x, (c, uc), 1234, 50)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res.setdefault("sched_on" if v else "sched_off", []).append(round(dt, 4))
        outs[v] = out.clone()
        print(f"rep{rep} EMB_SCHEDULE={v}: {dt:.4f} s/image", flush=True)
sampling.EMB_SCHEDULE = True
print("bitwise equal images:", bool(torch.equal(outs[True], outs[False])))
print(json.dumps(res))
