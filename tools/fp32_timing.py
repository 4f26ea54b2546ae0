"""What the fp32 service costs: one CFG-doubled UNet + control call at 1024^2 (latent 128^2, full depth) and the VAE encode / decode of a 1024^2
image in an fp32 scope next to the bf16 build, same process, same inputs; plus the rate of the general tile kernel on two plain GEMMs.
usage: python tools/fp32_timing.py [out.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supir_amd import ops  # noqa: E402
from supir_amd.configs import supir_v0_config  # noqa: E402
from supir_amd.plugin import instantiate_from_config  # noqa: E402
from supir_amd.synth import synth_param, synth_tensor  # noqa: E402

dev = "cuda"
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fp32_timing.json"


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {}
for (M, N, K) in [(8192, 1280, 1280), (8192, 5120, 1280), (4096, 4096, 4096)]:
    a, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * K ** -0.5
    ms = timed(lambda: ops.gemm(a, w), 5)
    res[f"gemm_{M}x{N}x{K}"] = {"ms": round(ms, 3), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}
x = torch.randn(1, 256, 256, 256, device=dev)
wk = torch.randn(256, 3, 3, 256, device=dev) * (9 * 256) ** -0.5
ms = timed(lambda: ops.conv3x3(x, wk), 5)
res["conv3x3_256x256_256to256"] = {"ms": round(ms, 3), "tflops": round(2.0 * 65536 * 256 * 2304 / ms / 1e9, 1)}
print(json.dumps(res), flush=True)

with torch.device(dev):
    model = instantiate_from_config(supir_v0_config(sampler_device=dev))
with torch.no_grad():
    for k, t in model.state_dict().items():
        if t.is_floating_point() and k != "denoiser.sigmas":
            t.copy_(synth_param(k, t.shape, device=dev))
B, L = 2, 128
xt = synth_tensor("xt128", (B, 4, L, L)).to(dev)
cond = {"crossattn": synth_tensor("context", (B, 77, 2048)).to(dev), "vector": synth_tensor("vector", (B, 2816)).to(dev),
        "control": synth_tensor("lq128", (B, 4, L, L)).to(dev)}
t = torch.tensor([500, 500], dtype=torch.int64, device=dev)
img = synth_tensor("img1024", (1, 3, 1024, 1024), scale=0.5).clamp(-1, 1).to(dev)
net = model.model
net.enable_graph(False)
with torch.no_grad():
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        net.dtype = model.ae_dtype = dt
        out = net(xt, t, cond, 1.0)
        res[f"network_call_1024px_{name}_ms"] = round(timed(lambda: net(xt, t, cond, 1.0), 3), 2)
        z = model.encode_first_stage_with_denoise(img, use_sample=False)
        res[f"vae_encode_1024px_{name}_ms"] = round(timed(lambda: model.encode_first_stage_with_denoise(img, use_sample=False), 2), 2)
        res[f"vae_decode_1024px_{name}_ms"] = round(timed(lambda: model.decode_first_stage(z), 2), 2)
        res[f"peak_mem_gb_{name}"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
        if name == "bf16":
            ref = out.clone()
        else:
            res["network_call_bf16_vs_fp32_rel_l2"] = float(((ref - out).norm() / out.norm()).item())
        print(json.dumps(res), flush=True)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
