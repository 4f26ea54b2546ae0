"""Produce supir_amd/tune_gfx950.json: the autotune winners (GEMM / conv tile per shape, fused-q|k|v choices) of THIS box for every
shape the five BASELINE configs launch, so that every later process runs the same kernels (ops.load_tuning at import).
Runs on the MI355X: python tools/make_tune.py [out.json]  (~2 minutes).  Starts from an empty state (SUPIR_TUNE_FILE=none is set here)."""
import os
import sys

os.environ["SUPIR_TUNE_FILE"] = "none"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from supir_amd import ops  # noqa: E402
from tests.helpers import build_unet, build_vae, synth_tensor  # noqa: E402

dev = "cuda"
# --vae-only: re-time the VAE's shapes alone (from an empty state, so new tile candidates get their chance) and overlay the picks on
# the packaged file -- every other pick stays what it was
VAE_ONLY = "--vae-only" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--vae-only"]
packaged = os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "tune_gfx950.json")
out_path = argv[0] if argv else packaged
wrap = None if VAE_ONLY else build_unet(device=dev)


def net_call(B, lat, dtype=torch.bfloat16, reps=2):
    x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
    cond = {"crossattn": synth_tensor("ctx", (B, 77, 2048)).to(dev), "vector": synth_tensor("y", (B, 2816)).to(dev),
            "control": synth_tensor("lq", (B, 4, lat, lat)).to(dev)}
    t = torch.full((B,), 500, dtype=torch.int64, device=dev)
    wrap.dtype = dtype
    try:
        for _ in range(reps):
            wrap(x, t, cond, 1.0)
    finally:
        wrap.dtype = torch.bfloat16
    torch.cuda.synchronize()
    print(f"  network B={B} latent={lat} {dtype}: {len(ops._TUNE)} tile picks, {len(ops._CHOICE)} choices", flush=True)


with torch.no_grad():
    if not VAE_ONLY:
        net_call(2, 128)                      # config 2 (the bench): one 1024^2 image, CFG-doubled
        net_call(8, 128)                      # 4 images per call / tile_batch 4 of the tiled sampler (config 3)
        net_call(2, 64)                       # config 1 (512^2)
        net_call(2, 128, torch.float16)       # config 5 (diff_dtype fp16)
        net_call(2, 32)                       # the reduced-size shapes the test suite uses
        net_call(2, 16)
    vae = build_vae(dev)
    for px in (1024, 512):
        img = synth_tensor("img", (1, 3, px, px), scale=0.5).to(dev)
        mom = vae.quant_conv(vae.denoise_encoder(img))
        z = mom[:, :4]
        vae.decoder(vae.post_quant_conv(z))
        vae.quant_conv(vae.encoder(img))
        torch.cuda.synchronize()
        print(f"  VAE {px}px: {len(ops._TUNE)} tile picks", flush=True)
if VAE_ONLY:
    fresh_t, fresh_c = dict(ops._TUNE), dict(ops._CHOICE)
    ops._TUNE.clear()
    ops._CHOICE.clear()
    ops.load_tuning(packaged)
    changed = {k: (ops._TUNE.get(k), v) for k, v in fresh_t.items() if ops._TUNE.get(k) != v}
    print(f"  overlay on {packaged}: {len(fresh_t)} VAE picks, {len(changed)} differ from the packaged ones", flush=True)
    for k, (old, new) in sorted(changed.items(), key=repr):
        print("   ", k, old, "->", new, flush=True)
    ops._TUNE.update(fresh_t)
    ops._CHOICE.update(fresh_c)
ops.save_tuning(out_path)
print("wrote", out_path, len(ops._TUNE), "tile picks,", len(ops._CHOICE), "choices")
