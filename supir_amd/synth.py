"""Deterministic synthetic weights / inputs ("random-init SDXL + SUPIR-control weights", BASELINE.json).

There is no network for checkpoints, and the reference's own random init is useless for parity: `zero_module`
zeroes every ResBlock out-conv, SpatialTransformer.proj_out, every ZeroSFT conv and the final UNet conv, so the network
output would be identically zero (SURVEY.md section 7 hard part 1).  Every parameter is therefore filled from an integer hash
of (key name, element index): bit-identical on CPU and GPU, independent of construction order and of torch's RNG
streams, cheap enough for 3.9 G parameters, and it needs no stored checkpoint -- a (key, shape) manifest is enough.
"""
import math
import zlib

import torch


def _hash_uniform(numel, seed, device):
    """16-bit uniform in [-1, 1) from a 32-bit integer hash; exact in fp32, identical on every device."""
    i = torch.arange(numel, dtype=torch.int64, device=device)
    h = (i * 2654435761 + (seed % 65521) * 40503 + 12345) & 0xFFFFFFFF
    h = h ^ (h >> 15)
    h = (h * 2246822519) & 0xFFFFFFFF
    h = h ^ (h >> 13)
    h = (h * 3266489917) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    u = (h >> 8) & 0xFFFF
    return (u.to(torch.float32) - 32768.0) * (1.0 / 32768.0)


_ZERO_INIT_MARKERS = (".out_layers.3.", ".proj_out.", ".zero_mul.", ".zero_add.", ".zero_conv.", ".out.2.",
                      "input_hint_block.0.")
# output projections of the other residual branches (attention to_out, feed-forward net.2): not zero-initialised by the
# reference, but given the same modest gain so that a 70-block-deep random network is not chaotic in bf16 (with O(1)
# branches the ATen-autocast bf16 path itself lands 23 % away from fp32 at full depth -- useless as a parity bar).
_BRANCH_OUT_MARKERS = (".to_out.0.", ".ff.net.2.",
                       # residual-branch outputs of the text towers (transformers CLIPTextModel / open_clip key names)
                       ".self_attn.out_proj.", ".attn.out_proj.", ".mlp.fc2.", ".mlp.c_proj.")


def _is_zero_init(key):
    """Parameters the reference zero-initialises (openaimodel.py:299-307,947-953; attention.py:606-611;
    SUPIR_v0.py:82-87,481-483).  VAE keys (first_stage_model.*) are never zero-initialised."""
    return (not key.startswith("first_stage_model.")) and any(m in key for m in _ZERO_INIT_MARKERS + _BRANCH_OUT_MARKERS)


def synth_param(key, shape, device="cpu", seed=0):
    """Value of parameter `key` (reference state-dict name) with `shape`."""
    shape = tuple(int(s) for s in shape)
    numel = int(math.prod(shape)) if len(shape) else 1
    v = _hash_uniform(numel, zlib.crc32(key.encode()) + 7919 * seed, device)
    if len(shape) >= 2:
        fan_in = int(math.prod(shape[1:]))
        gain = 1.7
        last = key.rsplit(".", 2)
        name = last[-2] if len(last) >= 2 else ""
        if name in ("to_q", "to_k", "q", "k"):
            gain = 2.2  # logits std ~2: a non-uniform softmax, so attention is a real test
        elif _is_zero_init(key):
            gain = 0.4  # the reference zero-initialises these (zero_module): trained residual branches stay modest
        v = v * torch.tensor(gain / math.sqrt(fan_in), dtype=torch.float32, device=device)
    elif key.endswith("weight"):  # 1-D weight == GroupNorm / LayerNorm scale
        v = 1.0 + 0.1 * v
    else:  # biases
        v = 0.05 * v
    return v.reshape(shape)


def fill_state_dict_(module_or_sd, device=None, seed=0, skip=("sigmas",)):
    """Overwrite every floating tensor of a module's state dict (or a {key: tensor} dict) in place."""
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    with torch.no_grad():
        for k, t in sd.items():
            if not t.is_floating_point() or any(k.endswith(s) for s in skip):
                continue
            t.copy_(synth_param(k, t.shape, device=device or t.device, seed=seed).to(t.dtype))
    return module_or_sd


def synth_tensor(name, shape, device="cpu", scale=1.0, seed=0):
    """Deterministic input tensor (sum of three hash uniforms ~ roughly bell shaped, std ~= scale)."""
    numel = int(math.prod(shape))
    s = zlib.crc32(name.encode()) + 104729 * seed
    v = _hash_uniform(numel, s, device) + _hash_uniform(numel, s + 1, device) + _hash_uniform(numel, s + 2, device)
    return (v * scale).reshape(tuple(shape))
