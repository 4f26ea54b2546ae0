"""Shared test helpers: synthetic state dicts from the committed manifests, golden loading, error metrics."""
import json
import os

import torch

from supir_amd.synth import synth_param, synth_tensor  # noqa: F401

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest(name="mini"):
    return json.load(open(os.path.join(GOLDEN_DIR, f"manifest_{name}.json")))


def synth_sd(man, device="cpu", prefixes=None):
    sd = {}
    for k, shape in man.items():
        if prefixes is None or any(k.startswith(p) for p in prefixes):
            sd[k] = synth_param(k, shape, device=device)
    return sd


_gold = {}


def golden():
    if "g" not in _gold:
        _gold["g"] = torch.load(os.path.join(GOLDEN_DIR, "golden_mini.pt"), map_location="cpu", weights_only=False)
    return _gold["g"]


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()
