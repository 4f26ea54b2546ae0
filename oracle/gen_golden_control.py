"""ORACLE -- TEST INFRASTRUCTURE ONLY.  The ten feature maps of the REFERENCE's GLVControl.forward (SUPIR/modules/SUPIR_v0.py:499-540),
as FULL tensors (VERDICT r03 "weak" 2: tests/golden/golden_mini.pt holds only a digest -- shape, std, first / last 32 values -- of
each map, which a permutation inside a map that cancels downstream would pass).

    python -m oracle.gen_golden_control        # build container (needs /root/reference); ~1 min; writes tests/golden/golden_control.pt

Same model and inputs as the `control_digest` entry of oracle/gen_golden.py: depth [1, 1, 2], real widths, supir_amd.synth weights by
key, latent 16 x 16, B = 2, timesteps [500, 37]; 0.72 M values, stored fp32 (2.9 MB)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import as R  # noqa: E402
from supir_amd.synth import fill_state_dict_, synth_tensor  # noqa: E402


def main():
    ns = R.load_reference()
    _, ctl, _ = R.unet_params(depth=[1, 1, 2])
    with R.quiet():
        ctrl = ns.GLVControl(**ctl).eval()
    fill_state_dict_({"model.control_model." + k: v for k, v in ctrl.state_dict().items()})
    B = 2
    x, lq = synth_tensor("xt", (B, 4, 16, 16)), synth_tensor("lq", (B, 4, 16, 16))
    y, ctx = synth_tensor("vector", (B, 2816)), synth_tensor("context", (B, 77, 2048))
    t = torch.tensor([500, 37], dtype=torch.int64)
    with torch.no_grad():
        hs = ctrl(x=lq, timesteps=t, xt=x, context=ctx, y=y)
    out = os.path.join(ROOT, "tests", "golden", "golden_control.pt")
    torch.save({"control_features": [h.float().contiguous().clone() for h in hs],
                "note": "reference GLVControl.forward, depth [1,1,2], latent 16x16, inputs as golden_mini.pt's control_digest"}, out)
    print("wrote", out, [tuple(h.shape) for h in hs])


if __name__ == "__main__":
    main()
