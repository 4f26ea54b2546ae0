"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Extra fixtures from the REAL reference (/root/reference) for options of the path that
`gen_golden.py` does not exercise: the sampler's `use_linear_control_scale` / `control_scale_start` (sampling.py:572-596) and
the AdaIN colour fix (SUPIR/utils/colorfix.py:59-70, selected by color_fix_type='AdaIn', SUPIR_model.py:132-134).

    python -m oracle.gen_golden_extra      # seconds; writes tests/golden/golden_extra.pt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import as R  # noqa: E402
from supir_amd.synth import synth_tensor  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "golden_extra.pt")


def main():
    R.load_reference()
    from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiserWithControl
    from sgm.modules.diffusionmodules.sampling import RestoreEDMSampler
    from SUPIR.utils.colorfix import adaptive_instance_normalization

    gold = {}
    den = DiscreteDenoiserWithControl(
        weighting_config={"target": "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting"},
        scaling_config={"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}, num_idx=1000,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"})
    sampler_cfg = dict(discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
                       guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearCFG",
                                      "params": {"scale": 1.0, "scale_min": 4.0}}, device="cpu")

    def fake_net(xin, tt, cc, cs):   # same analytic network as gen_golden.py: the control scale multiplies the output
        return torch.tanh(xin * 0.7 + cc["control"] * 0.1) * (1.0 + 0.001 * tt.view(-1, 1, 1, 1).float()) * cs

    ctx, y, lq = synth_tensor("context", (2, 77, 2048)), synth_tensor("vector", (2, 2816)), synth_tensor("lq", (2, 4, 16, 16))
    c = {"crossattn": ctx[:1], "vector": y[:1], "control": lq[:1]}
    uc = {"crossattn": ctx[1:], "vector": y[1:], "control": lq[:1]}
    xc, x0 = synth_tensor("x_center", (1, 4, 16, 16)), synth_tensor("noised_z", (1, 4, 16, 16))
    orig = torch.randn_like
    for name, steps, rcfg, cs, cs0 in (("lin_cs_12", 12, 4.0, 1.0, 0.0), ("lin_cs_8", 8, -1.0, 0.8, 0.3)):
        smp = RestoreEDMSampler(num_steps=steps, s_churn=5, s_noise=1.01, restore_cfg=rcfg, **sampler_cfg)
        it = iter([synth_tensor(f"{name}.eps{i}", (1, 4, 16, 16)) for i in range(steps)])
        torch.randn_like = lambda t_, **k2: next(it).to(t_)
        try:
            gold["sampler_" + name] = smp(lambda inp, sigma, cc, s_: den(fake_net, inp, sigma, cc, s_), x0.clone(), cond=dict(c),
                                          uc=dict(uc), x_center=xc, control_scale=cs, use_linear_control_scale=True,
                                          control_scale_start=cs0).clone()
        finally:
            torch.randn_like = orig
        print(name, gold["sampler_" + name].std().item())
    a, b = synth_tensor("wa", (2, 3, 24, 40)), synth_tensor("wb", (2, 3, 24, 40), scale=0.5) + 0.2
    gold["adain"] = adaptive_instance_normalization(a, b).clone()
    torch.save(gold, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
