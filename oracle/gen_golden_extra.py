"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Extra fixtures from the REAL reference (/root/reference) for options of the path that
`gen_golden.py` does not exercise: the sampler's `use_linear_control_scale` / `control_scale_start` (sampling.py:572-596) and
the AdaIN colour fix (SUPIR/utils/colorfix.py:59-70, selected by color_fix_type='AdaIn', SUPIR_model.py:132-134), and the DPM++ 2M restore
samplers constructing / querying their Brownian-tree noise sampler themselves (sampling.py:494-499, 687-692).

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
    # The reference's DPM++ 2M restore samplers (sampling.py:422-515, 663-730) constructing and querying the Brownian-tree noise
    # sampler THEMSELVES (:494 / :687 `BrownianTreeNoiseSampler(x, sigmas_min, sigmas_max)`, :499 / :692 `noise_sampler(s_in * sigmas[i],
    # s_in * sigmas[i + 1])`).  k-diffusion / torchsde are not installable: the name the reference imports is bound to the product's
    # restatement (supir_amd/modules/brownian.py, k-diffusion's interface over a virtual Brownian tree) and the schedule to the oracle's
    # Karras restatement.  What this fixture pins is the reference's CALL SITES -- constructor arguments (0-dim CPU tensors), query pattern,
    # consumption of the global generator (the tree's seed is its one draw) -- against the product samplers run under the same torch seed;
    # torchsde's own seed -> noise map stays parity-unpinned.
    import sgm.modules.diffusionmodules.sampling as S
    from oracle.supir_oracle import kdiff_get_sigmas_karras
    from supir_amd.modules.brownian import BrownianTreeNoiseSampler
    S.get_sigmas_karras = kdiff_get_sigmas_karras
    queries = []

    class Recording(BrownianTreeNoiseSampler):
        def __call__(self, sigma, sigma_next):
            queries.append((float(sigma.reshape(-1)[0]), float(sigma_next.reshape(-1)[0])))
            return super().__call__(sigma, sigma_next)

    S.BrownianTreeNoiseSampler = Recording
    for steps in (8, 4):
        dsm = S.RestoreDPMPP2MSampler(num_steps=steps, s_noise=1.003, eta=1.0, **sampler_cfg)
        torch.manual_seed(1234)
        gold[f"sampler_dpmpp_brownian_{steps}"] = dsm(lambda inp, sigma, cc, s_: den(fake_net, inp, sigma, cc, s_), x0.clone(),
                                                      cond=dict(c), uc=dict(uc), control_scale=0.9).clone()
        gold[f"sampler_dpmpp_brownian_{steps}_queries"] = torch.tensor(queries, dtype=torch.float64)
        queries.clear()
        print("dpmpp brownian", steps, gold[f"sampler_dpmpp_brownian_{steps}"].std().item())
    big = (1, 4, 24, 40)
    lqb = synth_tensor("lq_big", big)
    orig_tensor = S.torch.tensor
    S.torch.tensor = lambda *a_, **k: orig_tensor(*a_, **{kk: vv for kk, vv in k.items() if kk != "device"})   # gaussian_weights: device='cuda' literal (:750)
    try:
        tdsm = S.TiledRestoreDPMPP2MSampler(tile_size=16, tile_stride=8, num_steps=4, s_noise=1.003, eta=1.0, **sampler_cfg)
        torch.manual_seed(4321)
        gold["sampler_dpmpp_brownian_tiled_4"] = tdsm(lambda inp, sigma, cc, s_: den(fake_net, inp, sigma, cc, s_),
                                                      synth_tensor("noised_big", big), cond=dict(c, control=lqb),
                                                      uc=dict(uc, control=lqb), control_scale=1.0).clone()
    finally:
        S.torch.tensor = orig_tensor
    queries.clear()

    # Local (per-tile) prompts: `cond` is a LIST with one conditioning dict per latent tile (SUPIR_model.py:168-178 builds it, the tiled
    # samplers branch on isinstance(cond, list): sampling.py:609-616, 640-643 and 673-680, 705-708).  The analytic network here reads
    # crossattn and vector, so a tile's own prompt matters.
    def prompt_net(xin, tt, cc, cs):
        bias = 0.3 * cc["crossattn"].mean(dim=(1, 2)).view(-1, 1, 1, 1) * 40.0 + 0.2 * cc["vector"].mean(dim=1).view(-1, 1, 1, 1) * 40.0
        return torch.tanh(xin * 0.7 + cc["control"] * 0.1 + bias) * (1.0 + 0.001 * tt.view(-1, 1, 1, 1).float()) * cs

    n_tiles = len(S._sliding_windows(24, 40, 16, 8))
    local = [{"crossattn": synth_tensor(f"local.ctx{j}", (1, 77, 2048)), "vector": synth_tensor(f"local.vec{j}", (1, 2816)), "control": lqb}
             for j in range(n_tiles)]
    ucb = {"crossattn": ctx[1:], "vector": y[1:], "control": lqb}
    S.torch.tensor = lambda *a_, **k: orig_tensor(*a_, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        tsm = S.TiledRestoreEDMSampler(tile_size=16, tile_stride=8, num_steps=3, s_churn=5, s_noise=1.01, restore_cfg=4.0, **sampler_cfg)
        tdsm = S.TiledRestoreDPMPP2MSampler(tile_size=16, tile_stride=8, num_steps=4, s_noise=1.003, eta=1.0, **sampler_cfg)
    finally:
        S.torch.tensor = orig_tensor
    it = iter([synth_tensor(f"local.eps{i}", big) for i in range(3)])
    torch.randn_like = lambda t_, **k2: next(it).to(t_)
    try:
        gold["sampler_tiled_local_prompts"] = tsm(lambda inp, sigma, cc, s_: den(prompt_net, inp, sigma, cc, s_), synth_tensor("noised_big", big),
                                                  cond=[dict(cj) for cj in local], uc=dict(ucb), x_center=synth_tensor("xc_big", big),
                                                  control_scale=0.9).clone()
    finally:
        torch.randn_like = orig
    torch.manual_seed(777)
    gold["sampler_dpmpp_tiled_local_prompts"] = tdsm(lambda inp, sigma, cc, s_: den(prompt_net, inp, sigma, cc, s_),
                                                     synth_tensor("noised_big", big), cond=[dict(cj) for cj in local], uc=dict(ucb),
                                                     control_scale=1.0).clone()
    # the same run with ONE prompt for all tiles differs: the fixture really exercises the per-tile branch
    it = iter([synth_tensor(f"local.eps{i}", big) for i in range(3)])
    torch.randn_like = lambda t_, **k2: next(it).to(t_)
    try:
        single = tsm(lambda inp, sigma, cc, s_: den(prompt_net, inp, sigma, cc, s_), synth_tensor("noised_big", big), cond=dict(local[0]),
                     uc=dict(ucb), x_center=synth_tensor("xc_big", big), control_scale=0.9)
    finally:
        torch.randn_like = orig
    print("local prompts: tiled", gold["sampler_tiled_local_prompts"].std().item(), "vs one prompt rel",
          ((single - gold["sampler_tiled_local_prompts"]).norm() / single.norm()).item(),
          "dpmpp tiled", gold["sampler_dpmpp_tiled_local_prompts"].std().item())
    queries.clear()

    # SUPIRModel.prepare_condition (SUPIR/models/SUPIR_model.py:152-179) run UNBOUND on a stand-in `self` whose conditioner records what
    # it is asked: the batch keys / values it builds, the prompt concatenation, and -- for local prompts (p[0] a list) -- one conditioner
    # call per tile with the unconditional batch on the first only.
    import types
    import warnings
    from SUPIR.models.SUPIR_model import SUPIRModel

    class RecordingConditioner:
        def __init__(self):
            self.calls = []

        def get_unconditional_conditioning(self, batch, batch_uc=None):
            rec = lambda b_: None if b_ is None else {k: (v.tolist() if torch.is_tensor(v) and v.numel() <= 8 else (list(v.shape) if torch.is_tensor(v) else v))  # noqa: E731
                                                      for k, v in sorted(b_.items())}
            self.calls.append((rec(batch), rec(batch_uc)))
            n = len(self.calls)
            return {"tag": f"c{n}"}, (None if batch_uc is None else {"tag": f"uc{n}"})

    z4 = synth_tensor("lq", (2, 4, 16, 16))
    for name, p_, n_ in (("plain", ["a cat", "a dog"], 2), ("local", [["tile zero", "tile one", "tile two"]], 1)):
        stub = types.SimpleNamespace(conditioner=RecordingConditioner(), ae_dtype=torch.bfloat16)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c_, uc_ = SUPIRModel.prepare_condition(stub, z4[:n_], p_, ", best quality", "blurry", n_)
        gold["prepare_condition_" + name] = {"calls": stub.conditioner.calls, "c": c_, "uc": uc_}
    a, b = synth_tensor("wa", (2, 3, 24, 40)), synth_tensor("wb", (2, 3, 24, 40), scale=0.5) + 0.2
    gold["adain"] = adaptive_instance_normalization(a, b).clone()
    torch.save(gold, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
