"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/* by running the REAL reference (/root/reference) on CPU fp32.

    python -m oracle.gen_golden            # in the build container (needs /root/reference); ~5 min, ~12 GB RAM

The reference ships no tests / golden vectors / known-answer files for this path, so these fixtures --
outputs of the reference's own modules on deterministic inputs and deterministic weights -- are what pins parity.
Weights are NOT stored: every parameter is a pure function of its state-dict key and shape (supir_amd/synth.py), so a
fixture holds only outputs (+ the key/shape manifest, which doubles as the state-dict compatibility contract).

Model sizes: LightGLVUNet's channel tables are hard coded (mode 'XL-base'), so fixtures use the real widths
(320/640/1280, 2048-d context, 2816-d vector); what is reduced for the CPU-sized fixtures is transformer_depth
([1,2,10] -> [1,1,2]) and the spatial size.  The full-depth manifest is recorded from a meta-device construction.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import as R  # noqa: E402
from supir_amd.synth import fill_state_dict_, synth_tensor  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MINI_DEPTH = [1, 1, 2]


def manifest_of(module, prefix):
    return {prefix + k: list(v.shape) for k, v in module.state_dict().items()}


def fill(module, prefix):
    sd = module.state_dict()
    fill_state_dict_({prefix + k: v for k, v in sd.items()})
    return module


def digest(t):
    t = t.float()
    f = t.flatten()
    return dict(shape=list(t.shape), mean=f.mean().item(), std=f.std().item(), absmax=f.abs().max().item(),
                head=f[:32].clone(), tail=f[-32:].clone())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ns = R.load_reference()

    # ------------------------------------------------------------------ manifests (full depth, meta device)
    net, ctl, vae = R.unet_params()
    with R.quiet(), torch.device("meta"):
        unet_full = ns.LightGLVUNet(**net)
        ctrl_full = ns.GLVControl(**ctl)
        enc = ns.Encoder(**vae)
        dec = ns.Decoder(**vae)
    man = {}
    man.update(manifest_of(unet_full, "model.diffusion_model."))
    man.update(manifest_of(ctrl_full, "model.control_model."))
    man.update(manifest_of(enc, "first_stage_model.encoder."))
    man.update(manifest_of(enc, "first_stage_model.denoise_encoder."))
    man.update(manifest_of(dec, "first_stage_model.decoder."))
    man["first_stage_model.quant_conv.weight"] = [8, 8, 1, 1]
    man["first_stage_model.quant_conv.bias"] = [8]
    man["first_stage_model.post_quant_conv.weight"] = [4, 4, 1, 1]
    man["first_stage_model.post_quant_conv.bias"] = [4]
    json.dump(man, open(os.path.join(OUT, "manifest_full.json"), "w"), indent=0)
    print("manifest_full:", len(man), "tensors,", sum(int(torch.tensor(s).prod()) for s in man.values()) / 1e9, "G params")
    del unet_full, ctrl_full

    # ------------------------------------------------------------------ mini model (real widths, depth [1,1,2])
    net, ctl, vae = R.unet_params(depth=MINI_DEPTH)
    with R.quiet():
        unet = ns.LightGLVUNet(**net).eval()
        ctrl = ns.GLVControl(**ctl).eval()
    fill(unet, "model.diffusion_model.")
    fill(ctrl, "model.control_model.")
    mini_man = {}
    mini_man.update(manifest_of(unet, "model.diffusion_model."))
    mini_man.update(manifest_of(ctrl, "model.control_model."))

    with R.quiet():
        enc = ns.Encoder(**vae).eval()
        denc = ns.Encoder(**vae).eval()
        dec = ns.Decoder(**vae).eval()
    fill(enc, "first_stage_model.encoder.")
    fill(denc, "first_stage_model.denoise_encoder.")
    fill(dec, "first_stage_model.decoder.")
    quant = torch.nn.Conv2d(8, 8, 1)
    pquant = torch.nn.Conv2d(4, 4, 1)
    fill(quant, "first_stage_model.quant_conv.")
    fill(pquant, "first_stage_model.post_quant_conv.")
    for m, pfx in ((enc, "encoder."), (denc, "denoise_encoder."), (dec, "decoder."), (quant, "quant_conv."),
                   (pquant, "post_quant_conv.")):
        mini_man.update(manifest_of(m, "first_stage_model." + pfx))
    json.dump(mini_man, open(os.path.join(OUT, "manifest_mini.json"), "w"), indent=0)

    gold = {}
    with torch.no_grad():
        # -------------------------------------------------------------- schedules
        den = ns.DiscreteDenoiserWithControl(
            weighting_config={"target": "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting"},
            scaling_config={"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}, num_idx=1000,
            discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"})
        gold["denoiser_table"] = den.sigmas.clone()
        from sgm.modules.diffusionmodules.discretizer import LegacyDDPMDiscretization
        disc = LegacyDDPMDiscretization()
        gold["sigmas_50"] = disc(50, device="cpu").clone()
        gold["sigmas_2"] = disc(2, device="cpu").clone()
        gold["sigmas_8"] = disc(8, device="cpu").clone()
        gold["gaussian_weights_16"] = None  # filled below (needs the cuda literal patched)

        # -------------------------------------------------------------- module level (real widths, small spatial)
        B = 2
        emb = synth_tensor("emb", (B, 1280))
        ctx = synth_tensor("context", (B, 77, 2048))
        mods = {}

        def run_mod(name, mod, *args, **kw):
            out = mod(*args, **kw)
            mods[name] = out.clone()
            print(f"  {name}: {tuple(out.shape)} std {out.std().item():.4f}")

        x320 = synth_tensor("x320", (B, 320, 8, 8))
        x640 = synth_tensor("x640", (B, 640, 8, 8))
        x1280 = synth_tensor("x1280", (B, 1280, 4, 4))
        D = unet
        run_mod("res.input_blocks.1.0", D.input_blocks[1][0], x320, emb)                 # 320->320, identity skip
        run_mod("res.input_blocks.4.0", D.input_blocks[4][0], x320, emb)                 # 320->640, 1x1 skip
        run_mod("res.output_blocks.0.0", D.output_blocks[0][0], synth_tensor("x2560", (B, 2560, 4, 4)), emb)
        run_mod("down.input_blocks.3.0", D.input_blocks[3][0], x320)                     # Downsample conv s2
        run_mod("up.output_blocks.2.2", D.output_blocks[2][2], x1280)                    # Upsample nearest+conv
        run_mod("st.input_blocks.4.1", D.input_blocks[4][1], x640, ctx)                  # depth 1, 10 heads
        run_mod("st.middle_block.1", D.middle_block[1], x1280, ctx)                      # depth 2, 20 heads
        run_mod("btb.input_blocks.7.1.0", D.input_blocks[7][1].transformer_blocks[0],
                synth_tensor("tok1280", (B, 16, 1280)), ctx)
        # project_modules (after the two inserts): 11 = SFT(no h_ori), 10 = SFT, 7 = XAttn(1280 q, 640 ctx), 0 = SFT
        P = D.project_modules
        c1280 = synth_tensor("c1280", (B, 1280, 4, 4))
        c640 = synth_tensor("c640", (B, 640, 4, 4))
        c320 = synth_tensor("c320", (B, 320, 8, 8))
        run_mod("sft.11", P[11], c1280, x1280)
        run_mod("sft.10", P[10], c1280, x1280, synth_tensor("hori1280", (B, 1280, 4, 4)))
        run_mod("sft.10.cs0.7", P[10], c1280, x1280, synth_tensor("hori1280", (B, 1280, 4, 4)), control_scale=0.7)
        run_mod("sft.0", P[0], c320, x320, synth_tensor("hori320", (B, 320, 8, 8)))
        run_mod("sft.11.cs0.6", P[11], c1280, x1280, control_scale=0.6)
        run_mod("xattn.7.cs0.6", P[7], c640, x1280, control_scale=0.6)
        run_mod("xattn.7", P[7], c640, x1280)
        run_mod("xattn.3", P[3], c320, x640)
        gold["modules"] = mods

        # -------------------------------------------------------------- full ControlWrapper forward (mini), latent 16x16
        wrap = ns.ControlWrapper(unet, dtype=torch.float32)
        wrap.load_control_model(ctrl)
        x = synth_tensor("xt", (B, 4, 16, 16))
        lq = synth_tensor("lq", (B, 4, 16, 16))
        y = synth_tensor("vector", (B, 2816))
        t = torch.tensor([500, 37], dtype=torch.int64)
        cond = {"crossattn": ctx, "vector": y, "control": lq}
        hs = ctrl(x=lq, timesteps=t, xt=x, context=ctx, y=y)
        gold["control_digest"] = [digest(h) for h in hs]
        eps = wrap(x, t, cond, 1.0)
        gold["wrapper_eps"] = eps.clone()
        gold["wrapper_eps_cs0.5"] = wrap(x, t, cond, 0.5).clone()
        print("  wrapper eps std", eps.std().item())

        # -------------------------------------------------------------- denoiser + guider + sampler on the mini model
        from sgm.modules.diffusionmodules.sampling import RestoreEDMSampler
        sampler_cfg = dict(discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
                           guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearCFG",
                                          "params": {"scale": 1.0, "scale_min": 4.0}}, device="cpu")
        smp = RestoreEDMSampler(num_steps=2, s_churn=5, s_noise=1.01, restore_cfg=4.0, **sampler_cfg)
        N = 1
        c = {"crossattn": ctx[:1], "vector": y[:1], "control": lq[:1]}
        uc = {"crossattn": ctx[1:], "vector": y[1:], "control": lq[:1]}
        xc = synth_tensor("x_center", (N, 4, 16, 16))
        x0 = synth_tensor("noised_z", (N, 4, 16, 16))
        noises = [synth_tensor(f"eps{i}", (N, 4, 16, 16)) for i in range(2)]
        it = iter(noises)
        orig = torch.randn_like
        torch.randn_like = lambda t_, **kw: next(it).to(t_)  # inject the churn noise (reference draws it at sampling.py:555)
        try:
            denoiser = lambda inp, sigma, cc, cs: den(wrap, inp, sigma, cc, cs)
            out = smp(denoiser, x0.clone(), cond=dict(c), uc=dict(uc), x_center=xc, control_scale=1.0)
        finally:
            torch.randn_like = orig
        gold["sampler_2step"] = out.clone()
        print("  sampler 2-step std", out.std().item())

        # sampler numerics alone (analytic network, many steps, restore_cfg -1 and 4, both guider settings)
        def fake_net(xin, tt, cc, cs):
            return torch.tanh(xin * 0.7 + cc["control"] * 0.1) * (1.0 + 0.001 * tt.view(-1, 1, 1, 1).float()) * cs

        for name, kw in (("fake_50_r-1", dict(num_steps=50, restore_cfg=-1.0)), ("fake_50_r4", dict(num_steps=50, restore_cfg=4.0)),
                         ("fake_8_r2", dict(num_steps=8, restore_cfg=2.0))):
            smp = RestoreEDMSampler(s_churn=5, s_noise=1.01, **kw, **sampler_cfg)
            ns_ = [synth_tensor(f"{name}.eps{i}", (N, 4, 16, 16)) for i in range(kw["num_steps"])]
            it = iter(ns_)
            torch.randn_like = lambda t_, **k2: next(it).to(t_)
            try:
                denoiser = lambda inp, sigma, cc, cs: den(fake_net, inp, sigma, cc, cs)
                gold["sampler_" + name] = smp(denoiser, x0.clone(), cond=dict(c), uc=dict(uc), x_center=xc,
                                              control_scale=0.9).clone()
            finally:
                torch.randn_like = orig

        # tiled sampler (gaussian_weights has a literal device='cuda': patch torch.tensor's device kw)
        from sgm.modules.diffusionmodules import sampling as S
        orig_tensor = torch.tensor
        S.torch.tensor = lambda *a, **k: orig_tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
        try:
            gold["gaussian_weights_16"] = S.gaussian_weights(16, 16, 1).clone()
            tsm = S.TiledRestoreEDMSampler(tile_size=16, tile_stride=8, num_steps=3, s_churn=5, s_noise=1.01, restore_cfg=4.0,
                                           **sampler_cfg)
        finally:
            S.torch.tensor = orig_tensor
        big = (N, 4, 24, 40)
        lqb = synth_tensor("lq_big", big)
        cb = {"crossattn": ctx[:1], "vector": y[:1], "control": lqb}
        ucb = {"crossattn": ctx[1:], "vector": y[1:], "control": lqb}
        ns_ = [synth_tensor(f"tiled.eps{i}", big) for i in range(3)]
        it = iter(ns_)
        torch.randn_like = lambda t_, **k2: next(it).to(t_)
        try:
            denoiser = lambda inp, sigma, cc, cs: den(fake_net, inp, sigma, cc, cs)
            gold["sampler_tiled_fake"] = tsm(denoiser, synth_tensor("noised_big", big), cond=cb, uc=ucb,
                                             x_center=synth_tensor("xc_big", big), control_scale=1.0).clone()
        finally:
            torch.randn_like = orig
        gold["sliding_windows_24_40_16_8"] = S._sliding_windows(24, 40, 16, 8)

        # DPM++ 2M restore sampler (config 5). k_diffusion is not installed: the reference class is run with the ORACLE's
        # restatement of the published k-diffusion 0.1.1 `get_sigmas_karras` (oracle/supir_oracle.py kdiff_get_sigmas_karras,
        # called exactly as the reference calls it: 0-dim fp32 tensors) -- NOT with the product's function, which
        # tests/test_host_logic.py checks against the same restatement -- and a scripted noise sampler injected, so the solver
        # arithmetic and the schedule formula are pinned while the Brownian-tree seed -> noise map stays parity-unpinned.
        from oracle.supir_oracle import kdiff_get_sigmas_karras
        S.get_sigmas_karras = kdiff_get_sigmas_karras

        class ScriptedNoise:
            def __init__(self, x, smin, smax):
                self.i = 0
                self.shape = tuple(x.shape)

            def __call__(self, s, s_next):
                self.i += 1
                return synth_tensor(f"dpm.eps{self.i}.{self.shape[-1]}", self.shape)

        S.BrownianTreeNoiseSampler = ScriptedNoise
        for steps in (8, 4):
            dsm = S.RestoreDPMPP2MSampler(num_steps=steps, s_noise=1.003, eta=1.0, **sampler_cfg)
            denoiser = lambda inp, sigma, cc, cs: den(fake_net, inp, sigma, cc, cs)
            gold[f"sampler_dpmpp_{steps}"] = dsm(denoiser, x0.clone(), cond=dict(c), uc=dict(uc), control_scale=0.9).clone()
        S.torch.tensor = lambda *a, **k: orig_tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
        try:
            tdsm = S.TiledRestoreDPMPP2MSampler(tile_size=16, tile_stride=8, num_steps=4, s_noise=1.003, eta=1.0, **sampler_cfg)
        finally:
            S.torch.tensor = orig_tensor
        denoiser = lambda inp, sigma, cc, cs: den(fake_net, inp, sigma, cc, cs)
        # fresh dicts: the reference's tiled samplers overwrite cond['control'] / uc['control'] per tile
        gold["sampler_dpmpp_tiled_4"] = tdsm(denoiser, synth_tensor("noised_big", big), cond=dict(cb, control=lqb),
                                             uc=dict(ucb, control=lqb), control_scale=1.0).clone()

        # -------------------------------------------------------------- VAE (64x64 px)
        img = synth_tensor("img", (1, 3, 64, 64), scale=0.5)
        h = denc(img)
        mom = quant(h)
        gold["vae_denoise_moments"] = mom.clone()
        z = ns.DiagonalGaussianDistribution(mom).mode() * 0.13025
        gold["vae_z"] = z.clone()
        xs1 = dec(pquant(z / 0.13025))
        gold["vae_x_stage1"] = xs1.clone()
        mom2 = quant(enc(xs1))
        gold["vae_moments2"] = mom2.clone()
        pn = synth_tensor("posterior_noise", (1, 4, 8, 8))
        post = ns.DiagonalGaussianDistribution(mom2)
        gold["vae_z_stage1"] = (post.mean + post.std * pn) * 0.13025
        print("  vae x_stage1 std", xs1.std().item())

        # -------------------------------------------------------------- tiled VAE (the reference's VAEHook, CPU)
        import types
        xf = types.ModuleType("xformers")
        xf.ops = types.ModuleType("xformers.ops")

        def _mea(q, k, v, attn_bias=None, op=None):     # [B*H, T, D] -> SDPA (what xformers computes)
            return torch.nn.functional.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]

        xf.ops.memory_efficient_attention = _mea
        sys.modules["xformers"], sys.modules["xformers.ops"] = xf, xf.ops
        from SUPIR.utils import devices as ref_devices
        ref_devices.device = torch.device("cpu")
        ref_devices.get_optimal_device = lambda: torch.device("cpu")
        from SUPIR.utils import tilevae as TV
        TV.xformers = xf
        TV.devices.device = torch.device("cpu")
        for net in (denc, dec):
            net.original_forward = net.forward
            net.mid.attn_1.attention_op = None   # attribute of MemoryEfficientAttnBlock (attn_type vanilla-xformers)
        with R.quiet():
            img_t = synth_tensor("img_tiled", (1, 3, 192, 160), scale=0.5)
            gold["tiled_enc_192x160_t64"] = TV.VAEHook(denc, 64, is_decoder=False, fast_decoder=False, fast_encoder=False,
                                                       color_fix=False, to_gpu=False)(img_t).clone()
            z_t = synth_tensor("z_tiled", (1, 4, 40, 32), scale=1.0)
            gold["tiled_dec_40x32_t8"] = TV.VAEHook(dec, 8, is_decoder=True, fast_decoder=False, fast_encoder=False,
                                                    color_fix=False, to_gpu=False)(z_t).clone()
        print("  tiled enc", tuple(gold["tiled_enc_192x160_t64"].shape), "tiled dec", tuple(gold["tiled_dec_40x32_t8"].shape))

        # colour fix
        sys.modules["torchvision.transforms"].ToTensor = lambda: None
        from SUPIR.utils.colorfix import wavelet_reconstruction
        gold["wavelet"] = wavelet_reconstruction(synth_tensor("wa", (1, 3, 64, 64)), synth_tensor("wb", (1, 3, 64, 64))).clone()

    gold["meta"] = dict(mini_depth=MINI_DEPTH, torch=torch.__version__,
                        note="outputs of the reference modules (CPU fp32) on supir_amd.synth inputs/weights")
    torch.save(gold, os.path.join(OUT, "golden_mini.pt"))
    sz = os.path.getsize(os.path.join(OUT, "golden_mini.pt"))
    print("wrote golden_mini.pt", sz / 1e6, "MB")


if __name__ == "__main__":
    main()
