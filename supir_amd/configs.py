"""The hot-path part of options/SUPIR_v0.yaml as a plain dict (same `target:` strings and params; the YAML itself is the
reference's file and is not copied).  `supir_v0_config()` is what bench.py / smoke build the model from; a user with the
reference checkout simply loads the YAML and calls plugin.instantiate_from_config on `config.model`."""
import copy

_DDPM = {"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"}

_UNET = dict(adm_in_channels=2816, num_classes="sequential", use_checkpoint=True, in_channels=4, out_channels=4,
             model_channels=320, attention_resolutions=[4, 2], num_res_blocks=2, channel_mult=[1, 2, 4],
             num_head_channels=64, use_spatial_transformer=True, use_linear_in_transformer=True,
             transformer_depth=[1, 2, 10], context_dim=2048, spatial_transformer_attn_type="softmax-xformers", legacy=False)

_VAE_DD = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
               ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def supir_v0_config(transformer_depth=None, sampler="RestoreEDMSampler", sampler_device="cuda", **sampler_extra):
    unet = dict(_UNET)
    if transformer_depth is not None:
        unet["transformer_depth"] = list(transformer_depth)
    cfg = {
        "target": "SUPIR.models.SUPIR_model.SUPIRModel",
        "params": {
            # the YAML says diffusion_dtype fp16; the HIP path computes bf16 (modules/wrappers.py): asked for explicitly here
            "ae_dtype": "bf16", "diffusion_dtype": "bf16", "scale_factor": 0.13025, "disable_first_stage_autocast": True,
            "network_wrapper": "sgm.modules.diffusionmodules.wrappers.ControlWrapper",
            "denoiser_config": {"target": "sgm.modules.diffusionmodules.denoiser.DiscreteDenoiserWithControl",
                                "params": {"num_idx": 1000, "discretization_config": _DDPM,
                                           "weighting_config": {"target": "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting"},
                                           "scaling_config": {"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}}},
            "control_stage_config": {"target": "SUPIR.modules.SUPIR_v0.GLVControl", "params": dict(unet, input_upscale=1)},
            "network_config": {"target": "SUPIR.modules.SUPIR_v0.LightGLVUNet",
                               "params": dict(unet, mode="XL-base", project_type="ZeroSFT", project_channel_scale=2)},
            "conditioner_config": None,  # bench / smoke pass prepared cond=(c, uc) (no tokeniser vocabulary in this image); the YAML's
            # conditioner_config builds supir_amd.modules.conditioner through the plugin like every other target
            "first_stage_config": {"target": "sgm.models.autoencoder.AutoencoderKLInferenceWrapper",
                                   "params": {"ckpt_path": None, "embed_dim": 4, "monitor": "val/rec_loss",
                                              "ddconfig": dict(_VAE_DD), "lossconfig": {"target": "torch.nn.Identity"}}},
            "sampler_config": {"target": f"sgm.modules.diffusionmodules.sampling.{sampler}",
                               "params": dict({"num_steps": 100, "restore_cfg": 4.0, "s_churn": 0, "s_noise": 1.003,
                                               "device": sampler_device, "discretization_config": _DDPM,
                                               "guider_config": {"target": "sgm.modules.diffusionmodules.guiders.LinearCFG",
                                                                 "params": {"scale": 7.5, "scale_min": 4.0}}},
                                              **sampler_extra)},
        },
    }
    return copy.deepcopy(cfg)
