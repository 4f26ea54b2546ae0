"""Boundary proof: the REFERENCE's own construction path -- `SUPIR.util.create_SUPIR_model`
(/root/reference/SUPIR/util.py:34-51: OmegaConf.load -> instantiate_from_config(config.model) -> load_state_dict(strict=False))
-- run on the reference's real `options/SUPIR_v0.yaml` and `options/SUPIR_v0_tiled.yaml` after `supir_amd.plugin.install()`
must build THIS package's classes, accept reference-keyed state dicts, and expose the attribute protocol test.py uses
(test.py:62-68).  The YAML is used as is, text conditioner included (sgm.modules.GeneralConditionerWithControl with the CLIP-L /
OpenCLIP-bigG embedders resolves to supir_amd.modules.conditioner); one parametrisation swaps in a stub conditioner to show
that a user-supplied conditioner object still plugs in.

CPU tier; needs the reference checkout (present in the build container, absent on the GPU box -> skipped there).  The pip
packages the reference imports but this image lacks (omegaconf, cv2, ...) come from the oracle's inert import stubs: test
infrastructure, never touched by the product path.
"""
import os
import sys

import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not mounted")


class StubConditioner(torch.nn.Module):
    """Stands in for sgm.modules.GeneralConditionerWithControl (sgm/modules/encoders/modules.py:193-243): same call surface."""

    def __init__(self, emb_models=None):
        super().__init__()
        self.n_embedders = len(emb_models or [])

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None):
        n = len(batch_c["txt"])
        mk = lambda: {"crossattn": torch.zeros(n, 77, 2048), "vector": torch.zeros(n, 2816), "control": batch_c["control"]}
        return mk(), (mk() if batch_uc is not None else None)


@pytest.fixture(scope="module")
def ref_util():
    """The reference's SUPIR.util module, imported for real (with stubs for missing pip packages), AFTER plugin.install()."""
    ref_import._install_stubs()
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    # an earlier test may have called plugin.install() WITHOUT the reference on sys.path, which registers empty alias packages
    # named sgm / SUPIR: drop those so that the real packages are imported (and patched in place) this time
    for name in [n for n in sys.modules if n == "sgm" or n.startswith("sgm.") or n == "SUPIR" or n.startswith("SUPIR.")]:
        if getattr(sys.modules[name], "__file__", None) is None:
            del sys.modules[name]
    from supir_amd import plugin
    aliased = plugin.install()
    assert "SUPIR.models.SUPIR_model.SUPIRModel" in aliased
    with ref_import.quiet():
        import SUPIR.util as ref_util_mod
    return ref_util_mod


def _yaml_without_checkpoints(tmp_path, name, stub_conditioner=False):
    """The reference YAML, byte for byte, except: checkpoint paths -> null (no weights here) [and the conditioner target -> stub]."""
    import yaml
    src = os.path.join(ref_import.REF_ROOT, "options", name)
    cfg = yaml.safe_load(open(src))
    for k in ("SDXL_CKPT", "SUPIR_CKPT", "SUPIR_CKPT_F", "SUPIR_CKPT_Q"):
        assert k in cfg
        cfg[k] = None
    cond = cfg["model"]["params"]["conditioner_config"]
    assert cond["target"] == "sgm.modules.GeneralConditionerWithControl"
    if stub_conditioner:
        cond["target"] = "tests.test_boundary_reference_yaml.StubConditioner"
    out = tmp_path / name
    yaml.safe_dump(cfg, open(out, "w"))
    return str(out), cfg


@pytest.mark.parametrize("name,sampler", [("SUPIR_v0.yaml", "RestoreEDMSampler"), ("SUPIR_v0_tiled.yaml", "TiledRestoreEDMSampler")])
def test_create_supir_model_from_reference_yaml_builds_this_package(ref_util, tmp_path, name, sampler):
    import supir_amd.models.supir_model as M
    import supir_amd.modules.sampling as S
    import supir_amd.modules.supir_v0 as V
    import supir_amd.modules.vae as VA
    import supir_amd.modules.wrappers as W
    path, cfg = _yaml_without_checkpoints(tmp_path, name)
    with ref_import.quiet():
        model = ref_util.create_SUPIR_model(path, SUPIR_sign=None)       # the reference's function, unmodified
    assert type(model) is M.SUPIRModel
    assert type(model.model) is W.ControlWrapper
    assert type(model.model.diffusion_model) is V.LightGLVUNet and type(model.model.control_model) is V.GLVControl
    assert type(model.first_stage_model) is VA.AutoencoderKLInferenceWrapper
    assert type(model.denoiser) is S.DiscreteDenoiserWithControl
    assert type(model.sampler).__name__ == sampler and type(model.sampler).__module__ == S.__name__
    if sampler.startswith("Tiled"):
        assert model.sampler.tile_size == 128 and model.sampler.tile_stride == 64
    import supir_amd.modules.conditioner as CD
    assert type(model.conditioner) is CD.GeneralConditionerWithControl and len(model.conditioner.embedders) == 5
    assert type(model.conditioner.embedders[0]) is CD.FrozenCLIPEmbedder and model.conditioner.embedders[0].layer_idx == 11
    assert type(model.conditioner.embedders[1]) is CD.FrozenOpenCLIPEmbedder2 and model.conditioner.embedders[1].heads == 20
    assert [e.input_key for e in model.conditioner.embedders] == ["txt", "txt", "original_size_as_tuple", "crop_coords_top_left",
                                                                  "target_size_as_tuple"]
    # YAML params arrived: dtypes, scale factor (options/SUPIR_v0.yaml:4-6)
    assert model.ae_dtype == torch.bfloat16 and model.model.dtype == torch.float16 and model.scale_factor == 0.13025
    for p in model.parameters():
        assert p.device.type == "cpu"

    # state-dict round trip with the reference's keys, strict=False exactly like SUPIR/util.py:38-47
    keys = ["model.diffusion_model.input_blocks.4.1.transformer_blocks.0.attn1.to_q.weight",
            "model.diffusion_model.project_modules.0.zero_conv.weight", "model.diffusion_model.out.2.bias",
            "model.control_model.input_hint_block.0.weight", "model.control_model.middle_block.1.proj_in.weight",
            "first_stage_model.decoder.conv_out.weight", "first_stage_model.denoise_encoder.down.0.block.0.conv1.weight",
            "first_stage_model.quant_conv.bias"]
    own = model.state_dict()
    sd = {}
    g = torch.Generator().manual_seed(0)
    for k in keys:
        assert k in own, k
        sd[k] = torch.randn(own[k].shape, generator=g)
    keys += ["conditioner.embedders.0.transformer.text_model.final_layer_norm.weight",
             "conditioner.embedders.1.model.transformer.resblocks.31.attn.in_proj_weight", "conditioner.embedders.1.model.text_projection"]
    for k in keys[-3:]:
        assert k in own, k
        sd[k] = torch.randn(own[k].shape, generator=g)
    sd["conditioner.embedders.0.transformer.text_model.embeddings.position_ids"] = torch.arange(77)[None]   # older transformers saved it
    res = model.load_state_dict(sd, strict=False)
    assert res.unexpected_keys == ["conditioner.embedders.0.transformer.text_model.embeddings.position_ids"]
    after = model.state_dict()
    for k in keys:
        assert torch.equal(after[k], sd[k]), k

    # the attribute protocol of test.py:62-68
    model.init_tile_vae(encoder_tile_size=512, decoder_tile_size=64)
    model.ae_dtype = torch.bfloat16
    model.model.dtype = torch.bfloat16
    assert callable(model.batchify_sample) and callable(model.model.load_control_model)


def test_reference_default_setting_block_is_served(ref_util, tmp_path):
    """create_SUPIR_model(..., load_default_setting=True) (SUPIR/util.py:48-50) returns the YAML's default_setting untouched; a
    user-supplied conditioner target (here a stub) still plugs in through the same `target:` mechanism."""
    path, cfg = _yaml_without_checkpoints(tmp_path, "SUPIR_v0.yaml", stub_conditioner=True)
    with ref_import.quiet():
        model, default = ref_util.create_SUPIR_model(path, load_default_setting=True)
    assert default["edm_steps"] == 50 and default["s_cfg_Quality"] == 7.5
    assert isinstance(model.conditioner, StubConditioner) and model.conditioner.n_embedders == 5
