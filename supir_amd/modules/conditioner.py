"""The text conditioner on the HIP kernels: same class names / constructor kwargs / state-dict keys / call
surface as sgm/modules/encoders/modules.py (GeneralConditionerWithControl :193-243, FrozenCLIPEmbedder :445-510,
FrozenOpenCLIPEmbedder2 :513-609, ConcatTimestepEmbedderND :1027-1043), for the embedder set options/SUPIR_v0.yaml:66-106 uses.

Both text towers are pre-LN transformers over 77 tokens with head dim 64 (CLIP-L: 768 wide, 12 heads, QuickGELU, 11 of its 12 layers
are needed for hidden_states[11]; OpenCLIP bigG: 1280 wide, 20 heads, erf GELU, 32 layers): LayerNorm kernel, bias GEMMs with the
activation in the epilogue, V^T projection, `supir_flash_attn_d64_ex` with the causal flag.  Runs once per image (M = 77 N rows):
latency-trivial next to the 50 network steps, built so that `prepare_condition` never leaves the GPU.

Tokenisation is host string work and needs the CLIP BPE vocabulary files, which do not ship with this package: pass token id
tensors (what the parity tests do), or point SUPIR_CLIP_TOKENIZER at a directory holding vocab.json / merges.txt (loaded through
transformers.CLIPTokenizer, the same class the reference uses; the OpenCLIP tokeniser shares the vocabulary and pads with 0).
"""
import math
import os

import torch
import torch.nn as nn

from .. import ops
from .. import weights as Wt
from .base import cdt, Linear, Norm, Prep

MAX_LENGTH = 77


def _tokenize(texts, pad_zero):
    path = os.environ.get("SUPIR_CLIP_TOKENIZER")
    if not path:
        raise RuntimeError("tokenising prompts needs the CLIP BPE vocabulary (vocab.json / merges.txt): set SUPIR_CLIP_TOKENIZER to "
                           "a directory holding them, or pass token id tensors instead of strings")
    from transformers import CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(path)
    ids = tok(list(texts), truncation=True, max_length=MAX_LENGTH, padding="max_length", return_tensors="pt")["input_ids"]
    if pad_zero:   # open_clip.tokenize: [sot] tokens [eot] then zeros
        eot = ids.argmax(dim=-1)
        ids = ids * (torch.arange(MAX_LENGTH)[None] <= eot[:, None])
    return ids


class _Block(nn.Module):
    """Pre-LN residual block; parameter containers only (names are set by the owning tower so the keys match the reference)."""

    def __init__(self, width):
        super().__init__()
        self.width = width
        object.__setattr__(self, "_qkv", Prep())


def _run_block(x, ln1, w_qk, b_qk, w_v, b_v, out, ln2, fc1, fc2, heads, act):
    """x [B, T, D] bf16 -> x + attn(LN(x)); x + mlp(LN(x)).  T = 77: K / V^T padded to 128 keys (masked by the causal flag)."""
    B, T, D = x.shape
    n = ops.layernorm(x, ln1.g32(), ln1.b32(), ln1.eps)
    qk = ops.gemm(n, w_qk, b_qk)                                    # [B, T, 2D]
    Tp = (T + 63) // 64 * 64
    vt = ops.gemm_t(n, w_v, b_v, B, T, Tp)                          # [B, D, Tp]
    a = ops.flash_attn(qk[:, :, :D], qk[:, :, D:], vt, B, heads, T, T, causal=True)
    x = ops.gemm(a, out.w(), out.b32(), residual=x)
    n = ops.layernorm(x, ln2.g32(), ln2.b32(), ln2.eps)
    h = ops.gemm(n, fc1.w(), fc1.b32(), act=act)
    return ops.gemm(h, fc2.w(), fc2.b32(), residual=x)


class _Table(nn.Module):
    """nn.Embedding's parameter (`weight` [rows, dim]) without its random initialisation: real use loads a checkpoint."""

    def __init__(self, rows, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(rows, dim), requires_grad=False)


class AbstractEmbModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.is_trainable, self.ucg_rate, self.input_key = False, 0.0, None
        self.legacy_ucg_val = None


# ------------------------------------------------------------------------------------------------ CLIP-L (transformers layout)
class _HFAttn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = Linear(d, d), Linear(d, d), Linear(d, d), Linear(d, d)


class _HFMLP(nn.Module):
    def __init__(self, d, inner):
        super().__init__()
        self.fc1, self.fc2 = Linear(d, inner), Linear(inner, d)


class _HFLayer(nn.Module):
    def __init__(self, d, inner):
        super().__init__()
        self.self_attn = _HFAttn(d)
        self.layer_norm1 = Norm(d, 1e-5)
        self.mlp = _HFMLP(d, inner)
        self.layer_norm2 = Norm(d, 1e-5)
        object.__setattr__(self, "_qk", Prep())

    def packed(self):
        a = self.self_attn
        return self._qk.get((a.q_proj.weight, a.k_proj.weight, a.q_proj.bias, a.k_proj.bias), lambda: (
            torch.cat([Wt.linear_w(a.q_proj.weight), Wt.linear_w(a.k_proj.weight)], 0).contiguous(),
            torch.cat([Wt.f32(a.q_proj.bias), Wt.f32(a.k_proj.bias)], 0).contiguous()))


class _HFEmbeddings(nn.Module):
    def __init__(self, vocab, d, n_pos):
        super().__init__()
        self.token_embedding = _Table(vocab, d)
        self.position_embedding = _Table(n_pos, d)
        self.register_buffer("position_ids", torch.arange(n_pos).unsqueeze(0), persistent=False)


class _HFEncoder(nn.Module):
    def __init__(self, d, inner, layers):
        super().__init__()
        self.layers = nn.ModuleList([_HFLayer(d, inner) for _ in range(layers)])


class _HFTextModel(nn.Module):
    def __init__(self, vocab=49408, d=768, inner=3072, layers=12, n_pos=77):
        super().__init__()
        self.embeddings = _HFEmbeddings(vocab, d, n_pos)
        self.encoder = _HFEncoder(d, inner, layers)
        self.final_layer_norm = Norm(d, 1e-5)


class _HFWrapper(nn.Module):
    def __init__(self):
        super().__init__()
        self.text_model = _HFTextModel()


class FrozenCLIPEmbedder(AbstractEmbModel):
    """CLIP ViT-L/14 text tower; keys `transformer.text_model.*` as saved by transformers' CLIPTextModel (modules.py:445-510)."""
    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None, always_return_pooled=False):
        super().__init__()
        assert layer in self.LAYERS
        if layer != "hidden" or always_return_pooled:
            raise NotImplementedError("SUPIR uses layer='hidden', layer_idx=11 (options/SUPIR_v0.yaml:70-76)")
        assert layer_idx is not None and 0 <= abs(layer_idx) <= 12
        self.transformer = _HFWrapper()
        self.device, self.max_length, self.layer, self.layer_idx = device, max_length, layer, layer_idx

    def encode_tokens(self, tokens):
        """tokens int64 [N, 77] -> hidden_states[layer_idx] fp32 [N, 77, 768]."""
        tm = self.transformer.text_model
        dev = tm.embeddings.token_embedding.weight.device
        tokens = tokens.to(dev)
        x = (tm.embeddings.token_embedding.weight[tokens] + tm.embeddings.position_embedding.weight[None, :tokens.shape[1]]).to(cdt())
        x = x.contiguous()
        n_layers = self.layer_idx if self.layer_idx >= 0 else len(tm.encoder.layers) + 1 + self.layer_idx
        for lyr in list(tm.encoder.layers)[:n_layers]:
            wqk, bqk = lyr.packed()
            a = lyr.self_attn
            x = _run_block(x, lyr.layer_norm1, wqk, bqk, a.v_proj.w(), a.v_proj.b32(), a.out_proj, lyr.layer_norm2, lyr.mlp.fc1,
                           lyr.mlp.fc2, heads=12, act=4)
        return x.float()

    @torch.no_grad()
    def forward(self, text):
        tokens = text if torch.is_tensor(text) else _tokenize(text, pad_zero=False)
        return self.encode_tokens(tokens)

    def encode(self, text):
        return self(text)


# ------------------------------------------------------------------------------------------------ OpenCLIP bigG (open_clip layout)
class _MHA(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d), requires_grad=False)
        self.in_proj_bias = nn.Parameter(torch.empty(3 * d), requires_grad=False)
        self.out_proj = Linear(d, d)


class _OCMLP(nn.Module):
    def __init__(self, d, inner):
        super().__init__()
        self.c_fc, self.c_proj = Linear(d, inner), Linear(inner, d)


class _OCBlock(nn.Module):
    def __init__(self, d, inner):
        super().__init__()
        self.ln_1 = Norm(d, 1e-5)
        self.attn = _MHA(d)
        self.ln_2 = Norm(d, 1e-5)
        self.mlp = _OCMLP(d, inner)
        object.__setattr__(self, "_p", Prep())

    def packed(self):
        a = self.attn
        d = a.in_proj_weight.shape[1]
        return self._p.get((a.in_proj_weight, a.in_proj_bias), lambda: (
            Wt.linear_w(a.in_proj_weight[:2 * d]), Wt.f32(a.in_proj_bias[:2 * d]),
            Wt.linear_w(a.in_proj_weight[2 * d:]), Wt.f32(a.in_proj_bias[2 * d:])))


class _OCTransformer(nn.Module):
    def __init__(self, d, inner, layers):
        super().__init__()
        self.resblocks = nn.ModuleList([_OCBlock(d, inner) for _ in range(layers)])


class _OCModel(nn.Module):
    def __init__(self, vocab, d, inner, layers, out_dim, n_pos=77):
        super().__init__()
        self.token_embedding = _Table(vocab, d)
        self.positional_embedding = nn.Parameter(torch.empty(n_pos, d), requires_grad=False)
        self.transformer = _OCTransformer(d, inner, layers)
        self.ln_final = Norm(d, 1e-5)
        self.text_projection = nn.Parameter(torch.empty(d, out_dim), requires_grad=False)
        self.logit_scale = nn.Parameter(torch.empty(()), requires_grad=False)
        self.register_buffer("attn_mask", torch.full((n_pos, n_pos), float("-inf")).triu_(1), persistent=False)


_OC_ARCH = {"ViT-bigG-14": dict(vocab=49408, d=1280, inner=5120, layers=32, out_dim=1280, heads=20),
            "ViT-H-14": dict(vocab=49408, d=1024, inner=4096, layers=24, out_dim=1024, heads=16)}


class FrozenOpenCLIPEmbedder2(AbstractEmbModel):
    """OpenCLIP text tower; keys `model.*` as open_clip saves them (modules.py:513-609)."""
    LAYERS = ["pooled", "last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 always_return_pooled=False, legacy=True):
        super().__init__()
        assert layer in self.LAYERS and arch in _OC_ARCH
        if not (layer == "penultimate" and always_return_pooled and not legacy):
            raise NotImplementedError("SUPIR uses layer='penultimate', always_return_pooled=True, legacy=False (SUPIR_v0.yaml:78-87)")
        cfg = dict(_OC_ARCH[arch])
        self.heads = cfg.pop("heads")
        self.model = _OCModel(**cfg)
        self.device, self.max_length, self.layer, self.return_pooled, self.legacy = device, max_length, layer, True, legacy

    def encode_tokens(self, tokens):
        """tokens int64 [N, 77] -> (penultimate fp32 [N, 77, D], pooled fp32 [N, out_dim])."""
        m = self.model
        dev = m.positional_embedding.device
        tokens = tokens.to(dev)
        x = (m.token_embedding.weight[tokens] + m.positional_embedding[None]).to(cdt()).contiguous()
        blocks = list(m.transformer.resblocks)
        pen = None
        for i, blk in enumerate(blocks):
            if i == len(blocks) - 1:
                pen = x
            wqk, bqk, wv, bv = blk.packed()
            x = _run_block(x, blk.ln_1, wqk, bqk, wv, bv, blk.attn.out_proj, blk.ln_2, blk.mlp.c_fc, blk.mlp.c_proj, heads=self.heads,
                           act=3)
        o = ops.layernorm(x, m.ln_final.g32(), m.ln_final.b32(), m.ln_final.eps)
        eot = o[torch.arange(o.shape[0], device=dev), tokens.argmax(dim=-1)]             # [N, D] rows at the eot token
        # pooled = eot @ text_projection: M = N rows -- the GEMM kernel wants K-contiguous weights: text_projection^T [out, D]
        wp = Wt.linear_w(m.text_projection.t())
        pooled = ops.gemm(eot.contiguous(), wp, None, out_dtype=torch.float32)
        return pen.float(), pooled

    @torch.no_grad()
    def forward(self, text):
        tokens = text if torch.is_tensor(text) else _tokenize(text, pad_zero=True)
        return self.encode_tokens(tokens)

    def encode(self, text):
        return self(text)


class Timestep(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class ConcatTimestepEmbedderND(AbstractEmbModel):
    """modules.py:1027-1043: scalar prep on [N, 2] (host-side arithmetic, like timestep_embedding of the UNet)."""

    def __init__(self, outdim):
        super().__init__()
        self.timestep, self.outdim = Timestep(outdim), outdim

    def forward(self, x):
        if x.ndim == 1:
            x = x[:, None]
        b, dims = x.shape
        return self.timestep(x.reshape(-1)).reshape(b, dims * self.outdim)


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1, "control_vector": 1}

    def __init__(self, emb_models):
        super().__init__()
        from ..plugin import instantiate_from_config
        embedders = []
        for cfg in emb_models:
            e = instantiate_from_config(cfg)
            e.is_trainable = cfg.get("is_trainable", False)
            e.ucg_rate = cfg.get("ucg_rate", 0.0)
            if "input_key" in cfg:
                e.input_key = cfg["input_key"]
            else:
                raise KeyError(f"need 'input_key' for embedder {type(e).__name__}")
            e.legacy_ucg_val = cfg.get("legacy_ucg_value", None)
            embedders.append(e.eval())
        self.embedders = nn.ModuleList(embedders)

    def _key_of(self, embedder, emb):
        return self.OUTPUT_DIM2KEYS[emb.dim()]

    @torch.no_grad()
    def forward(self, batch, force_zero_embeddings=None):
        output = {}
        force_zero_embeddings = force_zero_embeddings or []
        for e in self.embedders:
            if e.ucg_rate > 0.0:
                # training-time conditioning dropout (modules.py:217-228; legacy_ucg_value only acts together with ucg_rate > 0,
                # :121-134): never active on the inference path, where get_unconditional_conditioning sets every rate to 0
                # (:177-186); a direct forward() with a live rate is refused rather than silently ignored
                raise NotImplementedError("ucg_rate > 0 is a training-time option of the reference (conditioning dropout)")
            out = e(batch[e.input_key])
            for emb in (out if isinstance(out, (list, tuple)) else [out]):
                key = self._key_of(e, emb)
                if e.input_key in force_zero_embeddings:
                    emb = torch.zeros_like(emb)
                output[key] = torch.cat((output[key], emb), self.KEY2CATDIM[key]) if key in output else emb
        return output

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None):
        """modules.py:177-191: every embedder's ucg_rate is zeroed for the two encodes and restored afterwards, so a config that
        carries training-time rates still serves inference."""
        rates = [e.ucg_rate for e in self.embedders]
        for e in self.embedders:
            e.ucg_rate = 0.0
        try:
            c = self(batch_c)
            uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings or [])
        finally:
            for e, r in zip(self.embedders, rates):
                e.ucg_rate = r
        return c, uc


class GeneralConditionerWithControl(GeneralConditioner):
    """modules.py:193-243: as GeneralConditioner, `control_vector` inputs keep their own key, and `control` is passed through."""

    def _key_of(self, embedder, emb):
        return "control_vector" if "control_vector" in embedder.input_key else self.OUTPUT_DIM2KEYS[emb.dim()]

    def forward(self, batch, force_zero_embeddings=None):
        output = super().forward(batch, force_zero_embeddings)
        output["control"] = batch["control"]
        return output
