"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Fixtures for the text conditioner from the REFERENCE'S OWN classes (/root/reference):

    GeneralConditionerWithControl    sgm/modules/encoders/modules.py:193-243 (forward; get_unconditional_conditioning :167-190)
    FrozenCLIPEmbedder               :445-510   (layer='hidden', layer_idx=11 -- options/SUPIR_v0.yaml:70-76)
    FrozenOpenCLIPEmbedder2          :513-609   (encode_with_transformer, text_transformer_forward, pool, legacy on / off)
    ConcatTimestepEmbedderND         :1027-1043

The classes are instantiated through the reference's own instantiate_from_config from the embedder list of options/SUPIR_v0.yaml
and run on CPU.  What they delegate to third-party packages is supplied as follows:
  * transformers.CLIPTextModel -- installed: the REAL class, built from a CLIPTextConfig instead of `from_pretrained` (no checkpoint
    files here); the tokeniser (vocabulary files absent) is replaced by a table text -> token ids;
  * open_clip (open-clip-torch==2.17.1) -- not installed: `create_model_and_transforms` returns a stand-in text tower with
    open_clip's attribute and state-dict names whose ResidualAttentionBlock is torch.nn.MultiheadAttention + LayerNorm + GELU MLP
    (what open_clip's block wraps).  So the reference's routing / pooling / permutes / legacy / ucg / force-zero code is pinned by
    these fixtures; open_clip's own block arithmetic remains a restatement (stated in DESIGN.md).
Widths are reduced (CLIP-L 192 x 12 layers x 12 heads, "bigG" 320 x 6 layers x 20 heads): the code paths are width-independent and
the fixture stays small.  Weights = supir_amd.synth by reference key name.

    python -m oracle.gen_golden_cond      # seconds; writes tests/golden/golden_cond.pt
"""
import collections
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import as R  # noqa: E402
from supir_amd.synth import synth_param  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "golden_cond.pt")
L_WIDTH, L_LAYERS, L_HEADS = 192, 12, 12
G_WIDTH, G_LAYERS, G_HEADS, G_PROJ = 320, 6, 20, 320
VOCAB = 49408


def tokens_for(n, seed, eot_pos):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(1000, 40000, (n, 77), generator=g)
    t[:, 0] = 49406
    for i in range(n):
        e = eot_pos[i % len(eot_pos)]
        t[i, e] = 49407
        t[i, e + 1:] = 0
    return t


TEXTS = {"a photo of a cat": tokens_for(1, 1, (9,))[0], "cinematic, high detail, 8k": tokens_for(1, 2, (30,))[0],
         "": tokens_for(1, 3, (1,))[0], "blurry, low quality": tokens_for(1, 4, (76,))[0]}


def _lookup(texts):
    return torch.stack([TEXTS[t] for t in texts])


# ------------------------------------------------------------------------------------------ stand-ins for the absent third parties
class _ResBlock(nn.Module):
    """open_clip.transformer.ResidualAttentionBlock (no layer scale): x + attn(ln_1 x); x + mlp(ln_2 x); sequence-first."""

    def __init__(self, d, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(collections.OrderedDict([("c_fc", nn.Linear(d, 4 * d)), ("gelu", nn.GELU()), ("c_proj", nn.Linear(4 * d, d))]))

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class _Transformer(nn.Module):
    def __init__(self, d, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(d, heads) for _ in range(layers)])
        self.grad_checkpointing = False


class _FakeOpenClipModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.visual = nn.Identity()     # the reference deletes it (modules.py:538)
        self.token_embedding = nn.Embedding(VOCAB, G_WIDTH)
        self.positional_embedding = nn.Parameter(torch.empty(77, G_WIDTH))
        self.transformer = _Transformer(G_WIDTH, G_LAYERS, G_HEADS)
        self.ln_final = nn.LayerNorm(G_WIDTH)
        self.text_projection = nn.Parameter(torch.empty(G_WIDTH, G_PROJ))
        self.register_buffer("attn_mask", torch.full((77, 77), float("-inf")).triu_(1), persistent=False)


def install_third_party_stand_ins():
    """After ref_import.load_reference(): give the reference's encoders module a working open_clip / CLIP loader."""
    import sgm.modules.encoders.modules as M
    from transformers import CLIPTextConfig, CLIPTextModel

    oc = sys.modules["open_clip"]
    oc.create_model_and_transforms = lambda arch, device=None, pretrained=None: (_FakeOpenClipModel(), None, None)
    oc.tokenize = _lookup
    M.open_clip = oc

    class Tok:
        @classmethod
        def from_pretrained(cls, *a, **k):
            return cls()

        def __call__(self, text, **kw):
            return {"input_ids": _lookup(text)}

    class HF:
        @classmethod
        def from_pretrained(cls, *a, **k):
            cfg = CLIPTextConfig(vocab_size=VOCAB, hidden_size=L_WIDTH, intermediate_size=4 * L_WIDTH, num_hidden_layers=L_LAYERS,
                                 num_attention_heads=L_HEADS, max_position_embeddings=77, hidden_act="quick_gelu")
            return CLIPTextModel(cfg)

    M.CLIPTokenizer, M.CLIPTextModel = Tok, HF
    return M


def canonical_key(k):
    """Reference state-dict key of a conditioner parameter (transformers >= 5 dropped the `text_model.` level inside CLIPTextModel)."""
    p = "embedders.0.transformer."
    if k.startswith(p) and not k.startswith(p + "text_model."):
        k = p + "text_model." + k[len(p):]
    return k


def fill(module):
    sd = {}
    with torch.no_grad():
        for k, t in module.state_dict().items():
            ck = canonical_key(k)
            if t.is_floating_point():
                t.copy_(synth_param("conditioner." + ck, t.shape))
            sd[ck] = t.detach().clone()
    return sd


def embedder_configs(legacy_g=False, layer_g="penultimate"):
    P = "sgm.modules.encoders.modules."
    g = {"arch": "ViT-bigG-14", "version": "laion2b_s39b_b160k", "freeze": True, "layer": layer_g, "device": "cpu",
         "always_return_pooled": not legacy_g, "legacy": legacy_g}
    nd = lambda key: {"is_trainable": False, "input_key": key, "target": P + "ConcatTimestepEmbedderND", "params": {"outdim": 256}}  # noqa: E731
    return [{"is_trainable": False, "input_key": "txt", "target": P + "FrozenCLIPEmbedder",
             "params": {"layer": "hidden", "layer_idx": 11, "device": "cpu"}},
            {"is_trainable": False, "input_key": "txt", "target": P + "FrozenOpenCLIPEmbedder2", "params": g},
            nd("original_size_as_tuple"), nd("crop_coords_top_left"), nd("target_size_as_tuple")]


def batches():
    c = {"txt": ["a photo of a cat", "cinematic, high detail, 8k"],
         "original_size_as_tuple": torch.tensor([[1024, 1024], [768, 512]]), "crop_coords_top_left": torch.tensor([[0, 0], [16, 32]]),
         "target_size_as_tuple": torch.tensor([[1024, 1024], [1536, 1024]]),
         "control": torch.arange(2 * 4 * 4 * 4, dtype=torch.float32).reshape(2, 4, 4, 4)}
    uc = dict(c, txt=["", "blurry, low quality"])
    return c, uc


def run_reference():
    ns = R.load_reference()
    import warnings
    warnings.filterwarnings("ignore")
    M = install_third_party_stand_ins()
    gold = {"texts": {k: v.clone() for k, v in TEXTS.items()}}
    with R.quiet():
        cond = M.GeneralConditionerWithControl(ns.OmegaConf.create(embedder_configs()))
    sd = fill(cond)
    c_b, uc_b = batches()
    with torch.no_grad():
        c, uc = cond.get_unconditional_conditioning(dict(c_b), dict(uc_b))                       # SUPIR_model.py:166
        _, uc0 = cond.get_unconditional_conditioning(dict(c_b), dict(uc_b), force_uc_zero_embeddings=["txt"])
    for name, d in (("c", c), ("uc", uc), ("uc_force_zero_txt", uc0)):
        for k, v in d.items():
            gold[f"{name}.{k}"] = v.clone()
    # the legacy branch of FrozenOpenCLIPEmbedder2 (modules.py:565-568: ln_final of the chosen layer, no pooled output)
    for layer in ("last", "penultimate"):
        with R.quiet():
            g = M.FrozenOpenCLIPEmbedder2(arch="ViT-bigG-14", device="cpu", layer=layer, legacy=True)
        g.load_state_dict({k[len("embedders.1."):]: v for k, v in sd.items() if k.startswith("embedders.1.")})
        with torch.no_grad():
            gold[f"g_legacy_{layer}"] = g(c_b["txt"]).clone()
    return gold, sd


def main():
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT
    gold, _ = run_reference()
    torch.save(gold, out)
    print("wrote", out, os.path.getsize(out), "bytes;", {k: tuple(v.shape) for k, v in gold.items() if torch.is_tensor(v)})


if __name__ == "__main__":
    main()
