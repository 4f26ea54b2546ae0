"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by supir_amd/).

fp32 PyTorch restatement of the text conditioner that feeds the hot path:
  GeneralConditionerWithControl.forward       sgm/modules/encoders/modules.py:193-243
  FrozenCLIPEmbedder (layer='hidden', idx 11) sgm/modules/encoders/modules.py:445-510   -> transformers.CLIPTextModel
  FrozenOpenCLIPEmbedder2 (bigG, penultimate + pooled, legacy=False)  :513-609           -> open_clip text transformer
  ConcatTimestepEmbedderND                    :1027-1043
on a flat {reference state-dict key: tensor} dict, tokens given (the BPE tokenisers need vocabulary files that are not in this
image).

Pinning.  tests/golden/golden_cond.pt holds outputs of the REFERENCE'S OWN GeneralConditionerWithControl / FrozenCLIPEmbedder /
FrozenOpenCLIPEmbedder2 / ConcatTimestepEmbedderND (instantiated from the embedder list of options/SUPIR_v0.yaml and run on CPU by
oracle/gen_golden_cond.py: c, uc, uc with force_uc_zero_embeddings, the legacy branch); tests/test_conditioner.py holds this file to
them (<= 2e-5).  That pins the reference's routing, concatenation order, pooling at the argmax token, layer selection, permutes,
force-zero and legacy handling.  The arithmetic of both towers lives in third-party packages:
  * transformers (CLIPTextModel; reference pin `transformers==4.28.1`, requirements.txt): INSTALLED here -- tests/test_conditioner.py
    checks `clip_l_hidden` against a real CLIPTextModel built from the ViT-L/14 text config with the same weights: pinned.
  * open_clip_torch (`open-clip-torch==2.17.1`, requirements.txt): NOT installed.  Its text tower is restated from the published
    architecture (pre-LN residual blocks around torch.nn.MultiheadAttention with an additive causal mask, erf-GELU MLP, ln_final,
    text_projection at the eot token) and from the reference's own call sites (:560-603).  The block restatement is checked against
    torch.nn.MultiheadAttention itself (what open_clip's ResidualAttentionBlock wraps); the assembly order comes from the reference
    code, and the whole tower -- at ViT-bigG-14's width and head geometry -- against transformers.CLIPTextModelWithProjection, a second
    implementation of the same architecture (the class SDXL's second text encoder, this checkpoint converted, is loaded into):
    tests/test_conditioner.py::test_bigg_*.  open_clip proper never runs here: "parity unpinned" for that package, stated in DESIGN.md.
"""
import math

import torch
import torch.nn.functional as F


def causal_mask(n, device):
    return torch.full((n, n), float("-inf"), device=device).triu_(1)


def _attn(x, wq, bq, wk, bk, wv, bv, wo, bo, heads):
    b, n, d = x.shape
    q, k, v = F.linear(x, wq, bq), F.linear(x, wk, bk), F.linear(x, wv, bv)
    q, k, v = (t.reshape(b, n, heads, d // heads).permute(0, 2, 1, 3) for t in (q, k, v))
    s = q @ k.transpose(-1, -2) / math.sqrt(d // heads) + causal_mask(n, x.device)
    o = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(b, n, d)
    return F.linear(o, wo, bo)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def clip_l_hidden(sd, tokens, p="transformer.text_model.", layer_idx=11, heads=12):
    """FrozenCLIPEmbedder(layer='hidden', layer_idx=11): hidden_states[11] of transformers.CLIPTextModel = the residual stream after
    11 encoder layers (hidden_states[0] is the embedding output; final_layer_norm is NOT applied to hidden_states)."""
    x = sd[p + "embeddings.token_embedding.weight"][tokens] + sd[p + "embeddings.position_embedding.weight"][None, :tokens.shape[1]]
    for i in range(layer_idx):
        q = f"{p}encoder.layers.{i}."
        h = F.layer_norm(x, x.shape[-1:], sd[q + "layer_norm1.weight"], sd[q + "layer_norm1.bias"], 1e-5)
        x = x + _attn(h, sd[q + "self_attn.q_proj.weight"], sd[q + "self_attn.q_proj.bias"], sd[q + "self_attn.k_proj.weight"],
                      sd[q + "self_attn.k_proj.bias"], sd[q + "self_attn.v_proj.weight"], sd[q + "self_attn.v_proj.bias"],
                      sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"], heads)
        h = F.layer_norm(x, x.shape[-1:], sd[q + "layer_norm2.weight"], sd[q + "layer_norm2.bias"], 1e-5)
        x = x + F.linear(quick_gelu(F.linear(h, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"])), sd[q + "mlp.fc2.weight"],
                         sd[q + "mlp.fc2.bias"])
    return x


def openclip_block(sd, q, x, heads):
    """open_clip ResidualAttentionBlock: x + attn(ln_1(x)); x + mlp(ln_2(x)); attn = nn.MultiheadAttention (packed in_proj)."""
    d = x.shape[-1]
    w, b = sd[q + "attn.in_proj_weight"], sd[q + "attn.in_proj_bias"]
    h = F.layer_norm(x, (d,), sd[q + "ln_1.weight"], sd[q + "ln_1.bias"], 1e-5)
    x = x + _attn(h, w[:d], b[:d], w[d:2 * d], b[d:2 * d], w[2 * d:], b[2 * d:], sd[q + "attn.out_proj.weight"],
                  sd[q + "attn.out_proj.bias"], heads)
    h = F.layer_norm(x, (d,), sd[q + "ln_2.weight"], sd[q + "ln_2.bias"], 1e-5)
    return x + F.linear(F.gelu(F.linear(h, sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"])), sd[q + "mlp.c_proj.weight"],
                        sd[q + "mlp.c_proj.bias"])


def openclip_g_penultimate_pooled(sd, tokens, p="model.", heads=20):
    """FrozenOpenCLIPEmbedder2(layer='penultimate', always_return_pooled=True, legacy=False), modules.py:560-603:
    penultimate = residual stream entering the LAST block (no ln_final); pooled = ln_final(last)[eot] @ text_projection, eot =
    argmax of the token ids (:581-586)."""
    x = sd[p + "token_embedding.weight"][tokens] + sd[p + "positional_embedding"]
    pre = p + "transformer.resblocks."
    n_layers = 1 + max(int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre))
    pen = None
    for i in range(n_layers):
        if i == n_layers - 1:
            pen = x
        x = openclip_block(sd, f"{p}transformer.resblocks.{i}.", x, heads)
    o = F.layer_norm(x, x.shape[-1:], sd[p + "ln_final.weight"], sd[p + "ln_final.bias"], 1e-5)
    pooled = o[torch.arange(o.shape[0]), tokens.argmax(dim=-1)] @ sd[p + "text_projection"]
    return pen, pooled


def openclip_g_legacy(sd, tokens, layer, p="model.", heads=20):
    """FrozenOpenCLIPEmbedder2(legacy=True) (modules.py:565-568, 592-601): ln_final applied to the chosen layer's residual stream
    ('last' = after all blocks, 'penultimate' = entering the last block); no pooled output."""
    x = sd[p + "token_embedding.weight"][tokens] + sd[p + "positional_embedding"]
    pre = p + "transformer.resblocks."
    n_layers = 1 + max(int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre))
    for i in range(n_layers):
        if i == n_layers - 1 and layer == "penultimate":
            break
        x = openclip_block(sd, f"{p}transformer.resblocks.{i}.", x, heads)
    return F.layer_norm(x, x.shape[-1:], sd[p + "ln_final.weight"], sd[p + "ln_final.bias"], 1e-5)


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def concat_timestep_embedder_nd(x, outdim=256):
    """ConcatTimestepEmbedderND (modules.py:1027-1043): each coordinate embedded separately, concatenated per sample."""
    if x.ndim == 1:
        x = x[:, None]
    b, dims = x.shape
    return timestep_embedding(x.reshape(-1), outdim).reshape(b, dims * outdim)


def general_conditioner_with_control(sd, batch, tokens_l, tokens_g, force_zero_embeddings=(), heads_l=12, heads_g=20):
    """GeneralConditionerWithControl.forward for options/SUPIR_v0.yaml:66-106 (embedders 0..4): crossattn = CLIP-L hidden[11] ||
    bigG penultimate (dim 2), vector = bigG pooled || 3 x ND(2 x 256) (dim 1), control passed through (:242).  An input key listed in
    force_zero_embeddings has EVERY output of EVERY embedder reading that key zeroed before concatenation (:229-233): 'txt' zeroes
    both text towers (crossattn entirely, the pooled part of vector) and leaves the size embeddings."""
    z_l = clip_l_hidden(sd, tokens_l, p="embedders.0.transformer.text_model.", heads=heads_l)
    pen, pooled = openclip_g_penultimate_pooled(sd, tokens_g, p="embedders.1.model.", heads=heads_g)
    if "txt" in force_zero_embeddings:
        z_l, pen, pooled = torch.zeros_like(z_l), torch.zeros_like(pen), torch.zeros_like(pooled)
    nd_keys = ("original_size_as_tuple", "crop_coords_top_left", "target_size_as_tuple")
    vec = [pooled] + [torch.zeros_like(e) if k in force_zero_embeddings else e for k, e in
                      ((k, concat_timestep_embedder_nd(batch[k])) for k in nd_keys)]
    return {"crossattn": torch.cat([z_l, pen], 2), "vector": torch.cat(vec, 1), "control": batch["control"]}
