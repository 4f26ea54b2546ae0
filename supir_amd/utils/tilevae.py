"""Tiled VAE with cross-tile GroupNorm, natively on the GPU.

Same interface and semantics as SUPIR/utils/tilevae.py (VAEHook :677-970, split_tiles :717-774, GroupNormParam :599-648,
custom_group_norm :524-553, crop_valid_region :556-567, build_task_queue :472-499), as SUPIRModel.init_tile_vae installs it
(SUPIR/models/SUPIR_model.py:138-150: fast_encoder = fast_decoder = False, color_fix = False):

  * the input is split into tiles padded by 32 px (encoder) / 11 latent px (decoder);
  * every tile runs the whole encoder / decoder; at each GroupNorm the per-tile statistics are pooled over all tiles
    -- pixel-weighted mean of the per-tile means and of the per-tile (biased) variances, exactly the reference's
    `summary()` (which is NOT the exact pooled variance) -- and every tile is normalised with the pooled values;
  * the mid-block attention is tile-local (the reference always takes its xformers branch);
  * each tile's valid region is cropped and pasted into the result.

What differs is the execution plan, not the math: the reference walks a per-tile task queue and ping-pongs tiles
between CPU and GPU at every GroupNorm ("limited by both the GPU and the CPU", tilevae.py:25-27); here all tiles stay
resident in HBM (a 4096^2 decode keeps ~8 GB per layer-wide tensor live, 288 GB available) and the network is executed
layer-major over the tiles, so a GroupNorm is two kernel launches per tile plus 64-float pooling arithmetic.
"""
import math

import torch

from .. import ops
from ..modules.base import cdt, to_nchw, to_nhwc


def get_best_tile_size(lowerbound, upperbound):
    divider = 32
    while divider >= 2:
        remainer = lowerbound % divider
        if remainer == 0:
            return lowerbound
        candidate = lowerbound - remainer + divider
        if candidate <= upperbound:
            return candidate
        divider //= 2
    return lowerbound


def split_tiles(h, w, tile_size, pad, is_decoder):
    """tilevae.py:717-774: returns (input bboxes, output bboxes), bbox = [x1, x2, y1, y2]."""
    nh = max(math.ceil((h - 2 * pad) / tile_size), 1)
    nw = max(math.ceil((w - 2 * pad) / tile_size), 1)
    rth = get_best_tile_size(math.ceil((h - 2 * pad) / nh), tile_size)
    rtw = get_best_tile_size(math.ceil((w - 2 * pad) / nw), tile_size)
    in_b, out_b = [], []
    for i in range(nh):
        for j in range(nw):
            ib = [pad + j * rtw, min(pad + (j + 1) * rtw, w), pad + i * rth, min(pad + (i + 1) * rth, h)]
            ob = [ib[0] if ib[0] > pad else 0, ib[1] if ib[1] < w - pad else w,
                  ib[2] if ib[2] > pad else 0, ib[3] if ib[3] < h - pad else h]
            out_b.append([x * 8 if is_decoder else x // 8 for x in ob])
            in_b.append([max(0, ib[0] - pad), min(w, ib[1] + pad), max(0, ib[2] - pad), min(h, ib[3] + pad)])
    return in_b, out_b


def crop_valid_region(x, input_bbox, target_bbox, is_decoder):
    padded = [i * 8 if is_decoder else i // 8 for i in input_bbox]
    m = [target_bbox[i] - padded[i] for i in range(4)]
    return x[:, :, m[2]:x.size(2) + m[3], m[0]:x.size(3) + m[1]]


def pooled_groupnorm(tiles, norm, silu):
    """GroupNorm of every tile ([B,h,w,C] bf16) with statistics pooled over all tiles (GroupNormParam.summary)."""
    C = tiles[0].shape[-1]
    cpg = C // 32
    pix = [t.shape[1] * t.shape[2] for t in tiles]
    tot = float(sum(pix))
    mean = var = None
    for t, p in zip(tiles, pix):
        s = ops.groupnorm_stats(t).double()
        n = float(p * cpg)
        m_i = s[..., 0] / n
        v_i = (s[..., 1] / n - m_i * m_i).clamp_min(0.0)      # biased variance (torch.var_mean(unbiased=False))
        w = p / tot
        mean = m_i * w if mean is None else mean + m_i * w
        var = v_i * w if var is None else var + v_i * w
    given = torch.stack([mean, var], dim=-1).float().contiguous()
    return [ops.groupnorm(t, norm.g32(), norm.b32(), norm.eps, silu=silu, given=given) for t in tiles]


def _resblock(tiles, blk):
    """ResnetBlock over all tiles (resblock2task, tilevae.py:374-402): shortcut, pooled GN+SiLU, conv1, pooled GN+SiLU,
    conv2 + residual."""
    if blk.in_channels != blk.out_channels:
        res = [ops.gemm(t, blk.nin_shortcut.w(), blk.nin_shortcut.b32()) for t in tiles]
    else:
        res = tiles
    h = pooled_groupnorm(tiles, blk.norm1, True)
    h = [ops.conv3x3(t, blk.conv1.w(), blk.conv1.b32()) for t in h]
    h = pooled_groupnorm(h, blk.norm2, True)
    return [ops.conv3x3(t, blk.conv2.w(), blk.conv2.b32(), residual=r) for t, r in zip(h, res)]


def _attn(tiles, att):
    """attn2task (tilevae.py:349-372): residual + proj_out(tile-local attention(pooled GN(tile)))."""
    n = pooled_groupnorm(tiles, att.norm, False)
    out = []
    for t, nt in zip(tiles, n):
        B, H, W, C = t.shape
        o = att.attend(nt.view(B, H * W, C))
        out.append(ops.gemm(o, att.proj_out.w(), att.proj_out.b32(), residual=t.view(B, H * W, C)).view(B, H, W, C))
    return out


class VAEHook:
    def __init__(self, net, tile_size, is_decoder, fast_decoder=False, fast_encoder=False, color_fix=False, to_gpu=False):
        if fast_decoder or fast_encoder or color_fix:
            raise NotImplementedError("SUPIR installs the hook with fast modes and color_fix off (SUPIR_model.py:142-150)")
        self.net = net
        self.tile_size = tile_size
        self.is_decoder = is_decoder
        self.pad = 11 if is_decoder else 32

    def __call__(self, x):
        H, W = x.shape[2], x.shape[3]
        if max(H, W) <= self.pad * 2 + self.tile_size:
            return self.net.original_forward(x)        # "the input size is tiny and unnecessary to tile"
        return self.vae_tile_forward(x)

    @torch.no_grad()
    def vae_tile_forward(self, z):
        net, dec = self.net, self.is_decoder
        N, height, width = z.shape[0], z.shape[2], z.shape[3]
        in_b, out_b = split_tiles(height, width, self.tile_size, self.pad, dec)
        z = z.float()
        tiles = [ops.conv3x3_smallcin(z[:, :, b[2]:b[3], b[0]:b[1]].contiguous(), net.conv_in.wf32(), net.conv_in.b32(), dtype=cdt())
                 for b in in_b]
        if dec:
            tiles = _resblock(tiles, net.mid.block_1)
            tiles = _attn(tiles, net.mid.attn_1)
            tiles = _resblock(tiles, net.mid.block_2)
            for lvl in reversed(range(net.num_resolutions)):
                for blk in net.up[lvl].block:
                    tiles = _resblock(tiles, blk)
                if lvl != 0:
                    up = net.up[lvl].upsample
                    tiles = [ops.conv3x3(t, up.conv.w(), up.conv.b32(), upsample=True) for t in tiles]
        else:
            for lvl in range(net.num_resolutions):
                for blk in net.down[lvl].block:
                    tiles = _resblock(tiles, blk)
                if lvl != net.num_resolutions - 1:
                    dn = net.down[lvl].downsample
                    tiles = [ops.conv3x3(t, dn.conv.w(), dn.conv.b32(), stride=2, pad=(0, 0),
                                         out_hw=(t.shape[1] // 2, t.shape[2] // 2)) for t in tiles]
            tiles = _resblock(tiles, net.mid.block_1)
            tiles = _attn(tiles, net.mid.attn_1)
            tiles = _resblock(tiles, net.mid.block_2)
        tiles = pooled_groupnorm(tiles, net.norm_out, True)
        outs = [ops.conv3x3_smallcout(t, net.conv_out.w9(), net.conv_out.b32()) for t in tiles]
        oh, ow = (height * 8, width * 8) if dec else (height // 8, width // 8)
        result = torch.zeros(N, outs[0].shape[1], oh, ow, device=z.device, dtype=torch.float32)
        for o, ib, ob in zip(outs, in_b, out_b):
            result[:, :, ob[2]:ob[3], ob[0]:ob[1]] = crop_valid_region(o, ib, ob, dec)
        return result
