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


# ------------------------------------------------------------------------------------------------ execution plan
# Tiles of the same shape are STACKED along the batch dimension (a 4096^2 decode has 64 tiles in at most 9 shapes: interior, four
# edge kinds, corners), so a layer is one launch per shape group instead of one per tile, and the per-tile GroupNorm statistics of a
# group come out of ONE statistics launch ([tiles * N, 32, 2]).  A "stack" below is a tensor [T_g * N, h, w, C] holding the T_g tiles
# of one shape group, tile-major.
#
# Tile-parallel across ranks ("Tiles"): each rank runs a subset of the tiles; the only exchange inside the network is
# the pooled statistic of every GroupNorm -- per image and group the pixel-weighted sums  sum_i w_i mean_i  and  sum_i w_i var_i  over
# ALL tiles are additive over ranks: one all-reduce of [N, 32, 2] doubles per GroupNorm (about 60 per pass, a few hundred bytes
# each) -- and the result canvas, assembled by one all-reduce of disjoint valid regions.
def assign_tiles(n_tiles, rank, world):
    """Tile i -> rank i % world (round robin: neighbouring tiles, which share a shape group, spread over the ranks)."""
    return [i for i in range(n_tiles) if i % world == rank]


def pool_statistics(sums, pix, tot_pix, cpg, n_images, group=None):
    """GroupNormParam.summary (tilevae.py:610-640) over this rank's tiles, completed across ranks.
    sums: list over shape groups of fp32 [T_g * N, 32, 2] (sum, sum of squares) per tile and image; pix: pixels per tile of each
    group; tot_pix: pixels of ALL tiles of all ranks.  Returns fp32 [N, 32, 2] = (pooled mean, pooled biased variance): the
    pixel-weighted mean of the per-tile means and of the per-tile variances (NOT the exact pooled variance -- the reference's)."""
    acc = None
    for s, p in zip(sums, pix):
        s = s.double().view(-1, n_images, 32, 2)
        n = float(p * cpg)
        m_i = s[..., 0] / n
        v_i = (s[..., 1] / n - m_i * m_i).clamp_min(0.0)      # biased variance (torch.var_mean(unbiased=False))
        part = torch.stack([m_i, v_i], dim=-1).sum(dim=0) * (p / float(tot_pix))   # [N, 32, 2]
        acc = part if acc is None else acc + part
    if group is not None:
        import torch.distributed as dist
        if acc is None:
            raise RuntimeError("a rank without tiles cannot take part in the pooled statistics (fewer tiles than ranks)")
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc.float().contiguous()


class _Tiles:
    """The tiles this rank runs, as shape-group stacks, plus what pooling needs to know about everybody's tiles: `all_hw` = the
    current (h, w) of EVERY tile of every rank (the pooling weights are pixel counts at the current resolution, tilevae.py:618-631)."""

    def __init__(self, stacks, all_hw, n_images, group):
        self.stacks, self.all_hw, self.n, self.group = stacks, all_hw, n_images, group

    @property
    def pix(self):
        return [t.shape[1] * t.shape[2] for t in self.stacks]

    @property
    def tot_pix(self):
        return sum(h * w for h, w in self.all_hw)

    def like(self, stacks):
        return _Tiles(stacks, self.all_hw, self.n, self.group)

    def map(self, fn, hw_fn=None):
        """Apply a layer to every stack; hw_fn maps a tile's (h, w) when the layer changes the resolution."""
        return _Tiles([fn(t) for t in self.stacks], self.all_hw if hw_fn is None else [hw_fn(h, w) for h, w in self.all_hw], self.n,
                      self.group)


def pooled_groupnorm(tiles, norm, silu):
    """GroupNorm of every tile with statistics pooled over all tiles of all ranks (custom_group_norm, tilevae.py:524-553)."""
    C = tiles.stacks[0].shape[-1]
    given = pool_statistics([ops.groupnorm_stats(t) for t in tiles.stacks], tiles.pix, tiles.tot_pix, C // 32, tiles.n, tiles.group)
    out = []
    for t in tiles.stacks:
        g = given.repeat(t.shape[0] // tiles.n, 1, 1) if t.shape[0] != tiles.n else given      # tile-major stack: every tile the same
        out.append(ops.groupnorm(t, norm.g32(), norm.b32(), norm.eps, silu=silu, given=g.contiguous()))
    return tiles.like(out)


def _resblock(tiles, blk):
    """ResnetBlock over all tiles (resblock2task, tilevae.py:374-402): shortcut, pooled GN+SiLU, conv1, pooled GN+SiLU,
    conv2 + residual."""
    if blk.in_channels != blk.out_channels:
        res = [ops.gemm(t, blk.nin_shortcut.w(), blk.nin_shortcut.b32()) for t in tiles.stacks]
    else:
        res = tiles.stacks
    h = pooled_groupnorm(tiles, blk.norm1, True)
    h = h.map(lambda t: ops.conv3x3(t, blk.conv1.w(), blk.conv1.b32()))
    h = pooled_groupnorm(h, blk.norm2, True)
    return tiles.like([ops.conv3x3(t, blk.conv2.w(), blk.conv2.b32(), residual=r) for t, r in zip(h.stacks, res)])


def _attn(tiles, att):
    """attn2task (tilevae.py:349-372): residual + proj_out(tile-local attention(pooled GN(tile)))."""
    n = pooled_groupnorm(tiles, att.norm, False)
    out = []
    for t, nt in zip(tiles.stacks, n.stacks):
        B, H, W, C = t.shape
        o = att.attend(nt.view(B, H * W, C))
        out.append(ops.gemm(o, att.proj_out.w(), att.proj_out.b32(), residual=t.view(B, H * W, C)).view(B, H, W, C))
    return tiles.like(out)


class VAEHook:
    def __init__(self, net, tile_size, is_decoder, fast_decoder=False, fast_encoder=False, color_fix=False, to_gpu=False,
                 tile_parallel=False, process_group=None):
        """tile_parallel=True (not in the reference, which has no distributed code): the tiles are dealt round robin to the ranks of
        `process_group` (default: the world); every rank returns the complete result."""
        if fast_decoder or fast_encoder or color_fix:
            raise NotImplementedError("SUPIR installs the hook with fast modes and color_fix off (SUPIR_model.py:142-150)")
        self.net = net
        self.tile_size = tile_size
        self.is_decoder = is_decoder
        self.pad = 11 if is_decoder else 32
        self.tile_parallel = tile_parallel
        self.process_group = process_group

    def __call__(self, x):
        H, W = x.shape[2], x.shape[3]
        if max(H, W) <= self.pad * 2 + self.tile_size:
            return self.net.original_forward(x)        # "the input size is tiny and unnecessary to tile"
        return self.vae_tile_forward(x)

    def _ranks(self):
        if not self.tile_parallel:
            return 0, 1, None
        import torch.distributed as dist
        from ..parallel import collectives_active
        if not collectives_active(self.process_group):
            return 0, 1, None
        return dist.get_rank(self.process_group), dist.get_world_size(self.process_group), (self.process_group or dist.group.WORLD)

    @torch.no_grad()
    def vae_tile_forward(self, z):
        net, dec = self.net, self.is_decoder
        N, height, width = z.shape[0], z.shape[2], z.shape[3]
        in_b, out_b = split_tiles(height, width, self.tile_size, self.pad, dec)
        rank, world, group = self._ranks()
        mine = assign_tiles(len(in_b), rank, world)
        if not mine:
            raise RuntimeError(f"tile-parallel tiled VAE: {len(in_b)} tiles for {world} ranks leaves rank {rank} without work; use a "
                               "process group no larger than the tile count")
        z = z.float()
        # shape groups of this rank's tiles, in order of first appearance
        groups = {}
        for i in mine:
            b = in_b[i]
            groups.setdefault((b[3] - b[2], b[1] - b[0]), []).append(i)
        order = [i for idx in groups.values() for i in idx]
        stacks = [ops.conv3x3_smallcin(torch.cat([z[:, :, in_b[i][2]:in_b[i][3], in_b[i][0]:in_b[i][1]] for i in idx], 0).contiguous(),
                                       net.conv_in.wf32(), net.conv_in.b32(), dtype=cdt()) for idx in groups.values()]
        tiles = _Tiles(stacks, [(b[3] - b[2], b[1] - b[0]) for b in in_b], N, group)
        if dec:
            tiles = _resblock(tiles, net.mid.block_1)
            tiles = _attn(tiles, net.mid.attn_1)
            tiles = _resblock(tiles, net.mid.block_2)
            for lvl in reversed(range(net.num_resolutions)):
                for blk in net.up[lvl].block:
                    tiles = _resblock(tiles, blk)
                if lvl != 0:
                    up = net.up[lvl].upsample
                    tiles = tiles.map(lambda t: ops.conv3x3(t, up.conv.w(), up.conv.b32(), upsample=True), lambda h, w: (2 * h, 2 * w))
        else:
            for lvl in range(net.num_resolutions):
                for blk in net.down[lvl].block:
                    tiles = _resblock(tiles, blk)
                if lvl != net.num_resolutions - 1:
                    dn = net.down[lvl].downsample
                    tiles = tiles.map(lambda t: ops.conv3x3(t, dn.conv.w(), dn.conv.b32(), stride=2, pad=(0, 0),
                                                            out_hw=(t.shape[1] // 2, t.shape[2] // 2)), lambda h, w: (h // 2, w // 2))
            tiles = _resblock(tiles, net.mid.block_1)
            tiles = _attn(tiles, net.mid.attn_1)
            tiles = _resblock(tiles, net.mid.block_2)
        tiles = pooled_groupnorm(tiles, net.norm_out, True)
        outs = [ops.conv3x3_smallcout(t, net.conv_out.w9(), net.conv_out.b32()) for t in tiles.stacks]
        oh, ow = (height * 8, width * 8) if dec else (height // 8, width // 8)
        result = torch.zeros(N, outs[0].shape[1], oh, ow, device=z.device, dtype=torch.float32)
        k = 0
        for o, idx in zip(outs, groups.values()):
            for j, i in enumerate(idx):
                result[:, :, out_b[i][2]:out_b[i][3], out_b[i][0]:out_b[i][1]] = crop_valid_region(o[j * N:(j + 1) * N], in_b[i], out_b[i], dec)
                k += 1
        assert k == len(order)
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(result, op=dist.ReduceOp.SUM, group=group)     # valid regions are disjoint: the sum IS the assembly
        return result
