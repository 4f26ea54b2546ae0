"""SDXL UNet building blocks on the HIP kernels; constructor kwargs / state-dict keys / forward signatures follow
sgm/modules/diffusionmodules/openaimodel.py (TimestepEmbedSequential :81-105, Upsample :108-151, Downsample :170-210,
ResBlock :213-356, UNetModel :506-1013) for the option subset SUPIR's configs use (options/SUPIR_v0*.yaml):
dims=2, conv_resample, no scale-shift norm, no resblock_updown, spatial transformers with linear projections.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .. import weights as Wt
from .attention import SpatialTransformer
from .base import cdt, Conv3x3, GroupNorm32, Linear, Passthrough, Prep, attach_gn_part, gn_part_of, to_nchw, to_nhwc


def timestep_embedding(timesteps, dim, max_period=10000):
    """cos || sin sinusoidal embedding (sgm/modules/diffusionmodules/util.py:206-230). Host-side scalar prep on [B]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class EmbBundle:
    """Time embedding handed down the block stack: raw emb [B,1280] plus, when the owning model pre-projected it, the
    per-ResBlock `emb_layers` outputs as column slices of ONE [B, sum(Cout)] GEMM (M=B rows: batching 17+ GEMVs)."""

    __slots__ = ("raw", "proj")

    def __init__(self, raw, proj=None):
        self.raw = raw
        self.proj = proj or {}


class TimestepBlock(nn.Module):
    pass


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, x, emb, context=None, *args, **kwargs):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    """nearest-2x + conv3x3 in ONE kernel: the upsample is folded into the implicit-GEMM gather
    (reference does interpolate in fp32 then conv, openaimodel.py:131-151)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_up=False):
        super().__init__()
        assert dims == 2 and use_conv and padding == 1
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = Conv3x3(channels, self.out_channels)

    def forward(self, x):
        out, po = ops.conv3x3(to_nhwc(x), self.conv.w(), self.conv.b32(), upsample=True, gn_part=True)
        return attach_gn_part(to_nchw(out), po)


class Downsample(nn.Module):
    """conv3x3 stride 2 pad 1 (openaimodel.py:196-210)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_down=False):
        super().__init__()
        assert dims == 2 and use_conv and padding == 1
        self.channels, self.out_channels = channels, out_channels or channels
        self.op = Conv3x3(channels, self.out_channels)

    def forward(self, x):
        out, po = ops.conv3x3(to_nhwc(x), self.op.w(), self.op.b32(), stride=2, pad=(1, 1), gn_part=True)
        return attach_gn_part(to_nchw(out), po)


class ResBlock(TimestepBlock):
    """GN+SiLU (1 fused pass) -> conv3x3 with the time-embedding add in its epilogue -> GN+SiLU -> conv3x3 with the skip
    add in its epilogue.  Sequential indices mirror the reference so keys match: in_layers.{0,2}, emb_layers.1,
    out_layers.{0,3}, skip_connection (openaimodel.py:260-321)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False, kernel_size=3, exchange_temb_dims=False,
                 skip_t_emb=False):
        super().__init__()
        assert dims == 2 and kernel_size == 3 and not (up or down or use_scale_shift_norm or use_conv or skip_t_emb)
        self.channels, self.emb_channels = channels, emb_channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(GroupNorm32(channels), Passthrough(), Conv3x3(channels, self.out_channels))
        self.emb_layers = nn.Sequential(Passthrough(), Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(self.out_channels), Passthrough(), Passthrough(),
                                        Conv3x3(self.out_channels, self.out_channels))
        if self.out_channels == channels:
            self.skip_connection = Passthrough()
        else:
            self.skip_connection = Linear(channels, self.out_channels, conv1x1=True)

    def forward(self, x, emb):
        xh = to_nhwc(x)
        B, H, W, C = xh.shape
        if isinstance(emb, EmbBundle):
            e = emb.proj.get(id(self))
            raw = emb.raw
        else:
            e, raw = None, emb
        if e is None:  # stand-alone use: project this block's embedding here
            el = self.emb_layers[1]
            e = ops.gemm(torch.nn.functional.silu(raw.float()).to(cdt()), el.w(), el.b32())
        n0, c0 = self.in_layers[0], self.in_layers[2]
        # GroupNorm statistics ride the producer's epilogue where there is one (ops.GnPart): the input's from whichever module made
        # it, conv1's output for out_layers, and this block's output for whoever normalises it next
        h = ops.groupnorm(xh, n0.g32(), n0.b32(), n0.eps, silu=True, part=gn_part_of(x))
        h, p1 = ops.conv3x3(h, c0.w(), c0.b32(), rowbias=e, gn_part=True)
        n1, c1 = self.out_layers[0], self.out_layers[3]
        h = ops.groupnorm(h, n1.g32(), n1.b32(), n1.eps, silu=True, out=h, part=p1)
        if isinstance(self.skip_connection, Linear):
            sk = ops.gemm(xh, self.skip_connection.w(), self.skip_connection.b32())
        else:
            sk = xh
        out, po = ops.conv3x3(h, c1.w(), c1.b32(), residual=sk, gn_part=True)
        return attach_gn_part(to_nchw(out), po)


def _as_list(v):
    return v if v is None or isinstance(v, int) else [int(i) for i in v]


class UNetModel(nn.Module):
    """Structure (and therefore state-dict keys) of the reference UNetModel for SUPIR's SDXL-base configuration."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, use_spatial_transformer=False,
                 transformer_depth=1, context_dim=None, n_embed=None, legacy=True, disable_self_attentions=None,
                 num_attention_blocks=None, disable_middle_self_attn=False, use_linear_in_transformer=False,
                 spatial_transformer_attn_type="softmax", adm_in_channels=None, use_fairscale_checkpoint=False,
                 offload_to_cpu=False, transformer_depth_middle=None, build_decoder=True, **ignored):
        super().__init__()
        assert dims == 2 and use_spatial_transformer and context_dim is not None and num_head_channels == 64
        assert not (resblock_updown or use_scale_shift_norm or legacy or n_embed is not None)
        assert num_classes == "sequential" and adm_in_channels is not None
        channel_mult = _as_list(channel_mult)
        attention_resolutions = _as_list(attention_resolutions)
        transformer_depth = _as_list(transformer_depth)
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        if transformer_depth_middle is None:
            transformer_depth_middle = transformer_depth[-1]
        if isinstance(num_res_blocks, int):
            num_res_blocks = len(channel_mult) * [num_res_blocks]
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.num_classes = list(num_res_blocks), num_classes
        self.predict_codebook_ids = False
        ted = model_channels * 4

        def st(ch, depth):
            return SpatialTransformer(ch, ch // num_head_channels, num_head_channels, depth=depth, context_dim=context_dim,
                                      use_linear=use_linear_in_transformer, attn_type=spatial_transformer_attn_type)

        self.time_embed = nn.Sequential(Linear(model_channels, ted), Passthrough(), Linear(ted, ted))
        self.label_emb = nn.Sequential(nn.Sequential(Linear(adm_in_channels, ted), Passthrough(), Linear(ted, ted)))
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(Conv3x3(in_channels, model_channels))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(st(ch, transformer_depth[level]))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, dropout), st(ch, transformer_depth_middle),
                                                    ResBlock(ch, ted, dropout))
        if build_decoder:
            self.output_blocks = nn.ModuleList([])
            for level, mult in list(enumerate(channel_mult))[::-1]:
                for i in range(self.num_res_blocks[level] + 1):
                    ich = chans.pop()
                    layers = [ResBlock(ch + ich, ted, dropout, out_channels=model_channels * mult)]
                    ch = model_channels * mult
                    if ds in attention_resolutions:
                        layers.append(st(ch, transformer_depth[level]))
                    if level and i == self.num_res_blocks[level]:
                        layers.append(Upsample(ch, conv_resample, out_channels=ch))
                        ds //= 2
                    self.output_blocks.append(TimestepEmbedSequential(*layers))
            self.out = nn.Sequential(GroupNorm32(ch), Passthrough(), Conv3x3(model_channels, out_channels))
        object.__setattr__(self, "_emb_w", Prep())
        object.__setattr__(self, "_label_cache", None)
        object.__setattr__(self, "_schedule", None)         # the embedding table the next announced call reads (prepare_schedule)
        object.__setattr__(self, "_schedules", {})          # every table prepared so far: (batch, element type, device) -> table
        object.__setattr__(self, "_sched_counter", [0])

    # ------------------------------------------------------------------ embeddings
    def _res_blocks(self):
        return [m for m in self.modules() if isinstance(m, ResBlock)]

    def _emb_project(self, emb):
        """ONE GEMM projects SiLU(emb) [rows, 1280] for every ResBlock (reference: per-block emb_layers Linear,
        openaimodel.py:287-293,343): returns (proj_all [rows, sum(Cout)], blocks, widths)."""
        blocks = self._res_blocks()
        srcs = [p for b in blocks for p in (b.emb_layers[1].weight, b.emb_layers[1].bias)]
        w_all, b_all, offs = self._emb_w.get(srcs, lambda: (
            torch.cat([Wt.linear_w(b.emb_layers[1].weight) for b in blocks], 0).contiguous(),
            torch.cat([Wt.f32(b.emb_layers[1].bias) for b in blocks], 0).contiguous(),
            [b.out_channels for b in blocks]))
        return ops.gemm(torch.nn.functional.silu(emb).to(cdt()), w_all, b_all), blocks, offs

    def _time_emb(self, timesteps):
        te = timestep_embedding(timesteps, self.model_channels).to(cdt())
        l0, l2 = self.time_embed[0], self.time_embed[2]
        return ops.gemm(ops.gemm(te, l0.w(), l0.b32(), act=1), l2.w(), l2.b32(), out_dtype=torch.float32)

    def _embed(self, timesteps, y):
        """emb = time_embed(t_emb) + label_emb(y), projected for every ResBlock.  With a prepared schedule (prepare_schedule: the
        sampler knows all of an image's timesteps in advance) the whole thing is ONE row gather out of a per-image table."""
        sch = self._schedule
        if sch is not None and sch["active"] and sch["B"] == y.shape[0] and sch["cdt"] == cdt():
            proj_all = sch["proj"].index_select(0, sch["row"])[0]              # [B, sum(Cout)]
            raw = sch["raw"].index_select(0, sch["row"])[0]
            blocks, offs = sch["blocks"], sch["offs"]
        else:
            emb = self._time_emb(timesteps) + self._label(y)
            proj_all, blocks, offs = self._emb_project(emb)
            raw = emb.to(cdt())
        proj, o = {}, 0
        for b, n in zip(blocks, offs):
            proj[id(b)] = proj_all[:, o:o + n]
            o += n
        return EmbBundle(raw, proj)

    def prepare_schedule(self, t_values, y, row):
        """All of one image's timestep embeddings at once (the sigma schedule is known on the host before the loop starts):
        time_embed for the n distinct timesteps, + label_emb(y), SiLU, the all-ResBlocks projection -- three GEMMs with M = n and
        M = n B rows per image instead of three M = B GEMVs and a dozen elementwise launches at the head of EVERY step (they sit on
        the critical path of both chains: nothing else can run until the first ResBlock has its row bias).  `row` = a device
        int64 [1] owned by the caller that selects the step; tables live in persistent buffers refreshed in place (captured
        graphs keep pointing at them).  Same arithmetic, same K order: bitwise the per-step values.

        One table per (batch, element type, device), allocated at a CAPACITY of rows (>= 64): a caller that alternates step counts
        or serves several batch sizes at once (the tiled sampler's full and remainder tile groups) keeps every table -- and with it
        every captured graph, whose key carries the table's version -- alive; the version changes on real reallocations only."""
        n, B = len(t_values), y.shape[0]
        t = torch.tensor(list(t_values), dtype=torch.int64, device=y.device)
        emb = self._time_emb(t)[:, None, :] + self._label(y)[None, :, :]             # [n, B, 1280] fp32
        proj_all, blocks, offs = self._emb_project(emb.reshape(n * B, -1))
        key = (B, cdt(), y.device)
        sch = self._schedules.get(key)
        if sch is None or sch["proj"].shape[0] < n or sch["proj"].shape[2] != proj_all.shape[-1]:
            cap = max(64, n)
            sch = dict(proj=torch.empty(cap, B, proj_all.shape[-1], dtype=cdt(), device=y.device),
                       raw=torch.empty(cap, B, emb.shape[-1], dtype=cdt(), device=y.device))
            self._sched_counter[0] += 1
            sch["version"] = self._sched_counter[0]
            self._schedules[key] = sch
        sch["proj"][:n].copy_(proj_all.view(n, B, -1))
        sch["raw"][:n].copy_(emb.to(cdt()))
        sch.update(row=row, blocks=blocks, offs=offs, B=B, cdt=cdt(), active=True, n=n, t_values=tuple(int(v) for v in t_values))
        object.__setattr__(self, "_schedule", sch)
        return sch["version"]

    def use_schedule(self, B, on):
        """Select the table prepared for batch `B` (None -> keep the current one) and switch it on / off for the next call."""
        if B is not None:
            sch = self._schedules.get((B, cdt(), self._schedule["proj"].device)) if self._schedule is not None else None
            if sch is not None:
                if self._schedule is not sch:
                    self._schedule["active"] = False
                object.__setattr__(self, "_schedule", sch)
            else:
                on = False
        if self._schedule is not None:
            self._schedule["active"] = bool(on)
        return bool(on)

    def end_schedule(self):
        for sch in self._schedules.values():
            sch["active"] = False

    def _label(self, y):
        """label_emb(y): constant over the sampling loop -> cached per y-shape on tensor identity/version and refreshed in
        place (captured hipGraphs point at these buffers; see CrossAttention._context_kv)."""
        cache = self._label_cache
        if cache is None:
            cache = {}
            object.__setattr__(self, "_label_cache", cache)
        key = (tuple(y.shape), y.device) + ops._k(cdt())
        c = cache.get(key)
        if c is None or c[0] is not y or c[1] != y._version:
            m0, m2 = self.label_emb[0][0], self.label_emb[0][2]
            lab = ops.gemm(ops.gemm(y.to(cdt()).contiguous(), m0.w(), m0.b32(), act=1), m2.w(), m2.b32(),
                           out_dtype=torch.float32, out=None if c is None else c[2])
            cache[key] = (y, y._version, lab)
        return cache[key][2]

    def refresh_static_conditioning(self, context, y):
        """Recompute every step-independent quantity (text K / V^T of all cross-attention layers, label embedding) for a
        new conditioning, into the SAME buffers: what a captured hipGraph needs before it can be replayed."""
        from .attention import BasicTransformerBlock
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock) and not m.attn2.is_self:
                m.attn2._context_kv(context)
        self._label(y)

    def _conv_in(self, x, add=None):
        c = self.input_blocks[0][0]
        return to_nchw(ops.conv3x3_smallcin(x.float(), c.wf32(), c.b32(), add=add, dtype=cdt()))

    def _out(self, h):
        n, c = self.out[0], self.out[2]
        hn = ops.groupnorm(to_nhwc(h), n.g32(), n.b32(), n.eps, silu=True, part=gn_part_of(h))
        return ops.conv3x3_smallcout(hn, c.w9(), c.b32())

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """Plain SDXL UNet forward (openaimodel.py:981-1013): skip connections by channel concat."""
        emb = self._embed(timesteps, y)
        hs = []
        h = self._conv_in(x)
        hs.append(h)
        for module in list(self.input_blocks)[1:]:
            h = module(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        for module in self.output_blocks:
            h = torch.cat([h, hs.pop()], dim=1).contiguous(memory_format=torch.channels_last)
            h = module(h, emb, context)
        return self._out(h)
