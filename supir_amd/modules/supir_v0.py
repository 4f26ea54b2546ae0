"""SUPIR's control branch and adapted SDXL UNet on the HIP kernels; same constructor kwargs / state-dict keys / forward
signatures as SUPIR/modules/SUPIR_v0.py (ZeroSFT :62-113, ZeroCrossAttn :116-152, GLVControl :155-540,
LightGLVUNet :543-666).
"""
import torch
import torch.nn as nn

from .. import ops
from .. import weights as Wt
from .attention import CrossAttention, MemoryEfficientCrossAttention, SpatialTransformer
from .base import cdt, Conv3x3, GroupNorm32, Linear, Passthrough, Prep, attach_gn_part, gn_part_of, to_nchw, to_nhwc
from .openaimodel import TimestepBlock, TimestepEmbedSequential, UNetModel, Upsample


class ZeroSFT(nn.Module):
    """h' = h + zero_conv(c);  out = GN(cat[h_ori, h']) * (zero_mul(a) + 1) + zero_add(a),  a = SiLU(mlp_shared(c));
    lerp with control_scale against cat[h_ori, h].

    Launches: 1x1 GEMM with the skip as residual epilogue; conv3x3(Cc->128)+SiLU epilogue; ONE conv3x3(128 -> 2*C) for
    gamma|beta (weights concatenated); GroupNorm stats; GroupNorm apply fused with the modulation, the lerp and the
    channel concat (two source pointers) -- the reference's torch.cat never materialises."""

    def __init__(self, label_nc, norm_nc, concat_channels=0, norm=True, mask=False):
        super().__init__()
        assert norm and not mask
        self.param_free_norm = GroupNorm32(norm_nc + concat_channels)  # affine despite the name
        self.mlp_shared = nn.Sequential(Conv3x3(label_nc, 128), Passthrough())
        self.zero_mul = Conv3x3(128, norm_nc + concat_channels)
        self.zero_add = Conv3x3(128, norm_nc + concat_channels)
        self.zero_conv = Linear(label_nc, norm_nc, conv1x1=True)
        self.pre_concat = bool(concat_channels != 0)
        self.mask = mask
        object.__setattr__(self, "_gb", Prep())

    def _w_gamma_beta(self):
        srcs = (self.zero_mul.weight, self.zero_add.weight, self.zero_mul.bias, self.zero_add.bias)
        return self._gb.get(srcs, lambda: (
            torch.cat([Wt.conv3x3_w(self.zero_mul.weight), Wt.conv3x3_w(self.zero_add.weight)], 0).contiguous(),
            torch.cat([Wt.f32(self.zero_mul.bias), Wt.f32(self.zero_add.bias)], 0).contiguous()))

    def control_side(self, c):
        """gamma|beta maps: depend on the control feature only, so they can be produced ahead of the decoder (side stream)."""
        ch = to_nhwc(c)
        m = self.mlp_shared[0]
        actv = ops.conv3x3(ch, m.w(), m.b32(), act=1)
        wgb, bgb = self._w_gamma_beta()
        return ops.conv3x3(actv, wgb, bgb)                                               # [B,H,W,2*Ccat]

    def forward(self, c, h, h_ori=None, control_scale=1, pre=None):
        assert self.mask is False
        ch, hh = to_nhwc(c), to_nhwc(h)
        B, H, W, Cs = hh.shape
        hz, pz = ops.gemm(ch, self.zero_conv.w(), self.zero_conv.b32(), residual=hh, rows_per_batch=H * W, gn_part=True)   # h + zero_conv(c)
        gb = pre if pre is not None else self.control_side(c)
        n = self.param_free_norm
        Ccat = n.num_channels
        cs = float(control_scale)
        if h_ori is not None and self.pre_concat:
            ho = to_nhwc(h_ori)
            out = ops.groupnorm(ho, n.g32(), n.b32(), n.eps, x2=hz, mod_g=gb[..., :Ccat], mod_b=gb[..., Ccat:],
                                control_scale=cs, x2raw=hh if cs != 1.0 else None, part=gn_part_of(h_ori), part2=pz)
        else:
            assert h_ori is None, "h_ori without pre_concat is not built by LightGLVUNet"
            out = ops.groupnorm(hz, n.g32(), n.b32(), n.eps, mod_g=gb[..., :Ccat], mod_b=gb[..., Ccat:],
                                control_scale=cs, x1raw=hh if cs != 1.0 else None, part=pz)
        return to_nchw(out)


class ZeroCrossAttn(nn.Module):
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, context_dim, query_dim, zero_out=True, mask=False):
        super().__init__()
        self.attn = CrossAttention(query_dim=query_dim, context_dim=context_dim, heads=query_dim // 64, dim_head=64)
        self.norm1 = GroupNorm32(query_dim)
        self.norm2 = GroupNorm32(context_dim)
        self.mask = mask

    def control_side(self, context):
        """K and V^T of the (group-normalised) control feature: independent of the decoder state."""
        chh = to_nhwc(context)
        B, Cc = chh.shape[0], chh.shape[-1]
        cn = ops.groupnorm(chh, self.norm2.g32(), self.norm2.b32(), self.norm2.eps, part=gn_part_of(context)).view(B, -1, Cc)
        a, Tk = self.attn, cn.shape[1]
        return ops.gemm(cn, a.to_k.w()), ops.gemm_t(cn, a.to_v.w(), None, B, Tk, (Tk + 63) // 64 * 64)

    def forward(self, context, x, control_scale=1, pre=None):
        """x + attn(GN(x), GN(context)) * control_scale (SUPIR_v0.py:138-152). The control feature changes every step,
        so its K / V^T are projected every step (no cross-step cache)."""
        assert self.mask is False
        xh = to_nhwc(x)
        B, H, W, C = xh.shape
        xn = ops.groupnorm(xh, self.norm1.g32(), self.norm1.b32(), self.norm1.eps, part=gn_part_of(x)).view(B, H * W, C)
        k, vt = pre if pre is not None else self.control_side(context)
        a = self.attn
        T, Tk = H * W, k.shape[1]
        q = ops.gemm(xn, a.to_q.w())
        o = ops.flash_attn(q, k, vt, B, a.heads, T, Tk)
        out = ops.gemm(o, a.to_out[0].w(), a.to_out[0].b32(), residual=xh.view(B, T, C), alpha=float(control_scale))
        return to_nchw(out.view(B, H, W, C))


class GLVControl(UNetModel):
    """SDXL encoder + middle block run on the noisy latent `xt`, with the LQ latent `x` entering through
    input_hint_block; returns the 10 multi-scale feature maps (SUPIR_v0.py:499-540)."""

    def __init__(self, *args, input_upscale=1, **kwargs):
        kwargs.pop("build_decoder", None)
        super().__init__(*args, build_decoder=False, **kwargs)
        assert input_upscale == 1
        self.input_upscale = input_upscale
        self.input_hint_block = TimestepEmbedSequential(Conv3x3(self.in_channels, self.model_channels))

    def _guided_hint(self, x, out=None):
        hint = self.input_hint_block[0]
        return ops.conv3x3_smallcin(x.float(), hint.wf32(), hint.b32(), dtype=cdt(), out=out)

    def prologue(self, x, timesteps, xt, y=None):
        """Embeddings + the two <= 8-channel input convolutions: (emb, h0).  Kept apart from `body` because it mixes torch
        elementwise ops with kernel launches (it cannot be recorded by ops.paired_run)."""
        emb = self._embed(timesteps, y)
        sch = self._schedule
        if sch is not None and sch["active"] and sch["cdt"] == cdt() and sch.get("hint") is not None and sch["hint"].shape[0] == x.shape[0]:
            guided_hint = sch["hint"]          # input_hint_block(LQ latent): the same every step of an image (prepare_schedule)
        else:
            guided_hint = self._guided_hint(x)
        return emb, self._conv_in(xt, add=guided_hint)          # input_blocks[0](xt) + guided_hint in one kernel

    def prepare_schedule(self, t_values, y, row, control=None):
        """UNetModel.prepare_schedule + the control branch's own step-invariant: guided_hint = input_hint_block(x) depends on the LQ
        latent only (SUPIR_v0.py:504-505), which a sampler passes unchanged to every step of an image -- computed once per image
        into a persistent buffer (refreshed in place: captured graphs point at it) instead of once per step at the head of the
        control chain."""
        ver = super().prepare_schedule(t_values, y, row)
        sch = self._schedule
        old = sch.get("hint")
        if control is not None:
            shape = (control.shape[0], control.shape[2], control.shape[3], self.model_channels)
            if old is None or tuple(old.shape) != shape or old.dtype != cdt() or old.device != control.device:
                old = torch.empty(shape, dtype=cdt(), device=control.device)
                self._sched_counter[0] += 1
                sch["version"] = ver = self._sched_counter[0]
            sch["hint"] = self._guided_hint(control, out=old)
        else:
            if old is not None:
                # a graph captured while the hint was cached reads that buffer instead of convolving c["control"]: without the
                # cache the launches differ, so the table's version (part of the graph key) must change with the hint's PRESENCE
                self._sched_counter[0] += 1
                sch["version"] = ver = self._sched_counter[0]
            sch["hint"] = None
        return ver

    def body(self, h, emb, context=None):
        """input_blocks[1:] + middle_block on h0: the 10 feature maps.  The same stack of layers, shape for shape, as
        LightGLVUNet.encode_body (ControlWrapper issues the two as grouped launches)."""
        hs = [h]
        for module in list(self.input_blocks)[1:]:
            h = module(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        hs.append(h)
        return hs

    def forward(self, x, timesteps, xt, context=None, y=None, **kwargs):
        emb, h = self.prologue(x, timesteps, xt, y)
        return self.body(h, emb, context)


class LightGLVUNet(UNetModel):
    def __init__(self, mode="", project_type="ZeroSFT", project_channel_scale=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if mode != "XL-base" or project_type != "ZeroSFT":
            raise NotImplementedError("SUPIR ships mode='XL-base', project_type='ZeroSFT' (options/SUPIR_v0*.yaml)")
        cond_output_channels = [320] * 4 + [640] * 3 + [1280] * 3
        project_channels = [int(c * project_channel_scale) for c in [160] * 4 + [320] * 3 + [640] * 3]
        concat_channels = [320] * 2 + [640] * 3 + [1280] * 4 + [0]
        cross_attn_insert_idx = [6, 3]
        self.progressive_mask_nums = [0, 3, 7, 11]
        self.project_modules = nn.ModuleList()
        for i in range(len(cond_output_channels)):
            self.project_modules.append(ZeroSFT(project_channels[i], cond_output_channels[i],
                                                concat_channels=concat_channels[i]))
        for i in cross_attn_insert_idx:
            self.project_modules.insert(i, ZeroCrossAttn(cond_output_channels[i], concat_channels[i]))

    def encode_prologue(self, x, timesteps=None, y=None):
        return self._embed(timesteps, y), self._conv_in(x)

    def encode_body(self, h, emb, context=None):
        hs = [h]
        for module in list(self.input_blocks)[1:]:
            h = module(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        return emb, hs, h

    def encode(self, x, timesteps=None, context=None, y=None):
        """Encoder + middle block: independent of the control branch (it can run concurrently with GLVControl, or -- the two
        being the same stack of layers -- be issued with it as grouped launches: ControlWrapper.pair_branches)."""
        emb, h = self.encode_prologue(x, timesteps, y)
        return self.encode_body(h, emb, context)

    def adapter_control_sides(self, control, on_done=None):
        """Control-side half of every adapter (ZeroSFT gamma|beta maps, ZeroCrossAttn K / V^T) in the order the decoder
        consumes them.  `on_done(idx, tensors)` is called after each one (ControlWrapper records a stream event there)."""
        pre = {}
        adapter_idx, control_idx = len(self.project_modules) - 1, len(control) - 1
        pre[adapter_idx] = self.project_modules[adapter_idx].control_side(control[control_idx])
        if on_done:
            on_done(adapter_idx, pre[adapter_idx])
        adapter_idx -= 1
        control_idx -= 1
        for module in self.output_blocks:
            pre[adapter_idx] = self.project_modules[adapter_idx].control_side(control[control_idx])
            if on_done:
                on_done(adapter_idx, pre[adapter_idx])
            adapter_idx -= 1
            if len(module) == 3:
                pre[adapter_idx] = self.project_modules[adapter_idx].control_side(control[control_idx])
                if on_done:
                    on_done(adapter_idx, pre[adapter_idx])
                adapter_idx -= 1
            control_idx -= 1
        return pre

    def forward(self, x, timesteps=None, context=None, y=None, control=None, control_scale=1, encoded=None,
                adapter_pre=None, **kwargs):
        """SUPIR_v0.py:600-666. Skip concat replaced by ZeroSFT; ZeroCrossAttn before the Upsample of 3-child blocks.
        `encoded` = a precomputed encode() result."""
        emb, hs, h = encoded if encoded is not None else self.encode(x, timesteps, context, y)
        hs = list(hs)
        adapter_idx = len(self.project_modules) - 1
        control_idx = len(control) - 1
        pre = adapter_pre if adapter_pre is not None else (lambda i: None)
        h = self.project_modules[adapter_idx](control[control_idx], h, control_scale=control_scale, pre=pre(adapter_idx))
        adapter_idx -= 1
        control_idx -= 1
        for module in self.output_blocks:
            _h = hs.pop()
            h = self.project_modules[adapter_idx](control[control_idx], _h, h, control_scale=control_scale,
                                                  pre=pre(adapter_idx))
            adapter_idx -= 1
            if len(module) == 3:
                assert isinstance(module[2], Upsample)
                for layer in module[:2]:
                    if isinstance(layer, TimestepBlock):
                        h = layer(h, emb)
                    elif isinstance(layer, SpatialTransformer):
                        h = layer(h, context)
                    else:
                        h = layer(h)
                h = self.project_modules[adapter_idx](control[control_idx], h, control_scale=control_scale,
                                                      pre=pre(adapter_idx))
                adapter_idx -= 1
                h = module[2](h)
            else:
                h = module(h, emb, context)
            control_idx -= 1
        return self._out(h)
