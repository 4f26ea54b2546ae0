"""SpatialTransformer stack on the HIP kernels; same constructor kwargs / state-dict keys / call signatures as
sgm/modules/attention.py (CrossAttention :196-285, MemoryEfficientCrossAttention :288-373, GEGLU/FeedForward :84-110,
BasicTransformerBlock :376-486, SpatialTransformer :533-635).

Per transformer block: 3 LayerNorm launches, 8 GEMM launches (fused q|k projection, transposed v projection, GEGLU
epilogue, residual epilogues) and 2 flash-attention launches; the token stream stays bf16 [B*T, C] throughout.
K / V^T of the text context are computed once per context tensor and reused across all sampling steps.
"""
import torch
import torch.nn as nn

from .. import ops
from .. import weights as Wt
from .base import cdt, Linear, Norm, Normalize, Passthrough, Prep, attach_gn_part, gn_part_of, to_nchw, to_nhwc, tokens_bf16


FOLD_LAYERNORM = True   # SpatialTransformer runs its blocks through BasicTransformerBlock.forward_fused
FINALIZE_STATS = False  # (measured equal to summing in the consumer; off = fewer launches)
# when True: reduce the row-statistic partials once per LayerNorm (tiny launch) instead of in every consumer


def _pad64(t):
    return (t + 63) // 64 * 64


class CrossAttention(nn.Module):
    """q = to_q(x), k/v = to_k/to_v(context or x), softmax(q k^T / sqrt(64)) v, to_out.0 (+bias).
    Head split 'b n (h d) -> b h n d' (attention.py:254) is the column layout the flash kernel reads directly."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, backend=None, **kwargs):
        super().__init__()
        if dim_head != 64:
            raise ValueError("the gfx950 flash-attention kernel is built for head dim 64 (SDXL / SUPIR)")
        inner = dim_head * heads
        self.is_self = context_dim is None
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(context_dim, inner, bias=False)
        self.to_v = Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(Linear(inner, query_dim), Passthrough())
        object.__setattr__(self, "_qk", Prep())
        object.__setattr__(self, "_kv_cache", None)

    def _w_qk(self):
        return self._qk.get((self.to_q.weight, self.to_k.weight),
                            lambda: torch.cat([Wt.linear_w(self.to_q.weight), Wt.linear_w(self.to_k.weight)], 0).contiguous())

    def _context_kv(self, context):
        """K [B,Tk,C] and V^T [B,C,Tpad] of a context tensor.  One cache entry PER CONTEXT SHAPE, keyed on tensor identity +
        version and refreshed IN PLACE: a captured hipGraph holds raw pointers to these buffers, so an entry is never
        reallocated or evicted once created (switching between batch sizes -- e.g. tile-batched and single-tile calls of the
        tiled sampler -- alternates between entries instead of freeing the other graph's buffers)."""
        wk, wv = self.to_k.w(), self.to_v.w()
        cache = self._kv_cache
        if cache is None:
            cache = {}
            object.__setattr__(self, "_kv_cache", cache)
        key = (tuple(context.shape), context.device) + ops._k(cdt())   # fp16 scopes keep their own buffers (other element type)
        c = cache.get(key)
        if c is not None and c[0] is context and c[1] == context._version and c[2] is wk and c[3] is wv:
            return c[4], c[5]
        ctx = tokens_bf16(context)
        B, Tk, _ = ctx.shape
        k_old = vt_old = None
        if c is not None:
            k_old, vt_old = c[4], c[5]
        k = ops.gemm(ctx, wk, out=k_old)
        vt = ops.gemm_t(ctx, wv, None, B, Tk, _pad64(Tk), out=vt_old)
        cache[key] = (context, context._version, wk, wv, k, vt)
        return k, vt

    def attend(self, x, context=None, residual=None, alpha=1.0, inplace=False):
        """x [B,T,C] bf16 tokens -> to_out(attention) (* alpha) (+ residual); inplace writes into `residual`."""
        B, T, C = x.shape
        H = self.heads
        inner = H * 64
        if context is None:
            qk = ops.gemm(x, self._w_qk())                      # [B,T,2*inner]
            vt = ops.gemm_t(x, self.to_v.w(), None, B, T, _pad64(T))
            a = ops.flash_attn(qk[:, :, :inner], qk[:, :, inner:], vt, B, H, T, T)
        else:
            q = ops.gemm(x, self.to_q.w())
            k, vt = self._context_kv(context)
            a = ops.flash_attn(q, k, vt, B, H, T, k.shape[1])
        out = residual if (inplace and residual is not None) else None
        return ops.gemm(a, self.to_out[0].w(), self.to_out[0].b32(), residual=residual, alpha=alpha, out=out)

    def forward(self, x, context=None, mask=None, additional_tokens=None, n_times_crossframe_attn_in_self=0):
        if mask is not None or additional_tokens is not None or n_times_crossframe_attn_in_self:
            raise NotImplementedError("mask / additional_tokens / crossframe attention are not on SUPIR's path")
        return self.attend(tokens_bf16(x), context)


MemoryEfficientCrossAttention = CrossAttention  # the xformers variant is the same math (attention.py:313-373)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)
        object.__setattr__(self, "_il", Prep())
        object.__setattr__(self, "_il16", Prep())

    def w_interleaved(self, block=32):
        """value / gate rows interleaved per `block` rows: 32 for the gemm.hip tiles, 16 for tile 34 of gemm16.hip."""
        prep = self._il if block == 32 else self._il16
        return prep.get((self.proj.weight, self.proj.bias),
                        lambda: Wt.interleave_geglu(Wt.linear_w(self.proj.weight), Wt.f32(self.proj.bias), block))

    def forward(self, x):
        w, b = self.w_interleaved()
        return ops.gemm(tokens_bf16(x), w, b, act=2, alt16=self.w_interleaved(16) if ops.USE_GEMM16 else None)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("SUPIR uses gated_ff=True everywhere")
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), Passthrough(), Linear(inner, dim if dim_out is None else dim_out))

    def forward(self, x, residual=None, inplace=False):
        h = self.net[0](x)
        return ops.gemm(h, self.net[2].w(), self.net[2].b32(), residual=residual, out=residual if inplace else None)


class BasicTransformerBlock(nn.Module):
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attn_mode="softmax", sdp_backend=None):
        super().__init__()
        assert attn_mode in self.ATTENTION_MODES
        self.disable_self_attn = disable_self_attn
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head,
                                    context_dim=context_dim if disable_self_attn else None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head)
        self.norm1 = Norm(dim, 1e-5)
        self.norm2 = Norm(dim, 1e-5)
        self.norm3 = Norm(dim, 1e-5)
        object.__setattr__(self, "_fold", Prep())

    # ------------------------------------------------------------------ LayerNorm-folded fast path
    def _folded(self):
        """Weights of the four LayerNorm consumers with the norm folded in (W' = gamma (.) W, column sums, b')."""
        a1, a2, ff = self.attn1, self.attn2, self.ff.net[0]
        srcs = (self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, self.norm3.weight, self.norm3.bias,
                a1.to_q.weight, a1.to_k.weight, a1.to_v.weight, a2.to_q.weight, ff.proj.weight, ff.proj.bias)

        def build():
            n1, n2, n3 = self.norm1, self.norm2, self.norm3
            wqk = torch.cat([a1.to_q.weight.detach().float(), a1.to_k.weight.detach().float()], 0)
            qk = Wt.fold_layernorm(wqk, None, n1.weight, n1.bias)
            v = Wt.fold_layernorm(a1.to_v.weight, None, n1.weight, n1.bias)
            # fused q|k|v (supir_gemm_bf16_qkv): the same folded matrices stacked [q; k; v]
            qkv = (torch.cat([qk[0], v[0]], 0).contiguous(), torch.cat([qk[1], v[1]], 0).contiguous(),
                   torch.cat([qk[2], v[2]], 0).contiguous())
            q2 = Wt.fold_layernorm(a2.to_q.weight, None, n2.weight, n2.bias)
            gw, gc, gb = Wt.fold_layernorm(ff.proj.weight, ff.proj.bias, n3.weight, n3.bias)
            gwi, gbi = Wt.interleave_geglu(gw, gb)
            _, gci = Wt.interleave_geglu(gw, gc)
            gw16, gb16 = Wt.interleave_geglu(gw, gb, 16)      # tile 34 (csrc/gemm16.hip) pairs value / gate per 16 rows
            _, gc16 = Wt.interleave_geglu(gw, gc, 16)
            return dict(qk=qk, v=v, q2=q2, qkv=qkv, geglu=(gwi, gci, gbi), geglu16=(gw16, gc16, gb16))

        return self._fold.get(srcs, build)

    def forward_fused(self, x, stats, context):
        """Token stream x [B,T,C] (updated in place) + the RowStats its producer emitted -> (x, RowStats).
        No LayerNorm launches: every norm is folded into the GEMM that consumes it (supir_gemm_bf16_ln); each residual
        GEMM emits the row statistics the next norm needs."""
        f = self._folded()
        B, T, C = x.shape
        H = self.attn1.heads
        inner = H * 64
        e1, e2, e3 = self.norm1.eps, self.norm2.eps, self.norm3.eps
        if FINALIZE_STATS:
            stats = ops.rowstats_finalize(stats, C, e1)
        def qkv_separate():
            w, cs, b = f["qk"]
            qk_ = ops.gemm_ln(x, w, b, ln=stats, colsum=cs, ln_eps=e1)
            w, cs, b = f["v"]
            return qk_, ops.gemm_ln(x, w, b, ln=stats, colsum=cs, ln_eps=e1, trans=(B, T, _pad64(T)))

        def qkv_fused():
            w, cs, b = f["qkv"]
            return ops.gemm_qkv(x, w, b, B, T, 2 * inner, ln=stats, colsum=cs, ln_eps=e1)

        if T % 64 == 0 and ops.gemm_qkv_supported(B * T, 3 * inner, 2 * inner, C, T):
            # one launch for q | k | v^T instead of two when it is faster for this shape (timed once, outside graph capture)
            which = ops.choose(("qkv", B * T, 3 * inner, C) + ops._k(x.dtype), (qkv_separate, qkv_fused), prefer=1)
            qk, vt = qkv_fused() if which == 1 else qkv_separate()
        else:
            qk, vt = qkv_separate()
        a = ops.flash_attn(qk[:, :, :inner], qk[:, :, inner:], vt, B, H, T, T)
        o1 = self.attn1.to_out[0]
        x, stats = ops.gemm_ln(a, o1.w(), o1.b32(), residual=x, out=x, emit_stats=True)
        if FINALIZE_STATS:
            stats = ops.rowstats_finalize(stats, C, e2)
        w, cs, b = f["q2"]
        k, vt2 = self.attn2._context_kv(context)
        Tk = k.shape[1]
        st2 = stats

        def xattn_separate():
            q = ops.gemm_ln(x, w, b, ln=st2, colsum=cs, ln_eps=e2)
            return ops.flash_attn(q, k, vt2, B, H, T, Tk)

        def xattn_fused():
            return ops.xattn_q(x, w, b, k, vt2, B, H, T, Tk, ln=st2, colsum=cs, ln_eps=e2)

        if ops.xattn_q_supported(B, T, C, H, Tk):
            # to_q + the attention over the (<= 128) text keys as ONE launch where that is faster (timed once, outside graph capture)
            which = ops.choose(("xattn", B, T, C, H, Tk) + ops._k(x.dtype), (xattn_separate, xattn_fused), prefer=1)
            a = xattn_fused() if which == 1 else xattn_separate()
        else:
            a = xattn_separate()
        o2 = self.attn2.to_out[0]
        x, stats = ops.gemm_ln(a, o2.w(), o2.b32(), residual=x, out=x, emit_stats=True)
        if FINALIZE_STATS:
            stats = ops.rowstats_finalize(stats, C, e3)
        w, cs, b = f["geglu"]
        g = ops.gemm_ln(x, w, b, act=2, ln=stats, colsum=cs, ln_eps=e3, alt16=f["geglu16"] if ops.USE_GEMM16 else None)
        l2 = self.ff.net[2]
        return ops.gemm_ln(g, l2.w(), l2.b32(), residual=x, out=x, emit_stats=True)

    def forward(self, x, context=None, additional_tokens=None, n_times_crossframe_attn_in_self=0, inplace=False):
        """x [B,T,C] -> x + attn1(LN x) ... (attention.py:465-486). inplace=True updates the token stream in place
        (SpatialTransformer owns it)."""
        x = tokens_bf16(x)
        n = ops.layernorm(x, self.norm1.g32(), self.norm1.b32(), self.norm1.eps)
        x = self.attn1.attend(n, context if self.disable_self_attn else None, residual=x, inplace=inplace)
        n = ops.layernorm(x, self.norm2.g32(), self.norm2.b32(), self.norm2.eps, out=n)
        x = self.attn2.attend(n, context, residual=x, inplace=True)
        n = ops.layernorm(x, self.norm3.g32(), self.norm3.b32(), self.norm3.eps, out=n)
        return self.ff(n, residual=x, inplace=True)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, disable_self_attn=False,
                 use_linear=False, attn_type="softmax", use_checkpoint=True, sdp_backend=None):
        super().__init__()
        if isinstance(context_dim, (list, tuple)):
            context_dim = list(context_dim)
            assert all(c == context_dim[0] for c in context_dim)
            context_dim = context_dim[0]
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.use_linear = use_linear
        self.proj_in = Linear(in_channels, inner, conv1x1=not use_linear)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim,
                                  disable_self_attn=disable_self_attn, attn_mode=attn_type, checkpoint=use_checkpoint)
            for _ in range(depth)])
        self.proj_out = Linear(inner, in_channels, conv1x1=not use_linear)

    def forward(self, x, context=None):
        """x logical [B,C,H,W] -> same (attention.py:614-635; linear and 1x1-conv projections are the same GEMM)."""
        if isinstance(context, list):
            context = context[0]
        xh = to_nhwc(x)
        B, H, W, C = xh.shape
        n = ops.groupnorm(xh, self.norm.g32(), self.norm.b32(), self.norm.eps, part=gn_part_of(x))
        fused = FOLD_LAYERNORM and ops.has_fused(cdt()) and context is not None and all(
            (not b.disable_self_attn) and (not b.attn2.is_self) for b in self.transformer_blocks)
        if fused:
            t, stats = ops.gemm_ln(n.view(B, H * W, C), self.proj_in.w(), self.proj_in.b32(), emit_stats=True)
            for blk in self.transformer_blocks:
                t, stats = blk.forward_fused(t, stats, context)
        else:
            t = ops.gemm(n.view(B, H * W, C), self.proj_in.w(), self.proj_in.b32())
            for blk in self.transformer_blocks:
                t = blk(t, context=context, inplace=True)
        out, po = ops.gemm(t, self.proj_out.w(), self.proj_out.b32(), residual=xh.view(B, H * W, C), rows_per_batch=H * W, gn_part=True)
        return attach_gn_part(to_nchw(out.view(B, H, W, C)), po)
