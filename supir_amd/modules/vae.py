"""SDXL VAE (first stage) on the HIP kernels; constructor kwargs / state-dict keys / call signatures follow
sgm/modules/diffusionmodules/model.py (ResnetBlock :91-148, AttnBlock :158-198, MemoryEfficientAttnBlock :201-262,
Upsample/Downsample :55-88, Encoder :482-596, Decoder :599-743), sgm/models/autoencoder.py:282-321 (AutoencoderKL,
AutoencoderKLInferenceWrapper) and sgm/modules/distributions/distributions.py:24-72.

Activations are NHWC bf16 between the fp32 NCHW image / latent boundaries.  The mid-block single-head attention
(head dim 512, up to 16384 tokens at 1024 px) is one flash-attention launch with the keys split over workgroup sets
(csrc/attention_d512.hip) from 1024 tokens up; smaller maps materialise their fp32 score matrix and run as GEMM -> row softmax ->
GEMM on the MFMA GEMM kernel.
"""
import torch
import torch.nn as nn

from .. import ops
from .base import attach_gn_part, cdt, Conv3x3, gn_part_of, Linear, Normalize, Prep, to_nchw, to_nhwc
from .. import weights as Wt


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512):
        super().__init__()
        assert not conv_shortcut and temb_channels == 0
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv3x3(in_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.conv2 = Conv3x3(out_channels, out_channels)
        if in_channels != out_channels:
            self.nin_shortcut = Linear(in_channels, out_channels, conv1x1=True)

    def forward(self, x, temb=None):
        """model.py:126-148.  Every convolution epilogue also leaves the GroupNorm statistics of what it stored (ops.GnPart, where
        the tile that runs can emit them): norm2 and the NEXT module's first norm then skip their pass over the tensor."""
        xh = to_nhwc(x)
        h = ops.groupnorm(xh, self.norm1.g32(), self.norm1.b32(), self.norm1.eps, silu=True, part=gn_part_of(x))
        h, p1 = ops.conv3x3(h, self.conv1.w(), self.conv1.b32(), gn_part=True)
        h = ops.groupnorm(h, self.norm2.g32(), self.norm2.b32(), self.norm2.eps, silu=True, out=h, part=p1)
        sk = xh
        if self.in_channels != self.out_channels:
            sk = ops.gemm(xh, self.nin_shortcut.w(), self.nin_shortcut.b32())
        out, po = ops.conv3x3(h, self.conv2.w(), self.conv2.b32(), residual=sk, gn_part=True)
        return attach_gn_part(to_nchw(out), po)


class AttnBlock(nn.Module):
    """Single-head self attention over H*W tokens, head dim = C (model.py:177-192 == :228-256)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Linear(in_channels, in_channels, conv1x1=True)
        self.k = Linear(in_channels, in_channels, conv1x1=True)
        self.v = Linear(in_channels, in_channels, conv1x1=True)
        self.proj_out = Linear(in_channels, in_channels, conv1x1=True)

    def attend(self, n):
        """n: normalised tokens [B, T, C] bf16 -> proj_out-less attention output [B, T, C].  Any T.
        C == 512 (every SDXL / SUPIR VAE) and ops.use_flash_d512(T) (T >= 1024 tokens: 256^2 px images, tiled-VAE tiles and everything
        larger, where it measures faster): q, k and v^T projections + ONE flash-attention launch with key splits + its merge
        (csrc/attention_d512.hip), no score matrix.  Otherwise (small maps) the materialised form: the key axis is padded to a multiple of 64
        with zero K rows / zero V^T columns, a GEMM writes fp32 scores [T, Tp], softmax_rows masks the padding, a GEMM applies P."""
        B, T, C = n.shape
        Tp = (T + 63) // 64 * 64
        q = ops.gemm(n, self.q.w(), self.q.b32())
        if C == 512 and ops.use_flash_d512(T):
            k = ops.gemm(n, self.k.w(), self.k.b32())
            vt = ops.gemm_t(n, self.v.w(), self.v.b32(), B, T, Tp)   # [B, C, Tp], zero padded
            return ops.flash_attn_d512(q, k, vt, T)
        if Tp == T:
            k = ops.gemm(n, self.k.w(), self.k.b32())
        else:
            # the padded buffer's batch stride (Tp*C) differs from its dense row extent (T*C), so a [B, T, C] view of it is not
            # a uniform-stride row matrix: project one batch element at a time (tiled-VAE tiles with B > 1 hit this)
            k = torch.zeros(B, Tp, C, dtype=cdt(), device=n.device)
            for b in range(B):
                ops.gemm(n[b], self.k.w(), self.k.b32(), out=k[b, :T])
        vt = ops.gemm_t(n, self.v.w(), self.v.b32(), B, T, Tp)       # [B, C, Tp], zero padded
        o = torch.empty(B, T, C, dtype=cdt(), device=n.device)
        for b in range(B):
            s = ops.gemm(q[b], k[b], out_dtype=torch.float32)       # [T, Tp] fp32 scores
            p = ops.softmax_rows(s, C ** -0.5, valid=T, dtype=cdt())
            ops.gemm(p, vt[b], out=o[b])                            # P [T,Tp] . (V^T [C,Tp])^T
        return o

    def forward(self, x, **kwargs):
        xh = to_nhwc(x)
        B, H, W, C = xh.shape
        T = H * W
        n = ops.groupnorm(xh, self.norm.g32(), self.norm.b32(), self.norm.eps, part=gn_part_of(x)).view(B, T, C)
        o = self.attend(n)
        out, po = ops.gemm(o, self.proj_out.w(), self.proj_out.b32(), residual=xh.view(B, T, C), rows_per_batch=T, gn_part=True)
        return attach_gn_part(to_nchw(out.view(B, H, W, C)), po)


MemoryEfficientAttnBlock = AttnBlock


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None):
    if attn_type in ("vanilla", "vanilla-xformers"):
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity()
    raise NotImplementedError(attn_type)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = Conv3x3(in_channels, in_channels)

    def forward(self, x):
        out, po = ops.conv3x3(to_nhwc(x), self.conv.w(), self.conv.b32(), upsample=True, gn_part=True)
        return attach_gn_part(to_nchw(out), po)


class Downsample(nn.Module):
    """F.pad(x, (0,1,0,1)) + conv s2 p0 (model.py:81-86) == taps beyond the bottom/right edge read zero."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = Conv3x3(in_channels, in_channels)

    def forward(self, x):
        xh = to_nhwc(x)
        H, W = xh.shape[1:3]
        out, po = ops.conv3x3(xh, self.conv.w(), self.conv.b32(), stride=2, pad=(0, 0), out_hw=(H // 2, W // 2), gn_part=True)
        return attach_gn_part(to_nchw(out), po)


class _Level(nn.Module):
    pass


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        assert len(attn_resolutions) == 0
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.conv_in = Conv3x3(in_channels, ch)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            down = _Level()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv3x3(block_in, 2 * z_channels if double_z else z_channels)

    def forward(self, x):
        """fp32 NCHW image [N,3,H,W] -> fp32 NCHW moments-before-quant [N,8,H/8,W/8] (model.py:571-596)."""
        h = to_nchw(ops.conv3x3_smallcin(x.float(), self.conv_in.wf32(), self.conv_in.b32(), dtype=cdt()))
        for i_level in range(self.num_resolutions):
            for blk in self.down[i_level].block:
                h = blk(h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        hn = ops.groupnorm(to_nhwc(h), self.norm_out.g32(), self.norm_out.b32(), self.norm_out.eps, silu=True, part=gn_part_of(h))
        return ops.conv3x3_smallcout(hn, self.conv_out.w9(), self.conv_out.b32())


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        assert len(attn_resolutions) == 0 and not give_pre_end and not tanh_out
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = Conv3x3(z_channels, block_in)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            up = _Level()
            up.block = block
            up.attn = nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv3x3(block_in, out_ch)

    def forward(self, z, **kwargs):
        """fp32 NCHW [N,4,h,w] (after post_quant_conv) -> fp32 NCHW image [N,3,8h,8w] (model.py:710-743)."""
        h = to_nchw(ops.conv3x3_smallcin(z.float(), self.conv_in.wf32(), self.conv_in.b32(), dtype=cdt()))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_resolutions)):
            for blk in self.up[i_level].block:
                h = blk(h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        hn = ops.groupnorm(to_nhwc(h), self.norm_out.g32(), self.norm_out.b32(), self.norm_out.eps, silu=True, part=gn_part_of(h))
        return ops.conv3x3_smallcout(hn, self.conv_out.w9(), self.conv_out.b32())


class DiagonalGaussianDistribution:
    """distributions.py:24-72 (mean / logvar split, clamp, sample = mean + std * randn drawn on the CPU generator then
    moved -- the reference's RNG behaviour, SURVEY q1 -- , mode)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape)
        return self.mean + self.std * noise.to(device=self.parameters.device, dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class PointwiseConv(nn.Module):
    """Holder + forward for the 1x1 quant convs on fp32 NCHW latents (autoencoder.py:297-298)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout), requires_grad=False)

    def forward(self, x, in_scale=1.0):
        return ops.pointwise_nchw(x.float(), self.weight.float(), self.bias.float(), in_scale=in_scale)


class AutoencoderKL(nn.Module):
    def __init__(self, embed_dim=4, ddconfig=None, ckpt_path=None, lossconfig=None, monitor=None, **kwargs):
        super().__init__()
        dd = dict(ddconfig)
        assert dd["double_z"]
        self.encoder = Encoder(**dd)
        self.decoder = Decoder(**dd)
        self.quant_conv = PointwiseConv(2 * dd["z_channels"], 2 * embed_dim)
        self.post_quant_conv = PointwiseConv(embed_dim, dd["z_channels"])
        self.embed_dim = embed_dim

    def encode(self, x):
        return DiagonalGaussianDistribution(self.quant_conv(self.encoder(x)))

    def decode(self, z, **kw):
        return self.decoder(self.post_quant_conv(z))


class AutoencoderKLInferenceWrapper(AutoencoderKL):
    def encode(self, x):
        return super().encode(x).sample()
