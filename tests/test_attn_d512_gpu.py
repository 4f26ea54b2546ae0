"""supir_flash_attn_d512 (csrc/attention_d512.hip): the VAE mid-block attention -- one head of dimension 512
(sgm/modules/diffusionmodules/model.py:177-192 AttnBlock.attention == :228-256 MemoryEfficientAttnBlock.attention) -- without a
materialised score matrix, against F.scaled_dot_product_attention in fp32 on the same bf16-rounded operands, and the VAE
AttnBlock with the flash path against the materialised-score path (which the reference-golden VAE tests pin).

Tolerance: rel-L2 <= 4e-3 (the bar of the head-dim-64 kernel, tests/test_kernels_gpu.py): P is rounded to bf16 before P.V.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from supir_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _ref(q, k, v):
    return F.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None])[:, 0]


def _vt(v, Tp):
    B, T, C = v.shape
    vt = torch.zeros(B, C, Tp, dtype=v.dtype, device=DEV)
    vt[:, :, :T] = v.permute(0, 2, 1)
    return vt


def _check(out, ref, rel, name):
    assert torch.isfinite(out).all(), name
    err = ((out.float() - ref).norm() / ref.norm()).item()
    mx = (out.float() - ref).abs().max().item()
    assert err <= rel, f"{name}: rel-L2 {err:.3e} > {rel} (max-abs {mx:.3e})"
    return err


# T = 16384 is the mid block at 1024 x 1024 px (BASELINE config 2); 4096 at 512 px (config 1); 7396 = an 86 x 86 tiled-VAE
# decoder tile (ragged: 7396 = 231 * 32 + 4); 63 / 1 / 33: fewer keys than a tile, a single key, one key into the second tile
@pytest.mark.parametrize("B,T", [(1, 16384), (1, 4096), (2, 7396), (2, 1024), (1, 63), (1, 1), (3, 33), (1, 200)])
def test_flash_attn_d512_vs_sdpa(B, T):
    # scale 3: logits spread over ~ +-15, so the softmax is far from uniform and the row maxima matter
    q, k, v = rnd(B, T, 512, scale=3.0).to(BF), rnd(B, T, 512, seed=1).to(BF), rnd(B, T, 512, seed=2).to(BF)
    Tp = (T + 63) // 64 * 64
    out = ops.flash_attn_d512(q, k, _vt(v, Tp), T)
    _check(out, _ref(q, k, v), 4e-3, f"d512 B{B} T{T}")


# key-split form (supir_flash_attn_d512_split): explicit split counts incl. ones that do not divide the key tiles (4096 keys = 128 tiles
# over 3), more splits than sensible (16 over 32 tiles), ragged last tiles (7396, 200, 33, 63 keys) and a split holding ONE key (33 keys
# = 2 tiles over 2 splits).  Against fp32 SDPA at the kernel's bar, against the single-pass launch (both round the same fp32 value to
# bf16 after different fp32 summation orders), and bitwise reproducible.
@pytest.mark.parametrize("B,T,splits", [(1, 16384, 2), (1, 4096, 8), (1, 4096, 3), (2, 7396, 2), (2, 1024, 16), (1, 200, 4), (3, 33, 2),
                                        (1, 63, 2), (1, 512, 0)])
def test_flash_attn_d512_key_splits(B, T, splits):
    from supir_amd import _lib
    q, k, v = rnd(B, T, 512, scale=3.0).to(BF), rnd(B, T, 512, seed=1).to(BF), rnd(B, T, 512, seed=2).to(BF)
    vt = _vt(v, (T + 63) // 64 * 64)
    want = _lib.load().supir_flash_attn_d512_workspace(B, T, T, splits)
    assert want > 0, "case does not exercise the split form"
    ops.start_trace()
    out = ops.flash_attn_d512(q, k, vt, T, splits=splits)
    rec = [r for r in ops.stop_trace() if r["kernel"] == "attn_d512"]
    assert len(rec) == 1 and rec[0]["splits"] == want // (B * T * 514 * 4) and rec[0]["splits"] > 1
    single = ops.flash_attn_d512(q, k, vt, T, splits=1)
    _check(out, _ref(q, k, v), 4e-3, f"d512 split B{B} T{T} x{splits}")
    e = ((out.float() - single.float()).norm() / single.float().norm()).item()
    assert e <= 3e-3, f"split vs single pass: {e:.3e}"
    assert torch.equal(out, ops.flash_attn_d512(q, k, vt, T, splits=splits))


def test_flash_attn_d512_split_extreme_logits_across_splits():
    """The dominating key sits in the LAST split and one query is near one-hot: the other splits' partial outputs must vanish in the
    merge (weights 2^((m_h - m) c) ~ 0), not pollute it."""
    B, T = 1, 1024
    q, k, v = rnd(B, T, 512).to(BF), rnd(B, T, 512, seed=1).to(BF), rnd(B, T, 512, seed=2).to(BF)
    k[:, 1000] = q[:, 7] * 8.0
    q[:, 9] *= 10.0
    out = ops.flash_attn_d512(q, k, _vt(v, 1024), T, splits=4)
    _check(out, _ref(q, k, v), 4e-3, "d512 split extreme")


def test_flash_attn_d512_cross_lengths_and_strides():
    """Tq != Tk, operands that are column slices of wider buffers (row stride > 512), V^T padded wider than needed."""
    B, Tq, Tk = 2, 300, 1000
    wide_q, wide_k = rnd(B, Tq, 1024).to(BF), rnd(B, Tk, 768, seed=1).to(BF)
    q, k = wide_q[:, :, 512:], wide_k[:, :, :512]
    v = rnd(B, Tk, 512, seed=2).to(BF)
    lib_out = torch.zeros(B, Tq, 640, dtype=BF, device=DEV)
    from supir_amd import _lib
    lib = _lib.load()
    vt = _vt(v, 1088)
    rc = lib.supir_flash_attn_d512(q.data_ptr(), k.data_ptr(), vt.data_ptr(), lib_out.data_ptr(), B, Tq, Tk, 1024, 768, 1088, 640,
                                   512 ** -0.5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    _check(lib_out[:, :, :512], _ref(q, k, v), 4e-3, "d512 strided")
    assert lib_out[:, :, 512:].abs().max().item() == 0      # nothing written beyond the 512 output channels


def test_flash_attn_d512_extreme_logits():
    """Row maxima far above the first key tile's and a huge dynamic range: the exact first-pass maximum keeps P <= 1."""
    B, T = 1, 512
    q, k, v = rnd(B, T, 512).to(BF), rnd(B, T, 512, seed=1).to(BF), rnd(B, T, 512, seed=2).to(BF)
    k[:, 400] = q[:, 7] * 8.0          # one late key that dominates query 7 by a wide margin
    q[:, 9] *= 10.0                    # one query with logits of +-25 nats: a near one-hot softmax
    out = ops.flash_attn_d512(q, k, _vt(v, 512), T)
    _check(out, _ref(q, k, v), 4e-3, "d512 extreme")


def test_vae_attnblock_flash_matches_materialised(monkeypatch):
    from tests.helpers import build_vae
    vae = build_vae(DEV)
    att = vae.decoder.mid.attn_1
    for shape in ((1, 512, 64, 64), (2, 512, 9, 7), (2, 512, 43, 43)):
        x = rnd(*shape)
        with torch.no_grad():
            monkeypatch.setattr(ops, "USE_FLASH_D512", False)
            a = att(x).float()
            monkeypatch.setattr(ops, "USE_FLASH_D512", True)
            ops.start_trace()
            b = att(x).float()
            names = [r["kernel"] for r in ops.stop_trace()]
        assert "attn_d512" in names and "softmax" not in names
        e = ((a - b).norm() / a.norm()).item()
        print(f"AttnBlock {shape}: flash vs materialised rel-L2 {e:.3e}")
        assert e <= 6e-3, e


def test_flash_attn_d512_timing_report():
    """Not a pass / fail test of speed: prints the launch time at the production sizes (T = 16 384: 1024^2 px, T = 4096: 512^2 px)
    next to the materialised path's and leaves both in gpurun_out/attn_d512_timing.json (what ops.FLASH_D512_MIN_TOKENS is set from)."""
    def timed(fn, n=5):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    res = {}
    for B, T in ((1, 16384), (1, 4096), (4, 4096)):
        q, k, v = rnd(B, T, 512).to(BF), rnd(B, T, 512, seed=1).to(BF), rnd(B, T, 512, seed=2).to(BF)
        vt = _vt(v, T)
        t_single = timed(lambda: ops.flash_attn_d512(q, k, vt, T, splits=1))
        t_flash = timed(lambda: ops.flash_attn_d512(q, k, vt, T))          # the library's split choice

        def materialised():
            for b in range(B):
                s = ops.gemm(q[b], k[b], out_dtype=torch.float32)
                p = ops.softmax_rows(s, 512 ** -0.5, valid=T)
                ops.gemm(p, vt[b])

        t_mat = timed(materialised)
        fl = 4.0 * B * T * T * 512
        print(f"[d512] B={B} T={T}: flash {t_flash:.0f} us = {fl / t_flash / 1e6:.0f} TFLOP/s (one pass per workgroup: {t_single:.0f} us); "
              f"materialised scores {t_mat:.0f} us")
        res[f"B{B}_T{T}"] = {"flash_us": t_flash, "flash_tflops": fl / t_flash / 1e6, "flash_single_pass_us": t_single, "materialised_us": t_mat}
    try:
        import json
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(res, open(os.path.join(out, "attn_d512_timing.json"), "w"), indent=1)
    except OSError:
        pass
