"""Per-phase s_memtime breakdown of the gemm16 kernels (prologue incl. row-statistics fetch / main loop / K-group exchange /
epilogue), per wave, averaged.  Builds a second copy of the library with -DSUPIR_G16_TIMELINE (the product build never defines
it) and calls it directly through ctypes.   python tools/probes/g16_timeline.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

CSRC = os.path.join(ROOT, "supir_amd", "csrc")
# built in-tree (git-ignored, travels with the gpurun snapshot: hipcc cross-compiles in the build container, so no GPU-box time is
# spent compiling); `--build-only` stops after the build
OUT = os.path.join(ROOT, "supir_amd", "libsupir_hip_tl.so")
srcs = [os.path.join(CSRC, f) for f in ("gemm.hip", "gemm16.hip", "gemm_big.hip", "attention.hip", "xattn.hip", "attention_d512.hip", "norm.hip", "edge.hip",
                                        "sampler.hip", "api.hip")]
if not os.path.exists(OUT) or any(os.path.getmtime(f) > os.path.getmtime(OUT) for f in srcs):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-DSUPIR_G16_TIMELINE"] + srcs + ["-o", OUT]
    subprocess.check_call(cmd)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(OUT)
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
lib.supir_gemm_bf16.argtypes = [P, P, P, I, I, I, I, I, P, P, I, I, P, I, I, I, F, I, P]
lib.supir_g16_tl_set.argtypes = [P]
lib.supir_big_tl_set.argtypes = [P]
BF = torch.bfloat16
CASES = [(2048, 1280, 1280, 35), (2048, 1280, 1280, 32), (2048, 1280, 5120, 35), (2048, 2560, 1280, 33), (2048, 10240, 1280, 34)]
if len(sys.argv) > 1 and sys.argv[1] == "large":   # round 5: the large-M regime (tile batches / --num_samples, VAE-sized GEMMs)
    CASES = [(8192, 1280, 1280, 34), (8192, 1280, 5120, 34), (65536, 512, 4608, 40),
             (65536, 512, 4608, 42), (262144, 256, 2304, 42), (1048576, 128, 1152, 39), (1048576, 128, 1152, 45), (262144, 256, 2304, 45)]
if len(sys.argv) > 1 and sys.argv[1] == "geglu":   # the GEGLU projection on the 256 x 160 and the 256 x 320 tile, several K and M
    CASES = [(2048, 10240, 1280, 34), (2048, 10240, 1280, 37), (2048, 10240, 640, 37), (2048, 10240, 2560, 37), (4096, 10240, 1280, 37),
             (8192, 5120, 640, 37)]
if len(sys.argv) > 1 and sys.argv[1] == "round6":
    # round 6: where a K step of the 128 x 80 / 128 x 160 loops goes -- counted wait / barrier / reads + loads + MFMAs -- on convolutions in
    # the implicit-GEMM form (35, 33) and the halo form (48, 49), and on the transformer GEMMs.  (profiles/r06/g16_timeline_loop_forms.log is
    # the run that also held the ping-pong / software-pipelined loops, since removed.)
    lib.supir_conv3x3_bf16.argtypes = [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P, P, I, P, I, I, I, F, I, P]
    for (B, H, W, Cin, Cout, tile) in [(2, 32, 32, 1280, 1280, 35), (2, 32, 32, 1280, 1280, 48), (2, 64, 64, 640, 640, 33), (2, 64, 64, 640, 640, 49)]:
        x = torch.randn(B, H, W, Cin, device="cuda").to(BF)
        w = (torch.randn(Cout, 3, 3, Cin, device="cuda") * (9 * Cin) ** -0.5).to(BF)
        bias = torch.randn(Cout, device="cuda")
        out = torch.empty(B, H, W, Cout, device="cuda", dtype=BF)
        bm, bn = {33: (128, 160), 49: (128, 160)}.get(tile, (128, 80))
        nwg = (B * H * W // bm) * (Cout // bn)
        buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream

        def runc():
            return lib.supir_conv3x3_bf16(x.data_ptr(), w.data_ptr(), out.data_ptr(), B, H, W, Cin, Cin, Cout, Cout, H, W, 1, 1, 1, 0, bias.data_ptr(),
                                          None, 0, None, 0, 0, 0, 1.0, tile, st)
        for _ in range(3):
            assert runc() == 0
        torch.cuda.synchronize()
        lib.supir_g16_tl_set(buf.data_ptr())
        assert runc() == 0
        torch.cuda.synchronize()
        lib.supir_g16_tl_set(None)
        t = buf.view(nwg * 8, 16).cpu().double()
        nk = (9 * Cin // 64) // 2
        print(f"conv B={B} {H}x{W} {Cin}->{Cout} tile={tile} wgs={nwg} | per wave (cycles): prologue {t[:, 1].mean():.0f}  loop {t[:, 2].mean():.0f} "
              f"({t[:, 2].mean() / nk:.0f} per iteration x {nk}: wait {t[:, 8].mean() / nk:.0f}, barrier {t[:, 9].mean() / nk:.0f}, reads+loads+MFMA "
              f"{t[:, 10].mean() / nk:.0f}; ideal MFMA {bm * bn * 64 * 2 * 2 / 4096:.0f})  exchange {t[:, 3].mean():.0f}  epilogue {t[:, 4].mean():.0f}  "
              f"total {t[:, 5].mean():.0f} (max {t[:, 5].max():.0f}); first start -> last end {t[:, 6].max() - t[:, 0].min():.0f}", flush=True)
    CASES = [(2048, 1280, 1280, 35), (2048, 1280, 5120, 35), (2048, 2560, 1280, 33), (2048, 10240, 1280, 34)]
if len(sys.argv) > 1 and sys.argv[1] == "qkv":
    # round 6 (late): the fused q | k | v projection (256 x 128 tile, normal + transposed epilogue) at the step's shape
    lib.supir_gemm_bf16_qkv.argtypes = [P, P, P, P, I, I, I, I, I, I, I, I, P, P, I, I, P, F, P]
    for (M, N, K, T) in [(2048, 3840, 1280, 1024), (8192, 1920, 640, 4096)]:
        a = torch.randn(M, K, device="cuda").to(BF)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
        ns = 2 * N // 3
        cqk = torch.empty(M, ns, device="cuda", dtype=BF)
        cvt = torch.empty(M // T, N - ns, T, device="cuda", dtype=BF)
        nwg = (M // 256) * (N // 128)
        buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream

        def runq():
            return lib.supir_gemm_bf16_qkv(a.data_ptr(), w.data_ptr(), cqk.data_ptr(), cvt.data_ptr(), M, N, ns, K, K, ns, T, T, None, None, 0, 0, None, 1e-5, st)
        for _ in range(5):
            assert runq() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            runq()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        lib.supir_g16_tl_set(buf.data_ptr())
        assert runq() == 0
        torch.cuda.synchronize()
        lib.supir_g16_tl_set(None)
        t = buf.view(nwg * 8, 16).cpu().double()
        t = t[t[:, 5] > 0]          # the workgroups of the transposed (V^T) third return through another epilogue that is not stamped
        nk = K // 64
        print(f"qkv M={M} N={N} K={K} wgs={nwg} | event {us:.1f} us (untimed build) | per wave (cycles): prologue {t[:, 1].mean():.0f}  loop {t[:, 2].mean():.0f} "
              f"({t[:, 2].mean() / nk:.0f} per K step x {nk}: wait {t[:, 8].mean() / nk:.0f}, barrier {t[:, 9].mean() / nk:.0f}, reads+loads+MFMA "
              f"{t[:, 10].mean() / nk:.0f}; ideal MFMA 1024)  epilogue {t[:, 4].mean():.0f}  total {t[:, 5].mean():.0f} (max {t[:, 5].max():.0f}); "
              f"first start -> last end {t[:, 6].max() - t[:, 0].min():.0f}", flush=True)
    sys.exit(0)
for (M, N, K, tile) in CASES:
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    res = torch.randn(M, N, device="cuda").to(BF)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=BF)
    geglu = len(sys.argv) > 1 and sys.argv[1] == "geglu"
    bm, bn = {32: (128, 80), 33: (128, 160), 34: (256, 160), 35: (128, 80), 37: (256, 320), 39: (256, 128), 40: (256, 256), 42: (256, 256), 45: (512, 128),
              }[tile]
    nwg = (M // bm) * (N // bn)
    SL = 8 if tile == 37 else 16      # slots per wave (gemm16: + the in-loop split of round 6)
    buf = torch.zeros(nwg * 8 * SL, dtype=torch.int64, device="cuda")
    (lib.supir_big_tl_set if tile == 37 else lib.supir_g16_tl_set)(buf.data_ptr())
    st = torch.cuda.current_stream().cuda_stream

    def run():
        if geglu:
            return lib.supir_gemm_bf16(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, K, N // 2, bias.data_ptr(), None, 0, 0,
                                       None, 0, 2, 0, 1.0, tile, st)
        return lib.supir_gemm_bf16(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, K, N, bias.data_ptr(), None, 0, 0,
                                   res.data_ptr(), N, 0, 0, 1.0, tile, st)
    for _ in range(5):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    t = buf.view(nwg * 8, SL).cpu().double()
    nk = (K // 64) // (1 if tile in (34, 37, 39, 40, 42, 45) else 2)
    if SL == 16:
        print(f"    in-loop split per K step: counted wait {t[:, 8].mean() / nk:.0f}, barrier {t[:, 9].mean() / nk:.0f}, reads + loads + MFMA {t[:, 10].mean() / nk:.0f}", flush=True)
    print(f"M={M} N={N} K={K} tile={tile} wgs={nwg} | event {us:.1f} us | per wave (cycles): prologue "
          f"{t[:, 1].mean():.0f}  loop {t[:, 2].mean():.0f} ({t[:, 2].mean() / nk:.0f} per K step x {nk})  exchange {t[:, 3].mean():.0f}  "
          f"epilogue {t[:, 4].mean():.0f}  total {t[:, 5].mean():.0f} (max {t[:, 5].max():.0f})", flush=True)
    (lib.supir_big_tl_set if tile == 37 else lib.supir_g16_tl_set)(None)
    t0 = t[:, 0]
    print(f"    start skew over waves: {t0.max() - t0.min():.0f} ticks; end skew {t[:, 6].max() - t[:, 6].min():.0f}; first start -> last end {t[:, 6].max() - t0.min():.0f}", flush=True)
