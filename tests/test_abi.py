"""The C-ABI shared library loads (no GPU needed for dlopen) and exports every symbol include/supir_hip.h declares, with
the argument counts the ctypes binding uses."""
import os
import re

from supir_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(name="supir_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    fns = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(supir_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        fns[m.group(1)] = 0 if args == "void" else len([a for a in args.split(",") if a.strip()])
    return fns


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    fns = _header_functions()
    assert len(fns) >= 14
    for name in fns:
        assert hasattr(lib, name), f"{name} declared in supir_hip.h but not exported"
    assert lib.supir_abi_version() == _lib.ABI_VERSION == 2
    assert lib.supir_target_arch() == b"gfx950"
    assert lib.supir_elem_type() == b"bf16"


def test_f16_library_exports_the_same_surface():
    """libsupir_hip_f16.so (same sources, -DSUPIR_F16) loads next to the bf16 library, exports every declared symbol, reports its
    element type and validates arguments the same way; the two handles are distinct objects with their own state."""
    import torch
    lib16 = _lib.load(torch.float16)
    lib = _lib.load()
    assert lib16 is not lib and _lib.load(torch.bfloat16) is lib and _lib.load(torch.float16) is lib16
    for name in _header_functions():
        assert hasattr(lib16, name), f"{name} not exported by the f16 build"
    assert lib16.supir_abi_version() == _lib.ABI_VERSION and lib16.supir_target_arch() == b"gfx950" and lib16.supir_elem_type() == b"f16"
    assert lib16.supir_gemm_bf16(None, None, None, 64, 64, 64, 64, 64, None, None, 0, 0, None, 0, 0, 0, 1.0, -1, None) == -1
    fake = 0x10000
    assert lib16.supir_gemm_bf16(fake, fake, fake, 200, 80, 128, 128, 80, None, None, 0, 0, None, 0, 0, 0, 1.0, 32, None) == -2


def test_f32_library_exports_its_own_table():
    """libsupir_hip_f32.so (csrc/f32, include/supir_hip_f32.h): loads next to the other two, exports every symbol its header declares with
    the argument counts of the ctypes mirror, the struct mirror has the C struct's size, and arguments are validated before any launch."""
    import ctypes
    import torch
    lib32 = _lib.load(torch.float32)
    assert lib32 is _lib.load_f32() and lib32 is not _lib.load() and lib32 in _lib.loaded()
    fns = _header_functions("supir_hip_f32.h")
    assert set(fns) == set(_lib.SIGNATURES_F32) | {"supir_abi_version", "supir_target_arch", "supir_elem_type", "supir_last_hip_error",
                                                   "supir_hip_error_string"}
    for name, n in fns.items():
        assert hasattr(lib32, name), f"{name} declared in supir_hip_f32.h but not exported"
        if name in _lib.SIGNATURES_F32:
            assert len(_lib.SIGNATURES_F32[name]) == n, (name, n)
    assert lib32.supir_abi_version() == _lib.ABI_VERSION and lib32.supir_target_arch() == b"gfx950" and lib32.supir_elem_type() == b"f32"
    assert not hasattr(lib32, "supir_gemm_bf16")              # the 16-bit table is not here: nothing can reach it with fp32 buffers
    # struct layout: 6 pointers, 10 ints, 2 ints + float + 2 ints, 6 longs, 10 ints  (natural alignment)
    assert ctypes.sizeof(_lib.F32GemmDesc) == 6 * 8 + 10 * 4 + 5 * 4 + 4 + 6 * 8 + 10 * 4
    fake = 0x10000
    d = _lib.F32GemmDesc(A=fake, W=fake, C=fake, kind=0, M=64, N=64, K=64, lda=64, ldw=64, ldc=64, nz0=1, nz1=1, alpha=1.0)
    d.K = 0
    assert lib32.supir_f32_gemm(ctypes.byref(d), None) == -1
    d.K, d.lda = 64, 32
    assert lib32.supir_f32_gemm(ctypes.byref(d), None) == -2            # leading dimension below K
    d.lda, d.act = 64, 2
    assert lib32.supir_f32_gemm(ctypes.byref(d), None) == -1            # GEGLU is supir_f32_geglu, not an epilogue
    d.act, d.nz0, d.bias = 0, 2, fake
    assert lib32.supir_f32_gemm(ctypes.byref(d), None) == -1            # batched launches take no bias / residual
    assert lib32.supir_f32_gemm(None, None) == -1
    assert lib32.supir_f32_geglu(fake, fake, 8, 130, 130, 65, 32, None) == -2
    assert lib32.supir_f32_softmax_rows(fake, fake, 4, 100, 64, 128, 128, 1.0, 0, None) == -1
    assert lib32.supir_f32_softmax_rows(fake, fake, 10, 64, 64, 64, 64, 1.0, 4, None) == -2    # causal: whole blocks of queries
    assert lib32.supir_f32_layernorm(None, None, None, None, 4, 64, 64, 64, 1e-5, None) == -1
    assert lib32.supir_f32_groupnorm(fake, None, None, None, 1, 16, 48, 48, 48, 0, fake, fake, 1e-5, 0, None, None, 0, 1.0, fake, 48, fake, 1 << 20,
                                     None, None) == -2                    # C % 32
    assert lib32.supir_f32_groupnorm(fake, None, None, None, 1, 16, 64, 64, 64, 0, fake, fake, 1e-5, 0, None, None, 0, 1.0, fake, 64, fake, 64,
                                     None, None) == -1                    # workspace too small
    assert lib32.supir_f32_groupnorm_stats(fake, 1, 16, 64, 32, fake, fake, 1 << 20, None) == -2     # leading dimension below C


def test_ctypes_signatures_match_header():
    fns = _header_functions()
    for name, argtypes in list(_lib.SIGNATURES.items()) + list(_lib.SIZE_SIGNATURES.items()):
        assert name in fns, name
        assert len(argtypes) == fns[name], (name, len(argtypes), fns[name])
    assert set(fns) - set(_lib.SIGNATURES) - set(_lib.SIZE_SIGNATURES) == {"supir_abi_version", "supir_target_arch", "supir_elem_type",
                                                 "supir_last_hip_error", "supir_hip_error_string"}


def test_bad_arguments_return_error_codes_without_a_gpu():
    lib = _lib.load()
    # null pointers are rejected before any launch
    rc = lib.supir_gemm_bf16(None, None, None, 64, 64, 64, 64, 64, None, None, 0, 0, None, 0, 0, 0, 1.0, -1, None)
    assert rc == -1
    rc = lib.supir_layernorm(None, None, None, None, 4, 64, 64, 64, 1e-5, None)
    assert rc == -1


def test_new_entry_points_validate_arguments_without_a_gpu():
    """supir_launch_hints / supir_wavelet_level: argument errors are reported before any HIP call; the thread-local one-shot
    setters of ABI 1 are gone (ABI 2: the library keeps no request state)."""
    import ctypes
    lib = _lib.load()
    assert lib.supir_abi_version() == 2
    for gone in ("supir_set_next_prefetch", "supir_set_next_gn_partials"):
        assert not hasattr(lib, gone), gone
    fake = 0x10000
    bad = _lib.LaunchHints(next_weight=None, next_weight_bytes=4096, gn_partials_out=None)      # bytes without a pointer
    assert lib.supir_gemm_bf16_ex(fake, fake, fake, 64, 64, 64, 64, 64, None, None, 0, 0, None, 0, 0, 0, 1.0, -1, ctypes.byref(bad), None) == -1
    assert lib.supir_wavelet_level(None, None, None, 3, 8, 8, 1, 1, None) == -1
    fake = 0x10000                                                  # never dereferenced: validation comes first
    assert lib.supir_wavelet_level(fake, fake, fake + 4096, 3, 8, 8, 1, 1, None) == -1      # img aliases low
    assert lib.supir_wavelet_level(fake, fake + 4096, fake + 8192, 3, 8, 8, 0, 1, None) == -2   # radius 0: shape error
    assert lib.supir_wavelet_level(fake, fake + 4096, fake + 8192, 70000, 8, 8, 1, 1, None) == -2   # planes > grid limit


def test_round2_entry_points_validate_arguments_without_a_gpu():
    """supir_gemm_bf16_qkv / supir_flash_attn_d64_ex / supir_resample_u8 / supir_bicubic_f32 and the gemm16 tile indices: argument
    and shape errors are reported before any HIP call (null pointers, inexact shapes, bad flags)."""
    lib = _lib.load()
    fake = 0x10000
    assert lib.supir_gemm_bf16_qkv(None, None, None, None, 2048, 3840, 2560, 1280, 1280, 2560, 1024, 1024, None, None, 0, 0, None, 1e-5, None) == -1
    # M not a multiple of 256 -> SUPIR_ERR_SHAPE (the caller then issues the two separate projections)
    assert lib.supir_gemm_bf16_qkv(fake, fake, fake, fake, 2000, 3840, 2560, 1280, 1280, 2560, 1000, 1000, None, None, 0, 0, None, 1e-5, None) == -2
    assert lib.supir_flash_attn_d64_ex(None, None, None, None, 1, 1, 64, 64, 64, 64, 64, 64, 0.125, 1, None) == -1
    assert lib.supir_flash_attn_d64_ex(fake, fake, fake, fake, 1, 1, 64, 128, 64, 64, 128, 64, 0.125, 1, None) == -2   # causal needs Tq == Tk
    assert lib.supir_flash_attn_d64_ex(fake, fake, fake, fake, 1, 1, 64, 64, 64, 64, 64, 64, 0.125, 6, None) == -1    # unknown flag bits
    assert lib.supir_resample_u8(None, None, None, None, None, None, 5, 8, 8, 8, 16, 3, 0, None) == -1
    assert lib.supir_resample_u8(fake, None, None, None, fake, fake, 5, 8, 8, 8, 16, 3, 0, None) == -1                  # no output at all
    assert lib.supir_resample_u8(fake, fake, None, None, fake, fake, 5, 8, 8, 9, 16, 3, 0, None) == -2                  # horizontal pass keeps H
    assert lib.supir_bicubic_f32(None, None, None, 3, 8, 8, 4, 4, None) == -1
    # supir_flash_attn_d512: null pointers, row strides below the head dim / misaligned, V^T narrower than the key count
    assert lib.supir_flash_attn_d512(None, None, None, None, 1, 64, 64, 512, 512, 64, 512, 0.044, None) == -1
    assert lib.supir_flash_attn_d512(fake, fake, fake, fake, 0, 64, 64, 512, 512, 64, 512, 0.044, None) == -1
    assert lib.supir_flash_attn_d512(fake, fake, fake, fake, 1, 64, 64, 256, 512, 64, 512, 0.044, None) == -2
    assert lib.supir_flash_attn_d512(fake, fake, fake, fake, 1, 64, 64, 512, 516, 64, 512, 0.044, None) == -2
    assert lib.supir_flash_attn_d512(fake, fake, fake, fake, 1, 64, 100, 512, 512, 96, 512, 0.044, None) == -2
    # key-split form: the workspace the library asks for (auto: 2 splits at T = 16 384, 8 at 4096, none for 256 query blocks or < 16 key
    # tiles; explicit counts clamp to 16 and to the key tiles, and never leave a split empty), and its argument checks
    ws = lib.supir_flash_attn_d512_workspace
    assert ws(1, 16384, 16384, 0) == 2 * 16384 * 514 * 4 and ws(1, 4096, 4096, 0) == 8 * 4096 * 514 * 4
    assert ws(2, 16384, 16384, 0) == 0 and ws(1, 200, 200, 0) == 0 and ws(1, 16384, 16384, 1) == 0 and ws(0, 64, 64, 0) == 0
    assert ws(1, 128, 16384, 64) == 16 * 128 * 514 * 4 and ws(1, 128, 96, 8) == 3 * 128 * 514 * 4
    assert ws(1, 128, 9 * 32, 4) == 3 * 128 * 514 * 4          # 9 tiles / 4 -> 3 per split -> 3 splits, none empty
    # the library's choice follows rounds-of-256-workgroups x tiles per split (profiles/r04/micro_attn_d512_key_split_sweep.log): eight
    # stacked 5184-token tiled-VAE tiles are 328 workgroups = 2 rounds unsplit, 4 rounds of a third with 3 splits; two full rounds stay whole
    per = 514 * 4
    assert ws(8, 5184, 5184, 0) == 3 * 8 * 5184 * per and ws(16, 4096, 4096, 0) == 0 and ws(8, 7396, 7396, 0) == 0
    assert ws(1, 5184, 5184, 0) == 6 * 5184 * per and ws(1, 7396, 7396, 0) == 4 * 7396 * per
    assert ws(1, 1024, 1024, 0) == 8 * 1024 * per and ws(16, 1024, 1024, 0) == 2 * 16 * 1024 * per and ws(1, 480, 480, 0) == 0
    split = lib.supir_flash_attn_d512_split
    assert split(None, None, None, None, 1, 64, 64, 512, 512, 64, 512, 0.044, 0, None, 0, None) == -1
    assert split(fake, fake, fake, fake, 1, 4096, 4096, 512, 512, 4096, 512, 0.044, 0, None, 0, None) == -1          # needs a workspace
    assert split(fake, fake, fake, fake, 1, 4096, 4096, 512, 512, 4096, 512, 0.044, 0, fake, 1024, None) == -1       # too small
    assert split(fake, fake, fake, fake, 1, 4096, 4096, 512, 512, 4096, 512, 0.044, 0, fake + 4, 1 << 30, None) == -1   # misaligned
    assert split(fake, fake, fake, fake, 1, 4096, 4096, 256, 512, 4096, 512, 0.044, 0, fake, 1 << 30, None) == -2     # ldq < 512
    # tile 32 (128 x 80) on an inexact shape is refused, not silently run on another tile
    assert lib.supir_gemm_bf16(fake, fake, fake, 200, 80, 128, 128, 80, None, None, 0, 0, None, 0, 0, 0, 1.0, 32, None) == -2
    assert lib.supir_gemm_bf16(fake, fake, fake, 128, 80, 128, 128, 80, None, None, 0, 0, None, 0, 0, 0, 1.0, 36, None) == -1   # no such tile


def test_grouped_launch_structs_match_the_header_layout(tmp_path):
    """The host-side structs of the grouped-launch entry points (supir_gemm_problem / supir_gemm_shape / supir_attn_problem /
    supir_gn_problem): the ctypes mirrors in supir_amd/_lib.py against what a C compiler makes of include/supir_hip.h (size and the
    offset of every field), and argument validation without a GPU."""
    import ctypes
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no C compiler")
    pairs = {"supir_gemm_problem": _lib.GemmProblem, "supir_gemm_shape": _lib.GemmShape, "supir_attn_problem": _lib.AttnProblem,
             "supir_gn_problem": _lib.GnProblem, "supir_launch_hints": _lib.LaunchHints}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "supir_hip.h")}"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-DSUPIR_EXPERIMENTAL", str(src), "-o", str(exe)], check=True)   # also: the header is plain C
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)
    lib = _lib.load()
    sh = _lib.GemmShape(kind=0, tile=35, M=2048, N=1280, K=1280, alpha=1.0)
    pr = (_lib.GemmProblem * 2)()
    assert lib.supir_gemm_grouped(None, pr, 2, None) == -1
    assert lib.supir_gemm_grouped(ctypes.byref(sh), pr, 3, None) == -1            # at most two problems
    assert lib.supir_gemm_grouped(ctypes.byref(sh), pr, 2, None) == -1            # null operands
    fake = 0x10000
    for q in pr:
        q.A, q.W, q.C, q.lda, q.ldc = fake, fake, fake, 1280, 1280
    sh.tile = 32
    assert lib.supir_gemm_grouped(ctypes.byref(sh), pr, 2, None) == -2            # tile 32 has no two-problem form
    sh.tile, sh.M = 35, 2000
    assert lib.supir_gemm_grouped(ctypes.byref(sh), pr, 2, None) == -2            # inexact shape
    assert lib.supir_flash_attn_d64_grouped(None, 2, 2, 20, 1024, 0.125, None) == -1
    assert lib.supir_groupnorm_grouped(None, 2, 2, 1024, 1280, 1e-5, 1, None) == -1


def test_hinted_entry_points_take_their_requests_as_an_argument():
    """Round 4: supir_*_ex carry the next-weight prefetch / GroupNorm-partials requests in a supir_launch_hints argument (the
    thread-local one-shot setters are deprecated shims).  Validation without a GPU: NULL hints are fine, a byte count without a pointer
    is an argument error, the fused q|k|v launch refuses a GroupNorm-partials request, and a legacy call after an _ex call does not
    inherit anything (no state left behind)."""
    import ctypes
    lib = _lib.load()
    fake = 0x10000
    g = (fake, fake, fake, 200, 80, 128, 128, 80, None, None, 0, 0, None, 0, 0, 0, 1.0, 32)
    assert lib.supir_gemm_bf16_ex(*g, None, None) == -2                                   # inexact shape for tile 32: reached the dispatcher
    bad = _lib.LaunchHints(next_weight=None, next_weight_bytes=4096, gn_partials_out=None)
    assert lib.supir_gemm_bf16_ex(*g, ctypes.byref(bad), None) == -1                      # bytes without a pointer
    assert lib.supir_gemm_bf16_ex(None, None, None, 64, 64, 64, 64, 64, None, None, 0, 0, None, 0, 0, 0, 1.0, -1, None, None) == -1
    ok = _lib.LaunchHints(next_weight=fake, next_weight_bytes=4096, gn_partials_out=fake)
    q = (fake, fake, fake, fake, 2048, 3840, 2560, 1280, 1280, 2560, 1024, 1024, None, None, 0, 0, None, 1e-5)
    assert lib.supir_gemm_bf16_qkv_ex(*q, ctypes.byref(ok), None) == -1                   # q|k|v emits no GroupNorm partials
    c = (fake, fake, fake, 2, 8, 8, 64, 64, 64, 64, 8, 8, 3, 1, 1, 0, None, None, 0, None, 0, 0, 0, 1.0, -1)
    assert lib.supir_conv3x3_bf16_ex(*c, ctypes.byref(ok), None) == -2                    # stride 3: shape error, after the hints were read
    ln = (fake, fake, fake, 200, 80, 128, 128, 80, None, None, 0, 0, 0, 0, 1.0, 32, None, 0, None, 0, 0, None, 1e-5)
    assert lib.supir_gemm_bf16_ln_ex(*ln, ctypes.byref(bad), None) == -1
    assert lib.supir_gemm_bf16_ln_ex(*ln, None, None) == -2


def test_arithmetic_choices_are_arguments_and_variant_switches_are_tools_only():
    """VERDICT r05 item 6.  The GELU of a GEGLU epilogue is chosen by the activation CODE of each launch (SUPIR_ACT_GEGLU = fitted,
    SUPIR_ACT_GEGLU_ERF = the reference's erf, sgm/modules/attention.py:89-91), both declared in the header and validated by every GEMM entry
    point; the process-global kernel-variant switches (`supir_debug_knob`) exist only in libsupir_hip_tools.so -- `nm -D` of the three product
    libraries shows no such symbol -- so the header's "only mutable state is the last-error code" is true of what ships."""
    import subprocess
    src = open(os.path.join(ROOT, "include", "supir_hip.h")).read()
    codes = dict(re.findall(r"#define (SUPIR_ACT_\w+) (\d+)", src))
    assert codes["SUPIR_ACT_GEGLU"] == "2" and codes["SUPIR_ACT_GEGLU_ERF"] == "5" and len(set(codes.values())) == len(codes) == 6
    assert "supir_debug_knob" not in _header_functions()
    for path in (_lib.LIB_PATH, _lib.LIB_PATH_F16, _lib.LIB_PATH_F32):
        syms = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
        assert "supir_abi_version" in syms and "debug_knob" not in syms, path
    lib = _lib.load()
    assert not hasattr(lib, "supir_debug_knob")
    tools = _lib.load_tools()
    assert tools is not lib and tools.supir_debug_knob(0, 0) == 0 and tools.supir_debug_knob(8, 1) == -1
    for name in _header_functions():
        assert hasattr(tools, name), name
    fake = 0x10000
    # code 5 passes validation and reaches the tile's shape check (tile 34 does not fit M = 200); code 6 is an argument error
    assert lib.supir_gemm_bf16(fake, fake, fake, 200, 320, 128, 128, 160, None, None, 0, 0, None, 0, 5, 0, 1.0, 34, None) == -2
    assert lib.supir_gemm_bf16(fake, fake, fake, 200, 320, 128, 128, 160, None, None, 0, 0, None, 0, 6, 0, 1.0, 34, None) == -1
    assert lib.supir_gemm_bf16_ln(fake, fake, fake, 256, 320, 128, 128, 160, None, None, 0, 6, 0, 256, 1.0, 34, None, 0, None, 0, 0, None, 1e-5, None) == -1
    assert lib.supir_gemm_tile_for(2048, 10240, 5) == lib.supir_gemm_tile_for(2048, 10240, 2)
    # the tools context hands bf16 operands to the tools build and gives the product library back
    with _lib.tools_knob(6, 1) as t:
        assert _lib.load() is t is tools
    assert _lib.load() is lib


def test_exact_gelu_switch_of_the_host_mirror(monkeypatch):
    from supir_amd import ops
    monkeypatch.setattr(ops, "EXACT_GELU", False)
    assert [ops._abi_act(a) for a in range(5)] == [0, 1, 2, 3, 4]
    monkeypatch.setattr(ops, "EXACT_GELU", True)
    assert [ops._abi_act(a) for a in range(5)] == [0, 1, 5, 3, 4]
