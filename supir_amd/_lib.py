"""ctypes binding of libsupir_hip.so (C ABI declared in include/supir_hip.h).

There is no fallback: if the shared library is missing or an entry point returns an error code the call raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_void_p

# torch bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  It MUST be in the process before our library is
# dlopen'ed, otherwise the loader resolves our DT_NEEDED to the system runtime and the process ends up with kernels
# registered in one HIP runtime and streams created by another (every launch then fails).
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsupir_hip.so")
# the same sources built with -DSUPIR_F16: every 16-bit buffer of the ABI holds IEEE binary16, MFMA operands are fp16 (csrc/common.h)
LIB_PATH_F16 = os.path.join(_HERE, "libsupir_hip_f16.so")
# the fp32 service (csrc/f32, include/supir_hip_f32.h): its own, much smaller, entry-point table
LIB_PATH_F32 = os.path.join(_HERE, "libsupir_hip_f32.so")
# the bf16 sources with -DSUPIR_TOOLS: the only build that contains `supir_debug_knob` (process-global kernel-VARIANT switches for A/B
# measurements and variant-vs-variant tests).  Never loaded by the product path; `tools_knob` swaps it in for the duration of a block.
LIB_PATH_TOOLS = os.path.join(_HERE, "libsupir_hip_tools.so")

_ERR = {-1: "SUPIR_ERR_ARG (null pointer / bad size)", -2: "SUPIR_ERR_SHAPE (unsupported shape or alignment)",
        -3: "SUPIR_ERR_HIP (launch failed)"}

P, I, F, L = c_void_p, c_int, c_float, c_long

# name -> argtypes; mirrors include/supir_hip.h one to one (tests/test_abi.py cross-checks against the header)
SIGNATURES = {
    "supir_gemm_bf16": [P, P, P, I, I, I, I, I, P, P, I, I, P, I, I, I, F, I, P],
    "supir_conv3x3_bf16": [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P, P, I, P, I, I, I, F, I, P],
    "supir_flash_attn_d64": [P, P, P, P, I, I, I, I, I, I, I, I, F, P],
    "supir_flash_attn_d64_ex": [P, P, P, P, I, I, I, I, I, I, I, I, F, I, P],
    "supir_flash_attn_d512": [P, P, P, P, I, I, I, I, I, I, I, F, P],
    "supir_softmax_rows": [P, P, I, I, I, L, L, F, P],
    "supir_groupnorm_nhwc": [P, P, P, P, I, I, I, I, I, I, P, P, F, I, P, P, I, F, P, I, P, c_size_t, P, P],
    "supir_groupnorm_stats": [P, P, I, I, I, I, I, I, P, P, c_size_t, P],
    "supir_groupnorm_parts_finalize": [P, I, I, I, I, I, P, P],
    "supir_layernorm": [P, P, P, P, I, I, I, I, F, P],
    "supir_conv3x3_smallcin": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "supir_conv3x3_smallcout": [P, P, P, P, I, I, I, I, I, I, P],
    "supir_pointwise_nchw": [P, P, P, P, I, I, I, L, F, P],
    "supir_wavelet_level": [P, P, P, I, I, I, I, I, P],
    "supir_resample_u8": [P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "supir_bicubic_f32": [P, P, P, I, I, I, I, I, P],
    "supir_edm_step_pre": [P, P, F, F, F, P, P, L, I, P],
    "supir_edm_step_post": [P, P, P, F, F, F, F, F, F, P, L, I, P],
    # x, eps, s_noise, noise_mul, c_in, x_hat, net_in, tile_hw (host int[2k]), k, b, C, Hc, Wc, T, reps, stream
    "supir_edm_step_pre_tiles": [P, P, F, F, F, P, P, P, I, I, I, I, I, I, I, P],
    # tiles, weights (fp64 [T][T]), canvas, tile_hw, k, b, C, Hc, Wc, T, stream
    "supir_tile_blend": [P, P, P, P, I, I, I, I, I, I, P],
    "supir_gemm_tile_for": [I, I, I],
    "supir_prefetch": [P, c_size_t, P, P],
    "supir_groupnorm_nhwc_parts": [P, P, P, P, I, I, I, I, I, I, P, P, F, I, P, P, I, F, P, I, P, I, P, I, P],
    "supir_rowstats_finalize": [P, P, I, I, I, I, F, P],
    "supir_gemm_bf16_qkv": [P, P, P, P, I, I, I, I, I, I, I, I, P, P, I, I, P, F, P],
    "supir_gemm_bf16_ln": [P, P, P, I, I, I, I, I, P, P, I, I, I, I, F, I, P, I, P, I, I, P, F, P],
}



# ---- grouped launches (include/supir_hip.h, "Grouped launches"): host-side structs, field for field
class GemmProblem(ctypes.Structure):
    _fields_ = [("A", P), ("W", P), ("C", P), ("C2", P), ("bias", P), ("rowbias", P), ("residual", P), ("rowstats_out", P),
                ("ln_stats", P), ("ln_colsum", P), ("gn_partials_out", P), ("prefetch", P), ("prefetch_bytes", c_size_t),
                ("lda", I), ("ldc", I), ("ldc2", I), ("ldr", I), ("ld_rowbias", I), ("rs_ld", I), ("ln_ld", I), ("ln_slots", I)]


class GemmShape(ctypes.Structure):
    _fields_ = [("kind", I), ("tile", I), ("M", I), ("N", I), ("K", I), ("rows_per_batch", I), ("act", I), ("out_mode", I),
                ("n_split", I), ("alpha", F), ("ln_eps", F), ("B", I), ("H", I), ("W", I), ("Cin", I), ("Cout", I), ("OH", I),
                ("OW", I), ("stride", I), ("pad_t", I), ("pad_l", I), ("upsample", I)]


class AttnProblem(ctypes.Structure):
    _fields_ = [("Q", P), ("K", P), ("Vt", P), ("O", P), ("Tk", I), ("ldq", I), ("ldk", I), ("ldvt", I), ("ldo", I), ("flags", I)]


class GnProblem(ctypes.Structure):
    _fields_ = [("x1", P), ("x2", P), ("x1raw", P), ("x2raw", P), ("gamma", P), ("beta", P), ("mod_g", P), ("mod_b", P), ("out", P),
                ("part1", P), ("part2", P), ("workspace", P), ("C1", I), ("ld1", I), ("ld2", I), ("ldm", I), ("ldo", I),
                ("nchunk1", I), ("nchunk2", I), ("control_scale", F)]


class LaunchHints(ctypes.Structure):
    """supir_launch_hints: optional per-launch requests of the *_ex GEMM-family entry points (next-weight prefetch, GroupNorm partials)."""
    _fields_ = [("next_weight", P), ("next_weight_bytes", c_size_t), ("gn_partials_out", P)]


GROUP_GEMM, GROUP_CONV3X3, GROUP_QKV = 0, 1, 2
SIGNATURES.update({
    "supir_gemm_bf16_ex": SIGNATURES["supir_gemm_bf16"][:-1] + [P, P],
    "supir_gemm_bf16_ln_ex": SIGNATURES["supir_gemm_bf16_ln"][:-1] + [P, P],
    "supir_gemm_bf16_qkv_ex": SIGNATURES["supir_gemm_bf16_qkv"][:-1] + [P, P],
    "supir_conv3x3_bf16_ex": SIGNATURES["supir_conv3x3_bf16"][:-1] + [P, P],
    # X, Wq, bias, K, Vt, O, B, H, T, Tk, C, ldx, ldk, ldvt, ldo, ln_stats, ln_ld, ln_slots, ln_colsum, ln_eps, scale, hints, stream
    "supir_xattn_q_d64": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P, I, I, P, F, F, P, P],
    "supir_gemm_grouped": [P, P, I, P],
    "supir_conv3x3_bf16_splitk": [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "supir_splitk_finalize": [P, I, I, I, P, I, P, I, P],
    "supir_flash_attn_d64_grouped": [P, I, I, I, I, F, P],
    # Q, K, Vt, O, B, Tq, Tk, ldq, ldk, ldvt, ldo, scale, splits, workspace, workspace_bytes, stream
    "supir_flash_attn_d512_split": [P, P, P, P, I, I, I, I, I, I, I, F, I, P, c_size_t, P],
    "supir_groupnorm_grouped": [P, I, I, I, I, F, I, P],
})

# entry points that return a byte count (size_t) instead of a status
SIZE_SIGNATURES = {"supir_flash_attn_d512_workspace": [I, I, I, I]}

class F32GemmDesc(ctypes.Structure):
    """supir_f32_gemm_desc (include/supir_hip_f32.h), field for field."""
    _fields_ = [("A", P), ("W", P), ("C", P), ("bias", P), ("rowbias", P), ("residual", P),
                ("kind", I), ("M", I), ("N", I), ("K", I),
                ("lda", I), ("ldw", I), ("ldc", I), ("ldr", I), ("ld_rowbias", I), ("rows_per_batch", I),
                ("act", I), ("out_mode", I), ("alpha", F), ("nz0", I), ("nz1", I),
                ("a_s0", L), ("a_s1", L), ("w_s0", L), ("w_s1", L), ("c_s0", L), ("c_s1", L),
                ("B", I), ("H", I), ("Wd", I), ("Cin", I), ("OH", I), ("OW", I), ("stride", I), ("pad_t", I), ("pad_l", I), ("upsample", I)]


F32_GEMM, F32_CONV3X3 = 0, 1
# libsupir_hip_f32.so: name -> argtypes; mirrors include/supir_hip_f32.h one to one (tests/test_abi.py cross-checks)
SIGNATURES_F32 = {
    "supir_f32_gemm": [P, P],
    "supir_f32_geglu": [P, P, I, I, I, I, I, P],
    "supir_f32_softmax_rows": [P, P, L, I, I, L, L, F, I, P],
    "supir_f32_groupnorm": [P, P, P, P, I, I, I, I, I, I, P, P, F, I, P, P, I, F, P, I, P, c_size_t, P, P],
    "supir_f32_groupnorm_stats": [P, I, I, I, I, P, P, c_size_t, P],
    "supir_f32_layernorm": [P, P, P, P, I, I, I, I, F, P],
}

ABI_VERSION = 2   # include/supir_hip.h: round 5 removed the thread-local one-shot setters, added the tiled-sampler edges
_lib = None       # the bf16 library (the product default)
_lib_f16 = None   # the fp16 build, loaded on first use
_lib_f32 = None   # the fp32 service, loaded on first use
_lib_tools = None  # the tools build (tests / tools only)


class SupirHipError(RuntimeError):
    pass


def _bind(path, want):
    lib = ctypes.CDLL(path)   # RTLD_LOCAL: the builds export the same names and must not see each other's symbols
    lib.supir_abi_version.restype = c_int
    lib.supir_abi_version.argtypes = []
    lib.supir_target_arch.restype = c_char_p
    lib.supir_target_arch.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, argtypes in SIZE_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_size_t
    lib.supir_last_hip_error.restype = c_int
    lib.supir_last_hip_error.argtypes = []
    lib.supir_hip_error_string.restype = c_char_p
    lib.supir_hip_error_string.argtypes = [c_int]
    lib.supir_elem_type.restype = c_char_p
    lib.supir_elem_type.argtypes = []
    if lib.supir_abi_version() != ABI_VERSION:
        raise SupirHipError(f"{os.path.basename(path)} ABI version mismatch")
    if lib.supir_elem_type() != want:
        raise SupirHipError(f"{os.path.basename(path)} was built for {lib.supir_elem_type()!r} elements, expected {want!r}")
    return lib


def load_tools():
    """libsupir_hip_tools.so: the bf16 sources + `supir_debug_knob(which, value)` (-DSUPIR_TOOLS).  Tests and tools only."""
    global _lib_tools
    if _lib_tools is None:
        if not os.path.exists(LIB_PATH_TOOLS):
            raise SupirHipError(f"{LIB_PATH_TOOLS} not found: run `python -m supir_amd.build`")
        _lib_tools = _bind(LIB_PATH_TOOLS, b"bf16")
        _lib_tools.supir_debug_knob.argtypes = [c_int, c_int]
        _lib_tools.supir_debug_knob.restype = c_int
    return _lib_tools


import contextlib  # noqa: E402


@contextlib.contextmanager
def tools_knob(which=None, value=0):
    """Inside the block, bf16 operands reach the TOOLS build (`load()` returns it) with variant switch `which` set to `value`; on exit the
    switch is cleared and the product library is back.  Yields the tools library (for direct entry-point calls / further knob calls)."""
    global _lib
    load()
    tools = load_tools()
    prev, _lib = _lib, tools
    try:
        if which is not None:
            check(tools.supir_debug_knob(which, value), "supir_debug_knob", tools)
        yield tools
    finally:
        for k in range(8):
            tools.supir_debug_knob(k, 0)
        _lib = prev


def load(dtype=None):
    """Load the library (building is __graft_entry__.build()'s / supir_amd.build's job, never done implicitly).
    dtype: element type of the 16-bit operands the caller is about to pass -- torch.float16 selects libsupir_hip_f16.so,
    anything else (None, torch.bfloat16) the bf16 library."""
    global _lib, _lib_f16
    if dtype is torch.float32:
        return load_f32()
    f16 = dtype is torch.float16
    cur = _lib_f16 if f16 else _lib
    if cur is not None:
        return cur
    path = LIB_PATH_F16 if f16 else LIB_PATH
    if not os.path.exists(path):
        raise SupirHipError(
            f"{path} not found: run `python -m supir_amd.build` (hipcc, gfx950). The HIP extension is mandatory; "
            "there is no CPU / PyTorch fallback on the product path.")
    lib = _bind(path, b"f16" if f16 else b"bf16")
    if f16:
        _lib_f16 = lib
    else:
        _lib = lib
    return lib


def load_f32():
    """libsupir_hip_f32.so (include/supir_hip_f32.h): what fp32 operands reach.  Same rules: built by supir_amd.build, no fallback."""
    global _lib_f32
    if _lib_f32 is not None:
        return _lib_f32
    if not os.path.exists(LIB_PATH_F32):
        raise SupirHipError(
            f"{LIB_PATH_F32} not found: run `python -m supir_amd.build` (hipcc, gfx950). The HIP extension is mandatory; "
            "there is no CPU / PyTorch fallback on the product path.")
    lib = ctypes.CDLL(LIB_PATH_F32)
    for name, restype in (("supir_abi_version", c_int), ("supir_target_arch", c_char_p), ("supir_elem_type", c_char_p),
                          ("supir_last_hip_error", c_int)):
        getattr(lib, name).restype = restype
        getattr(lib, name).argtypes = []
    lib.supir_hip_error_string.restype = c_char_p
    lib.supir_hip_error_string.argtypes = [c_int]
    for name, argtypes in SIGNATURES_F32.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    if lib.supir_abi_version() != ABI_VERSION:
        raise SupirHipError("libsupir_hip_f32.so ABI version mismatch")
    if lib.supir_elem_type() != b"f32":
        raise SupirHipError(f"libsupir_hip_f32.so was built for {lib.supir_elem_type()!r} elements, expected b'f32'")
    _lib_f32 = lib
    return lib


def loaded():
    """The libraries this process has dlopen'ed so far (bf16 first)."""
    return [lib for lib in (_lib, _lib_f16, _lib_f32) if lib is not None]


def check(rc, name, lib=None):
    if rc != 0:
        detail = ""
        lib = lib if lib is not None else _lib
        if rc == -3 and lib is not None:
            code = lib.supir_last_hip_error()
            detail = f" [hipError {code}: {lib.supir_hip_error_string(code).decode()}]"
        raise SupirHipError(f"{name} failed: {_ERR.get(rc, rc)}{detail}")
