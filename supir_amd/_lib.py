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

_ERR = {-1: "SUPIR_ERR_ARG (null pointer / bad size)", -2: "SUPIR_ERR_SHAPE (unsupported shape or alignment)",
        -3: "SUPIR_ERR_HIP (launch failed)"}

P, I, F, L = c_void_p, c_int, c_float, c_long

# name -> argtypes; mirrors include/supir_hip.h one to one (tests/test_abi.py cross-checks against the header)
SIGNATURES = {
    "supir_gemm_bf16": [P, P, P, I, I, I, I, I, P, P, I, I, P, I, I, I, F, I, P],
    "supir_conv3x3_bf16": [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P, P, I, P, I, I, I, F, I, P],
    "supir_flash_attn_d64": [P, P, P, P, I, I, I, I, I, I, I, I, F, P],
    "supir_flash_attn_d64_ex": [P, P, P, P, I, I, I, I, I, I, I, I, F, I, P],
    "supir_softmax_rows": [P, P, I, I, I, L, L, F, P],
    "supir_groupnorm_nhwc": [P, P, P, P, I, I, I, I, I, I, P, P, F, I, P, P, I, F, P, I, P, c_size_t, P, P],
    "supir_groupnorm_stats": [P, P, I, I, I, I, I, I, P, P, c_size_t, P],
    "supir_layernorm": [P, P, P, P, I, I, I, I, F, P],
    "supir_conv3x3_smallcin": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "supir_conv3x3_smallcout": [P, P, P, P, I, I, I, I, I, I, P],
    "supir_pointwise_nchw": [P, P, P, P, I, I, I, L, F, P],
    "supir_wavelet_level": [P, P, P, I, I, I, I, I, P],
    "supir_resample_u8": [P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "supir_bicubic_f32": [P, P, P, I, I, I, I, I, P],
    "supir_gemm_tile_for": [I, I, I],
    "supir_prefetch": [P, c_size_t, P, P],
    "supir_set_next_prefetch": [P, c_size_t],
    "supir_set_next_gn_partials": [P],
    "supir_groupnorm_nhwc_parts": [P, P, P, P, I, I, I, I, I, I, P, P, F, I, P, P, I, F, P, I, P, I, P, I, P],
    "supir_rowstats_finalize": [P, P, I, I, I, I, F, P],
    "supir_gemm_bf16_qkv": [P, P, P, P, I, I, I, I, I, I, I, I, P, P, I, I, P, F, P],
    "supir_gemm_bf16_ln": [P, P, P, I, I, I, I, I, P, P, I, I, I, I, F, I, P, I, P, I, I, P, F, P],
}

_lib = None


class SupirHipError(RuntimeError):
    pass


def load():
    """Load the library (building is __graft_entry__.build()'s / supir_amd.build's job, never done implicitly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SupirHipError(
            f"{LIB_PATH} not found: run `python -m supir_amd.build` (hipcc, gfx950). The HIP extension is mandatory; "
            "there is no CPU / PyTorch fallback on the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.supir_abi_version.restype = c_int
    lib.supir_abi_version.argtypes = []
    lib.supir_target_arch.restype = c_char_p
    lib.supir_target_arch.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.supir_last_hip_error.restype = c_int
    lib.supir_last_hip_error.argtypes = []
    lib.supir_hip_error_string.restype = c_char_p
    lib.supir_hip_error_string.argtypes = [c_int]
    if lib.supir_abi_version() != 1:
        raise SupirHipError("libsupir_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        detail = ""
        if rc == -3 and _lib is not None:
            code = _lib.supir_last_hip_error()
            detail = f" [hipError {code}: {_lib.supir_hip_error_string(code).decode()}]"
        raise SupirHipError(f"{name} failed: {_ERR.get(rc, rc)}{detail}")
