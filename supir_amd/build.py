"""Build libsupir_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The .so is git-ignored but travels to the GPU box with the working-tree snapshot; nothing is JIT-compiled at run time.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "attention.hip", "norm.hip", "edge.hip", "api.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "supir_hip.h")]
LIB = os.path.join(HERE, "libsupir_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print("[supir_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
