"""Build libsupir_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The .so is git-ignored but travels to the GPU box with the working-tree snapshot; nothing is JIT-compiled at run time.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "gemm16.hip", "gemm_big.hip", "attention.hip", "norm.hip", "edge.hip", "api.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "supir_hip.h")]
LIB = os.path.join(HERE, "libsupir_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


# per-source extra flags.  attention.hip: keep the MFMA accumulators in architectural VGPRs -- the default allocation put
# O and S^T in AGPRs and spent 176 v_accvgpr moves per KV tile on the online-softmax rescale (37 % of the loop's instructions).
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = base + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[supir_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"]
    if verbose:
        print("[supir_amd.build]", " ".join(link), flush=True)
    subprocess.check_call(link)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
