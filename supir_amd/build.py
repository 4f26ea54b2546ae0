"""Build libsupir_hip.so (bf16 elements, the product default), libsupir_hip_f16.so (the same sources with -DSUPIR_F16: fp16
elements and MFMA operands, for callers that request the reference's default diff_dtype) and libsupir_hip_f32.so (csrc/f32: the
fp32 service of `--diff_dtype fp32` / `--ae_dtype fp32` requests, include/supir_hip_f32.h) in-tree with hipcc for gfx950
(cross-compiles without a GPU).

The .so files are git-ignored but travel to the GPU box with the working-tree snapshot; nothing is JIT-compiled at run time.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "gemm16.hip", "gemm_big.hip", "attention.hip", "xattn.hip", "attention_d512.hip", "norm.hip", "edge.hip", "sampler.hip", "api.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "supir_hip.h")]
LIB = os.path.join(HERE, "libsupir_hip.so")
LIB_F16 = os.path.join(HERE, "libsupir_hip_f16.so")
LIB_F32 = os.path.join(HERE, "libsupir_hip_f32.so")
LIB_TOOLS = os.path.join(HERE, "libsupir_hip_tools.so")   # the bf16 sources with -DSUPIR_TOOLS: adds supir_debug_knob (kernel-variant switches for
                                                          # A/B measurements and variant-vs-variant tests); never loaded by the product path
SOURCES_F32 = [os.path.join("f32", "f32.hip")]
HEADERS_F32 = [os.path.join("..", "..", "include", "supir_hip.h"), os.path.join("..", "..", "include", "supir_hip_f32.h")]
# (library, object sub-directory, extra compile flags, extra link flags, sources, headers).  -Bsymbolic on the fp16 / fp32 builds: the
# libraries define the same symbols; each must bind to its own even if a host application loads them RTLD_GLOBAL.
VARIANTS = [(LIB, "", [], [], SOURCES, HEADERS), (LIB_F16, "f16", ["-DSUPIR_F16"], ["-Wl,-Bsymbolic"], SOURCES, HEADERS),
            (LIB_F32, "f32", [], ["-Wl,-Bsymbolic"], SOURCES_F32, HEADERS_F32),
            (LIB_TOOLS, "tools", ["-DSUPIR_TOOLS"], ["-Wl,-Bsymbolic"], SOURCES, HEADERS)]


def _stale(lib=LIB, sources=SOURCES, headers=HEADERS):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in sources + headers)


# per-source extra flags.  attention.hip: keep the MFMA accumulators in architectural VGPRs -- the default allocation put
# O and S^T in AGPRs and spent 176 v_accvgpr moves per KV tile on the online-softmax rescale (37 % of the loop's instructions).
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "xattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _obj_stale(src, obj, headers=HEADERS):
    """An object is rebuilt when its source or any shared header is newer (headers are few and included everywhere)."""
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in [src] + headers)


def build(force=False, verbose=True):
    """Compile every stale translation unit of every stale variant (in parallel), link, return the bf16 library's path."""
    todo = [v for v in VARIANTS if force or _stale(v[0], v[4], v[5])]
    if not todo:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
    procs, objs = [], {}
    for lib, sub, cflags, _, sources, headers in todo:
        objdir = os.path.join(CSRC, "_obj", sub)
        os.makedirs(objdir, exist_ok=True)
        objs[lib] = []
        for src in sources:
            obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
            objs[lib].append(obj)
            if not force and not _obj_stale(src, obj, headers):
                continue
            cmd = base + cflags + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print("[supir_amd.build]", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    for lib, _, _, lflags, _, _ in todo:
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + lflags + objs[lib] + ["-o", lib + ".tmp"]
        if verbose:
            print("[supir_amd.build]", " ".join(link), flush=True)
        subprocess.check_call(link)
        os.replace(lib + ".tmp", lib)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
