"""Does the cache policy of the epilogues' OUTPUT stores change what a dependent kernel boundary costs?  (round 6, late)

A kernel that leaves B bytes dirty in the XCDs' L2s pays B / (a few TB/s) at its end-of-kernel release before the next kernel of the
stream may start (guide: "boundary" row, +2.8-3.8 us behind 12.6-16.8 MB).  The step has ~1250 launches whose outputs are 5-16 MB each.
csrc/common.h's supir_store16 / supir_store8 take a build-time policy (SUPIR_STORE_POLICY: 0 plain = the product build, 1 sc1 = write-through,
2 nt, 3 sc0 sc1); this tool builds one extra library per other policy (in-tree, git-ignored, never loaded by the product path), swaps it in for the
bf16 library inside ONE process and times the replayed 1024^2 step with the shipped kernel picks, interleaved, three times, and holds
every variant's output bitwise to the product's.  Result (profiles/r06/store_policy_ab.json): no policy beats plain stores.

    python tools/store_policy_ab.py --build-only        (here: hipcc cross-compiles)
    python tools/store_policy_ab.py [1 3]              (GPU box)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from supir_amd import build as B  # noqa: E402

POLICIES = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 3]     # the product library is policy 0 (plain stores)


def lib_path(pol):
    return os.path.join(B.HERE, f"libsupir_hip_sp{pol}.so")


def build_all():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
    procs, objs = [], {}
    for pol in POLICIES:
        lib = lib_path(pol)
        if not B._stale(lib):
            continue
        objdir = os.path.join(B.CSRC, "_obj", f"sp{pol}")
        os.makedirs(objdir, exist_ok=True)
        objs[pol] = []
        for src in B.SOURCES:
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            objs[pol].append(obj)
            cmd = base + [f"-DSUPIR_STORE_POLICY={pol}"] + B.EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(B.CSRC, src), "-o", obj]
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    for pol, oo in objs.items():
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic"] + oo + ["-o", lib_path(pol)])


build_all()
if "--build-only" in sys.argv:
    sys.exit(0)

import torch  # noqa: E402

from supir_amd import _lib  # noqa: E402
from tests.helpers import build_unet, synth_tensor  # noqa: E402

dev = "cuda"
base_lib = _lib.load()
PRODUCT = 0
assert PRODUCT not in POLICIES
libs = {PRODUCT: base_lib}
for pol in POLICIES:
    libs[pol] = _lib._bind(lib_path(pol), b"bf16")
wrap = build_unet(device=dev)
Bn, lat = 2, 128
x = synth_tensor("x", (Bn, 4, lat, lat)).to(dev)
cond = {"crossattn": synth_tensor("ctx", (Bn, 77, 2048)).to(dev), "vector": synth_tensor("y", (Bn, 2816)).to(dev),
        "control": synth_tensor("lq", (Bn, 4, lat, lat)).to(dev)}
t = torch.full((Bn,), 500, dtype=torch.int64, device=dev)
res, outs = {}, {}
with torch.no_grad():
    for rep in range(3):
        for pol in [PRODUCT] + POLICIES:
            _lib._lib = libs[pol]
            wrap.enable_graph(False)
            for _ in range(2):
                o = wrap(x, t, cond, 1.0)
            wrap.enable_graph(True)
            for _ in range(3):
                o = wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            n = 20
            t0 = time.time()
            for _ in range(n):
                o = wrap(x, t, cond, 1.0)
            torch.cuda.synchronize()
            ms = (time.time() - t0) / n * 1e3
            res.setdefault(pol, []).append(round(ms, 3))
            outs[pol] = o.float().clone()
            print(f"rep{rep} policy {pol}: {ms:.3f} ms/step", flush=True)
    wrap.enable_graph(False)
    _lib._lib = base_lib
print(f"product library: output finite {bool(torch.isfinite(outs[PRODUCT]).all())}, rms {outs[PRODUCT].float().pow(2).mean().sqrt().item():.4f}")
bitwise = {}
for pol in POLICIES:
    bitwise[str(pol)] = bool(torch.equal(outs[pol].view(torch.int32), outs[PRODUCT].view(torch.int32)))
    print(f"policy {pol} vs the product library: bitwise equal {bitwise[str(pol)]} (same kernels, same order, same arithmetic: must be True)")
out = {"what": "replayed 1024^2 CFG-doubled step (ms), shipped picks, one process, interleaved; store policy of the epilogue outputs",
       "policy_names": {"0": "plain", "1": "sc1", "2": "nt", "3": "sc0 sc1"}, "ms_per_step": {str(k): v for k, v in res.items()}, "bitwise_equal_to_product": bitwise}
print(json.dumps(out))
go = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(go):
    json.dump(out, open(os.path.join(go, "store_policy_ab.json"), "w"), indent=1)
