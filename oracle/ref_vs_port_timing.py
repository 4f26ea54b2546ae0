"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Same-box comparison of the two CPU baselines bench.py can report:

  kind "reference": the REFERENCE's own modules (ControlWrapper(LightGLVUNet) + GLVControl from /root/reference, FULL depth
                    [1, 2, 10], fp32, imported through oracle/ref_import.py) -- what `north_star` names as the CPU path;
  kind "port"     : oracle/supir_oracle.py (the restatement bench.py times on the GPU box, where /root/reference does not exist).

Both run the SAME network call (one CFG-doubled UNet + control forward, same synthetic weights by key, same inputs) on the same
host cores, so the file this writes (profiles/r04/cpu_baseline_oracle_vs_reference_same_box.json) shows (a) how good a proxy the
port's timing is for the reference's, and (b) the port's parity with the reference at FULL depth (the committed goldens are at
depth [1, 1, 2]; this closes the "depth-agnostic by assumption" gap for the network call).

    python -m oracle.ref_vs_port_timing [--latent 32 64] [--threads 8]        # build container only (needs /root/reference)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import as R  # noqa: E402
from oracle import supir_oracle as O  # noqa: E402
from supir_amd.synth import fill_state_dict_, synth_tensor  # noqa: E402


def build_reference(depth=None):
    ns = R.load_reference()
    net, ctl, _ = R.unet_params(depth=depth)
    with R.quiet():
        unet = ns.LightGLVUNet(**net).eval()
        ctrl = ns.GLVControl(**ctl).eval()
    fill_state_dict_({"model.diffusion_model." + k: v for k, v in unet.state_dict().items()})
    fill_state_dict_({"model.control_model." + k: v for k, v in ctrl.state_dict().items()})
    wrap = ns.ControlWrapper(unet, dtype=torch.float32)
    wrap.load_control_model(ctrl)
    sd = {"model.diffusion_model." + k: v for k, v in unet.state_dict().items()}
    sd.update({"model.control_model." + k: v for k, v in ctrl.state_dict().items()})
    return wrap, sd


def time_call(fn, repeat=1):
    best, out = None, None
    for _ in range(repeat):
        t0 = time.time()
        out = fn()
        dt = time.time() - t0
        best = dt if best is None or dt < best else best
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, nargs="+", default=[32, 64])
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--depth", type=int, nargs=3, default=None, help="transformer depth (default: the YAML's [1, 2, 10])")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "cpu_baseline_oracle_vs_reference_same_box.json"))
    a = ap.parse_args()
    if not R.available():
        raise SystemExit("needs the reference checkout (/root/reference)")
    torch.set_num_threads(a.threads)
    t0 = time.time()
    wrap, sd = build_reference(a.depth)
    build_s = time.time() - t0
    res = {"host_threads": a.threads, "host_threads_available": os.cpu_count(), "transformer_depth": a.depth or [1, 2, 10],
           "reference_build_and_fill_s": round(build_s, 1), "torch": torch.__version__, "calls": []}
    B = 2
    t = torch.tensor([500, 37], dtype=torch.int64)
    for lat in a.latent:
        x, lq = synth_tensor("xt", (B, 4, lat, lat)), synth_tensor("lq", (B, 4, lat, lat))
        cond = {"crossattn": synth_tensor("context", (B, 77, 2048)), "vector": synth_tensor("vector", (B, 2816)), "control": lq}
        with torch.no_grad():
            # warm-up of both (first-touch of 15 GB of weights), then one timed call each, interleaved
            wrap(x, t, dict(cond), 1.0)
            O.control_wrapper(sd, x, t, cond, 1.0)
            t_ref, y_ref = time_call(lambda: wrap(x, t, dict(cond), 1.0))
            t_port, y_port = time_call(lambda: O.control_wrapper(sd, x, t, cond, 1.0))
        err = ((y_port - y_ref).norm() / y_ref.norm()).item()
        res["calls"].append({"latent": lat, "pixels": lat * 8, "batch": B, "reference_s": round(t_ref, 3), "port_s": round(t_port, 3),
                             "port_over_reference": round(t_port / t_ref, 3), "port_vs_reference_rel_l2": err})
        print(res["calls"][-1], flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
