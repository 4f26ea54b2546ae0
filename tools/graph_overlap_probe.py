"""Do the two branches of a step actually overlap under hipGraph replay?  GLVControl and the UNet encoder captured as two SEPARATE graphs:
replayed alone, one after the other on one stream, and concurrently on two streams; next to the same two branches captured as ONE
two-stream graph (what ControlWrapper does).  Usage: python tools/graph_overlap_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from supir_amd import weights as Wt
from tests.helpers import build_unet, synth_tensor

dev = "cuda"
wrap = build_unet(device=dev)
B, lat = 2, 128
x = synth_tensor("x", (B, 4, lat, lat)).to(dev)
ctx, vec = synth_tensor("ctx", (B, 77, 2048)).to(dev), synth_tensor("y", (B, 2816)).to(dev)
lq = synth_tensor("lq", (B, 4, lat, lat)).to(dev)
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
cm, dm = wrap.control_model, wrap.diffusion_model
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def run_c():
    return cm(x=lq, timesteps=t, xt=x, context=ctx, y=vec)


def run_e():
    return dm.encode(x, timesteps=t, context=ctx, y=vec)


def capture(fn, stream):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            out = fn()
    return g, out


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


with torch.no_grad(), Wt.compute_dtype(torch.bfloat16):
    wrap(x, t, {"crossattn": ctx, "vector": vec, "control": lq}, 1.0)     # caches / autotune
    gc_, keep1 = capture(run_c, sA)
    ge_, keep2 = capture(run_e, sB)

    def both_one_graph():
        main = torch.cuda.current_stream()
        sA.wait_stream(main)
        with torch.cuda.stream(sA):
            c = run_c()
        e = run_e()
        main.wait_stream(sA)
        return c, e

    g2, keep3 = capture(both_one_graph, sB)

    def alone_c():
        with torch.cuda.stream(sA):
            gc_.replay()

    def alone_e():
        with torch.cuda.stream(sB):
            ge_.replay()

    def sequential():
        with torch.cuda.stream(sA):
            gc_.replay()
            ge_.replay()

    def concurrent():
        with torch.cuda.stream(sA):
            gc_.replay()
        with torch.cuda.stream(sB):
            ge_.replay()

    def one_graph():
        with torch.cuda.stream(sB):
            g2.replay()

    for rep in range(2):
        print(f"rep{rep}: control alone {timed(alone_c):.2f} ms | encoder alone {timed(alone_e):.2f} ms | one after the other {timed(sequential):.2f} ms | "
              f"two graphs on two streams {timed(concurrent):.2f} ms | ONE two-stream graph {timed(one_graph):.2f} ms", flush=True)

    # ---- more chains: the two CFG samples are independent too.  Four graphs (control / encoder x sample 0 / 1, B = 1 each) on four
    # streams, and -- the shape of the decoder phase, one chain today -- the encoder alone as two B = 1 graphs on two streams
    ss = [torch.cuda.Stream() for _ in range(4)]
    xs = [x[b:b + 1].contiguous() for b in range(B)]
    lqs = [lq[b:b + 1].contiguous() for b in range(B)]
    cs = [ctx[b:b + 1].contiguous() for b in range(B)]
    vs = [vec[b:b + 1].contiguous() for b in range(B)]
    ts = [t[b:b + 1].contiguous() for b in range(B)]
    fns = []
    for b in range(B):
        fns.append(lambda b=b: cm(x=lqs[b], timesteps=ts[b], xt=xs[b], context=cs[b], y=vs[b]))
        fns.append(lambda b=b: dm.encode(xs[b], timesteps=ts[b], context=cs[b], y=vs[b]))
    graphs = [capture(f, s) for f, s in zip(fns, ss)]

    def four():
        for (g, _), s in zip(graphs, ss):
            with torch.cuda.stream(s):
                g.replay()

    def enc_two_b1():
        for i in (1, 3):
            with torch.cuda.stream(ss[i]):
                graphs[i][0].replay()

    def enc_b1_alone():
        with torch.cuda.stream(ss[1]):
            graphs[1][0].replay()

    for rep in range(2):
        print(f"rep{rep}: four B=1 graphs on four streams {timed(four):.2f} ms (vs two B=2 graphs above) | encoder: one B=1 graph {timed(enc_b1_alone):.2f} ms, "
              f"two B=1 graphs on two streams {timed(enc_two_b1):.2f} ms (vs one B=2 graph above)", flush=True)
