"""Worker of tests/test_rccl_single_rank_gpu.py (run as a subprocess: a wedged collective must not take the test session with it).

One rank, backend "nccl" (= RCCL on ROCm), with supir_amd.parallel.FORCE_COLLECTIVES on, so that every collective of the multi-GPU
design is ISSUED through RCCL on device tensors even though the group has one member: the bucketed weight broadcast, the autotune
object broadcast, the timing all-reduce, the tile-parallel sampler's per-step broadcast + all-reduce, the tile-parallel VAE's
pooled-statistics and canvas all-reduces -- and one hipGraph capture + replay with the process group's watchdog thread alive
(thread-local capture mode).  Each is compared bit for bit with the same computation without torch.distributed.  Prints one JSON
line."""
import json
import os
import sys

import torch


def main():
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from supir_amd import ops, parallel
    from supir_amd.modules import wrappers
    from supir_amd.modules.sampling import DiscreteDenoiserWithControl, LinearCFG, TiledRestoreEDMSampler
    from supir_amd.utils.tilevae import VAEHook
    from tests.helpers import build_unet, build_vae, synth_tensor
    dev = "cuda"
    res = {}

    def T(name, shape, **kw):
        return synth_tensor(name, shape, **kw).to(dev)

    wrap = build_unet(depth=(1, 1, 2), device=dev)
    vae = build_vae(dev)
    for net in (vae.denoise_encoder, vae.decoder):
        net.original_forward = net.forward
    x, lq = T("xt", (2, 4, 16, 16)), T("lq", (2, 4, 16, 16))
    y, ctx = T("vector", (2, 2816)), T("context", (2, 77, 2048))
    t = torch.tensor([500, 37], dtype=torch.int64, device=dev)
    cond = {"crossattn": ctx, "vector": y, "control": lq}
    den = DiscreteDenoiserWithControl().to(dev)
    h, w, steps = 48, 40, 2
    c1 = {"crossattn": ctx[:1], "vector": y[:1], "control": T("lq_tiled", (1, 4, h, w))}
    uc1 = {"crossattn": ctx[1:], "vector": y[1:], "control": c1["control"]}
    x0, xc = T("tiled.gpu.x0", (1, 4, h, w)), T("tiled.gpu.xc", (1, 4, h, w))
    z_t, img_t = T("z_tiled", (1, 4, 40, 32)), T("img_tiled", (1, 3, 192, 160), scale=0.5)

    def tiled_sample(parallel_on):
        torch.manual_seed(11)
        smp = TiledRestoreEDMSampler(tile_size=32, tile_stride=16, num_steps=steps, s_churn=5, s_noise=1.01, restore_cfg=4.0,
                                     guider_config=LinearCFG(1.0, 4.0), device=dev, tile_batch=2, tile_parallel=parallel_on)
        with torch.no_grad():
            return smp(lambda i, s, cc, cs: den(wrap, i, s, cc, cs), x0.clone(), cond=dict(c1), uc=dict(uc1), x_center=xc,
                       control_scale=1.0).float().clone()

    def tiled_vae(parallel_on):
        with torch.no_grad():
            d = VAEHook(vae.decoder, 8, is_decoder=True, tile_parallel=parallel_on)(z_t).clone()
            e = VAEHook(vae.denoise_encoder, 64, is_decoder=False, tile_parallel=parallel_on)(img_t).clone()
        return d, e

    # ---- reference results, no torch.distributed
    with torch.no_grad():
        wrap(x, t, cond, 1.0)
        eager = wrap(x, t, cond, 1.0).clone()
    ref_smp = tiled_sample(False)
    ref_dec, ref_enc = tiled_vae(False)
    sd_before = {k: v.clone() for k, v in vae.state_dict().items()}

    # ---- RCCL, one rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    parallel.FORCE_COLLECTIVES = True
    res["backend"] = dist.get_backend()
    res["capture_mode"] = wrappers._capture_mode()
    # bucketed weight broadcast on device tensors (several buckets: 1.4 G parameters of the mini UNet / control in 2^26-element buckets)
    n_vae = parallel.broadcast_module_(vae, src=0, bucket_elems=1 << 22)
    n_net = parallel.broadcast_module_(wrap, src=0, bucket_elems=1 << 26)
    torch.cuda.synchronize()
    res["broadcast_buckets"] = [n_vae, n_net]
    res["broadcast_identity"] = all(torch.equal(v, sd_before[k]) for k, v in vae.state_dict().items())
    tune_before = (dict(ops._TUNE), dict(ops._CHOICE))
    res["sync_autotune_changed"] = parallel.sync_autotune()
    res["sync_autotune_identity"] = (dict(ops._TUNE), dict(ops._CHOICE)) == tune_before
    res["max_over_ranks"] = parallel.max_over_ranks(1.25, device=dev)
    # replica start-up as bench.py does it for N > 1 (round 5): a module constructed on META + to_empty, every parameter AND buffer
    # received; here the "received" values are copied from the materialised model (a one-rank group has no second rank), then a
    # receive-everything broadcast runs through RCCL on it with one buffer kept on the HOST (rank 0 of the real job builds the denoiser's
    # sigma table there: RCCL cannot move host memory, the payload is staged on the device).  The replica must compute the same bits.
    from tests.helpers import SUPIR_NET
    from supir_amd.modules.supir_v0 import GLVControl, LightGLVUNet
    from supir_amd.modules.wrappers import ControlWrapper

    def factory():
        net = dict(SUPIR_NET, transformer_depth=[1, 1, 2])
        ctl = {k: v for k, v in net.items() if k not in ("mode", "project_type", "project_channel_scale")}
        w2 = ControlWrapper(LightGLVUNet(**net), dtype=torch.bfloat16)
        w2.load_control_model(GLVControl(**ctl, input_upscale=1))
        return w2

    rep = parallel.construct_replica(factory, dev, materialize=False)
    res["replica_on_device"] = all(v.is_cuda for v in rep.state_dict().values())
    with torch.no_grad():
        src_sd = wrap.state_dict()
        for k, v in rep.state_dict().items():
            v.copy_(src_sd[k])
    den_host = DiscreteDenoiserWithControl()                      # its sigma table lives on the host
    sig_before = den_host.sigmas.clone()
    res["host_buffer_buckets"] = parallel.broadcast_module_(den_host, src=0, skip=()) + parallel.broadcast_module_(rep, src=0, skip=(), bucket_elems=1 << 26)
    res["host_buffer_identity"] = bool((not den_host.sigmas.is_cuda) and torch.equal(den_host.sigmas, sig_before))
    with torch.no_grad():
        res["replica_equal"] = bool(torch.equal(rep(x, t, cond, 1.0), eager))
    del rep
    # hipGraph capture + replay with the NCCL watchdog alive
    with torch.no_grad():
        e2 = wrap(x, t, cond, 1.0).clone()
        wrap.enable_graph(True)
        g1 = wrap(x, t, cond, 1.0).clone()
        g2 = wrap(x, t, cond, 1.0).clone()
        wrap.enable_graph(False)
    res["eager_equal_after_broadcast"] = bool(torch.equal(e2, eager))
    res["graph_equal_eager"] = bool(torch.equal(g1, eager) and torch.equal(g2, eager))
    # tile-parallel sampler: per-step broadcast of the churn noise + all-reduce of the blended canvas, through RCCL
    res["tile_parallel_sampler_equal"] = bool(torch.equal(tiled_sample(True), ref_smp))
    d, e = tiled_vae(True)
    res["tile_parallel_vae_equal"] = bool(torch.equal(d, ref_dec) and torch.equal(e, ref_enc))
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
