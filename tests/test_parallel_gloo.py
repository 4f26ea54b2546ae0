"""world_size-2 gloo run of the multi-GPU plumbing on CPU: weight broadcast, image sharding, max-over-ranks timing."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from supir_amd import parallel
    from supir_amd.modules.vae import AutoencoderKLInferenceWrapper
    from supir_amd.synth import fill_state_dict_
    from tests.helpers import VAE_DD
    dd = dict(VAE_DD, ch=32, ch_mult=[1, 2])
    vae = AutoencoderKLInferenceWrapper(embed_dim=4, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"})
    with torch.no_grad():
        for p in vae.parameters():
            p.fill_(float(rank + 7))          # ranks start different
    if rank == 0:
        fill_state_dict_(vae)
    n_b = parallel.broadcast_module_(vae, src=0, bucket_elems=50_000)   # force several buckets
    ref = AutoencoderKLInferenceWrapper(embed_dim=4, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"})
    fill_state_dict_(ref)
    same = all(torch.equal(a, b) for a, b in zip(vae.state_dict().values(), ref.state_dict().values()))
    # bf16 payload (optional: half the bytes over xGMI): every rank, rank 0 included, ends with the SAME masters -- the big matrices
    # (ndim >= 2, >= 2^16 elements: what the kernels read as 16-bit copies) bf16-rounded, everything the kernels read as fp32 untouched
    vae2 = AutoencoderKLInferenceWrapper(embed_dim=4, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"})
    with torch.no_grad():
        for p in vae2.parameters():
            p.fill_(float(rank + 3))
    if rank == 0:
        fill_state_dict_(vae2)
    parallel.broadcast_module_(vae2, src=0, bucket_elems=50_000, payload_dtype=torch.bfloat16, round_min_elems=1 << 12)
    n_rounded = 0
    for (k, a), b in zip(vae2.state_dict().items(), ref.state_dict().values()):
        if not a.is_floating_point():
            continue
        big = a.dim() >= 2 and a.numel() >= (1 << 12)
        n_rounded += int(big)
        same = same and a.dtype == torch.float32 and torch.equal(a, b.to(torch.bfloat16).to(b.dtype) if big else b)
    same = same and n_rounded > 0
    # replica start-up as bench.py does it for N > 1 (parallel.construct_replica): rank 0 constructs for real and owns the weights, every
    # other rank constructs on META, allocates (to_empty) and RECEIVES everything -- parameters and constructor-computed buffers (the
    # denoiser's sigma table) alike; the received module must be the one rank 0 holds, bit for bit, buffers included
    from supir_amd.configs import supir_v0_config
    from supir_amd.plugin import instantiate_from_config
    cfg = supir_v0_config(transformer_depth=[0, 0, 1], sampler_device="cpu")
    cfg["params"]["first_stage_config"]["params"]["ddconfig"].update(ch=32, ch_mult=[1, 2])
    for key in ("control_stage_config", "network_config"):
        cfg["params"][key]["params"].update(model_channels=64, context_dim=64, adm_in_channels=64)
    try:
        mdl = parallel.construct_replica(lambda: instantiate_from_config(cfg), "cpu", materialize=(rank == 0))
        if rank == 0:
            fill_state_dict_(mdl)
        else:
            assert all(t.device.type == "cpu" for t in mdl.state_dict().values())
        parallel.broadcast_module_(mdl, src=0, bucket_elems=200_000, skip=())
        ref_m = instantiate_from_config(cfg)
        fill_state_dict_(ref_m)
        rsd = ref_m.state_dict()
        same = same and set(rsd) == set(mdl.state_dict()) and "denoiser.sigmas" in rsd
        same = same and all(torch.equal(v, rsd[k]) for k, v in mdl.state_dict().items())
    except NotImplementedError:
        same = False
    # persistent=False buffers (absent from state_dict: the text towers' position_ids / causal mask) travel too, and a plain
    # tensor ATTRIBUTE computed in a constructor -- which cannot travel -- makes the meta construction fail instead of leaving garbage
    class WithHidden(torch.nn.Module):
        def __init__(self, attr=False):
            super().__init__()
            self.lin = torch.nn.Linear(4, 4)
            self.register_buffer("mask", torch.arange(12.0).view(3, 4), persistent=False)
            if attr:
                self.table = torch.arange(5.0)
    m = parallel.construct_replica(WithHidden, "cpu", materialize=(rank == 0))
    if rank != 0:
        with torch.no_grad():
            m.mask.fill_(-1.0)          # whatever to_empty left there
    parallel.broadcast_module_(m, src=0, skip=())
    same = same and "mask" not in m.state_dict() and torch.equal(m.mask, torch.arange(12.0).view(3, 4))
    if rank != 0:
        try:
            parallel.construct_replica(lambda: WithHidden(attr=True), "cpu", materialize=False)
            same = False
        except RuntimeError as e:
            same = same and "table" in str(e)
    # autotune winners: ranks that tuned differently end with rank 0's picks
    from supir_amd import ops
    ops._TUNE.clear(); ops._CHOICE.clear()
    ops._TUNE[("gemm", 2048, 1280, 1280, 0, 0)] = 35 if rank == 0 else 32
    ops._TUNE[("gemm", 2048, 10240, 1280, 2, 0)] = 37
    if rank == 1:
        ops._TUNE[("gemm", 64, 64, 64, 0, 0)] = 3            # only rank 1 saw this shape: dropped (rank 0 will tune it when it meets it)
    ops._CHOICE[("qkv", 2048, 1280)] = 1 if rank == 0 else 0
    n_changed = parallel.sync_autotune()
    same = same and n_changed == (0 if rank == 0 else 3) and ops._TUNE == {("gemm", 2048, 1280, 1280, 0, 0): 35, ("gemm", 2048, 10240, 1280, 2, 0): 37} \
        and ops._CHOICE == {("qkv", 2048, 1280): 1}
    mine = parallel.shard_items(5)
    t = parallel.max_over_ranks(1.0 + rank)
    imgs = parallel.gather_images({i: torch.full((3, 2, 2), float(i)) for i in mine}, 5)
    q.put((rank, same, n_b, mine, t, [float(im.mean()) for im in imgs]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_shard_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, same0, nb0, m0, t0, g0), (r1, same1, nb1, m1, t1, g1) = res
    assert same0 and same1 and nb0 == nb1 and nb0 > 3
    assert m0 == [0, 2, 4] and m1 == [1, 3]
    assert t0 == t1 == 2.0
    assert g0 == g1 == [0.0, 1.0, 2.0, 3.0, 4.0]


def _tile_worker(rank, world, port, q, tile_batch):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from supir_amd.modules import sampling as S
    from tests.helpers import synth_tensor

    def fake_net(xin, tt, cc, cs):
        return torch.tanh(xin * 0.7 + cc["control"] * 0.1) * (1.0 + 0.001 * tt.view(-1, 1, 1, 1).float()) * cs

    big = (1, 4, 24, 40)
    lqb = synth_tensor("lq_big", big)
    ctx, y = synth_tensor("context", (2, 77, 2048)), synth_tensor("vector", (2, 2816))
    c = {"crossattn": ctx[:1], "vector": y[:1], "control": lqb}
    uc = {"crossattn": ctx[1:], "vector": y[1:], "control": lqb}
    den = S.DiscreteDenoiserWithControl()

    def run(parallel, seed, x0):
        smp = S.TiledRestoreEDMSampler(tile_size=16, tile_stride=8, num_steps=3, s_churn=5, s_noise=1.01, restore_cfg=4.0,
                                       device="cpu", guider_config=S.LinearCFG(1.0, 4.0), tile_batch=tile_batch,
                                       tile_parallel=parallel)
        torch.manual_seed(seed)
        return smp(lambda i, s, cc, cs: den(fake_net, i, s, cc, cs), x0.clone(), cond=dict(c), uc=dict(uc),
                   x_center=synth_tensor("xc_big", big), control_scale=1.0)

    x0 = synth_tensor("noised_big", big)
    serial = run(False, 123, x0)                                   # what one GPU computes from rank 0's start latent / RNG stream
    # rank 1 starts from a DIFFERENT latent and RNG stream: the broadcasts must pull it onto rank 0's trajectory
    par = run(True, 123 if rank == 0 else 999, x0 if rank == 0 else x0 * 0.5 + 1.0)
    err = ((par - serial).norm() / serial.norm()).item()
    q.put((rank, err, bool(torch.isfinite(par).all())))
    dist.barrier()
    dist.destroy_process_group()


def _run_tile_parallel(tile_batch):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, q, tile_batch)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_tile_parallel_sampler_matches_single_process():
    """tiles of a step dealt to 2 ranks + one all-reduce per step == the single-process tiled sampler (up to the
    fp32 summation order of overlapping tiles), with and without tile batching, on every rank."""
    for tile_batch in (1, 2):
        for rank, err, finite in _run_tile_parallel(tile_batch):
            assert finite and err <= 2e-6, (tile_batch, rank, err)


# ------------------------------------------------------------------------------------------------ tile-parallel tiled VAE
def _tiny_vae():
    """A narrow VAE (32 base channels, the reference's 4-level layout) with synthetic weights, on CPU."""
    from supir_amd.modules.vae import AutoencoderKLInferenceWrapper
    from tests.helpers import VAE_DD, fill_module
    import copy
    vae = AutoencoderKLInferenceWrapper(embed_dim=4, ddconfig=dict(VAE_DD, ch=32), lossconfig={"target": "torch.nn.Identity"})
    vae.denoise_encoder = copy.deepcopy(vae.encoder)
    return fill_module(vae, "first_stage_model.", "cpu")


def _vae_tile_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from supir_amd.utils.tilevae import VAEHook
    from tests import fake_ops
    from tests.helpers import synth_tensor
    fake_ops.install()
    vae = _tiny_vae()
    img, z = synth_tensor("img_tiled", (2, 3, 192, 160), scale=0.5), synth_tensor("z_tiled", (2, 4, 40, 32))
    res = {}
    with torch.no_grad():
        for name, net, size, dec, x in (("enc", vae.denoise_encoder, 64, False, img), ("dec", vae.decoder, 8, True, z)):
            serial = VAEHook(net, size, is_decoder=dec)(x)
            par = VAEHook(net, size, is_decoder=dec, tile_parallel=True)(x)
            res[name] = ((par - serial).norm() / serial.norm()).item()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_parallel_tiled_vae_matches_single_process():
    """VAEHook(tile_parallel=True) on two ranks (tiles dealt round robin; one all-reduce of the pooled GroupNorm statistics per
    norm layer, one of the assembled result) == the single-process tiled forward, batch of 2 images, encoder and decoder.  The
    kernels are replaced by tests/fake_ops.py (plain torch): this checks the host logic -- SURVEY.md 8(e), row "Tiles"."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_vae_tile_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, errs in res:
        assert errs["enc"] <= 2e-6 and errs["dec"] <= 2e-6, (rank, errs)
