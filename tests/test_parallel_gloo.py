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
