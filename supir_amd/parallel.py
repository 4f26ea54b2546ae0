"""Multi-GPU: one process per GPU, independent images sharded across ranks, ONE weight broadcast, no collective inside a
sample (the path shards at image granularity; the reference has no distributed code at all).

backend "nccl" is RCCL on ROCm (xGMI inside a node); the same code runs over gloo on CPU for the tests.
"""
import torch
import torch.distributed as dist


def broadcast_module_(module, src=0, bucket_elems=1 << 28, skip=("sigmas",), payload_dtype=None):
    """Broadcast every floating parameter/buffer of `module` from rank `src`, coalesced into buckets (few large
    collectives: xGMI rings are per-link bound, so size matters more than count). Returns the number of buckets sent.

    payload_dtype=torch.bfloat16: the fp32 masters travel as bf16 (7.9 GB instead of 15.9 GB for SDXL + control + VAE,
    SURVEY.md 8(e)).  The kernels only ever read bf16 copies of the weights, so nothing the compute path sees changes; to keep
    every rank's masters IDENTICAL (derived layouts such as LayerNorm-folded matrices are computed from the masters), rank
    `src` rounds its own masters to the payload precision as well."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    tensors = [t for k, t in module.state_dict().items() if t.is_floating_point() and not any(k.endswith(s) for s in skip)]
    groups = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    sent = 0
    for (dt, _), ts in groups.items():
        wire = payload_dtype if (payload_dtype is not None and dt == torch.float32) else dt
        bucket, size = [], 0
        for t in ts + [None]:
            if t is None or (bucket and size + t.numel() > bucket_elems):
                flat = torch.cat([b.reshape(-1).to(wire) for b in bucket])
                dist.broadcast(flat, src=src)
                o = 0
                with torch.no_grad():
                    for b in bucket:     # on rank src too: its masters become the rounded values every other rank receives
                        b.copy_(flat[o:o + b.numel()].view_as(b))
                        o += b.numel()
                sent += 1
                bucket, size = [], 0
            if t is not None:
                bucket.append(t)
                size += t.numel()
    return sent


def shard_items(n_items, rank=None, world=None):
    """Image i -> rank i % world (round robin, independent units, zero exchange)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
    return list(range(rank, n_items, world))


def max_over_ranks(seconds, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_images(local, n_items, device="cpu"):
    """Optional result gather: {index: [3,H,W] tensor} from every rank -> list on rank 0 (12 MB per 1024^2 image)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local[i] for i in range(n_items)]
    objs = [None] * dist.get_world_size()
    dist.all_gather_object(objs, {k: v.cpu() for k, v in local.items()})
    merged = {}
    for o in objs:
        merged.update(o)
    return [merged[i] for i in range(n_items)]


def sync_autotune(src=0, group=None):
    """Every rank adopts rank `src`'s autotune winners (ops._TUNE: tile per GEMM / conv shape; ops._CHOICE: launch-sequence
    alternatives such as fused vs separate q|k|v).  Tuning is per process and two near-equal candidates can win on different
    ranks, which makes the ranks' results differ at the bf16 noise floor; after this call they run the same kernels, so replicas
    (and the tiles of a tile-parallel sample) are bit-identical given identical inputs.  Call it after a warm-up pass has tuned
    the shapes in use and BEFORE graphs are captured for the timed / production calls (a changed pick invalidates captured
    graphs: re-run ControlWrapper.enable_graph).  Returns the number of entries that changed on this rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    from . import ops
    payload = [None]
    if dist.get_rank(group) == src:
        payload = [(list(ops._TUNE.items()), list(ops._CHOICE.items()))]
    dist.broadcast_object_list(payload, src=src, group=group)
    tune, choice = payload[0]
    changed = sum(1 for k, v in tune if ops._TUNE.get(k) != v) + sum(1 for k, v in choice if ops._CHOICE.get(k) != v)
    changed += sum(1 for k in list(ops._TUNE) if k not in dict(tune)) + sum(1 for k in list(ops._CHOICE) if k not in dict(choice))
    ops._TUNE.clear()
    ops._TUNE.update(dict(tune))
    ops._CHOICE.clear()
    ops._CHOICE.update(dict(choice))
    return changed
