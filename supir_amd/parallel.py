"""Multi-GPU: one process per GPU, independent images sharded across ranks, ONE weight broadcast, no collective inside a
sample (the path shards at image granularity; the reference has no distributed code at all).

backend "nccl" is RCCL on ROCm (xGMI inside a node); the same code runs over gloo on CPU for the tests.
"""
import os

import torch
import torch.distributed as dist

# A 1-rank group has nothing to exchange, so every helper below returns early for it.  FORCE_COLLECTIVES (env
# SUPIR_FORCE_COLLECTIVES=1) runs the collectives anyway: on a single-GPU box this is the only way to execute the RCCL code path
# (bucketed broadcasts, all-reduces on device tensors, graph capture next to a live process group) -- tests/test_rccl_single_rank_gpu.py.
FORCE_COLLECTIVES = os.environ.get("SUPIR_FORCE_COLLECTIVES", "0") == "1"


def collectives_active(group=None):
    """True when torch.distributed is initialised and the group has more than one rank (or FORCE_COLLECTIVES is set)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def broadcast_module_(module, src=0, bucket_elems=1 << 28, skip=("sigmas",), payload_dtype=None, round_min_elems=1 << 16):
    """Broadcast every floating parameter/buffer of `module` from rank `src`, coalesced into buckets (few large
    collectives: xGMI rings are per-link bound, so size matters more than count). Returns the number of buckets sent.

    payload_dtype=None (default): the masters travel in their own precision (15.9 GB of fp32 for SDXL + control + VAE, once,
    ~0.1 s per GB-link): every rank computes exactly what a single-GPU run computes.

    payload_dtype=torch.bfloat16 halves the bytes for links where that matters, at a price that must be known: only the big
    matrices (ndim >= 2 and >= 2^16 elements: Linear / 3x3-conv weights, which the kernels consume as 16-bit copies) travel
    rounded; biases, norm affine parameters and the <= 8-channel edge convolutions (read as fp32 by the kernels) stay fp32.  Rank
    `src` rounds its own copies too, so all ranks hold IDENTICAL masters -- but layouts DERIVED from them in fp32 (LayerNorm-folded
    W' = gamma (.) W, its column sums) then start from bf16-rounded W and differ from a single-GPU run at the bf16 noise floor,
    and an fp16 compute scope loses the three mantissa bits it would have kept.  Not used by bench.py."""
    if not collectives_active():
        return 0
    # floating tensors travel in dtype buckets; the few integer buffers (none on SUPIR's path today) would travel one by one.  Bucket
    # composition depends on names, shapes and dtypes only -- never on where a rank happens to keep a tensor (rank 0 builds the
    # denoiser's sigma table on the HOST, a meta-constructed replica has it on the device): every payload is staged on ONE
    # communication device (RCCL moves device memory only) and copied back to wherever the rank keeps the tensor
    comm = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    state = {k: t for k, t in module.state_dict().items() if not any(k.endswith(s) for s in skip)}
    # state_dict() omits persistent=False buffers (the text towers' position_ids / causal mask, conditioner.py): on a meta-constructed
    # replica those are uninitialised memory after to_empty() -- they travel too.  Same order on every rank: named_buffers()
    # follows module registration order, which the shared factory fixes.
    have = {id(t) for t in state.values()}
    for k, t in module.named_buffers():
        if id(t) not in have and not any(k.endswith(s) for s in skip):
            state["<buffer>." + k] = t
            have.add(id(t))
    for t in state.values():
        if not t.is_floating_point():
            w = t.to(comm)
            dist.broadcast(w, src=src)
            if w is not t:
                with torch.no_grad():
                    t.copy_(w)
    sent = 0
    split = {}
    for t in state.values():
        if not t.is_floating_point():
            continue
        rounded = payload_dtype is not None and t.dtype == torch.float32 and t.dim() >= 2 and t.numel() >= round_min_elems
        split.setdefault((t.dtype, payload_dtype if rounded else t.dtype), []).append(t)
    for (dt, wire_dt), ts in split.items():
        bucket, size = [], 0
        for t in ts + [None]:
            if t is None or (bucket and size + t.numel() > bucket_elems):
                flat = torch.cat([b.reshape(-1).to(device=comm, dtype=wire_dt) for b in bucket])
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


def construct_replica(factory, device, materialize):
    """Build the module `factory()` returns on `device`.

    materialize=True (the rank that owns the weights): an ordinary construction on the device.
    materialize=False (every other rank of a replicated job): construct on the META device -- no allocation, no initialiser kernels,
    no host RAM -- then `to_empty(device)`: parameters AND buffers exist uninitialised and must ALL be received
    (`broadcast_module_(module, skip=())`: buffers computed in constructors, e.g. the denoiser's sigma table, are not recomputed on
    this rank; non-persistent buffers travel too).  What cannot travel is a plain tensor ATTRIBUTE a constructor computes (neither parameter
    nor buffer): none may be left on meta -- asserted here, so such a module fails at construction instead of computing on garbage.
    SURVEY 8(e): weights replicated, one broadcast; this keeps rank > 0 start-up at allocation + receive."""
    if materialize:
        with torch.device(device):
            return factory()
    with torch.device("meta"):
        module = factory()
    module = module.to_empty(device=device)
    for name, sub in module.named_modules():
        for attr, val in vars(sub).items():
            if isinstance(val, torch.Tensor) and val.device.type == "meta":
                raise RuntimeError(f"construct_replica: {name or type(sub).__name__}.{attr} is a constructor-computed tensor attribute left on the "
                                   "meta device (register it as a buffer so that the weight broadcast carries it)")
    return module


def shard_items(n_items, rank=None, world=None):
    """Image i -> rank i % world (round robin, independent units, zero exchange)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
    return list(range(rank, n_items, world))


def max_over_ranks(seconds, device="cpu"):
    if not collectives_active():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_images(local, n_items, device="cpu"):
    """Optional result gather: {index: [3,H,W] tensor} from every rank -> list on rank 0 (12 MB per 1024^2 image)."""
    if not collectives_active():
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
    if not collectives_active(group):
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
