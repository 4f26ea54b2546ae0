"""Tensor-level wrappers over the C ABI (raw device pointers out of torch tensors; torch is only the allocator/stream).

Internal activation layout: bf16, channels last.  A feature map is a tensor of logical shape [B, H, W, C] whose last
dim is contiguous and whose pixel stride `ld` (>= C) is uniform (so channel slices of a wider buffer are valid
operands: this is how `torch.cat(dim=1)` of the reference disappears).  Token tensors are [B, T, C] / [M, C].
"""
import ctypes as _ct
import math

import torch

from . import _lib
from .weights import HALF_TYPES
from . import ops_f32 as _f32   # the fp32 forms (libsupir_hip_f32.so): every operator below hands fp32 operands to its namesake there

BF16 = torch.bfloat16


def _k(dtype):
    """Suffix for autotune / choice keys: the fp16 library's kernels are timed and cached separately from the bf16 ones (the bf16
    keys stay exactly what they were)."""
    return () if dtype == BF16 else ("f32",) if dtype == torch.float32 else ("f16",)

# --------------------------------------------------------------------------------------------- op trace (bench / roofline)
_TRACE = None


def stop_trace():
    global _TRACE, _TIMED
    t, _TRACE, _TIMED = _TRACE, None, False
    return t


_TIMED = False


def start_trace(timed=False):  # noqa: F811
    """Record every kernel launch (class, algorithmic FLOPs / bytes, shape).  timed=True additionally brackets each launch
    with HIP events on the launch stream (bench.py's live per-kernel durations)."""
    global _TRACE, _TIMED
    _TRACE, _TIMED = [], bool(timed)
    return _TRACE


def _ev():
    if _TRACE is None or not _TIMED:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _rec(kernel, flops, bytes_, ev=None, **shape):
    if _TRACE is not None:
        r = dict(kernel=kernel, flops=float(flops), bytes=float(bytes_), **shape)
        if ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            r["_ev"] = (ev, e1)
        _TRACE.append(r)


def finish_timing(trace):
    """After a synchronize: replace the event pairs by `us` (microseconds)."""
    for r in trace:
        ev = r.pop("_ev", None)
        if ev is not None:
            r["us"] = ev[0].elapsed_time(ev[1]) * 1e3
    return trace


def gemm_tile_name(M, N, act=0, conv=False, trans=False, tile=-1, group=1):
    """Kernel instantiation a GEMM-family trace record ran on; group = 2: the two-problem grouped form (",x2")."""
    g = ",x2" if group == 2 else ""
    if tile == 37:
        return f"geglu_big_kernel<256,320,4x2{g}>"
    if tile is not None and tile >= 64:      # tap-split convolution (supir_conv3x3_bf16_splitk + supir_splitk_finalize)
        name = ["128,128,2x2", "128,64,2x2", "64,128,2x2", "64,64,2x2"][tile - 64]
        return f"gemm_bf16_kernel<{name},conv,split9>+finalize"
    if tile is not None and tile >= 32:
        nm = {32: "128,80,2k,s2", 33: "128,160,2k,s2", 34: "256,160,1k,s3", 35: "128,80,2k,s3", 36: "256,160,1k,s3,qkv",
              38: "128,80,1k,s3,4w", 39: "256,128,1k,s3", 40: "256,256,1k,s2", 41: "256,128,1k,s3,qkv", 42: "256,256,1k,8ph", 45: "512,128,1k,8ph",
              48: "128,80,2k,s3,halo32", 49: "128,160,2k,s2,halo64", 50: "256,160,1k,s3,halo32", 51: "256,160,1k,s2,halo64"}[tile]
        return f"gemm16_kernel<{nm}{',conv' if conv else ''}{',T' if trans else ''}{g}>"
    t = (tile & 7) if tile is not None and tile >= 0 else _lib.load().supir_gemm_tile_for(M, N, act)
    name = ["128,128,2x2", "128,64,2x2", "64,128,2x2", "64,64,2x2", "256,128,4x2", "256,256,2x4", "256,128,2x2",
            "128,128,2x2x2k"][t]
    return f"gemm_bf16_kernel<{name},{'conv' if conv else 'plain'}{',T' if trans else ''}>"


# --------------------------------------------------------------------------------------------- helpers
def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _rows_ld(t):
    """(rows, C, ld) of a channels-last tensor whose leading dims are dense in units of the pixel stride."""
    assert t.stride(-1) == 1, "channel dim must be contiguous"
    C = t.shape[-1]
    if t.dim() == 1:
        return 1, C, C
    ld = t.stride(-2)
    rows = 1
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1:
            assert t.stride(d) == exp, f"non-uniform pixel stride {t.shape} {t.stride()}"
        exp *= t.shape[d]
        rows *= t.shape[d]
    return rows, C, ld


def _check_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SupirHipError("supir_amd ops need CUDA(HIP) tensors: the product path has no CPU fallback")


_WS = {}


def _gn_workspace(B, device):
    key = (B, device, torch.cuda.current_stream().cuda_stream)  # per stream: GroupNorms may run concurrently
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(B * 1024 * 64, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


# --------------------------------------------------------------------------------------------- tile autotuning
# The GEMM kernel has four tile shapes; which one wins depends on how the (M, N) grid quantises onto 256 CUs and on K.
# "Measure, don't guess": the first time a problem shape is seen (outside graph capture) every candidate is timed with HIP
# events on the launch stream and the winner is cached for the life of the process.  All tiles accumulate K in the same
# order, so the choice never changes a bit of a GEMM's own output; the row statistics a producer GEMM emits for a folded
# LayerNorm are summed per column tile, so THEIR fp32 summation order (and, through bf16 rounding flips, the network output at
# its noise floor) can differ between two processes that picked different tiles.  Within a process results are reproducible.
import os as _os

_TUNE = {}
_CHOICE = {}
AUTOTUNE = _os.environ.get("SUPIR_AUTOTUNE", "1") != "0"
# Winners are persisted: supir_amd/tune_gfx950.json ships the picks of one MI355X box for every shape of the five BASELINE configs
# (tools/make_tune.py) and is loaded at import, so that tests, bench and profiler runs execute the SAME kernels in every process --
# timing-based picks otherwise differ between processes wherever two candidates are within noise, and with them the fp32 summation
# order of the folded-LayerNorm statistics (bf16 rounding flips at the noise floor).  Shapes the file does not know are still tuned
# on first sight.  SUPIR_TUNE_FILE=<path> uses another file, SUPIR_TUNE_FILE=none starts empty (re-tune everything on this box).
_TUNE_DEFAULT = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tune_gfx950.json")
_TUNE_FILE = _os.environ.get("SUPIR_TUNE_FILE", _TUNE_DEFAULT)


def _packaged_tune_applies():
    """The packaged picks were timed on gfx950: on any other device (or an unknown one) every shape is tuned on first sight."""
    try:
        if torch.cuda.is_available():
            return str(getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "")).startswith("gfx950")
    except Exception:
        return False
    return True     # no device in this process (CPU tier): nothing is launched, the entries are inert


def load_tuning(path):
    """Merge a file written by save_tuning into the autotune state; returns the number of entries read."""
    import json as _json
    data = _json.load(open(path))
    if isinstance(data, list):      # round-2 format: tile picks only
        data = {"tune": data, "choice": []}
    _TUNE.update({tuple(k): v for k, v in data.get("tune", [])})
    _CHOICE.update({tuple(k): v for k, v in data.get("choice", [])})
    return len(data.get("tune", [])) + len(data.get("choice", []))


# The file is read at import (plain file I/O).  Whether the PACKAGED picks apply to this device is decided at the first autotune
# lookup, not here: asking the device for its architecture is a HIP call, and a HIP call as an import side effect breaks fork-based
# multiprocessing ("Cannot re-initialize CUDA in forked subprocess") and defeats environment settings that must precede the first HIP
# call of the process.  On a device the picks were not timed for, the packaged entries are dropped then and every shape is tuned.
_PACKAGED = (set(), set())
_ARCH_CHECKED = False
if _TUNE_FILE and _TUNE_FILE.lower() != "none" and _os.path.exists(_TUNE_FILE):
    _before = (set(_TUNE), set(_CHOICE))
    load_tuning(_TUNE_FILE)
    if _TUNE_FILE == _TUNE_DEFAULT:
        _PACKAGED = (set(_TUNE) - _before[0], set(_CHOICE) - _before[1])


def _check_packaged_tune():
    global _ARCH_CHECKED
    if _ARCH_CHECKED:
        return
    _ARCH_CHECKED = True
    if (_PACKAGED[0] or _PACKAGED[1]) and not _packaged_tune_applies():
        for k in _PACKAGED[0]:
            _TUNE.pop(k, None)
        for k in _PACKAGED[1]:
            _CHOICE.pop(k, None)


def save_tuning(path=None):
    """Write the autotune state to `path`, or -- with no argument -- to the file SUPIR_TUNE_FILE names EXPLICITLY.  Never to the
    packaged supir_amd/tune_gfx950.json by default: a tool run must not rewrite the product's picks behind the user's back
    (tools/make_tune.py passes that path on purpose)."""
    import json as _json
    path = path or _os.environ.get("SUPIR_TUNE_FILE")
    if path and path.lower() != "none":
        data = {"tune": sorted(([list(k), v] for k, v in _TUNE.items()), key=repr),
                "choice": sorted(([list(k), v] for k, v in _CHOICE.items()), key=repr)}
        _json.dump(data, open(path, "w"), indent=0)


def _time_options(options, run, repeat=None):
    """Time `run(option)` for every option (HIP events on the launch stream; min over three short rounds: one round of 5 launches
    picked a 15 % slower tile now and then).  An option whose first launch returns an error code is dropped -- the C side refuses
    shapes / alignments the Python predicates may not know about, and a refused launch would otherwise time at ~0 us and win.
    repeat(option, n), when given, issues the option n times back to back (options that span two streams fork / join once per
    round, not once per launch).  Returns [(ms, option)] of the options that ran."""
    if repeat is None:
        def repeat(t, n):
            for _ in range(n):
                run(t)
    times = []
    for t in options:
        if run(t) != 0:
            continue
        repeat(t, 2)
        best_t = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            repeat(t, 6)
            e1.record()
            e1.synchronize()
            dt = e0.elapsed_time(e1)
            best_t = dt if best_t is None or dt < best_t else best_t
        times.append((best_t, t))
    return times


def _autotune(key, candidates, launch):
    _check_packaged_tune()
    best = _TUNE.get(key)
    if best is not None:
        return best
    if not AUTOTUNE or _DEFER is not None or torch.cuda.is_current_stream_capturing():
        return -1
    times = _time_options(candidates, launch)
    best = min(times)[1] if times else -1
    _TUNE[key] = best
    return best


# --------------------------------------------------------------------------------------------- weight prefetch plan
class WeightPrefetch:
    """Record the order in which weight matrices are consumed during one network call; while the same call is replayed
    (normally: captured into a hipGraph) make the launch of op i pull the weight of op i+distance towards the caches.

    mode "inline" (default): supir_launch_hints.next_weight of the launch -- op i's own GEMM kernel touches the later weight after its last
    store; no extra launches, streams or graph edges.  mode "stream": a `supir_prefetch` launch per op on a dedicated stream
    behind an event edge (kept for tools/cold_probe.py; measured slower end to end: +1168 graph nodes per step)."""

    def __init__(self, distance=1, kind="inline"):
        self.distance = distance
        self.kind = kind
        self.plan = []
        self.mode = None
        self.idx = 0
        self.stream = None
        self.sink = None

    def begin_record(self):
        self.plan, self.mode, self.idx = [], "record", 0

    def begin_replay(self, device):
        self.mode, self.idx = "replay", 0
        if self.kind != "stream":
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=device)
            self.sink = torch.zeros(4, dtype=torch.int32, device=device)
        self.stream.wait_stream(torch.cuda.current_stream())
        lib = _lib.load()
        for j in range(min(self.distance, len(self.plan))):      # prime the first `distance` weights
            ptr, nb = self.plan[j][0]
            lib.supir_prefetch(ptr, nb, self.sink.data_ptr(), self.stream.cuda_stream)

    def end(self):
        if self.mode == "replay":
            if self.kind == "stream" and self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)     # join (required inside a capture)
        self.mode = None

    def touch_group(self, ws):
        """One launch consuming the weight matrices `ws` (one per problem of a grouped launch).  Recording: append the group to the
        plan.  Replaying: return, per problem, the (pointer, bytes) of the weight the group `distance` launches later consumes
        (problem k prefetches that group's k-th weight), or None."""
        cur = tuple((w.data_ptr(), w.numel() * w.element_size()) for w in ws)
        none = [None] * len(ws)
        if self.mode == "record":
            self.plan.append(cur)
            return none
        if self.mode != "replay":
            return none
        i = self.idx
        self.idx += 1
        # wrap around: the last ops of a step prefetch the first weights of the next step (same graph replayed 50 times)
        if i >= len(self.plan) or [q[0] for q in self.plan[i]] != [q[0] for q in cur]:
            return none
        if self.kind == "stream" and i + self.distance >= len(self.plan):
            return none
        nxt = self.plan[(i + self.distance) % len(self.plan)]
        return [nxt[k] if k < len(nxt) else None for k in range(len(ws))]

    def touch(self, w):
        """The (pointer, bytes) this launch should touch on its way out (inline kind: travels with the launch as
        supir_launch_hints.next_weight), or None (nothing planned / the stream kind, which issues its own prefetch launch here)."""
        nxt = self.touch_group([w])[0]
        if nxt is None:
            return None
        ptr, nb = nxt
        if self.kind == "stream":
            ev = torch.cuda.Event()
            ev.record()                       # on the op's own stream: "op i is next"
            self.stream.wait_event(ev)
            _lib.load().supir_prefetch(ptr, nb, self.sink.data_ptr(), self.stream.cuda_stream)
            return None
        return ptr, nb


_PF = None


def set_prefetch(pf):
    global _PF
    _PF = pf


def _pf(w):
    if _PF is not None and _PF.mode is not None:
        return _PF.touch(w)
    return None


def _pf_group(ws):
    if _PF is not None and _PF.mode is not None:
        return _PF.touch_group(ws)
    return [None] * len(ws)


# --------------------------------------------------------------------------------------------- GroupNorm statistics from producers
class GnPart:
    """What a producer GEMM / conv epilogue left behind for a GroupNorm over its output (supir_launch_hints.gn_partials_out): fp32
    [B, nchunk, C // unit, 2] = (sum, sum of squares) per batch, tile row and `unit`-channel unit of the bf16 values it stored.
    unit = 10 from the 80 / 160-column tiles (UNet / control net), 4 from the 128 / 256-column tiles 39 / 40 (VAE)."""
    __slots__ = ("buf", "nchunk", "C", "unit")

    def __init__(self, buf, nchunk, C, unit=10):
        self.buf, self.nchunk, self.C, self.unit = buf, nchunk, C, unit


USE_GN_PARTS = _os.environ.get("SUPIR_GN_PARTS", "1") != "0"   # producers emit GroupNorm statistics where the kernels support it


def _gn_part_alloc(tile, nbatch, rows_per_batch, N, device):
    """The buffer a launch on `tile` fills with GroupNorm partials of its output, if that tile can emit them; returns the GnPart
    or None.  (The request itself travels with the launch: supir_launch_hints.gn_partials_out for a single launch, the
    problem's gn_partials_out field for a grouped one.)"""
    if not USE_GN_PARTS or tile not in _G16:
        return None
    bm = _G16[tile][0]
    unit = 4 if tile in _G16_PLAIN_ONLY else 10     # csrc/gemm16.hip: GU by the tile's column count
    if rows_per_batch <= 0 or rows_per_batch % bm or N % unit:
        return None
    if unit == 4 and ((N // 32) % 4 or N % 32 or _DEFER is not None):
        return None                                   # VAE form: whole groups of 4-channel units; never inside paired_run
    nchunk = rows_per_batch // bm
    # recorded launches (paired_run) may be re-timed on other tiles of the family before they are issued: room for the finest tile rows
    alloc = max(nchunk, rows_per_batch // 128) if _DEFER is not None else nchunk
    buf = torch.empty(nbatch, alloc, N // unit, 2, dtype=torch.float32, device=device)
    return GnPart(buf, nchunk, N, unit)


# --------------------------------------------------------------------------------------------- deferred issue / paired launches
# GLVControl and the encoder half of LightGLVUNet are the same stack of layers on independent data (SUPIR/modules/SUPIR_v0.py:499-540
# next to :600-625).  paired_run() executes the two Python call trees one after the other with every launch RECORDED instead of
# issued (shapes, output allocation and tile choice happen as usual; control flow on this path never depends on tensor values), then
# walks the two launch lists in lockstep and issues each pair of identical problems as ONE grouped launch (supir_gemm_grouped,
# supir_flash_attn_d64_grouped, supir_groupnorm_grouped), anything else one by one in the recorded order.  Per-branch order is
# preserved and the two branches share no tensor, so any interleaving is valid.  Every tensor a recorded launch touches is kept
# alive until the lists have been issued: the caching allocator must not hand a buffer the first branch "freed" while recording to
# the second branch.
_DEFER = None
PAIR = _os.environ.get("SUPIR_PAIR", "1") != "0"            # paired_run groups launches (0: records and issues them one by one)
PAIR_TILES = (33, 34, 35, 37)                               # tiles with a two-problem form (csrc/gemm16.hip, csrc/gemm_big.hip)
PAIR_MIN_GAIN = float(_os.environ.get("SUPIR_PAIR_MIN_GAIN", "0.0"))   # grouped must beat the two overlapped singles by this fraction
PAIR_FORCE = False                                          # tools/pair_ab.py: drop the "two single launches" option from the timing
PAIR_KINDS = set(_os.environ.get("SUPIR_PAIR_KINDS", "gemm,conv,qkv,attn,gn").split(","))   # A/B runs: which kinds may group


class _Launch:
    """One kernel launch, ready to go: `call(tile, out_override)` issues it alone through its own entry point; `make(tile,
    out_override)` returns its (shared shape, per-problem struct) for a grouped launch.  kind None = no grouped form."""
    __slots__ = ("kind", "key", "tkey", "lib", "name", "tile", "single_tile", "cands", "call", "make", "w", "part", "trace",
                 "keep", "out", "inplace")

    def __init__(self, kind, lib, name, call, *, key=None, tkey=None, tile=-1, single_tile=-1, cands=(), make=None, w=None, part=None,
                 trace=None, keep=(), out=None, inplace=False):
        self.kind, self.lib, self.name, self.call = kind, lib, name, call
        self.key, self.tkey, self.tile, self.single_tile, self.cands, self.make = key, tkey, tile, single_tile, cands, make
        self.w, self.part, self.trace, self.keep, self.out, self.inplace = w, part, trace, keep, out, inplace


_NO_PF = object()


def _run_single(L, nxt=_NO_PF):
    # per-launch requests travel WITH the launch (supir_launch_hints of the *_ex entry points): no thread-local state in the library.
    # nxt: a prefetch hint the caller already drew from the WeightPrefetch plan for this launch (the pair fallback below)
    if nxt is _NO_PF:
        nxt = _pf(L.w) if L.w is not None else None
    hints = None
    if nxt is not None or L.part is not None:
        hints = _lib.LaunchHints(next_weight=nxt[0] if nxt else None, next_weight_bytes=nxt[1] if nxt else 0,
                                 gn_partials_out=L.part.buf.data_ptr() if L.part is not None else None)
    ev = _ev()
    rc = L.call(L.tile, None, hints) if hints is not None else L.call(L.tile, None)
    _lib.check(rc, L.name, L.lib)
    if L.trace is not None:
        _rec(L.trace[0], L.trace[1], L.trace[2], ev, **L.trace[3])


def _issue(L):
    if _DEFER is not None:
        _DEFER.append(L)
    else:
        _run_single(L)


def _no_defer(name):
    if _DEFER is not None:
        raise _lib.SupirHipError(f"ops.{name} inside paired_run: this op has no recorded form (it would run out of order)")


def _pair_tile(tkey, tile, cands):
    """Tile of a RECORDED launch: the winner of the pair autotune for this shape once there is one, else the single-launch tile."""
    if _DEFER is None or not PAIR:
        return tile
    pt = _TUNE.get(("pair",) + tkey)
    return pt if (pt is not None and pt >= 32 and pt in cands) else tile


def _scratch_like(t):
    return torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)


def _gemm_group_call(a, b, tile, oa=None, ob=None, prefetch=(None, None)):
    sh, pa = a.make(tile, oa)
    _, pb = b.make(tile, ob)
    probs = (_lib.GemmProblem * 2)(pa, pb)
    for q, pf in zip(probs, prefetch):
        if pf is not None:
            q.prefetch, q.prefetch_bytes = pf
    return a.lib.supir_gemm_grouped(_ct.byref(sh), probs, 2, _stream())


_PAIR_BM_BN = {33: (128, 160), 34: (256, 160), 35: (128, 80), 36: (256, 160), 37: (256, 320)}


def _group_ok(L, tile):
    """A two-problem grid gives each problem four XCDs: its tile count must divide over them (same check as the C side)."""
    if tile not in _PAIR_BM_BN:
        return False
    bm, bn = _PAIR_BM_BN[tile]
    M, N = L.trace[3]["M"], L.trace[3]["N"]
    return M % bm == 0 and N % bn == 0 and ((M // bm) * (N // bn)) % 4 == 0


def _group_call(a, b, tile, oa=None, ob=None, prefetch=(None, None)):
    """One grouped launch of the recorded launches a, b (same kind, same key)."""
    if a.kind in ("gemm", "conv", "qkv"):
        return _gemm_group_call(a, b, tile, oa, ob, prefetch)
    if a.kind == "attn":
        (B, H, Tq, scale), pa = a.make(oa)
        _, pb = b.make(ob)
        return a.lib.supir_flash_attn_d64_grouped((_lib.AttnProblem * 2)(pa, pb), 2, B, H, Tq, scale, _stream())
    (B, HW, C, eps, act), pa = a.make(oa)
    _, pb = b.make(ob)
    return a.lib.supir_groupnorm_grouped((_lib.GnProblem * 2)(pa, pb), 2, B, HW, C, eps, act, _stream())


def _pair_autotune(a, b, pkey, side):
    """Grouped launch (on each tile both problems can take) vs the two single launches on their own tiles (-2) -- the latter on two
    streams when the caller runs the branches that way --, timed in place where the pair sits in the issue order (operands are live;
    outputs of in-place accumulations go to scratch).  Everything a candidate writes is rewritten by the real launch that follows."""
    if a.kind == "qkv":
        opts = ([36] if _group_ok(a, 36) else []) + [-2]
    elif a.kind in ("gemm", "conv"):
        opts = [t for t in PAIR_TILES if t in a.cands and t in b.cands and _group_ok(a, t)] + [-2]
    else:
        opts = [0, -2]
    oa = _scratch_like(a.out) if a.inplace else None
    ob = _scratch_like(b.out) if b.inplace else None
    main = torch.cuda.current_stream()
    # recorded launches skipped their single-launch tuning (_autotune returns -1 while recording): tune them here, where the
    # operands are live, so that the "two singles" option is not timed on heuristic tiles against tuned grouped ones
    for L, o in ((a, oa), (b, ob)):
        if L.kind in ("gemm", "conv") and L.single_tile == -1 and L.cands and L.tkey is not None:
            st = _TUNE.get(L.tkey)
            if st is None:
                times = _time_options(L.cands, lambda t, L=L, o=o: L.call(t, o))
                st = min(times)[1] if times else -1
                _TUNE[L.tkey] = st
            if st in L.cands:
                L.single_tile = st

    def run(t):
        if t == -2:
            rc = a.call(a.single_tile, oa)
            return rc if rc != 0 else b.call(b.single_tile, ob)
        return _group_call(a, b, t, oa, ob)

    def repeat(t, n):
        if t != -2 or side is None:
            for _ in range(n):
                run(t)
            return
        side.wait_stream(main)
        for _ in range(n):
            a.call(a.single_tile, oa)
        with torch.cuda.stream(side):
            for _ in range(n):
                b.call(b.single_tile, ob)
        main.wait_stream(side)

    if PAIR_FORCE and len(opts) > 1:
        opts = opts[:-1]
    times = _time_options(opts, run, repeat)
    best = min(times)[1] if times else -2
    if best != -2 and PAIR_MIN_GAIN > 0.0:
        # a grouped launch is a join point of the two chains (cross-queue edges in the captured graph): it has to win by a margin
        sep = [ms for ms, t in times if t == -2]
        if sep and min(times)[0] > (1.0 - PAIR_MIN_GAIN) * sep[0]:
            best = -2
    _TUNE[pkey] = best
    return best


def _pair_choice(a, b, side, join):
    """How to run the recorded launches a, b (identical problems of the two branches): the tile of a grouped launch (0 for kinds
    without tiles), or None = one by one.  Timed once per shape (pair autotune; `join()` first: the candidates run on the current
    stream and both branches' operands must be there)."""
    if not PAIR or a.kind not in PAIR_KINDS:
        return None
    gemm_like = a.kind in ("gemm", "conv", "qkv")
    if gemm_like and a.kind != "qkv" and not a.cands and not b.cands:
        pt = a.tile if a.tile in PAIR_TILES else -2      # tiles forced by the caller: group on that tile, no timing
    else:
        pkey = ("pair",) + (a.tkey if gemm_like else a.key)
        pt = _TUNE.get(pkey)
        if pt is None and AUTOTUNE and not torch.cuda.is_current_stream_capturing():
            join()
            pt = _pair_autotune(a, b, pkey, side)
    if pt is None or pt == -2:
        return None
    if gemm_like and (a.tile != b.tile or a.tile != pt or not _group_ok(a, pt)):
        return None   # (a winner found after this pass was recorded is used from the next pass on: tile-dependent buffers)
    return pt


def _run_pair(a, b, tile):
    gemm_like = a.kind in ("gemm", "conv", "qkv")
    pfs = _pf_group([a.w, b.w]) if gemm_like else (None, None)
    ev = _ev()
    rc = _group_call(a, b, tile, prefetch=pfs)
    if rc in (-1, -2):
        # SUPIR_ERR_ARG / SUPIR_ERR_SHAPE: the C side refuses a pairing the Python key could not tell apart (a cached winner for a
        # key that under-describes the problems): nothing was launched -- run the two problems one by one, WITH the prefetch hints
        # already drawn for them (drawing again would advance the record / replay cursor twice and shift every later hint of the
        # pass), and do not retry the pairing on later steps
        _run_single(a, pfs[0])
        _run_single(b, pfs[1])
        _TUNE[("pair",) + (a.tkey if gemm_like else a.key)] = -2      # the key _pair_choice looks the pairing up under
        return
    _lib.check(rc, {"attn": "supir_flash_attn_d64_grouped", "gn": "supir_groupnorm_grouped"}.get(a.kind, "supir_gemm_grouped"), a.lib)
    if a.trace is not None:
        k, fl, by, kw = a.trace
        _rec(k, 2 * fl, 2 * by, ev, **dict(kw, group=2))


def paired_run(fn_a, fn_b, side=None):
    """Run fn_a() and fn_b() -- two independent branches built from the ops of this module -- with their launches recorded, then
    issue the two launch lists in lockstep: identical problems as ONE grouped launch on the current stream where that is the faster
    way to run them, everything else one by one -- branch A on the current stream and, when a `side` stream is given, branch B on
    that one (two half-machine launches overlap; a grouped launch is a join point of the two chains: in a captured hipGraph these
    are just edges).  On return the current stream is ordered after everything issued.  Returns (fn_a(), fn_b())."""
    global _DEFER
    assert _DEFER is None, "paired_run does not nest"
    lists = []
    results = []
    for fn in (fn_a, fn_b):
        _DEFER = cur = []
        try:
            results.append(fn())
        finally:
            _DEFER = None
        lists.append(cur)
    la, lb = lists
    main = torch.cuda.current_stream()
    state = {"b_on_side": False}   # which stream the tail of branch B's chain is on

    def join():
        if state["b_on_side"]:
            main.wait_stream(side)
            state["b_on_side"] = False

    for i in range(max(len(la), len(lb))):
        a = la[i] if i < len(la) else None
        b = lb[i] if i < len(lb) else None
        if a is not None and b is not None and a.kind is not None and a.kind == b.kind and a.key == b.key and a.lib is b.lib:
            tile = _pair_choice(a, b, side, join)
            if tile is not None:
                join()
                _run_pair(a, b, tile)
                continue
        if b is not None and side is not None and not state["b_on_side"]:
            side.wait_stream(main)       # B's next launch follows B's previous one (on main so far); taken BEFORE a is issued
            state["b_on_side"] = True
        if a is not None:
            _run_single(a)
        if b is not None:
            if state["b_on_side"]:
                with torch.cuda.stream(side):
                    _run_single(b)
            else:
                _run_single(b)
    join()
    return results[0], results[1]


# --------------------------------------------------------------------------------------------- GEMM family
# GEGLU arithmetic (include/supir_hip.h): act code 2 = SUPIR_ACT_GEGLU, the fitted GELU (|error| <= 2.5e-5, the speed default); code 5 =
# SUPIR_ACT_GEGLU_ERF, the reference's erf (sgm/modules/attention.py:89-91).  SUPIR_EXACT_GELU=1 (or ops.EXACT_GELU = True before the
# first call / graph capture) routes every GEGLU launch of the module layer to the erf form -- an argument of each launch, not a state
# of the library; same tiles, same layouts.
EXACT_GELU = _os.environ.get("SUPIR_EXACT_GELU", "0") == "1"


def _abi_act(act):
    return 5 if (act == 2 and EXACT_GELU) else act


def gemm(a, w, bias=None, *, rowbias=None, rows_per_batch=0, residual=None, act=0, alpha=1.0, out=None,
         out_dtype=None, tile=-1, alt16=None, gn_part=False):
    """out[M,N] = alpha*act(a[M,K] @ w[N,K]^T + bias + rowbias[batch]) + residual.  act=2 (GEGLU) -> N/2 columns.
    alt16 = (w, bias) in the 16-row GEGLU interleave: lets the autotuner also try tile 34 (csrc/gemm16.hip).
    gn_part=True: returns (out, GnPart or None) -- GroupNorm statistics of `out` per batch of `rows_per_batch` rows, when the tile
    that runs can emit them.  The element type (bf16 or fp16) is that of `a`; every 16-bit operand must share it."""
    DT = a.dtype
    if DT == torch.float32:
        return _f32.gemm(a, w, bias, rowbias=rowbias, rows_per_batch=rows_per_batch, residual=residual, act=act, alpha=alpha, out=out,
                         out_dtype=out_dtype, gn_part=gn_part)
    lib = _lib.load(DT)
    _check_dev(a, w)
    M, K, lda = _rows_ld(a)
    N = w.shape[0]
    assert DT in HALF_TYPES and w.shape[1] == K and w.is_contiguous() and w.dtype == DT
    n_out = N // 2 if act == 2 else N
    if out is None:
        out = torch.empty(*a.shape[:-1], n_out, dtype=DT if out_dtype is None else out_dtype, device=a.device)
    Mo, No, ldc = _rows_ld(out)
    assert Mo == M and No == n_out and out.dtype in (DT, torch.float32)
    ldr = 0
    if residual is not None:
        Mr, Nr, ldr = _rows_ld(residual)
        assert Mr == M and Nr == n_out and residual.dtype == DT
    ld_rb = 0
    if rowbias is not None:
        assert rowbias.dtype == DT and rowbias.stride(-1) == 1 and rowbias.shape[-1] == N
        ld_rb = rowbias.stride(0)
    om = 0 if out.dtype == DT else 1
    if gn_part and rows_per_batch <= 0:
        rows_per_batch = M   # one batch: the partials' row-chunk index is computed from it (never 0 in the kernel)

    def wb(t):
        return (alt16[0], alt16[1]) if (t in (34, 37) and act == 2 and alt16 is not None) else (w, bias)

    def launch(t, outp=None, hints=None):
        wq, bq = wb(t)
        return lib.supir_gemm_bf16_ex(a.data_ptr(), wq.data_ptr(), (out if outp is None else outp).data_ptr(), M, N, K, lda, ldc,
                                      _p(bq), _p(rowbias), ld_rb, rows_per_batch, _p(residual), ldr, _abi_act(act), om, alpha, t,
                                      None if hints is None else _ct.byref(hints), _stream())

    inplace = residual is not None and residual.data_ptr() == out.data_ptr()
    key = ("gemm", M, N, K, act, om) + _k(DT)
    cands = ()
    single_tile = tile
    if tile == -1:
        ok = (lda % 8 == 0 and out.data_ptr() % 16 == 0 and (residual is None or ldr % 4 == 0)
              and (rowbias is None or ld_rb % 4 == 0))
        big_ok = alpha == 1.0 and a.data_ptr() % 16 == 0 and (alt16 is None or alt16[0].data_ptr() % 16 == 0)
        cands = _gemm_candidates(M, N, K, act, om, ldc, epilogue_ok=ok, geglu16=alt16 is not None, big_ok=big_ok)
        if inplace:
            # in-place accumulate (x += f(x)): re-launching would change the data, so the candidates are timed into a scratch
            # output of the same strides (reads `residual`, never writes it)
            tile = _TUNE.get(key)
            if tile is None:
                if _DEFER is None:
                    scratch = _scratch_like(out)
                    tile = _autotune(key, cands, lambda t: launch(t, scratch))
                else:
                    tile = -1
        else:
            tile = _autotune(key, cands, launch)
        if tile >= 32 and tile not in cands:   # a winner cached for this shape under friendlier strides / layouts
            tile = -1
        single_tile = tile
        tile = _pair_tile(key, tile, cands)
    part = None
    if gn_part and om == 0 and act != 2:
        part = _gn_part_alloc(tile, M // rows_per_batch, rows_per_batch, N, a.device)

    def make(t, outp=None):
        wq, bq = wb(t)
        sh = _lib.GemmShape(kind=_lib.GROUP_GEMM, tile=t, M=M, N=N, K=K, rows_per_batch=rows_per_batch, act=_abi_act(act), out_mode=om, alpha=alpha)
        pr = _lib.GemmProblem(A=a.data_ptr(), W=wq.data_ptr(), C=(out if outp is None else outp).data_ptr(), bias=_p(bq),
                              rowbias=_p(rowbias), residual=_p(residual), gn_partials_out=None if part is None else part.buf.data_ptr(),
                              lda=lda, ldc=ldc, ldr=ldr, ld_rowbias=ld_rb)
        return sh, pr

    _issue(_Launch("gemm", lib, "supir_gemm_bf16", launch, key=key + (alpha, rows_per_batch, part is not None), tkey=key, tile=tile,
                   single_tile=single_tile, cands=cands, make=make, w=w, part=part, out=out, inplace=inplace,
                   trace=("gemm", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * n_out), dict(M=M, N=N, K=K, act=act, tile=tile)),
                   keep=(a, w, bias, rowbias, residual, out, alt16, part)))
    return (out, part) if gn_part else out


_TILE_BN_WN = {0: (128, 2), 1: (64, 2), 2: (128, 2), 3: (64, 2), 4: (128, 2), 5: (256, 4), 6: (128, 2), 7: (128, 2),
               32: (80, 1), 33: (160, 2), 34: (160, 1), 35: (80, 1), 37: (320, 2), 38: (80, 1), 39: (128, 2), 40: (256, 2), 42: (256, 4), 45: (128, 2),
               48: (80, 1), 49: (160, 2), 50: (160, 2), 51: (160, 2)}
G16_TILES = {32, 33, 34, 35, 39, 40, 42, 45, 48, 49, 50, 51}   # enabled members of the family (tools/step_ab.py switches them for A/B runs)
# tile: (BM, BN, K groups, ring).  38 = 128 x 80 with FOUR waves and a 78 KB ring (two workgroups per CU): known to the mirror, enabled
# by adding it to G16_TILES (tools/archive/step_ab4.py in the history at commit 5dd2326); not in the default lists -- see docs/roundlog.md section 3 for what it measured
# 39 / 40 = 256 x 128 and 256 x 256 (round 4): the VAE's 128 / 256 / 512-channel layers; ordinary epilogue only, and offered only where no
# 80-column tile fits (N % 80 != 0), so the candidate lists -- and with them the picks -- of the UNet's shapes are what they were
_G16 = {32: (128, 80, 2, 2), 33: (128, 160, 2, 2), 34: (256, 160, 1, 3), 35: (128, 80, 2, 3), 38: (128, 80, 1, 3),
        39: (256, 128, 1, 3), 40: (256, 256, 1, 2), 42: (256, 256, 1, 3), 45: (512, 128, 1, 3),
        48: (128, 80, 2, 3), 49: (128, 160, 2, 2), 50: (256, 160, 1, 3), 51: (256, 160, 1, 2)}
# 48-51 (round 6): the LDS-staged HALO form of the stride-1 3 x 3 convolutions (csrc/gemm16.hip): a tile is whole rows of a map of the
# width below; per 64-channel chunk its (rows + 2) x (W + 2) input pixels are staged once and the nine taps read them at shifted LDS
# addresses, K order (chunk, tap).  Convolutions only
_G16_HALO_W = {48: 32, 49: 64, 50: 32, 51: 64}
# 42 / 45 = 256 x 256 / 512 x 128 on the eight-phase ping-pong schedule (round 5; ring column = 3: they need at least two K-tiles);
# ordinary epilogue only; convolutions: whole tiles inside one batch element (OH * OW % BM == 0).
# (The same schedule as 256 x 160 and as a 256 x 320 GEGLU tile measured slower than / equal to tiles 34 / 37: not built.)
_G16_NO_TRANS = (39, 40, 42, 45)
_G16_PLAIN_ONLY = (39, 40, 42, 45)
USE_GEMM16 = _os.environ.get("SUPIR_GEMM16", "1") != "0"   # tiles 32 / 33 (csrc/gemm16.hip) in the autotune lists
USE_GEMM_BIG = _os.environ.get("SUPIR_GEMM_BIG", "1") != "0"   # tile 37 (csrc/gemm_big.hip) in the GEGLU autotune lists
USE_CONV_SPLIT = _os.environ.get("SUPIR_CONV_SPLIT", "1") != "0"   # tap-split conv3x3 (codes 64 + tile) in the autotune lists of small-grid convs


def _gemm_candidates(M, N, K, act, om, ldc, ln_slots=0, epilogue_ok=True, geglu16=False, big_ok=True):
    """Tile candidates for the autotuner.  Tiles 32-35 (csrc/gemm16.hip: 16x16x32 MFMA, tile grids that are exact multiples of
    the 256 CUs) take exact shapes only -- the same predicate as supir_gemm16_supported.  act = 2 (GEGLU) can use tile 34 when the
    caller also supplied the 16-row-interleaved weight layout (`geglu16`)."""
    base = (0, 2, 4, 5, 6) if act == 2 else (0, 1, 2, 3, 4, 5, 6)
    if not USE_GEMM16 or om == 1 or ln_slots > 32 or not epilogue_ok:
        return base
    if (om == 2 and ldc % 4) or (om == 0 and ldc % 8):
        return base
    extra = []
    for t, (bm, bn, ks, s) in _G16.items():
        if t not in G16_TILES or t in _G16_HALO_W:
            continue
        if M % bm or N % bn or K % (64 * ks) or (K // 64) // ks < s - 1:
            continue
        if act == 2 and not (t == 34 and geglu16):
            continue
        if t in _G16_PLAIN_ONLY and (om != 0 or N % 80 == 0):
            continue
        if t in _G16_NO_TRANS and om != 0:
            continue
        extra.append(t)
    # tile 37 (csrc/gemm_big.hip): 256 x 320, GEGLU only; same predicate as supir_gemm_big_supported (big_ok: alpha == 1 and 16-byte
    # aligned A / W / C, checked by the caller)
    if act == 2 and geglu16 and USE_GEMM_BIG and big_ok and om == 0 and M % 256 == 0 and N % 320 == 0 and K % 64 == 0 and K >= 128 and ln_slots <= 64:
        extra.append(37)
    return base + tuple(extra)


class RowStats:
    """Per-row statistics a producer GEMM left behind for the LayerNorm folded into its consumers: either partial
    (sum, sum of squares) per column tile [M, ld, 2] with `slots` valid slots, or (slots == 0) finalised (mean, rstd) [M, 2]."""
    __slots__ = ("buf", "slots", "ld")

    def __init__(self, buf, slots, ld):
        self.buf, self.slots, self.ld = buf, slots, ld


def rowstats_finalize(st, dim, eps):
    """partials -> (mean, rstd) per row, one tiny launch shared by all consumers of the same LayerNorm."""
    if st.slots == 0:
        return st
    lib = _lib.load()
    M = st.buf.shape[0]
    out = torch.empty(M, 2, dtype=torch.float32, device=st.buf.device)

    def launch(t, outp=None):
        return lib.supir_rowstats_finalize(st.buf.data_ptr(), out.data_ptr(), M, st.ld, st.slots, dim, eps, _stream())

    _issue(_Launch(None, lib, "supir_rowstats_finalize", launch, trace=("rowstats_finalize", 0, 8.0 * M * (st.slots + 1), {}),
                   keep=(st.buf, out)))
    return RowStats(out, 0, 0)


def gemm_ln(a, w, bias=None, *, residual=None, act=0, alpha=1.0, out=None, tile=-1, emit_stats=False, ln=None, colsum=None,
            ln_eps=1e-5, trans=None, alt16=None):
    """GEMM with LayerNorm folding (see supir_gemm_bf16_ln).  ln = RowStats of `a` (consumer side), emit_stats=True makes
    this call a producer and returns (out, RowStats).  trans=(B, T, Tpad) selects the transposed (V^T) output.
    alt16 = (w, colsum, bias) in the 16-row GEGLU interleave (act = 2): lets the autotuner also try tile 34."""
    DT = a.dtype
    lib = _lib.load(DT)
    _check_dev(a, w)
    M, K, lda = _rows_ld(a)
    N = w.shape[0]
    assert DT in HALF_TYPES and w.shape[1] == K and w.is_contiguous() and w.dtype == DT
    n_out = N // 2 if act == 2 else N
    if trans is not None:
        B, T, Tpad = trans
        assert M == B * T
        if out is None:
            if Tpad != T:
                _no_defer("gemm_ln(trans with padding)")
            out = torch.zeros(B, N, Tpad, dtype=DT, device=a.device) if Tpad != T else \
                torch.empty(B, N, Tpad, dtype=DT, device=a.device)
        ldc, om, rpb = Tpad, 2, T
    else:
        if out is None:
            out = torch.empty(*a.shape[:-1], n_out, dtype=DT, device=a.device)
        _, _, ldc = _rows_ld(out)
        om, rpb = 0, 0
    ldr = 0
    if residual is not None:
        _, _, ldr = _rows_ld(residual)
    stats = None
    rs_ld = 0
    if emit_stats:
        rs_ld = (N + 63) // 64          # finest column tile is 64 wide
        rs_ld += rs_ld & 1              # even: 16-byte aligned rows for the consumer's vector loads
        stats = torch.empty(M, rs_ld, 2, dtype=torch.float32, device=a.device)
    ln_p, ln_ld, ln_slots = 0, 0, 0
    if ln is not None:
        ln_p, ln_ld, ln_slots = ln.buf.data_ptr(), ln.ld, ln.slots
        assert colsum is not None and colsum.numel() == N

    def wcb(t):
        return alt16 if (t in (34, 37) and act == 2 and alt16 is not None) else (w, colsum, bias)

    def launch(t, outp=None, hints=None):
        wq, cq, bq = wcb(t)
        return lib.supir_gemm_bf16_ln_ex(a.data_ptr(), wq.data_ptr(), (out if outp is None else outp).data_ptr(), M, N, K, lda, ldc,
                                         _p(bq), _p(residual), ldr, _abi_act(act), om, rpb, alpha, t, _p(stats), rs_ld, ln_p, ln_ld, ln_slots,
                                         _p(cq), ln_eps, None if hints is None else _ct.byref(hints), _stream())

    inplace = residual is not None and residual.data_ptr() == out.data_ptr()
    key = ("gemm", M, N, K, act, om) + _k(DT)
    cands = ()
    single_tile = tile
    if tile == -1:
        ok = (lda % 8 == 0 and out.data_ptr() % 16 == 0 and (residual is None or ldr % 4 == 0)
              and (trans is None or rpb % 4 == 0))
        big_ok = alpha == 1.0 and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and (alt16 is None or alt16[0].data_ptr() % 16 == 0)
        cands = _gemm_candidates(M, N, K, act, om, ldc, ln_slots=ln_slots, epilogue_ok=ok, geglu16=alt16 is not None, big_ok=big_ok)
        if inplace:
            tile = _TUNE.get(key)
            if tile is None:   # in-place accumulate: time the candidates into a scratch output (see gemm())
                if _DEFER is None:
                    scratch = _scratch_like(out)
                    tile = _autotune(key, cands, lambda t: launch(t, scratch))
                else:
                    tile = -1
        else:
            tile = _autotune(key, cands, launch)
        if tile >= 32 and tile not in cands:
            tile = -1
        single_tile = tile
        tile = _pair_tile(key, tile, cands)

    def make(t, outp=None):
        wq, cq, bq = wcb(t)
        sh = _lib.GemmShape(kind=_lib.GROUP_GEMM, tile=t, M=M, N=N, K=K, rows_per_batch=rpb, act=_abi_act(act), out_mode=om, alpha=alpha, ln_eps=ln_eps)
        pr = _lib.GemmProblem(A=a.data_ptr(), W=wq.data_ptr(), C=(out if outp is None else outp).data_ptr(), bias=_p(bq),
                              residual=_p(residual), rowstats_out=_p(stats), ln_stats=ln_p, ln_colsum=_p(cq), lda=lda, ldc=ldc, ldr=ldr,
                              rs_ld=rs_ld, ln_ld=ln_ld, ln_slots=ln_slots)
        return sh, pr

    _issue(_Launch("gemm", lib, "supir_gemm_bf16_ln", launch, key=key + (alpha, rpb, ln_eps, emit_stats, ln is not None), tkey=key,
                   tile=tile, single_tile=single_tile, cands=cands, make=make, w=w, out=out, inplace=inplace,
                   trace=("gemm_t" if trans is not None else "gemm", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * n_out),
                          dict(M=M, N=N, K=K, act=act, tile=tile)),
                   keep=(a, w, bias, residual, out, stats, ln.buf if ln is not None else None, colsum, alt16)))
    if emit_stats:
        t_used = tile if tile >= 32 else (tile & 7) if tile >= 0 else lib.supir_gemm_tile_for(M, N, act)
        bn, _ = _TILE_BN_WN[t_used]
        return out, RowStats(stats, (N + bn - 1) // bn, rs_ld)
    return out


USE_QKV = _os.environ.get("SUPIR_FUSED_QKV", "1") != "0"


def has_fused(dtype):
    """Whether the fused forms (LayerNorm folded into its consumers, fused q|k|v, fused cross-attention, GroupNorm partials) exist for this
    element type: the 16-bit libraries have them, the fp32 service (ops_f32.py) is the plain operator sequence."""
    return dtype in HALF_TYPES


def gemm_qkv_supported(M, N, n_split, K, T):
    """Shape predicate of supir_gemm_bf16_qkv (256 x 160 or 256 x 128 tile of csrc/gemm16.hip)."""
    fits = (N % 160 == 0 and n_split % 160 == 0) or (N % 128 == 0 and n_split % 128 == 0)
    return (USE_QKV and USE_GEMM16 and 34 in G16_TILES and M % 256 == 0 and fits and 0 < n_split < N
            and K % 64 == 0 and K >= 128 and T % 4 == 0)


def qkv_tile_width(M, N, n_split):
    """The tile width the fused q|k|v launch takes (csrc/gemm16.hip g16_qkv_bn, same rule): 128 or 160 columns -- the one with fewer
    rounds of 256 workgroups x columns among those that divide N and n_split.  For trace records / kernel names only."""
    ok160, ok128 = N % 160 == 0 and n_split % 160 == 0, N % 128 == 0 and n_split % 128 == 0
    if not (ok128 and ok160):
        return 128 if ok128 else 160
    tm = M // 256
    c160, c128 = -(-tm * (N // 160) // 256) * 160, -(-tm * (N // 128) // 256) * 128
    return 128 if c128 < c160 else 160


def gemm_qkv(a, w, bias, B, T, n_split, *, ln=None, colsum=None, ln_eps=1e-5, out_qk=None, out_vt=None):
    """Fused q | k | v projection: a [B*T, K] -> (qk [B, T, n_split] bf16, v^T [B, N - n_split, T] bf16) in one launch; optional
    LayerNorm fold as in gemm_ln.  Callers check gemm_qkv_supported first."""
    DT = a.dtype
    lib = _lib.load(DT)
    _check_dev(a, w)
    M, K, lda = _rows_ld(a)
    N = w.shape[0]
    assert DT in HALF_TYPES and M == B * T and w.shape[1] == K and w.is_contiguous() and w.dtype == DT
    if out_qk is None:
        out_qk = torch.empty(B, T, n_split, dtype=DT, device=a.device)
    if out_vt is None:
        out_vt = torch.empty(B, N - n_split, T, dtype=DT, device=a.device)
    ln_p, ln_ld, ln_slots = 0, 0, 0
    if ln is not None:
        ln_p, ln_ld, ln_slots = ln.buf.data_ptr(), ln.ld, ln.slots
        assert colsum is not None and colsum.numel() == N

    def launch(t, outp=None, hints=None):
        return lib.supir_gemm_bf16_qkv_ex(a.data_ptr(), w.data_ptr(), out_qk.data_ptr(), out_vt.data_ptr(), M, N, n_split, K, lda, n_split,
                                          T, T, _p(bias), ln_p, ln_ld, ln_slots, _p(colsum), ln_eps,
                                          None if hints is None else _ct.byref(hints), _stream())

    def make(t, outp=None):
        sh = _lib.GemmShape(kind=_lib.GROUP_QKV, tile=34, M=M, N=N, K=K, rows_per_batch=T, n_split=n_split, alpha=1.0, ln_eps=ln_eps)
        pr = _lib.GemmProblem(A=a.data_ptr(), W=w.data_ptr(), C=out_qk.data_ptr(), C2=out_vt.data_ptr(), bias=_p(bias), ln_stats=ln_p,
                              ln_colsum=_p(colsum), lda=lda, ldc=n_split, ldc2=T, ln_ld=ln_ld, ln_slots=ln_slots)
        return sh, pr

    tkey = ("qkv", M, N, K) + _k(DT)
    _issue(_Launch("qkv", lib, "supir_gemm_bf16_qkv", launch, key=tkey + (n_split, T, ln_eps, ln is not None), tkey=tkey, tile=36,
                   single_tile=36, make=make, w=w, out=out_qk,
                   trace=("gemm", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N),
                          dict(M=M, N=N, K=K, act=0, tile=36 if qkv_tile_width(M, N, n_split) == 160 else 41)),
                   keep=(a, w, bias, out_qk, out_vt, ln.buf if ln is not None else None, colsum)))
    return out_qk, out_vt


def choose(key, fns, prefer=None, margin=0.1):
    """Time alternative implementations of the same step once (outside graph capture) and remember the faster one:
    returns the index into `fns`.  Used where two launch sequences compute the same tensors (fused q|k|v vs two projections).
    prefer = index of the alternative with FEWER launches: it is kept unless another one is faster by more than `margin` -- the
    timing here is back-to-back and hot, where a launch costs ~2 us; inside a step every launch also meets its weights cold
    (+4..6 us), so near-ties go to the shorter sequence (fused q|k|v at (2048, 3840, 1280): a tie here, -0.7 ms per step there,
    profiles/r02/step_ab_fused_qkv.log, profiles/r03/step_variants_*.log)."""
    _check_packaged_tune()
    c = _CHOICE.get(key)
    if c is not None:
        return c
    if not AUTOTUNE or _DEFER is not None or torch.cuda.is_current_stream_capturing():
        return 0
    times = []
    for i, fn in enumerate(fns):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        e1.synchronize()
        times.append((e0.elapsed_time(e1), i))
    c = min(times)[1]
    if prefer is not None and c != prefer and min(times)[0] > (1.0 - margin) * dict((i, t_) for t_, i in times)[prefer]:
        c = prefer
    _CHOICE[key] = c
    return c


def gemm_t(a, w, bias, B, T, Tpad, out=None, tile=-1):
    """Transposed projection: out[b][n][t] = (a[b*T+t] @ w[n]) (+bias); out is [B, N, Tpad] (zero padded)."""
    DT = a.dtype
    if DT == torch.float32:
        return _f32.gemm_t(a, w, bias, B, T, Tpad, out=out)
    lib = _lib.load(DT)
    _check_dev(a, w)
    M, K, lda = _rows_ld(a)
    assert M == B * T and DT in HALF_TYPES and w.dtype == DT
    N = w.shape[0]
    if out is None:
        if Tpad != T:
            _no_defer("gemm_t(with padding)")
        out = torch.zeros(B, N, Tpad, dtype=DT, device=a.device) if Tpad != T else \
            torch.empty(B, N, Tpad, dtype=DT, device=a.device)
    assert out.dtype == DT

    def launch(t, outp=None, hints=None):
        return lib.supir_gemm_bf16_ex(a.data_ptr(), w.data_ptr(), (out if outp is None else outp).data_ptr(), M, N, K, lda, Tpad, _p(bias),
                                      0, 0, T, 0, 0, 0, 2, 1.0, t, None if hints is None else _ct.byref(hints), _stream())

    key = ("gemm", M, N, K, 0, 2) + _k(DT)
    cands = ()
    single_tile = tile
    if tile == -1:
        cands = _gemm_candidates(M, N, K, 0, 2, Tpad, epilogue_ok=(lda % 8 == 0 and T % 4 == 0))
        tile = _autotune(key, cands, launch)
        if tile >= 32 and tile not in cands:
            tile = -1
        single_tile = tile
        tile = _pair_tile(key, tile, cands)

    def make(t, outp=None):
        sh = _lib.GemmShape(kind=_lib.GROUP_GEMM, tile=t, M=M, N=N, K=K, rows_per_batch=T, act=0, out_mode=2, alpha=1.0)
        pr = _lib.GemmProblem(A=a.data_ptr(), W=w.data_ptr(), C=(out if outp is None else outp).data_ptr(), bias=_p(bias), lda=lda, ldc=Tpad)
        return sh, pr

    _issue(_Launch("gemm", lib, "supir_gemm_bf16(T)", launch, key=key + (T, Tpad), tkey=key, tile=tile, single_tile=single_tile,
                   cands=cands, make=make, w=w, out=out,
                   trace=("gemm_t", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N), dict(M=M, N=N, K=K, act=0, tile=tile)),
                   keep=(a, w, bias, out)))
    return out


def conv3x3(x, w, bias=None, *, stride=1, pad=(1, 1), upsample=False, out_hw=None, rowbias=None, residual=None,
            act=0, alpha=1.0, out=None, tile=-1, gn_part=False):
    """x [B,H,W,Cin(ld)] bf16 -> [B,OH,OW,Cout]. w [Cout,3,3,Cin] bf16. pad=(top,left); bottom/right implied by out_hw.
    gn_part=True: returns (out, GnPart or None) -- GroupNorm statistics of `out` from the epilogue, when the tile that runs can emit them."""
    DT = x.dtype
    if DT == torch.float32:
        return _f32.conv3x3(x, w, bias, stride=stride, pad=pad, upsample=upsample, out_hw=out_hw, rowbias=rowbias, residual=residual, act=act,
                            alpha=alpha, out=out, gn_part=gn_part)
    lib = _lib.load(DT)
    _check_dev(x, w)
    B, H, W, Cin = x.shape
    _, _, ldx = _rows_ld(x)
    Cout = w.shape[0]
    assert DT in HALF_TYPES and w.shape[1:] == (3, 3, Cin) and w.is_contiguous() and w.dtype == DT
    if out_hw is None:
        if upsample:
            out_hw = (2 * H, 2 * W)
        elif stride == 1:
            out_hw = (H, W)
        else:
            out_hw = ((H + 2 * pad[0] - 3) // stride + 1, (W + 2 * pad[1] - 3) // stride + 1)
    OH, OW = out_hw
    if out is None:
        out = torch.empty(B, OH, OW, Cout, dtype=DT, device=x.device)
    _, _, ldy = _rows_ld(out)
    assert out.dtype in (DT, torch.float32)
    ldr = 0
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == DT
        _, _, ldr = _rows_ld(residual)
    ld_rb = 0
    if rowbias is not None:
        assert rowbias.dtype == DT and rowbias.shape == (B, Cout) and rowbias.stride(-1) == 1
        ld_rb = rowbias.stride(0)
    om = 0 if out.dtype == DT else 1

    M_all = B * OH * OW
    split_ws = []

    def launch(t, outp=None, hints=None):
        if t >= 64:
            # tap-split form (supir_conv3x3_bf16_splitk): nine fp32 partials + a finalize launch (bias, SiLU, bf16), codes 64 + tile
            if not split_ws:
                split_ws.append(torch.empty(9 * M_all * Cout, dtype=torch.float32, device=x.device))
            rc = lib.supir_conv3x3_bf16_splitk(x.data_ptr(), w.data_ptr(), split_ws[0].data_ptr(), B, H, W, Cin, ldx, Cout, OH, OW, stride,
                                               pad[0], pad[1], 1 if upsample else 0, 9, t - 64, _stream())
            if rc != 0:
                return rc
            return lib.supir_splitk_finalize(split_ws[0].data_ptr(), 9, M_all, Cout, _p(bias), act, (out if outp is None else outp).data_ptr(),
                                             ldy, _stream())
        return lib.supir_conv3x3_bf16_ex(x.data_ptr(), w.data_ptr(), (out if outp is None else outp).data_ptr(), B, H, W, Cin, ldx, Cout,
                                         ldy, OH, OW, stride, pad[0], pad[1], 1 if upsample else 0, _p(bias), _p(rowbias), ld_rb,
                                         _p(residual), ldr, act, om, alpha, t, None if hints is None else _ct.byref(hints), _stream())

    inplace = residual is not None and residual.data_ptr() == out.data_ptr()
    key = ("conv", B, H, W, Cin, Cout, stride, bool(upsample)) + _k(DT)
    cands = ()
    single_tile = tile
    if tile == -1:
        M_ = B * OH * OW
        cands = [0, 1, 2, 3, 4, 5, 6]
        if USE_GEMM16 and om == 0 and act != 2 and ldx % 8 == 0 and ldy % 8 == 0 and out.data_ptr() % 16 == 0 \
                and (residual is None or ldr % 4 == 0) and (rowbias is None or ld_rb % 4 == 0):
            for t, (bm, bn, ks, s_) in _G16.items():   # same predicate as supir_gemm16_supported(conv)
                if t in G16_TILES and M_ % bm == 0 and Cout % bn == 0 and Cin % (64 * ks) == 0 and (9 * Cin // 64) // ks >= s_ - 1 \
                        and not (t in _G16_PLAIN_ONLY and Cout % 80 == 0) and not (t == 42 and (OH * OW) % bm) and t != 45:   # 45: plain GEMMs only (it ties or loses on every convolution measured: profiles/r05/big_tiles_tile45.json)
                    if t in _G16_HALO_W and not (W == _G16_HALO_W[t] and stride == 1 and not upsample and tuple(pad) == (1, 1) and (OH, OW) == (H, W)
                                                 and (OH * OW) % bm == 0 and H * W * ldx < (1 << 31)):
                        continue      # halo form: whole rows of a map of its width, stride 1, pad 1 (same predicate as supir_gemm16_supported)
                    cands.append(t)
        # tap-split candidates for convolutions whose tile grid is a fraction of the machine: one workgroup set per filter tap
        if USE_CONV_SPLIT and om == 0 and act in (0, 1) and residual is None and rowbias is None and alpha == 1.0 and Cin % 64 == 0 \
                and Cout % 4 == 0 and ldy % 4 == 0 and ((M_ + 63) // 64) * ((Cout + 63) // 64) <= 128:
            cands += [64 + 1, 64 + 2, 64 + 3]
        cands = tuple(cands)
        if inplace:
            tile = _TUNE.get(key, -1)
        else:
            tile = _autotune(key, cands, launch)
        if tile >= 32 and tile not in cands:
            tile = -1
        single_tile = tile
        tile = _pair_tile(key, tile, cands)
    if tile >= 64 and (act not in (0, 1) or residual is not None or rowbias is not None or om != 0):
        raise _lib.SupirHipError("conv3x3: the tap-split tiles (64 + t) take no residual / row bias / fp32 output")
    part = _gn_part_alloc(tile, B, OH * OW, Cout, x.device) if (gn_part and om == 0) else None

    def make(t, outp=None):
        sh = _lib.GemmShape(kind=_lib.GROUP_CONV3X3, tile=t, act=act, out_mode=om, alpha=alpha, B=B, H=H, W=W, Cin=Cin, Cout=Cout, OH=OH,
                            OW=OW, stride=stride, pad_t=pad[0], pad_l=pad[1], upsample=1 if upsample else 0)
        pr = _lib.GemmProblem(A=x.data_ptr(), W=w.data_ptr(), C=(out if outp is None else outp).data_ptr(), bias=_p(bias),
                              rowbias=_p(rowbias), residual=_p(residual), gn_partials_out=None if part is None else part.buf.data_ptr(),
                              lda=ldx, ldc=ldy, ldr=ldr, ld_rowbias=ld_rb)
        return sh, pr

    M = B * OH * OW
    _issue(_Launch("conv", lib, "supir_conv3x3_bf16", launch, key=key + (OH, OW, pad, act, om, alpha, part is not None), tkey=key, tile=tile,
                   single_tile=single_tile, cands=cands, make=make, w=w, part=part, out=out, inplace=inplace,
                   trace=("conv3x3", 2.0 * M * Cout * 9 * Cin, 2.0 * (B * H * W * Cin + Cout * 9 * Cin + M * Cout),
                          dict(M=M, N=Cout, K=9 * Cin, act=act, tile=tile)),
                   keep=(x, w, bias, rowbias, residual, out, part, split_ws)))
    return (out, part) if gn_part else out


# --------------------------------------------------------------------------------------------- attention
def flash_attn(q, k, vt, B, H, Tq, Tk, out=None, causal=False):
    """q [B,Tq,>=H*64] k [B,Tk,>=H*64] (views with row stride), vt [B,H*64,Tpad]; returns [B,Tq,H*64].
    causal=True (text towers): query i sees keys j <= i."""
    DT = q.dtype
    if DT == torch.float32:
        return _f32.flash_attn(q, k, vt, B, H, Tq, Tk, out=out, causal=causal)
    lib = _lib.load(DT)
    _check_dev(q, k, vt)
    assert DT in HALF_TYPES and k.dtype == DT and vt.dtype == DT
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and vt.is_contiguous()
    ldq, ldk, ldvt = q.stride(-2), k.stride(-2), vt.shape[-1]
    assert q.shape[0] == B and q.stride(0) == Tq * ldq and k.stride(0) == Tk * ldk
    if out is None:
        out = torch.empty(B, Tq, H * 64, dtype=DT, device=q.device)
    assert out.dtype == DT
    ldo = out.stride(-2)

    def launch(t, outp=None):
        o = out if outp is None else outp
        if causal:
            return lib.supir_flash_attn_d64_ex(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, Tq, Tk, ldq, ldk, ldvt,
                                               ldo, 0.125, 1, _stream())
        return lib.supir_flash_attn_d64(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, Tq, Tk, ldq, ldk, ldvt,
                                        ldo, 0.125, _stream())

    def make(outp=None):
        return (B, H, Tq, 0.125), _lib.AttnProblem(Q=q.data_ptr(), K=k.data_ptr(), Vt=vt.data_ptr(), O=(out if outp is None else outp).data_ptr(), Tk=Tk, ldq=ldq,
                                                   ldk=ldk, ldvt=ldvt, ldo=ldo, flags=1 if causal else 0)

    _issue(_Launch("attn", lib, "supir_flash_attn_d64", launch, key=("attn", B, H, Tq) + _k(DT), make=make, out=out,
                   trace=("attn", 4.0 * B * H * Tq * Tk * 64, 2.0 * B * H * 64 * (2 * Tq + 2 * Tk), dict(B=B, H=H, Tq=Tq, Tk=Tk)),
                   keep=(q, k, vt, out)))
    return out


USE_XATTN = _os.environ.get("SUPIR_FUSED_XATTN", "1") != "0"


def xattn_q_supported(B, T, C, H, Tk):
    """Shape predicate of supir_xattn_q_d64 (csrc/xattn.hip): whole 128-token row blocks, K steps of 64 channels, <= 2 key tiles."""
    return USE_XATTN and T % 128 == 0 and C % 64 == 0 and C >= 192 and 0 < Tk <= 128 and H > 0


def xattn_q(x, wq, bias, k, vt, B, H, T, Tk, *, ln=None, colsum=None, ln_eps=1e-5, out=None):
    """to_q projection (optionally with the LayerNorm fold of gemm_ln) + attention over the Tk <= 128 cached text keys, one launch:
    x [B,T,C] tokens, wq [H*64,C], k [B,Tk,H*64], vt [B,H*64,Tpad] -> [B,T,H*64].  Callers check xattn_q_supported first."""
    DT = x.dtype
    lib = _lib.load(DT)
    _check_dev(x, wq, k, vt)
    M, C, ldx = _rows_ld(x)
    assert DT in HALF_TYPES and M == B * T and wq.shape == (H * 64, C) and wq.is_contiguous() and wq.dtype == DT
    assert k.dtype == DT and vt.dtype == DT and k.stride(-1) == 1 and vt.is_contiguous()
    ldk, ldvt = k.stride(-2), vt.shape[-1]
    assert k.shape[0] == B and k.stride(0) == Tk * ldk
    if out is None:
        out = torch.empty(B, T, H * 64, dtype=DT, device=x.device)
    assert out.dtype == DT
    _, _, ldo = _rows_ld(out)
    ln_p, ln_ld, ln_slots = 0, 0, 0
    if ln is not None:
        ln_p, ln_ld, ln_slots = ln.buf.data_ptr(), ln.ld, ln.slots
        assert colsum is not None and colsum.numel() == H * 64

    def launch(t, outp=None, hints=None):
        return lib.supir_xattn_q_d64(x.data_ptr(), wq.data_ptr(), _p(bias), k.data_ptr(), vt.data_ptr(),
                                     (out if outp is None else outp).data_ptr(), B, H, T, Tk, C, ldx, ldk, ldvt, ldo, ln_p, ln_ld, ln_slots,
                                     _p(colsum), ln_eps, 0.125, None if hints is None else _ct.byref(hints), _stream())

    N = H * 64
    _issue(_Launch(None, lib, "supir_xattn_q_d64", launch, w=wq, out=out,
                   trace=("xattn_q", 2.0 * M * N * C + 4.0 * M * N * Tk, 2.0 * (M * C + N * C + M * N + 2 * B * Tk * N),
                          dict(B=B, H=H, T=T, Tk=Tk, C=C)),
                   keep=(x, wq, bias, k, vt, out, ln.buf if ln is not None else None, colsum)))
    return out


# VAE mid-block attention (one head, dim 512): supir_flash_attn_d512 never forms the score matrix; the materialised form (GEMM -> fp32
# scores [T, T] -> softmax_rows -> GEMM) moves ~3 GB per call at T = 16 384 but runs on the tuned GEMM tiles.  Until round 4 the flash
# kernel had B * T / 128 workgroups of one wave per SIMD and lost below 16 384 tokens (T = 4096: 391 us vs 93 us materialised,
# profiles/r02/attn_d512_timing.json).  With the keys split over workgroup sets (supir_flash_attn_d512_split, the library picks the count)
# it fills the chip at every size -- measured on one box (profiles/r04/micro_attn_d512_key_split_sweep.log), flash / materialised in us:
# (B 1, T 16 384) 820 / 1 668, (1, 4096) 96 / 100, (4, 4096) 243 / 382, (16, 4096) 876 / 1 800, (1, 5184) 125 / 158, (8, 5184) 985 / 1 437,
# (1, 1024) 44 / 50, (16, 1024) 94 / 795 -- so it takes over from FLASH_D512_MIN_TOKENS = 1024 tokens per batch element upwards (env
# SUPIR_FLASH_D512_MIN_TOKENS); SUPIR_FLASH_D512 = 1 / 0 forces it on / off for every size.
USE_FLASH_D512 = {"1": True, "0": False}.get(_os.environ.get("SUPIR_FLASH_D512", "auto"), "auto")
FLASH_D512_MIN_TOKENS = int(_os.environ.get("SUPIR_FLASH_D512_MIN_TOKENS", "1024"))


def use_flash_d512(T):
    """Whether the head-dim-512 attention over T tokens takes the flash kernel (see USE_FLASH_D512)."""
    if USE_FLASH_D512 == "auto":
        return T >= FLASH_D512_MIN_TOKENS
    return bool(USE_FLASH_D512)


def flash_attn_d512(q, k, vt, Tk, out=None, splits=0):
    """Single-head, head-dim-512 attention without a score matrix (VAE mid block): q [B,Tq,512], k [B,Tk,512], vt [B,512,Tpad]
    (V transposed per batch, Tpad >= roundup(Tk, 32), zero padded) -> [B,Tq,512].  scale = 512^-0.5.
    splits: key splits of supir_flash_attn_d512_split (0 = the library's choice: enough to fill the 256 CUs -- 2 at 16 384 tokens, 8 at
    4096; 1 = one pass over all keys per workgroup)."""
    DT = q.dtype
    if DT == torch.float32:
        return _f32.flash_attn_d512(q, k, vt, Tk, out=out)
    lib = _lib.load(DT)
    _check_dev(q, k, vt)
    assert DT in HALF_TYPES and k.dtype == DT and vt.dtype == DT
    B, Tq, C = q.shape
    assert C == 512 and k.shape == (B, Tk, 512) and vt.shape[:2] == (B, 512) and vt.is_contiguous()
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and q.stride(0) == Tq * q.stride(1) and k.stride(0) == Tk * k.stride(1)
    if out is None:
        out = torch.empty(B, Tq, 512, dtype=DT, device=q.device)
    assert out.dtype == DT and out.stride(-1) == 1 and out.stride(0) == Tq * out.stride(1)
    _no_defer("flash_attn_d512")
    nbytes = lib.supir_flash_attn_d512_workspace(B, Tq, Tk, splits)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device) if nbytes else None
    ev = _ev()
    rc = lib.supir_flash_attn_d512_split(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, Tq, Tk, q.stride(1), k.stride(1),
                                         vt.shape[-1], out.stride(1), 512 ** -0.5, splits, _p(ws), nbytes, _stream())
    _lib.check(rc, "supir_flash_attn_d512_split", lib)
    # algorithmic FLOPs (Q.K^T and P.V once each; the kernel's own first pass over K is overhead, not work)
    _rec("attn_d512", 4.0 * B * Tq * Tk * 512, 2.0 * B * 512 * (2 * Tq + 2 * Tk), ev, B=B, H=1, Tq=Tq, Tk=Tk,
         splits=nbytes // (B * Tq * 514 * 4) if nbytes else 1)
    return out


def softmax_rows(s, scale, out=None, valid=None, dtype=None):
    """softmax over the first `valid` columns of fp32 scores [rows, Tpad]; remaining columns of the 16-bit output (`dtype`,
    default: that of `out`, else bf16) are zero."""
    DT = out.dtype if out is not None else (BF16 if dtype is None else dtype)
    if DT == torch.float32:
        return _f32.softmax_rows(s, scale, out=out, valid=valid)
    lib = _lib.load(DT)
    _check_dev(s)
    rows, Tp = s.shape
    T = Tp if valid is None else valid
    assert s.dtype == torch.float32 and s.stride(1) == 1 and DT in HALF_TYPES
    if out is None:
        out = torch.empty(rows, Tp, dtype=DT, device=s.device)
    _no_defer("softmax_rows")
    ev = _ev()
    rc = lib.supir_softmax_rows(s.data_ptr(), out.data_ptr(), rows, T, Tp, s.stride(0), out.stride(0), scale, _stream())
    _lib.check(rc, "supir_softmax_rows", lib)
    _rec("softmax", 0, 6.0 * rows * T, ev)
    return out


# --------------------------------------------------------------------------------------------- norms
def groupnorm_stats(x):
    """(sum, sum of squares) per (batch, group) of a channels-last tensor: fp32 [B, 32, 2] (fp64 from the fp32 service)."""
    if x.dtype == torch.float32:
        return _f32.groupnorm_stats(x)
    lib = _lib.load(x.dtype)
    _check_dev(x)
    assert x.dtype in HALF_TYPES
    B = x.shape[0]
    HW = int(math.prod(x.shape[1:-1]))
    _, C, ld = _rows_ld(x)
    _no_defer("groupnorm_stats")
    ws = _gn_workspace(B, x.device)
    out = torch.empty(B, 32, 2, dtype=torch.float32, device=x.device)
    rc = lib.supir_groupnorm_stats(x.data_ptr(), 0, B, HW, C, C, ld, 0, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream())
    _lib.check(rc, "supir_groupnorm_stats", lib)
    return out


def groupnorm(x, gamma, beta, eps, *, silu=False, x2=None, mod_g=None, mod_b=None, control_scale=1.0, x1raw=None,
              x2raw=None, out=None, given=None, part=None, part2=None):
    """GroupNorm(32) over channels-last x (optionally the channel concat [x | x2]); see supir_groupnorm_nhwc.
    part / part2 = GnPart of x / x2 from their producers: one launch, no statistics pass (supir_groupnorm_nhwc_parts)."""
    DT = x.dtype
    if DT == torch.float32:
        return _f32.groupnorm(x, gamma, beta, eps, silu=silu, x2=x2, mod_g=mod_g, mod_b=mod_b, control_scale=control_scale, x1raw=x1raw,
                              x2raw=x2raw, out=out, given=given)
    lib = _lib.load(DT)
    _check_dev(x, gamma, beta)
    assert DT in HALF_TYPES and all(t is None or t.dtype == DT for t in (x2, mod_g, mod_b, x1raw, x2raw, out))
    B = x.shape[0]
    HW = int(math.prod(x.shape[1:-1]))
    _, C1, ld1 = _rows_ld(x)
    C, ld2 = C1, 0
    if x2 is not None:
        _, C2, ld2 = _rows_ld(x2)
        assert x2.shape[:-1] == x.shape[:-1]
        C = C1 + C2
    assert gamma.numel() == C and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    if out is None:
        out = torch.empty(*x.shape[:-1], C, dtype=DT, device=x.device)
    _, Co, ldo = _rows_ld(out)
    assert Co == C
    ldm = 0
    if mod_g is not None:
        _, Cm, ldm = _rows_ld(mod_g)
        _, Cm2, ldm2 = _rows_ld(mod_b)
        assert Cm == C and Cm2 == C and ldm == ldm2
    if (part is not None and part.unit != 10 and given is None and x2 is None and part.C == C and part.buf.shape[0] == B
            and (C // 32) % part.unit == 0 and _DEFER is None):
        # VAE form (4-channel units from tiles 39 / 40, thousands of tile rows): one small launch reduces them to (mean, variance) per
        # group, the apply pass takes those as `given` -- no pass over the tensor for its statistics
        given = torch.empty(B, 32, 2, dtype=torch.float32, device=x.device)
        ev = _ev()
        _lib.check(lib.supir_groupnorm_parts_finalize(part.buf.data_ptr(), B, part.nchunk, C, part.unit, HW, given.data_ptr(), _stream()),
                   "supir_groupnorm_parts_finalize", lib)
        _rec("groupnorm_parts_finalize", 0, 8.0 * B * part.nchunk * (C // part.unit), ev, B=B, C=C, nchunk=part.nchunk)
    use_parts = (part is not None and part.unit == 10 and given is None and (x2 is None or part2 is not None) and (C // 32) % 10 == 0
                 and C1 % 10 == 0 and (part2 is None or part2.unit == 10)
                 and part.C == C1 and part.buf.shape[0] == B and (part2 is None or (part2.C == C - C1 and part2.buf.shape[0] == B)))
    act = 1 if silu else 0
    # own statistics: partial-sum workspace.  Recorded launches may run two to a grid (paired_run): each gets its own.
    ws = None if use_parts else (torch.empty(B * 1024 * 64, dtype=torch.float32, device=x.device) if _DEFER is not None
                                 else _gn_workspace(B, x.device))

    def launch(t, outp=None):
        o = out if outp is None else outp
        if use_parts:
            return lib.supir_groupnorm_nhwc_parts(x.data_ptr(), _p(x2), _p(x1raw), _p(x2raw), B, HW, C, C1, ld1, ld2, gamma.data_ptr(),
                                                  beta.data_ptr(), eps, act, _p(mod_g), _p(mod_b), ldm, control_scale,
                                                  o.data_ptr(), ldo, part.buf.data_ptr(), part.nchunk,
                                                  0 if part2 is None else part2.buf.data_ptr(), 0 if part2 is None else part2.nchunk,
                                                  _stream())
        return lib.supir_groupnorm_nhwc(x.data_ptr(), _p(x2), _p(x1raw), _p(x2raw), B, HW, C, C1, ld1, ld2, gamma.data_ptr(),
                                        beta.data_ptr(), eps, act, _p(mod_g), _p(mod_b), ldm, control_scale,
                                        o.data_ptr(), ldo, ws.data_ptr(), ws.numel() * 4, _p(given), _stream())

    def make(outp=None):
        pr = _lib.GnProblem(x1=x.data_ptr(), x2=_p(x2), x1raw=_p(x1raw), x2raw=_p(x2raw), gamma=gamma.data_ptr(), beta=beta.data_ptr(),
                            mod_g=_p(mod_g), mod_b=_p(mod_b), out=(out if outp is None else outp).data_ptr(),
                            part1=part.buf.data_ptr() if use_parts else None,
                            part2=part2.buf.data_ptr() if (use_parts and part2 is not None) else None, workspace=_p(ws), C1=C1, ld1=ld1,
                            ld2=ld2, ldm=ldm, ldo=ldo, nchunk1=part.nchunk if use_parts else 0,
                            nchunk2=part2.nchunk if (use_parts and part2 is not None) else 0, control_scale=control_scale)
        return (B, HW, C, eps, act), pr

    n = B * HW * C
    # in place (ResBlock: the second norm overwrites conv1's output): a candidate timed by the pair autotune must not be applied to
    # its own input over and over -- `inplace` sends its timing launches to a scratch output (make / launch take the override)
    inplace = any(t is not None and t.data_ptr() == out.data_ptr() for t in (x, x2, x1raw, x2raw))
    _issue(_Launch("gn" if given is None else None, lib, "supir_groupnorm_nhwc_parts" if use_parts else "supir_groupnorm_nhwc", launch,
                   key=("gn", B, HW, C, eps, act, use_parts, x2 is not None, mod_g is not None, x1raw is not None or x2raw is not None)
                   + _k(DT), make=make, out=out, inplace=inplace,
                   trace=("groupnorm", 0, 2.0 * n * (2 + (2 if mod_g is not None else 0)), dict(B=B, HW=HW, C=C, parts=use_parts)),
                   keep=(x, x2, x1raw, x2raw, gamma, beta, mod_g, mod_b, out, ws, given, part, part2)))
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    DT = x.dtype
    if DT == torch.float32:
        return _f32.layernorm(x, gamma, beta, eps, out=out)
    lib = _lib.load(DT)
    _check_dev(x, gamma, beta)
    assert DT in HALF_TYPES and (out is None or out.dtype == DT)
    rows, C, ldx = _rows_ld(x)
    if out is None:
        out = torch.empty(*x.shape, dtype=DT, device=x.device)
    _, _, ldy = _rows_ld(out)

    def launch(t, outp=None):
        return lib.supir_layernorm(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, C, ldx, ldy, eps, _stream())

    _issue(_Launch(None, lib, "supir_layernorm", launch, trace=("layernorm", 0, 4.0 * rows * C, dict(rows=rows, C=C)),
                   keep=(x, out, gamma, beta)))
    return out


# --------------------------------------------------------------------------------------------- boundary convs
def conv3x3_smallcin(x_nchw, w, bias, add=None, out=None, dtype=None):
    """fp32 NCHW [B,Cin<=8,H,W] -> 16-bit [B,H,W,Cout] (`dtype`; default: that of `out` / `add`, else bf16); w fp32 [Cout,Cin,3,3]."""
    DT = out.dtype if out is not None else add.dtype if add is not None else (BF16 if dtype is None else dtype)
    if DT == torch.float32:
        return _f32.conv3x3_smallcin(x_nchw, w, bias, add=add, out=out)
    lib = _lib.load(DT)
    assert DT in HALF_TYPES and (add is None or add.dtype == DT)
    _check_dev(x_nchw, w)
    x_nchw = x_nchw.contiguous()
    assert x_nchw.dtype == torch.float32 and w.dtype == torch.float32 and w.is_contiguous()
    B, Cin, H, W = x_nchw.shape
    Cout = w.shape[0]
    if out is None:
        out = torch.empty(B, H, W, Cout, dtype=DT, device=x_nchw.device)
    _, _, ldo = _rows_ld(out)
    ld_add = 0
    if add is not None:
        _, _, ld_add = _rows_ld(add)
    _no_defer("conv3x3_smallcin")
    ev = _ev()
    rc = lib.supir_conv3x3_smallcin(x_nchw.data_ptr(), w.data_ptr(), _p(bias), _p(add), out.data_ptr(), B, Cin, H, W, Cout,
                                    ld_add, ldo, _stream())
    _lib.check(rc, "supir_conv3x3_smallcin", lib)
    _rec("conv_smallcin", 2.0 * B * H * W * Cout * 9 * Cin, B * H * W * (4.0 * Cin + 2.0 * Cout), ev)
    return out


def conv3x3_smallcout(x, w9, bias, out=None):
    """bf16 [B,H,W,Cin] -> fp32 NCHW [B,Cout,H,W]; w9 bf16 [9,Cout,Cin]."""
    DT = x.dtype
    if DT == torch.float32:
        return _f32.conv3x3_smallcout(x, w9, bias, out=out)
    lib = _lib.load(DT)
    _check_dev(x, w9)
    B, H, W, Cin = x.shape
    _, _, ldx = _rows_ld(x)
    Cout = w9.shape[1]
    assert DT in HALF_TYPES and w9.shape == (9, Cout, Cin) and w9.dtype == DT and w9.is_contiguous()
    if out is None:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    _no_defer("conv3x3_smallcout")
    ev = _ev()
    rc = lib.supir_conv3x3_smallcout(x.data_ptr(), w9.data_ptr(), _p(bias), out.data_ptr(), B, Cin, H, W, Cout, ldx,
                                     _stream())
    _lib.check(rc, "supir_conv3x3_smallcout", lib)
    _rec("conv_smallcout", 2.0 * B * H * W * Cout * 9 * Cin, B * H * W * (2.0 * Cin + 4.0 * Cout), ev)
    return out


def pointwise_nchw(x, w, bias, in_scale=1.0):
    lib = _lib.load()
    _check_dev(x, w)
    x = x.contiguous()
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    w2 = w.reshape(Cout, Cin).contiguous()
    out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    _no_defer("pointwise_nchw")
    rc = lib.supir_pointwise_nchw(x.data_ptr(), w2.data_ptr(), _p(bias), out.data_ptr(), B, Cin, Cout, H * W, in_scale,
                                  _stream())
    _lib.check(rc, "supir_pointwise_nchw")
    return out


# --------------------------------------------------------------------------------------------- sampler step (elementwise halves)
def edm_step_pre(x, eps, s_noise, noise_mul, c_in, reps):
    """x_hat = x + (eps * s_noise) * noise_mul (eps None: x_hat is x itself); net_in = [x_hat * c_in] * reps along the batch.
    fp32 contiguous latents; returns (x_hat, net_in).  See supir_edm_step_pre."""
    lib = _lib.load()
    _check_dev(x, eps)
    assert x.dtype == torch.float32 and x.is_contiguous() and (eps is None or (eps.dtype == torch.float32 and eps.is_contiguous()
                                                                                and eps.shape == x.shape))
    n = x.numel()
    x_hat = torch.empty_like(x) if eps is not None else x
    net_in = torch.empty((reps * x.shape[0],) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    _no_defer("edm_step_pre")
    rc = lib.supir_edm_step_pre(x.data_ptr(), _p(eps), float(s_noise), float(noise_mul), float(c_in),
                                x_hat.data_ptr() if eps is not None else 0, net_in.data_ptr(), n, reps, _stream())
    _lib.check(rc, "supir_edm_step_pre", lib)
    return x_hat, net_in


def edm_step_post(net_out, x_hat, x_center, c_out, c_skip, cfg_scale, restore_mul, sigma_hat, dt, reps):
    """den = CFG(net_out * c_out + x_hat * c_skip), restoration guidance towards x_center (None: skipped), Euler step; returns
    x_next (fp32, new tensor).  See supir_edm_step_post."""
    lib = _lib.load()
    _check_dev(net_out, x_hat, x_center)
    n = x_hat.numel()
    assert net_out.dtype == torch.float32 and net_out.is_contiguous() and net_out.numel() == reps * n
    assert x_hat.dtype == torch.float32 and x_hat.is_contiguous()
    if x_center is not None:
        if x_center.dtype != torch.float32 or not x_center.is_contiguous():
            x_center = x_center.float().contiguous()
        assert x_center.shape == x_hat.shape
    out = torch.empty_like(x_hat)
    _no_defer("edm_step_post")
    rc = lib.supir_edm_step_post(net_out.data_ptr(), x_hat.data_ptr(), _p(x_center), float(c_out), float(c_skip), float(cfg_scale),
                                 float(restore_mul), float(sigma_hat), float(dt), out.data_ptr(), n, reps, _stream())
    _lib.check(rc, "supir_edm_step_post", lib)
    return out


def _tile_hw(tiles):
    arr = (_ct.c_int * (2 * len(tiles)))()
    for j, t in enumerate(tiles):
        arr[2 * j], arr[2 * j + 1] = int(t[0]), int(t[2])     # (hi, he, wi, we) windows of _sliding_windows
    return arr


def edm_step_pre_tiles(x, eps, tiles, T, s_noise, noise_mul, c_in, reps):
    """edm_step_pre on the k tile windows `tiles` = [(hi, he, wi, we)] of the fp32 canvases x / eps ([b, C, Hc, Wc]; eps None: no
    churn): returns (x_hat [k*b, C, T, T], net_in [reps*k*b, C, T, T]), tile j / sample i at row j*b + i -- torch.cat of the crops along
    dim 0, same arithmetic per element.  See supir_edm_step_pre_tiles."""
    lib = _lib.load()
    _check_dev(x, eps)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    assert eps is None or (eps.dtype == torch.float32 and eps.is_contiguous() and eps.shape == x.shape)
    b, C, Hc, Wc = x.shape
    k = len(tiles)
    x_hat = torch.empty((k * b, C, T, T), dtype=torch.float32, device=x.device)
    net_in = torch.empty((reps * k * b, C, T, T), dtype=torch.float32, device=x.device)
    _no_defer("edm_step_pre_tiles")
    rc = lib.supir_edm_step_pre_tiles(x.data_ptr(), _p(eps), float(s_noise), float(noise_mul), float(c_in), x_hat.data_ptr(),
                                      net_in.data_ptr(), _tile_hw(tiles), k, b, C, Hc, Wc, T, reps, _stream())
    _lib.check(rc, "supir_edm_step_pre_tiles", lib)
    return x_hat, net_in


def tile_blend(out_tiles, weights, canvas, tiles, T):
    """canvas[:, :, hi:he, wi:we] += out_tiles[j*b:(j+1)*b] * weights for the k windows of `tiles`, in tile order, one launch; weights =
    float64 [T, T] (TiledRestoreEDMSampler's Gaussian): bitwise the k sequential slice-adds torch evaluates in float64 per add.  In
    place on `canvas` (fp32 [b, C, Hc, Wc]).  See supir_tile_blend."""
    lib = _lib.load()
    _check_dev(out_tiles, weights, canvas)
    b, C, Hc, Wc = canvas.shape
    k = len(tiles)
    assert canvas.dtype == torch.float32 and canvas.is_contiguous()
    assert out_tiles.dtype == torch.float32 and out_tiles.is_contiguous() and tuple(out_tiles.shape) == (k * b, C, T, T)
    assert weights.dtype == torch.float64 and weights.is_contiguous() and tuple(weights.shape) == (T, T)
    _no_defer("tile_blend")
    rc = lib.supir_tile_blend(out_tiles.data_ptr(), weights.data_ptr(), canvas.data_ptr(), _tile_hw(tiles), k, b, C, Hc, Wc, T, _stream())
    _lib.check(rc, "supir_tile_blend", lib)
    return canvas


def wavelet_decomposition(img, levels=5, want_high=True):
    """(high, low) of SUPIR/utils/colorfix.py:96-107 on fp32 [N,3,H,W]: `levels` launches of supir_wavelet_level (radius 2^i),
    ping-ponging two low-pass buffers; `high` accumulates img_i - low_i in place.  want_high=False skips the high band (the style
    image of wavelet_reconstruction only contributes its low band): returns (None, low)."""
    lib = _lib.load()
    _check_dev(img)
    assert img.dtype == torch.float32 and img.dim() == 4
    cur = img.contiguous()
    N, C, H, W = cur.shape
    high = torch.empty_like(cur) if want_high else None
    bufs = [torch.empty_like(cur), torch.empty_like(cur)]
    for i in range(levels):
        low = bufs[i & 1]
        ev = _ev()
        rc = lib.supir_wavelet_level(cur.data_ptr(), low.data_ptr(), _p(high), N * C, H, W, 2 ** i, int(i == 0), _stream())
        _lib.check(rc, "supir_wavelet_level")
        _rec("wavelet_level", 0, 4.0 * N * C * H * W * (2 + (2 if (want_high and i) else 1 if want_high else 0)), ev)
        cur = low
    return high, cur
