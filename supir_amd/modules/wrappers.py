"""ControlWrapper (sgm/modules/diffusionmodules/wrappers.py:68-102): control_model -> diffusion_model -> fp32.

`dtype` follows the attribute protocol (`model.model.dtype = ...`, test.py:67-68) and selects the element type the kernels run in
(`effective_dtype`):
  * torch.bfloat16 -> bf16 MFMA operands, fp32 accumulation and epilogues (libsupir_hip.so): the reference's own arithmetic for
    that request (`torch.autocast("cuda", dtype=torch.bfloat16)`, wrappers.py:87);
  * torch.float16 (the reference's default `diff_dtype`, options/SUPIR_v0.yaml:5, test.py:68) -> the fp16 build of the same kernels
    (libsupir_hip_f16.so: fp16 MFMA operands and activations, fp32 accumulation and epilogues), again the reference's own precision.
    FP16_NATIVE = False (env SUPIR_FP16_NATIVE=0) serves fp16 requests by bf16 instead -- never silently (RuntimeWarning on the
    first call; SUPIR_STRICT_DTYPE=1 turns it into an error);
  * torch.float32 (`test.py --diff_dtype fp32`; also the constructor default): the reference computes this request in plain fp32
    (`torch.autocast("cuda", dtype=torch.float32)` disables itself: "In CUDA autocast, but the target dtype is not supported.
    Disabling autocast.") -> the fp32 service (libsupir_hip_f32.so, supir_amd/ops_f32.py: exact-fp32 MFMA, fp32 activations, the
    reference's own weights): the same module code, the plain operator sequence (no LayerNorm folding / fused q|k|v / GroupNorm
    partials), eager launches only (no hipGraph replay).  A correctness service at 1/16 of the bf16 MFMA rate.
    weights.FP32_NATIVE = False (env SUPIR_FP32_NATIVE=0) serves fp32 requests by bf16 instead -- never silently (RuntimeWarning on
    the first call; SUPIR_STRICT_DTYPE=1 turns it into an error).

Optional hipGraph replay: one CFG-doubled step is ~1700 kernel launches issued from Python; `enable_graph()` captures
them once per (shape, control_scale) and replays the graph on later steps (inputs copied into static buffers).
"""
import os

import torch
import torch.nn as nn

from .. import weights as Wt

# hipGraph capture mode.  "global" (torch's default) fails a capture when ANY thread of the process makes a capture-unsafe HIP
# call meanwhile; with torch.distributed initialised there is such a thread (ProcessGroupNCCL's watchdog polls events), so ranks of
# a multi-GPU job capture in "thread_local" mode, which only polices the capturing thread.  SUPIR_GRAPH_CAPTURE_MODE overrides.
CAPTURE_MODE = os.environ.get("SUPIR_GRAPH_CAPTURE_MODE", "auto")


def _capture_mode():
    if CAPTURE_MODE != "auto":
        return CAPTURE_MODE
    import torch.distributed as dist
    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


# fp16 requests run on the fp16 build of the kernels (True) or are served by bf16 with a warning (False)
FP16_NATIVE = os.environ.get("SUPIR_FP16_NATIVE", "1") == "1"


class ControlWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False, dtype=torch.float32):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.control_model = None
        self.dtype = dtype
        self._graph_on = False
        self._graphs = {}
        self._cs_miss = {}             # per input shape: consecutive calls that found no graph for their control scale
        # what the SHARED step-independent buffers (text K / V^T of every cross-attention, label embedding) currently hold,
        # per (context shape, vector shape): every graph of that shape and the eager path write the same buffers in place
        self._resident = {}
        self.overlap_branches = True   # GLVControl || UNet encoder on two HIP streams
        # GLVControl and the UNet encoder are the same stack of layers on independent data: ops.paired_run records both and can
        # issue a layer pair as ONE grouped launch (two problems, 128 x 160 / 256 x 160 tiles, problem q on XCDs [4q, 4q+4)).
        # Measured (profiles/r03/pair_ab_*.log, dual_bench_*.log): the grouped kernels are faster per FLOP where they unlock the
        # 256 x 160 tile ((2048, 2560, 1280): 18.9 -> 14.0 us per problem, (8192, 640, 2560): 32.2 -> 24.0, K = 5120: 31.5 -> 28.8,
        # attention at 1024 tokens: 2 x 29.3 -> 36.8 us) and equal at (2048, 1280, 1280) (12.4 vs 12.2-12.8: fixed-cost bound) --
        # but every grouped launch is a JOIN of the two chains, and two free-running chains of full-machine kernels hide each
        # other's launch gaps and cold starts: the step is 29.9 ms on two streams, 31.2 ms with every pair grouped on one stream,
        # 30.3-31.3 ms with the pairs chosen by timing (+20-27 us per join).  Off by default; SUPIR_PAIR_BRANCHES=1 enables it.
        self.pair_branches = os.environ.get("SUPIR_PAIR_BRANCHES", "0") == "1"
        # Weight prefetch inside captured graphs: op i's GEMM kernel touches the weight of op i+distance on its way out
        # (ops.WeightPrefetch, supir_set_next_prefetch).  The cold-weight penalty is +25..40 % per GEMM (tools/cold_probe.py).
        # The earlier form -- a prefetch launch per op on a third stream -- cost more in graph nodes than it saved
        # (44.3 -> 62-65 ms per step); this one adds no launches.  0 = off.
        self.prefetch_distance = 1     # measured: 1 -> -4..5.5 % per step, 2 -> -4.4 %, 4 -> -2 %, 8 -> +0.5 %
        self.prefetch_kind = "inline"
        self._side = None
        self._warm = False
        self._dtype_noted = None   # the dtype request _note_dtype last reported on
        self._last_cdt = None
        # per-image timestep-embedding schedule (prepare_schedule / select_step): row selector shared by both networks
        self._emb_row = None
        self._emb_rows = None
        self._sched = None             # the schedule prepared last; _scheds: one per batch size (same timesteps)
        self._scheds = {}
        self._sched_armed = False

    @property
    def effective_dtype(self):
        """What the kernels compute in for the current `dtype` request."""
        if self.dtype == torch.float32 and Wt.FP32_NATIVE:
            return torch.float32
        return torch.float16 if (self.dtype == torch.float16 and FP16_NATIVE) else torch.bfloat16

    def load_control_model(self, control_model):
        self.control_model = control_model

    # ------------------------------------------------------------------ per-image embedding schedule
    def prepare_schedule(self, t_values, vector, control=None):
        """Called by a sampler that knows every timestep (table index) of the image it is about to sample: both networks build
        their time + label embedding projections for ALL steps now (openaimodel.UNetModel.prepare_schedule: three GEMMs per
        network per image) instead of three M = B GEMVs and ~12 elementwise launches at the head of every step.  A step then
        announces itself with select_step(i) right before its forward call; a forward call that was not announced takes the normal
        path, so callers that know nothing about schedules are unaffected.  `vector` must be the tensor the calls will pass as
        c["vector"]; `control` (optional) the CFG-doubled LQ latent the calls will pass as c["control"]: the control branch's
        input_hint_block convolution of it is step-invariant too and is then computed once per image as well.

        One schedule per batch size may be prepared (the tiled sampler stacks k tiles per call and has a remainder group: two
        batch sizes, same timesteps); a call is served by the schedule prepared for ITS batch."""
        with torch.no_grad(), Wt.compute_dtype(self.effective_dtype):
            if self._emb_row is None or self._emb_row.device != vector.device:
                self._emb_row = torch.zeros(1, dtype=torch.int64, device=vector.device)
                self._emb_rows = torch.arange(1024, dtype=torch.int64, device=vector.device)
            assert len(t_values) <= self._emb_rows.numel()
            v1 = self.control_model.prepare_schedule(t_values, vector, self._emb_row, control=control)
            v2 = self.diffusion_model.prepare_schedule(t_values, vector, self._emb_row)
            self.control_model.end_schedule()       # inactive until a step announces itself
            self.diffusion_model.end_schedule()
        self._sched = (len(t_values), v1, v2, self.effective_dtype, tuple(int(v) for v in t_values))
        if self._scheds and next(iter(self._scheds.values()))[4] != self._sched[4]:
            self._scheds.clear()                    # another image's timesteps: those tables are stale
        self._scheds[int(vector.shape[0])] = self._sched
        self._sched_armed = False

    def select_step(self, i, expect_t=None):
        """The NEXT forward call is step `i` of the prepared schedule; its `t` must be the schedule's i-th timestep (`expect_t`, when
        given, is checked against it on the host)."""
        if self._sched is None or not (0 <= i < self._sched[0]):
            raise IndexError("select_step without a prepared schedule / out of range")
        if expect_t is not None and int(expect_t) != self._sched[4][i]:
            raise ValueError(f"step {i} of the prepared schedule is timestep {self._sched[4][i]}, the call is for {int(expect_t)}")
        self._emb_row.copy_(self._emb_rows[i:i + 1])
        self._sched_armed = True

    def end_schedule(self):
        self._sched, self._sched_armed = None, False
        self._scheds.clear()
        for m in (self.control_model, self.diffusion_model):
            if m is not None and hasattr(m, "end_schedule"):
                m.end_schedule()

    def _use_schedule(self, kwargs, B=None):
        """Consume the one-call announcement; switch both networks' tables (the ones prepared for batch B) on / off for this call.
        Returns the schedule entry that serves the call, or None."""
        ent = self._scheds.get(B) if B is not None else self._sched
        use = ent is not None and self._sched_armed and not kwargs and ent[3] == self.effective_dtype
        self._sched_armed = False
        for m in (self.control_model, self.diffusion_model):
            if m is not None and hasattr(m, "use_schedule"):
                use = m.use_schedule(B, use) and use
        if not use:
            for m in (self.control_model, self.diffusion_model):
                if m is not None and hasattr(m, "use_schedule"):
                    m.use_schedule(None, False)
        return ent if use else None

    # ------------------------------------------------------------------ eager
    def _forward_eager(self, x, t, c, control_scale, **kwargs):
        ctx, vec = c.get("crossattn", None), c.get("vector", None)
        ckw = dict(x=c.get("control", None), timesteps=t, xt=x, control_vector=c.get("control_vector", None),
                   mask_x=c.get("mask_x", None), context=ctx, y=vec)
        if self.overlap_branches and self._warm and x.is_cuda and not kwargs:
            # The control branch and the UNet encoder+middle are independent until the first ZeroSFT: run them on two
            # streams.  Each alone launches grids of only 160-640 workgroups (M = 2048 tokens at 32x32), i.e. 1-2 per CU
            # at 25-40 % MFMA utilisation; co-scheduled they fill each other's idle CUs / MFMA slots.
            main = torch.cuda.current_stream()
            if self._side is None or self._side.device != x.device:
                self._side = torch.cuda.Stream(device=x.device)
            side = self._side
            ready = {}
            enc = None
            if self.pair_branches:
                from .. import ops
                cm, dm = self.control_model, self.diffusion_model
                emb_c, h_c = cm.prologue(ckw["x"], t, x, vec)
                emb_u, h_u = dm.encode_prologue(x, t, vec)
                control, enc = ops.paired_run(lambda: cm.body(h_c, emb_c, ctx), lambda: dm.encode_body(h_u, emb_u, ctx), side=side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if enc is None:
                    control = self.control_model(**ckw)
                ev_control = torch.cuda.Event()
                ev_control.record(side)

                def on_done(idx, tensors):
                    ev = torch.cuda.Event()
                    ev.record(side)
                    ready[idx] = (tensors, ev)

                # control-side halves of the adapters keep the side stream busy while the main stream runs the decoder
                self.diffusion_model.adapter_control_sides(control, on_done)
            if enc is None:
                enc = self.diffusion_model.encode(x, timesteps=t, context=ctx, y=vec)
            main.wait_event(ev_control)
            for h in control:
                h.record_stream(main)

            def pre(idx):
                tensors, ev = ready[idx]
                main.wait_event(ev)
                for tt in (tensors if isinstance(tensors, tuple) else (tensors,)):
                    tt.record_stream(main)
                return tensors

            out = self.diffusion_model(x, timesteps=t, context=ctx, y=vec, control=control, control_scale=control_scale,
                                       encoded=enc, adapter_pre=pre)
            side.wait_stream(main)   # join: nothing of this step is still running on the side stream afterwards
        else:
            control = self.control_model(**ckw)
            out = self.diffusion_model(x, timesteps=t, context=ctx, y=vec, control=control, control_scale=control_scale,
                                       **kwargs)
            self._warm = True   # weight / context caches are now built (they are filled on the calling stream)
        if ctx is not None and vec is not None:
            # this call (re)filled the shared text-K/V^T / label buffers with ITS conditioning: graphs of the same shape that
            # were captured with another prompt must refresh before their next replay (_forward_graph checks this token)
            self._resident[(tuple(ctx.shape), tuple(vec.shape), Wt.cdt())] = (ctx, ctx._version, vec, vec._version)
        return out.float()

    # ------------------------------------------------------------------ hipGraph replay
    def enable_graph(self, on=True):
        self._graph_on = bool(on)
        if not on:
            self._graphs.clear()
            self._cs_miss.clear()

    def _forward_graph(self, x, t, c, control_scale, sched=None):
        ctx, vec, ctl = c["crossattn"], c["vector"], c["control"]
        # a step of a prepared embedding schedule reads its embeddings out of the per-image tables (other launches than a plain
        # call): its own graph, valid as long as the tables are the same buffers (version)
        key = (tuple(x.shape), float(control_scale), tuple(ctx.shape), tuple(vec.shape), Wt.cdt()) + ((sched[1:3],) if sched else ())
        g = self._graphs.get(key)
        if g is None:
            # control_scale is a launch argument baked into the captured kernels.  With use_linear_control_scale
            # (sampling.py:557-559) it changes on every step: capturing a graph per value would cost three network calls per
            # step, so after two CONSECUTIVE misses for a shape such calls run eagerly.  A hit resets the streak, so a new
            # constant scale on a later image (1.0, then 0.9, then 0.8 ...) still gets its graph.
            skey = (key[0], key[2], key[3], key[4]) + key[5:]
            self._cs_miss[skey] = self._cs_miss.get(skey, 0) + 1
            if self._cs_miss[skey] > 2:
                return self._forward_eager(x, t, c, control_scale)
            if len(self._graphs) >= 4:
                self._graphs.clear()
            sx, st, sc = x.clone(), t.clone(), ctl.clone()
            cond = {"crossattn": ctx, "vector": vec, "control": sc}
            # warm-up on a side stream: fills the weight / text-KV / label caches OUTSIDE the capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            from .. import ops
            pf = ops.WeightPrefetch(self.prefetch_distance, self.prefetch_kind) if self.prefetch_distance > 0 else None
            with torch.cuda.stream(s):
                self._forward_eager(sx, st, cond, control_scale)
                if pf is not None:            # second warm-up pass doubles as the recording pass (op order is static)
                    ops.set_prefetch(pf)
                    pf.begin_record()
                self._forward_eager(sx, st, cond, control_scale)
                if pf is not None:
                    pf.end()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode=_capture_mode()):
                if pf is not None:
                    pf.begin_replay(x.device)
                out = self._forward_eager(sx, st, cond, control_scale)
                if pf is not None:
                    pf.end()
            ops.set_prefetch(None)
            g = [graph, sx, st, sc, out]
            self._graphs[key] = g
        else:
            self._cs_miss[(key[0], key[2], key[3], key[4]) + key[5:]] = 0
        graph, sx, st, sc, out = g[:5]
        rkey = (key[2], key[3], key[4])
        res = self._resident.get(rkey)
        if res is None or res[0] is not ctx or res[1] != ctx._version or res[2] is not vec or res[3] != vec._version:
            # the shared buffers hold another prompt's K / V^T / label embedding (a new prompt, or another graph / an eager call
            # of the same shape ran in between): refresh them in place, keep the graph
            self.control_model.refresh_static_conditioning(ctx, vec)
            self.diffusion_model.refresh_static_conditioning(ctx, vec)
            self._resident[rkey] = (ctx, ctx._version, vec, vec._version)
        sx.copy_(x)
        st.copy_(t)
        sc.copy_(ctl)
        graph.replay()
        return out

    def _note_dtype(self):
        """Once per requested dtype: say when the kernels compute narrower than the request (module docstring)."""
        if self._dtype_noted == self.dtype:
            return
        self._dtype_noted = self.dtype
        if self.dtype == torch.float16 and not FP16_NATIVE:
            Wt.note_downgrade("ControlWrapper.dtype", "torch.float16 (the reference's default diff_dtype)", "torch.bfloat16",
                              "SUPIR_FP16_NATIVE=0 keeps fp16 requests off the fp16 build of the kernels; bf16 has 7 mantissa bits "
                              "instead of fp16's 10 (per-call rel-L2 vs fp32 ~7e-3 instead of ~1e-3).", stacklevel=4)
        elif self.dtype not in (torch.float16, torch.bfloat16) and self.effective_dtype != self.dtype:
            Wt.note_downgrade("ControlWrapper.dtype", f"{self.dtype} (test.py --diff_dtype fp32)", "torch.bfloat16",
                              "the reference computes this request in true fp32 (torch.autocast disables itself for float32, "
                              "sgm/modules/diffusionmodules/wrappers.py:87); SUPIR_FP32_NATIVE=0 keeps fp32 requests off the fp32 "
                              "service (per-call rel-L2 vs fp32 ~7e-3).", stacklevel=4)

    def forward(self, x, t, c, control_scale=1, **kwargs):
        self._note_dtype()
        if self.effective_dtype != self._last_cdt:
            # The derived weight layouts (base.Prep) exist in ONE element type at a time: a call in the other type rebuilds them
            # and frees the buffers the kernels of an earlier capture point at.  Graphs (and the record of what the shared text
            # K / V^T buffers hold) therefore do not survive a change of compute dtype; they are re-captured on demand.
            self._graphs.clear()
            self._cs_miss.clear()
            self._resident.clear()
            self._last_cdt = self.effective_dtype
        vec = c.get("vector", None) if isinstance(c, dict) else None
        with torch.no_grad(), Wt.compute_dtype(self.effective_dtype):
            # (inside the scope: the networks keep one table per element type and look theirs up by the scope's)
            sched = self._use_schedule(kwargs, None if vec is None else int(vec.shape[0]))   # the schedule entry serving this call, or None
            if self._graph_on and not kwargs and x.is_cuda and self.effective_dtype != torch.float32:
                return self._forward_graph(x, t, c, control_scale, sched)
            return self._forward_eager(x, t, c, control_scale, **kwargs)
