"""Training harness of the path (the build's counterpart of ywz/mywork/newtrain1.py:74-111): R-D loss, the
two-optimiser update order, and plain data parallelism -- one process per GPU, gradients averaged with RCCL
all-reduce over xGMI (``torch.distributed`` backend "nccl" on ROCm).

The reference has no multi-device code at all; pairs are independent, the loss normalises by the
LOCAL batch (newtrain1.py:45-47), so averaging rank gradients reproduces the single-process gradient of the
concatenated batch exactly.

Layout (round 2).  The parameters of an optimiser group live in ONE flat fp32 buffer and so do their gradients
(``FlatGroup``): ``p.data`` / ``p.grad`` are views.  That makes the bookkeeping of a step a handful of launches: one fill
clears all gradients, the weight-gradient kernels ADD into their slot in place (``functional.GradSlot``: no unpack, no
AccumulateGrad copy/add), the all-reduce runs in place on contiguous slices of the flat gradient (no ``torch.cat`` /
``copy_`` pack and unpack), one Adam launch updates everything.  The collective is issued per bucket as soon as the
bucket's last gradient has been written, so it overlaps the rest of the backward pass, and every piece is capturable:
``GraphedTrainer`` records the step -- collectives included -- into one HIP graph per rank.
"""
from __future__ import annotations

import ctypes as C
import os as _os

import torch
import torch.distributed as dist

from . import functional as Fn


class MultiTensorAdam(torch.optim.Adam):
    """``torch.optim.Adam(params, lr)`` with the update of all tensors in a handful of HIP launches
    (``hesic_adam_step``: 24 tensors per launch, descriptors in the kernel arguments).

    State layout and ``state_dict`` are those of ``torch.optim.Adam(capturable=True)`` (per-parameter fp32 ``step`` on the
    device, ``exp_avg``, ``exp_avg_sq``), so checkpoints move between the two.  Why: the multi-tensor (foreach) torch
    step falls back to one elementwise launch per parameter for its 0-dim step counters -- ~290 launches, 1.2 ms of the
    9 ms graphed training step.  Anything this kernel does not cover (CPU tensors, non-fp32, non-contiguous, amsgrad,
    weight decay, maximize) goes to the parent's step.  Self-contained: ``step()`` bumps the version counters of what it
    updated and starts a new epoch for the packed-weight caches, like any in-place optimiser."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, capturable=True, foreach=True)

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib as L
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            ok = (not group.get("amsgrad") and not group.get("weight_decay") and not group.get("maximize")
                  and not isinstance(group["lr"], torch.Tensor)
                  and all(p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and not p.grad.is_sparse
                          and p.is_contiguous() and p.grad.is_contiguous() for p in ps))
            if not ok:
                return super().step(closure)
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            b1, b2 = group["betas"]
            for i in range(0, len(ps), L.ADAM_MAX_TENSORS):
                part = ps[i:i + L.ADAM_MAX_TENSORS]
                c = L.AdamChunk()
                for j, p in enumerate(part):
                    st = self.state[p]
                    c.p[j], c.g[j], c.m[j], c.v[j] = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    c.step[j], c.numel[j] = st["step"].data_ptr(), p.numel()
                c.n, c.lr, c.beta1, c.beta2, c.eps = len(part), float(group["lr"]), float(b1), float(b2), float(group["eps"])
                L.call("hesic_adam_step", C.byref(c), L.stream())
            # the kernel writes through raw pointers: tell PyTorch (version counters, like any in-place optimiser) and the
            # packed-weight / GDN / bottleneck caches (keyed on those counters) that the parameters moved
            if ps:
                torch.autograd.graph.increment_version(ps)
        return loss


# ------------------------------------------------------------------------------------------------ flat buffers
class FlatGroup:
    """The parameters of one optimiser group as views into one flat fp32 buffer, their gradients as views into another.

    ``p.data`` and ``p.grad`` are re-pointed (values are kept); state-dict keys, shapes and ``nn.Module`` structure are
    untouched.  Every tensor starts on a 256-byte boundary (the kernels read biases / GDN parameters with 16-byte loads).
    Moving the module afterwards (``.to()``, ``.cuda()``) would break the aliasing: build the group last."""

    ALIGN = 64      # floats

    def __init__(self, params):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError("FlatGroup: no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatGroup: fp32 parameters on one device")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_views = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.detach())
                p.data = view
                g = self.flat_g[o:o + p.numel()].view(p.shape)
                p.grad = g
                self.grad_views.append(g)
        self.slots = [Fn.GradSlot(g, name=str(i)) for i, g in enumerate(self.grad_views)]
        if dev.type == "cuda":
            import weakref
            keys = Fn.register_grad_slots(zip(self.params, self.slots))
            weakref.finalize(self, Fn.clear_grad_slots, keys)       # a dead group must not keep its gradient buffer registered

    def zero_grad(self):
        """One fill for every gradient of the group (a kernel, not a memset: graph-safe); re-attaches ``p.grad`` views that
        user code dropped (``zero_grad(set_to_none=True)``)."""
        self.flat_g.fill_(0)
        for p, g, s in zip(self.params, self.grad_views, self.slots):
            if p.grad is not g:
                p.grad = g
            s.writes = 0

    def view_like_params(self, flat):
        return [flat[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]


class FlatAdam:
    """``torch.optim.Adam(params, lr)`` (newtrain1.py:294-295; no amsgrad / weight decay) over a ``FlatGroup``: ONE update
    launch for the whole group (``hesic_adam_step`` on the flat buffers; padding elements have zero gradient and stay
    zero).  ``state_dict()`` / ``load_state_dict()`` speak ``torch.optim.Adam``'s per-parameter format."""

    def __init__(self, group: FlatGroup, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.group, self.betas, self.eps = group, (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(group.flat_p)
        self.exp_avg_sq = torch.zeros_like(group.flat_p)
        self.step_count = torch.zeros((), dtype=torch.float32, device=group.flat_p.device)
        self.param_groups = [{"params": group.params, "lr": float(lr), "betas": self.betas, "eps": self.eps}]

    def zero_grad(self, set_to_none=False):
        self.group.zero_grad()

    @torch.no_grad()
    def step(self):
        from . import _lib as L
        g = self.group
        c = L.AdamChunk()
        c.p[0], c.g[0], c.m[0], c.v[0] = g.flat_p.data_ptr(), g.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        c.step[0], c.numel[0] = self.step_count.data_ptr(), g.numel
        c.n, c.lr, c.beta1, c.beta2, c.eps = 1, float(self.param_groups[0]["lr"]), self.betas[0], self.betas[1], self.eps
        L.call("hesic_adam_step", C.byref(c), L.stream())
        # the kernel writes through raw pointers: bump the version counters (like any in-place optimiser) -- every cache of
        # packed weights / GDN parameters / bottleneck tables is keyed on them
        torch.autograd.graph.increment_version(g.params)

    def state_dict(self):
        m, v = self.group.view_like_params(self.exp_avg), self.group.view_like_params(self.exp_avg_sq)
        state = {i: {"step": self.step_count.clone(), "exp_avg": m[i].clone(), "exp_avg_sq": v[i].clone()} for i in range(len(m))}
        pg = {"lr": self.param_groups[0]["lr"], "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
              "foreach": None, "capturable": True, "differentiable": False, "fused": None, "params": list(range(len(m)))}
        return {"state": state, "param_groups": [pg]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        m, v = self.group.view_like_params(self.exp_avg), self.group.view_like_params(self.exp_avg_sq)
        for i, st in sd["state"].items():
            m[int(i)].copy_(st["exp_avg"])
            v[int(i)].copy_(st["exp_avg_sq"])
            self.step_count.fill_(float(st["step"]))
        if sd.get("param_groups"):
            self.param_groups[0]["lr"] = sd["param_groups"][0].get("lr", self.param_groups[0]["lr"])


class FlatReducer:
    """Gradient averaging over the ranks, in place on a ``FlatGroup``'s flat gradient buffer.

    The buffer is cut into ~``bucket_mb`` buckets of whole parameters.  With ``overlap`` a bucket's all-reduce is issued
    (``async_op``: on the backend's own stream, after an event on the compute stream) the moment its last gradient has been
    written -- gradient kernels report through ``GradSlot.on_write``, parameters that go through autograd's AccumulateGrad
    through a post-accumulate hook -- so the collective runs under the rest of the backward pass.  How many writes complete
    a parameter (encoder1's weights receive two per step) is learned in the first step, which reduces everything at
    ``finish()``.  ``finish()`` launches what is left (parameters without gradient on this rank still take part: ranks stay
    in step), waits and -- for backends without an averaging reduction (gloo) -- divides.  world_size 1: all no-ops.

    ``collective`` (round 5; environment default ``HESIC_DP_COLLECTIVE``): "allreduce" -- one in-place all-reduce per bucket (a ring on
    RCCL: 2 (N-1)/N of the bucket over ONE xGMI link per hop, ~1.6 ms for the 140 MB of HESIC at N = 8) -- or "rsag": the same
    sum as a reduce-scatter into this rank's 1/N of the bucket followed by an all-gather of the shards, both in place on the flat buffer;
    on a fully connected xGMI node every rank then exchanges 1/N of the bucket with each of its 7 peers at once (~0.23 ms).  The part of a
    bucket beyond a multiple of N elements (< N values) goes through a small all-reduce.  Values are equal to "allreduce" up to the
    summation order (``tests/test_dp_gloo.py``); unmeasured on hardware until an 8-GPU node exists."""

    def __init__(self, group: FlatGroup, bucket_mb: float = 25.0, process_group=None, overlap=True, force=False, collective=None):
        self.group, self.pg, self.overlap = group, process_group, overlap
        self.collective = collective or _os.environ.get("HESIC_DP_COLLECTIVE", "allreduce")
        if self.collective not in ("allreduce", "rsag"):
            raise ValueError("FlatReducer: collective must be 'allreduce' or 'rsag'")
        up = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if up else 1
        self.buckets, self._work, self._expected, self._events = [], [], None, [0] * len(group.params)
        self.compute_stream = self.side_stream = None
        # ``force``: issue the collectives even in a 1-rank group (a 1-GPU box can then exercise RCCL bring-up and the capture
        # of the all-reduces into the step's HIP graph)
        self.active = self.world > 1 or (force and up)
        if not self.active:
            return
        self.avg_op = dist.get_backend(process_group) == "nccl"
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        lo, members = 0, []
        for i in range(len(group.params)):
            end = group.offsets[i + 1] if i + 1 < len(group.params) else group.numel
            members.append(i)
            if end - lo >= cap or i + 1 == len(group.params):
                self.buckets.append({"lo": lo, "hi": end, "members": members, "pending": len(members), "launched": False})
                lo, members = end, []
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for i in b["members"]:
                self._bucket_of[i] = bi
        # the closures hold the reducer only weakly and the hook handles are kept: a second reducer / Trainer on the same model
        # must not leave this one's hooks firing (and its flat buffers alive) -- ``close()`` removes them
        import weakref
        me = weakref.ref(self)

        def fire(i):
            r = me()
            if r is not None:
                r._event(i)
        self._handles = []
        for i, (p, s) in enumerate(zip(group.params, group.slots)):
            s.on_write = (lambda slot, i=i: fire(i))
            self._handles.append(p.register_post_accumulate_grad_hook(lambda p_, i=i: fire(i)))
        # diagnostics (``timing = True``, eager steps on RCCL only): every bucket's all-reduce runs synchronously on a communication
        # stream of its own between two events, ``report()`` gives per-bucket durations and the share hidden under the backward pass
        self.timing, self._comm, self._timed, self._bwd_end = False, None, [], None

    def close(self):
        """Detach from the parameters (hooks, slot callbacks)."""
        for h in getattr(self, "_handles", []):
            h.remove()
        self._handles = []
        for s in self.group.slots:
            s.on_write = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin(self):
        """Call before the backward pass of every step."""
        if not self.active:
            return
        self._events = [0] * len(self.group.params)
        self.compute_stream = torch.cuda.current_stream() if self.group.flat_g.is_cuda else None
        for b in self.buckets:
            b["pending"], b["launched"] = len(b["members"]), False

    def _event(self, i):
        self._events[i] += 1
        if self._expected is None or self._events[i] != self._expected[i]:
            return
        b = self.buckets[self._bucket_of[i]]
        b["pending"] -= 1
        if b["pending"] == 0 and self.overlap and not b["launched"]:
            self._launch(b)

    def _launch(self, b):
        b["launched"] = True
        op = dist.ReduceOp.AVG if self.avg_op else dist.ReduceOp.SUM
        buf = self.group.flat_g[b["lo"]:b["hi"]]
        if buf.is_cuda:
            # a bucket's gradients are written on the compute stream AND (weight gradients, train.Trainer) on ``side_stream``: the
            # stream the collective is issued from waits for both
            cur = torch.cuda.current_stream()
            for st in (self.compute_stream, self.side_stream):
                if st is not None and st != cur:
                    cur.wait_stream(st)
        if self.timing and self.avg_op and buf.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=buf.device)
            self._comm.wait_stream(torch.cuda.current_stream())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self._comm):
                e0.record()
                for w in self._collect(buf, op, False):              # synchronous on the communication stream: e1 is its completion
                    pass
                e1.record()
            self._timed.append((b["hi"] - b["lo"], e0, e1))
            return
        self._work.extend(self._collect(buf, op, True))

    def _collect(self, buf, op, async_op):
        """The bucket's sum over the ranks, in place: one all-reduce, or reduce-scatter + all-gather (see the class docstring)."""
        if self.collective == "allreduce" or self.world < 2:
            return [dist.all_reduce(buf, op=op, group=self.pg, async_op=async_op)]
        n, W = buf.numel(), self.world
        main = n - n % W
        works = []
        if main:
            rank = dist.get_rank(self.pg)
            shard = buf[rank * (main // W):(rank + 1) * (main // W)]
            # the all-gather reads the shard the reduce-scatter wrote: the two are issued in order on the backend's stream; with async_op the
            # first is waited for before the second is issued only on backends that run collectives on the caller's thread (gloo)
            w1 = dist.reduce_scatter_tensor(shard, buf[:main], op=op, group=self.pg, async_op=async_op)
            if async_op and not self.avg_op:
                w1.wait()
                w1 = None
            w2 = dist.all_gather_into_tensor(buf[:main], shard, group=self.pg, async_op=async_op)
            works += [w for w in (w1, w2) if w is not None]
        if main < n:
            works.append(dist.all_reduce(buf[main:], op=op, group=self.pg, async_op=async_op))
        return [w for w in works if w is not None]

    def finish(self):
        """Call once after the backward pass: launches the remaining buckets, waits, averages."""
        if not self.active:
            return
        if self.timing and self.avg_op and self.group.flat_g.is_cuda:
            self._bwd_end = torch.cuda.Event(enable_timing=True)
            self._bwd_end.record()                       # the backward pass ends here on the compute stream
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        if self._comm is not None and self._timed:
            torch.cuda.current_stream().wait_stream(self._comm)
        for w in self._work:
            w.wait()
        self._work.clear()
        if not self.avg_op and self.world > 1:
            self.group.flat_g.div_(self.world)
        if self._expected is None:
            # a parameter that saw no gradient event this step completes its bucket at finish() in later steps too
            self._expected = [e if e > 0 else -1 for e in self._events]


WGRAD_STREAM = False      # module switch (off): weight gradients on a stream of their own -- measured 12.34 vs 12.28 ms at B=8 512^2: the step is not gap-bound


class Trainer:
    """Holds the model, ``Adam(parameters, lr)`` + ``Adam(aux_parameters, aux_lr)`` (newtrain1.py:294-295) over flat
    parameter / gradient buffers and, when a process group is up, one in-place reducer per optimiser group."""

    def __init__(self, model, lr=1e-4, aux_lr=1e-3, lmbda=1e-2, bucket_mb=25.0, overlap=True, force_collectives=False, collective=None):
        self.model, self.lmbda = model, float(lmbda)
        main, aux = list(model.parameters()), list(model.aux_parameters())
        self.main_group, self.aux_group = FlatGroup(main), FlatGroup(aux)
        self.on_gpu = self.main_group.flat_p.is_cuda
        if self.on_gpu:
            self.optimizer = FlatAdam(self.main_group, lr=lr)
            self.aux_optimizer = FlatAdam(self.aux_group, lr=aux_lr)
        else:       # host modules (the gloo tests): torch's Adam on the same views
            self.optimizer = torch.optim.Adam(self.main_group.params, lr=lr)
            self.aux_optimizer = torch.optim.Adam(self.aux_group.params, lr=aux_lr)
        # EB matrices/biases/factors get their gradient from the main backward but are stepped by the aux
        # optimiser after the aux backward adds the quantile gradient: both groups are reduced,
        # the aux group only after the aux backward.
        self.main_reducer = FlatReducer(self.main_group, bucket_mb, overlap=overlap, force=force_collectives, collective=collective)
        self.aux_reducer = FlatReducer(self.aux_group, bucket_mb, overlap=False, force=force_collectives, collective=collective)
        self.world = self.main_reducer.world

    def _forward_loss(self, x1, x2, h_matrix, noise):
        out = self.model(x1, x2, h_matrix, noise=noise)
        return Fn.rd_loss(out, x1, x2, self.lmbda)

    def step(self, x1, x2, h_matrix, noise=None):
        """One iteration in the reference's order: zero both -> forward -> R-D loss backward -> (reduce) -> optimizer.step
        -> aux loss backward -> (reduce) -> aux_optimizer.step.  Returns the loss dict (device scalars, no sync)."""
        self.model.train()
        self.main_group.zero_grad()
        self.aux_group.zero_grad()
        self.main_reducer.begin()
        self.aux_reducer.begin()
        prev = Fn.train_pack_cache(True)          # packed conv weights persist across the step, one batched repack below
        scaled, Fn.SCALED_LOSS = Fn.SCALED_LOSS, False     # backward() starts at the unscaled loss: no g_loss multiplies
        slots = Fn.grad_slots_active(True)        # gradient kernels add straight into the flat buffer (cleared above) for THIS step only
        wst = None
        if self.on_gpu and WGRAD_STREAM:
            if getattr(self, "_wstream", None) is None:
                self._wstream = torch.cuda.Stream(device=self.main_group.flat_p.device)
            wst = self._wstream
            wst.wait_stream(torch.cuda.current_stream())       # behind zero_grad
            self.main_reducer.side_stream = wst
        wprev = Fn.set_wgrad_stream(wst)
        defer = Fn.defer_wgrad_finish(self.on_gpu and wst is None)     # finishing passes of the weight gradients: batches of 8 layers per launch
        try:
            crit = self._forward_loss(x1, x2, h_matrix, noise)
            crit["loss"].backward()
        finally:
            Fn.defer_wgrad_finish(defer)          # flushes what is still queued: every weight gradient is in (or on its way into) the flat buffer
            Fn.train_pack_cache(prev)
            Fn.SCALED_LOSS = scaled
            Fn.grad_slots_active(slots)
            Fn.set_wgrad_stream(wprev)
            if wst is not None:
                torch.cuda.current_stream().wait_stream(wst)    # every weight gradient is in the flat buffer from here on
        self.main_reducer.finish()
        self.optimizer.step()
        if self.on_gpu:
            Fn.repack_all()                       # every packed conv weight of the step refreshed (and re-tagged) in ONE launch
        aux = self.model.aux_loss()
        slots = Fn.grad_slots_active(True)
        try:
            aux.backward()
        finally:
            Fn.grad_slots_active(slots)
        self.aux_reducer.finish()
        self.aux_optimizer.step()
        # detached scalars only: a returned loss that still requires grad would keep this step's autograd graph (and its
        # AccumulateGrad nodes, bound to this step's stream) alive into the next one
        crit = {k: v.detach() for k, v in crit.items()}
        crit["aux_loss"] = aux.detach()
        return crit


class GraphedTrainer(Trainer):
    """``Trainer.step`` recorded once into a HIP graph and replayed -- with a process group up, one graph per rank with the
    bucketed RCCL all-reduces inside it (collectives are stream work like any kernel; their forked comm-stream nodes keep the
    overlap with the backward pass in the graph).

    One training step is a few hundred launches, most of them a few microseconds long: eagerly the Python/ctypes launch work
    exceeds the GPU time, so the step is host-bound.  The first ``warmup`` calls run eagerly on a side stream (they are real
    steps: they pack weights, size workspaces, let autograd allocate and teach the reducer how many writes complete each
    gradient), the next call captures zero_grad -> forward -> R-D backward -> reduce -> Adam -> aux backward -> reduce -> aux
    Adam and every later call is a copy of the inputs into the static buffers plus one graph launch.  The quantisation noise
    is drawn inside the graph by the graph-safe Philox generator unless a ``noise`` dict is given at capture time (then it is
    a static input too).  The returned dict holds the graph's static loss tensors (overwritten by the next call)."""

    def __init__(self, model, *args, warmup=3, **kw):
        super().__init__(model, *args, **kw)
        if not self.on_gpu:
            raise RuntimeError("GraphedTrainer needs the model on a ROCm device")
        if int(warmup) < 1:
            raise ValueError("GraphedTrainer: warmup >= 1 (the packed-weight registry and the reducer's write counts are created by an "
                             "eager step)")
        self.warmup, self.calls, self.graph = int(warmup), 0, None
        self._in = self._noise = self._out = None
        # only RCCL's collectives are stream work; gloo stages through the host (synchronises) and cannot be captured
        self.capturable = not (self.main_reducer.active and not self.main_reducer.avg_op)
        if not self.capturable:
            import warnings
            warnings.warn("GraphedTrainer: the process group's backend is not nccl/RCCL -- its collectives cannot be captured into a HIP "
                          "graph, steps run eagerly")

    def _stage(self, x1, x2, h_matrix, noise):
        if self._in is None:
            self._in = (x1.clone(), x2.clone(), h_matrix.clone())
            self._noise = None if noise is None else {k: v.clone() for k, v in noise.items()}
        else:
            for dst, src in zip(self._in, (x1, x2, h_matrix)):
                if src.data_ptr() != dst.data_ptr():
                    dst.copy_(src)
            if self._noise is not None:
                if noise is None:
                    raise RuntimeError("GraphedTrainer: the step was captured with explicit noise tensors")
                for k, dst in self._noise.items():
                    dst.copy_(noise[k])

    def step(self, x1, x2, h_matrix, noise=None):
        if not self.capturable:
            return super().step(x1, x2, h_matrix, noise=noise)
        self._stage(x1, x2, h_matrix, noise)
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup:
            side = torch.cuda.Stream(device=x1.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = super().step(*self._in, noise=self._noise)
            torch.cuda.current_stream().wait_stream(side)
            return out
        if self.graph is None:
            self.graph = torch.cuda.CUDAGraph()
            # The process group's watchdog thread polls the events of the warm-up steps' collectives; under the default ("global")
            # capture mode such a query from ANOTHER thread while this one captures is an error that takes the process down
            # ("operation not permitted when stream is capturing", seen with one RCCL rank).  Drain first, and confine the capture
            # rules to this thread.
            torch.cuda.synchronize()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._out = super().step(*self._in, noise=self._noise)
        self.graph.replay()
        # a replay runs no Python: the captured Adam kernels moved the parameters without touching their version counters, so
        # every cache keyed on them (bottleneck tables, packed GDN parameters, inference weight packs) must start a new epoch
        Fn.invalidate_weight_cache()
        return self._out


def comm_report(reducer):
    """Per-bucket all-reduce timings of the steps run with ``reducer.timing = True`` (synchronises the device): a list of
    {mb, ms, gbps} per launch order, the total, and ``hidden_frac`` = the share of the communication time that lay before the end
    of the backward pass on the compute stream (what the overlap hides; 0 for a reducer built with overlap=False)."""
    torch.cuda.synchronize()
    recs, total, hidden = [], 0.0, 0.0
    for numel, e0, e1 in reducer._timed:
        ms = e0.elapsed_time(e1)
        recs.append({"mb": round(numel * 4 / 1e6, 2), "ms": round(ms, 4), "gbps": round(numel * 4 / 1e6 / max(ms, 1e-6), 1)})
        total += ms
        if reducer._bwd_end is not None:
            # [t0, t1]: this all-reduce on a clock whose zero is the end of the backward pass (event times are comparable across
            # streams; negative = before it); the part before zero ran under the backward pass
            t0, t1 = reducer._bwd_end.elapsed_time(e0), reducer._bwd_end.elapsed_time(e1)
            hidden += max(0.0, min(t1, 0.0) - min(t0, 0.0))
    reducer._timed.clear()
    return {"buckets": recs, "total_ms": round(total, 4), "hidden_frac": round(hidden / total, 4) if total > 0 else None}


def init_distributed(backend=None):
    """Process-group bring-up from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, local_rank); a plain single-process run returns (0, 1, 0)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        # HESIC_DIST_BACKEND=gloo + HESIC_SINGLE_DEVICE=1: several ranks sharing ONE GPU (functional check of the multi-rank code
        # paths on a 1-GPU box; RCCL itself refuses two ranks on one device)
        backend = os.environ.get("HESIC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("HESIC_SINGLE_DEVICE"):
        local = 0
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class Stage2Trainer:
    """Stage 2 of the reference's schedule (ywz/mywork/newtrain6_real.py:120-167): the compression model is FROZEN in eval mode (rounded
    latents, no noise), the enhancement net ``Independent_EN`` trains on its reconstructions with
    ``loss = lambda * 255^2 * (MSE(x1_hat', x1) + MSE(x2_hat', x2))`` (:83-91, ``kind=0``) and one Adam over ITS parameters (:287); the
    bottlenecks' aux loss is computed there but never stepped (:169-171).  The frozen forward runs under ``no_grad`` in whatever 16-bit /
    fp32 mode is selected for inference; the enhancement net's forward / backward in ``train_dtype`` (bf16 or fp32)."""

    def __init__(self, model, enhancer, lr=1e-4, lmbda=1e-2, train_dtype=None):
        self.model, self.enhancer, self.lmbda = model, enhancer, float(lmbda)
        # training runs in bfloat16 (fp32 exponent range) or fp32: with float16 inference selected the enhancer needs an explicit training
        # format (with train_dtype=None the first step failed inside an operator instead of here)
        if train_dtype is None and Fn.compute_dtype() == torch.float16:
            train_dtype = torch.bfloat16
        if train_dtype not in (None, torch.bfloat16, torch.float32):
            raise ValueError("Stage2Trainer: train_dtype is None, torch.bfloat16 or torch.float32 (float16 is an inference format)")
        self.train_dtype = train_dtype
        self.optimizer = MultiTensorAdam(list(enhancer.parameters()), lr=lr) if next(enhancer.parameters()).is_cuda else \
            torch.optim.Adam(enhancer.parameters(), lr=lr)

    def step(self, x1, x2, h_matrix):
        self.model.eval()
        self.enhancer.train()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.no_grad():
            out = self.model(x1, x2, h_matrix)
        keep = Fn.compute_dtype()
        if self.train_dtype is not None and self.train_dtype != keep:
            Fn.set_compute_dtype(self.train_dtype)
        try:
            out2 = self.enhancer(out["x1_hat"].float(), out["x2_hat"].float(), h_matrix)
            mse = ((out2["x1_hat"].float() - x1) ** 2).mean() + ((out2["x2_hat"].float() - x2) ** 2).mean()
            loss = self.lmbda * 255 ** 2 * mse
            loss.backward()
        finally:
            if Fn.compute_dtype() != keep:
                Fn.set_compute_dtype(keep)
        self.optimizer.step()
        return {"loss": loss.detach(), "mse_loss": mse.detach()}
