"""Training harness of the path (the build's counterpart of ywz/mywork/newtrain1.py:74-111):
R-D loss, the two-optimiser update order, and plain data parallelism -- one process per GPU, gradients
summed with RCCL all-reduce over xGMI (``torch.distributed`` backend "nccl" on ROCm), bucketed and
launched from autograd hooks so the collective overlaps the rest of the backward pass.

The reference has no multi-device code at all; pairs are independent, the loss
normalises by the LOCAL batch (newtrain1.py:45-47), so averaging rank gradients reproduces the
single-process gradient of the concatenated batch exactly.
"""
from __future__ import annotations

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
    weight decay, maximize) goes to the parent's step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, capturable=True, foreach=True)

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib as L
        import ctypes as C
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
        Fn.invalidate_weight_cache()
        return loss


class GradBucketReducer:
    """Bucketed asynchronous gradient all-reduce (average).

    Parameters are grouped into ~``bucket_mb`` buckets in reverse registration order (the order autograd
    finishes them); a post-accumulate hook counts arrivals and, when a bucket is complete, packs it into one
    flat buffer and starts ``all_reduce(async_op=True)``.  ``finish()`` waits and scatters the averages back.
    With world_size == 1 (or no process group) everything is a no-op.
    """

    def __init__(self, params, bucket_mb: float = 25.0, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets, self._of, self._pending, self._work = [], {}, [], []
        if self.world == 1:
            return
        cap = int(bucket_mb * (1 << 20))
        cur, size = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * 4
            if cur and size + nbytes > cap:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._of[p] = bi
                p.register_post_accumulate_grad_hook(self._hook)
        self._pending = [len(b) for b in self.buckets]

    def _hook(self, p):
        bi = self._of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1).float() for p in ps])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._work.append((work, flat, ps))

    def finish(self):
        """Wait for every bucket (launching the ones whose params got no gradient hook call) and write the
        averaged gradients back.  Call once after each backward."""
        if self.world == 1:
            return
        for bi, left in enumerate(self._pending):
            if left > 0:          # parameters unused in this backward: reduce what exists so ranks stay in step
                self._launch(bi)
        for work, flat, ps in self._work:
            work.wait()
            flat.div_(self.world)
            off = 0
            for p in ps:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._work.clear()
        self._pending = [len(b) for b in self.buckets]


class Trainer:
    """Holds the model, ``Adam(parameters, lr)`` + ``Adam(aux_parameters, aux_lr)`` (newtrain1.py:294-295)
    and, when a process group is up, one reducer per optimiser group."""

    def __init__(self, model, lr=1e-4, aux_lr=1e-3, lmbda=1e-2, bucket_mb=25.0, fused=False, capturable=False, multi_tensor=True):
        self.model, self.lmbda = model, float(lmbda)
        main, aux = list(model.parameters()), list(model.aux_parameters())
        # multi-tensor (foreach) Adam by default: on this ROCm build the fused Adam kernel takes visibly
        # smaller first steps than the reference's plain Adam (measured: loss 220.4 -> 216.1 vs 220.3 -> 187.0)
        kw = {"fused": True} if fused else {}
        if capturable:            # step counters live on the device: the update can be recorded into a HIP graph
            kw["capturable"] = True
        on_gpu = all(p.is_cuda for p in main + aux) and len(main) > 0
        if on_gpu and not fused and multi_tensor:
            # same state layout as Adam(capturable=True); update of all tensors in ~6 launches per optimiser
            self.optimizer = MultiTensorAdam(main, lr=lr)
            self.aux_optimizer = MultiTensorAdam(aux, lr=aux_lr)
        else:
            self.optimizer = torch.optim.Adam(main, lr=lr, **kw)
            self.aux_optimizer = torch.optim.Adam(aux, lr=aux_lr, **kw)
        # EB matrices/biases/factors get their gradient from the main backward but are stepped by the aux
        # optimiser after the aux backward adds the quantile gradient: both groups are reduced,
        # the aux group only after the aux backward.
        self.main_reducer = GradBucketReducer(main, bucket_mb)
        self.aux_params = aux
        self.world = self.main_reducer.world

    def _reduce_aux(self):
        if self.world == 1:
            return
        gs = [p.grad for p in self.aux_params if p.grad is not None]
        if not gs:
            return
        flat = torch.cat([g.reshape(-1).float() for g in gs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(self.world)
        off = 0
        for g in gs:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def step(self, x1, x2, h_matrix, noise=None):
        """One iteration in the reference's order: zero both -> forward -> R-D loss backward -> optimizer.step
        -> aux loss backward -> aux_optimizer.step.  Returns the loss dict (device scalars, no sync)."""
        self.model.train()
        self.optimizer.zero_grad(set_to_none=True)
        self.aux_optimizer.zero_grad(set_to_none=True)
        prev = Fn.train_pack_cache(True)          # packed conv weights persist across the step, one batched repack below
        scaled, Fn.SCALED_LOSS = Fn.SCALED_LOSS, False     # backward() starts at the unscaled loss: no g_loss multiplies
        try:
            out = self.model(x1, x2, h_matrix, noise=noise)
            crit = Fn.rd_loss(out, x1, x2, self.lmbda)
            crit["loss"].backward()
        finally:
            Fn.train_pack_cache(prev)
            Fn.SCALED_LOSS = scaled
        self.main_reducer.finish()
        self.optimizer.step()
        Fn.invalidate_weight_cache()              # fused optimisers do not bump version counters: new epoch for every cache
        if x1.is_cuda:
            Fn.repack_all()
        aux = self.model.aux_loss()
        aux.backward()
        self._reduce_aux()
        self.aux_optimizer.step()
        # detached scalars only: a returned loss that still requires grad would keep this step's autograd graph (and its
        # AccumulateGrad nodes, bound to this step's stream) alive into the next one
        crit = {k: v.detach() for k, v in crit.items()}
        crit["aux_loss"] = aux.detach()
        return crit


class GraphedTrainer(Trainer):
    """``Trainer.step`` recorded once into a HIP graph and replayed (single process; with a process group the eager
    ``Trainer`` and its overlapped all-reduce is the path).

    One training step is ~600 launches, most of them a few microseconds long: eagerly the Python/ctypes launch work
    (~12 ms at B=8, 256x256) exceeds the GPU time (~9.6 ms), so the step is host-bound.  The first ``warmup`` calls run
    eagerly on a side stream (they are real steps: they pack weights, size workspaces and let autograd allocate), the next
    call captures zero_grad -> forward -> R-D backward -> Adam -> aux backward -> aux Adam and every later call is a copy
    of the inputs into the static buffers plus one graph launch.  The quantisation noise is drawn inside the graph by
    the graph-safe Philox generator unless a ``noise`` dict is given at capture time (then it is a static input too).
    The returned dict holds the graph's static loss tensors (overwritten by the next call)."""

    def __init__(self, model, *args, warmup=3, **kw):
        kw["capturable"] = True
        super().__init__(model, *args, **kw)
        if self.world != 1:
            raise RuntimeError("GraphedTrainer is single-process: use Trainer under torch.distributed")
        if int(warmup) < 1:
            raise ValueError("GraphedTrainer: warmup >= 1 (the optimiser state and the packed-weight registry are created by an eager step; "
                             "captured, their zero-initialisation would replay on every step)")
        self.warmup, self.calls, self.graph = int(warmup), 0, None
        self._in = self._noise = self._out = None

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
            with torch.cuda.graph(self.graph):
                self._out = super().step(*self._in, noise=self._noise)
        self.graph.replay()
        # a replay runs no Python: the captured Adam kernels moved the parameters without touching their version counters, so
        # every cache keyed on them (bottleneck tables, packed GDN parameters, inference weight packs) must start a new epoch
        Fn.invalidate_weight_cache()
        return self._out


def init_distributed(backend=None):
    """Process-group bring-up from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, local_rank); a plain single-process run returns (0, 1, 0)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
