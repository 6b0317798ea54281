"""Host-side operators over the C ABI (``include/hesic_hip.h``) as ``torch.autograd.Function``s.

Tensors keep the reference's logical NCHW shape; wide feature maps are stored ``channels_last``
(= NHWC in memory, what the kernels want), 3-channel images keep whatever strides they have.
PyTorch is used for device memory, streams and autograd bookkeeping only -- every number on the
path is produced by a HIP kernel of ``libhesic_hip.so``.
"""
from __future__ import annotations

import ctypes as C
import os as _os

import torch

from . import _lib as L

_CL = torch.channels_last
_compute_dtype = torch.float32


def set_compute_dtype(dtype):
    """Storage type of the wide feature maps: torch.float32 (exact-fp32 MFMA, parity mode), torch.bfloat16 (bf16 MFMA, fp32
    accumulate: training and inference) or torch.float16 (IEEE half operands on the matrix cores at the same rate, 11-bit
    significand: INFERENCE -- the default of ``bench.py``).  A 16-bit choice also selects the library built for that format
    (``_lib.use_h16``); packed-weight caches are dropped when the format changes."""
    global _compute_dtype, _cache_epoch
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError("compute dtype must be torch.float32, torch.bfloat16 or torch.float16")
    if dtype != torch.float32 and dtype != L.h16_dtype():
        L.use_h16(dtype)
        # A pure library switch moves the FORMAT half of the epoch only: packs whose cache key carries the 16-bit dtype (every conv weight:
        # PackedWeight) keep one entry per format and hit again when the format comes back -- Stage2Trainer alternates f16 inference with a
        # bf16 enhancer step and used to repack all frozen HSIC weights twice per step.  Caches keyed without the dtype see a
        # different tag and rebuild, as before.
        _cache_epoch = (str(dtype), _epoch_n)
    _compute_dtype = dtype


def _h16():
    """torch dtype of the active 16-bit storage format."""
    return L.h16_dtype()


def _is16():
    """True when the compute dtype is one of the 16-bit formats."""
    return _compute_dtype != torch.float32


def compute_dtype():
    return _compute_dtype


def _nhwc(x):
    return x if x.is_contiguous(memory_format=_CL) else x.contiguous(memory_format=_CL)


def _empty_nhwc(b, c, h, w, dtype, device):
    return torch.empty((b, c, h, w), dtype=dtype, device=device, memory_format=_CL)


def _is_narrow(c):
    return c <= 8


class _AllFalse:
    def __getitem__(self, i):
        return False


class _NoCtx:
    """Stand-in for the autograd context when grad mode is off: ``Function.apply`` costs ~10 us of host time per call (16+ calls
    per eval forward) to build a graph node nobody will use; ``_apply`` calls ``forward`` directly then."""
    needs_input_grad = _AllFalse()

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def mark_dirty(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


_autograd_depth = 0          # > 0 while an autograd.Function of this module runs its forward on behalf of a grad-enabled call


def _apply(fn, *args):
    global _autograd_depth
    if torch.is_grad_enabled():
        if _compute_dtype == torch.float16 and any(torch.is_tensor(a) and a.requires_grad for a in args):
            raise RuntimeError("hesic_amd: float16 is an inference format here (gradients need the fp32 exponent range): train with "
                               "set_compute_dtype(torch.bfloat16) or torch.float32, or run the forward under torch.no_grad()")
        _autograd_depth += 1
        try:
            return fn.apply(*args)
        finally:
            _autograd_depth -= 1
    return fn.forward(_NoCtx(), *args)


def inference_call():
    """True when the current operator call is a plain inference call: grad mode off, NOT the forward of an autograd.Function reached from a
    grad-enabled call (grad mode is off inside ``Function.forward`` too) and not a backward pass.  The error-feedback ("shaped") weight packs are
    inference-only: a training forward outside ``Trainer.step`` must multiply the plainly rounded weights its backward differentiates."""
    return not torch.is_grad_enabled() and _autograd_depth == 0 and torch._C._current_graph_task_id() < 0


FUSE_GDN3 = True      # module switch (tests / profiling): 3-channel (I)GDN inside the 6 -> 3 cat-conv launch


def _c(t):
    """``t`` for its pointer: contiguous storage without the ~5 us ``detach()`` + ``contiguous()`` make of a new tensor object per
    launch (parameters are contiguous leaves; ``data_ptr()`` needs no detach)."""
    return t if t.is_contiguous() else t.detach().contiguous()


def _nd(t):
    return t.detach() if t.requires_grad else t


def _zeros(shape, dtype, device):
    """Zero-filled scratch for kernels that ACCUMULATE into it.  ``torch.zeros`` is a hipMemsetAsync on ROCm, and a memset
    node recorded into a HIP graph did not reliably run before the accumulating kernel on replays (see csrc/wgrad.hip:
    zero_async); ``fill_`` is an ordinary kernel and keeps its place in the stream in eager and graph mode alike."""
    return torch.empty(shape, dtype=dtype, device=device).fill_(0)


# ------------------------------------------------------------------------------ flat gradient slots
# train.FlatGroup keeps the gradients of all parameters of an optimiser group in ONE flat fp32 buffer (cleared by one
# launch per step, reduced over ranks in place, read by one Adam launch); every ``p.grad`` is a view into it.  The
# backward passes below look the parameter up here (by storage address: saved tensors come back as fresh wrappers) and, if
# it has a slot, ADD their gradient into it directly -- K-slice reduce, layout change and accumulation in the finishing
# kernel -- and hand ``None`` to autograd: no unpack kernel, no AccumulateGrad copy / add per parameter.
class GradSlot:
    __slots__ = ("grad", "writes", "on_write", "name", "param")

    def __init__(self, grad, name=""):
        self.grad, self.writes, self.on_write, self.name, self.param = grad, 0, None, name, None


_grad_slots = {}


def register_grad_slots(slots):
    """``slots``: iterable of (parameter, GradSlot).  Replaces any slot registered for the same storage address; returns the
    registered addresses (for ``clear_grad_slots``)."""
    import weakref
    keys = []
    for p, s in slots:
        s.param = weakref.ref(p)
        _grad_slots[p.data_ptr()] = s
        keys.append(p.data_ptr())
    return keys


def clear_grad_slots(keys=None):
    if keys is None:
        _grad_slots.clear()
    else:
        for k in keys:
            _grad_slots.pop(k, None)


_slots_active = False
_wgrad_stream = None


def set_wgrad_stream(stream):
    """Stream for the weight-gradient launches of the flat-slot path (``_wide_conv_grads``); None = the current stream.  The caller
    joins it back (``current_stream().wait_stream(stream)``) before anything reads the flat gradient buffer."""
    global _wgrad_stream
    prev, _wgrad_stream = _wgrad_stream, stream
    return prev



# Deferred finishing passes of the wide weight gradients (see _wide_conv_grads): a list while train.Trainer.step collects them
_finish_queue = None
GDN_FINISH_BATCH = True      # module switch: False = one parameter finish per GDN backward (rounds 2-4)
WGRAD_FINISH_BATCH = 8     # module switch: 0 = one finishing launch per layer (rounds 2-3)
# round 5: the split-K launches of the queued layers are deferred too and share grids (hesic_conv2d_wgrad_partial_batched); 0 = each layer's
# split-K launch where its backward runs, only the finishing pass batched (A/B)
WGRAD_PARTIAL_BATCH = True


def defer_wgrad_finish(on):
    """Gate of the batched finishing pass; returns the previous state.  Turning it off flushes what is queued."""
    global _finish_queue
    prev = _finish_queue is not None
    if not on:
        flush_wgrad_finish()
        _finish_queue = None
    elif _finish_queue is None and WGRAD_FINISH_BATCH > 0:
        _finish_queue = []
        _gdn_finish_queue.clear()          # jobs a failed step left behind must not be finished into this one
    return prev


# a queued conv weight gradient: descriptor, split-K workspace, dY, gradient slots of weight / bias, the weight slot's address (two jobs on one
# gradient never share a launch), and -- when the split-K launch itself is deferred -- the conv input, the workspace size and the K-slice count
_WgJob = __import__("collections").namedtuple("_WgJob", "desc ws gy wslot bslot dwp x nws nsplit")
# a fused GDN backward whose parameter finish is pending
_GdnJob = __import__("collections").namedtuple("_GdnJob", "ws P beta gamma slot_beta slot_gamma beta_min")
_gdn_finish_queue = []


def flush_gdn_finish():
    """One ``hesic_gdn_param_finish_batched`` call for the queued GDN backwards (round 5: 15 six-microsecond launches per step before)."""
    if not _gdn_finish_queue:
        return
    q = list(_gdn_finish_queue)
    _gdn_finish_queue.clear()
    n = len(q)
    vp, i64, f32 = C.c_void_p * n, C.c_int64 * n, C.c_float * n
    L.call("hesic_gdn_param_finish_batched", n, vp(*[j.ws.data_ptr() for j in q]), i64(*[j.P for j in q]), vp(*[j.beta.data_ptr() for j in q]),
           vp(*[j.gamma.data_ptr() for j in q]), vp(*[j.slot_gamma.grad.data_ptr() for j in q]), vp(*[j.slot_beta.grad.data_ptr() for j in q]),
           f32(*[j.beta_min for j in q]), 1, L.stream())
    for j in q:
        _slot_done(j.slot_beta)
        _slot_done(j.slot_gamma)


def flush_wgrad_finish():
    """One ``hesic_conv2d_wgrad_finish_batched`` call for the queued layers, on the current stream (the stream their split-K launches
    went to); the gradient slots report afterwards, so a bucket's all-reduce is still issued behind its last finishing launch."""
    flush_gdn_finish()
    if not _finish_queue:
        return
    q = list(_finish_queue)
    _finish_queue.clear()          # first: a call that fails below must not leave its jobs for the next step
    n = len(q)
    descs = (L.ConvDesc * n)(*[j.desc for j in q])
    vp = C.c_void_p * n
    ws = vp(*[j.ws.data_ptr() for j in q])
    dy = vp(*[j.gy.data_ptr() for j in q])
    pend = [j for j in q if j.x is not None]
    if pend:          # jobs whose split-K launch was deferred: the shared grids of hesic_conv2d_wgrad_partial_batched
        m = len(pend)
        vm = C.c_void_p * m
        L.call("hesic_conv2d_wgrad_partial_batched", m, (L.ConvDesc * m)(*[j.desc for j in pend]), vm(*[j.x.data_ptr() for j in pend]),
               vm(*[j.gy.data_ptr() for j in pend]), vm(*[j.ws.data_ptr() for j in pend]), (C.c_int64 * m)(*[j.nws for j in pend]),
               (C.c_int32 * m)(*[j.nsplit for j in pend]), L.stream())
    dw = vp(*[j.wslot.grad.data_ptr() for j in q])
    db = vp(*[(j.bslot.grad.data_ptr() if j.bslot is not None else None) for j in q])
    L.call("hesic_conv2d_wgrad_finish_batched_n", n, descs, ws, dy, dw, db, 1, (C.c_int32 * n)(*[j.nsplit for j in q]), L.stream())
    for j in q:
        _slot_done(j.wslot)
        if j.bslot is not None:
            _slot_done(j.bslot)


def grad_slots_active(on):
    """Gate of the direct-write path.  Only inside ``train.Trainer.step`` (which clears the flat buffer first and reads it with
    its own optimiser) do the gradient kernels add into the slots and hand ``None`` to autograd; everywhere else -- a plain
    ``loss.backward()``, ``torch.autograd.grad``, tensor / DDP hooks on the parameters -- gradients are returned to autograd as
    usual (``p.grad`` is still the slot view, so AccumulateGrad lands in the flat buffer all the same)."""
    global _slots_active
    prev, _slots_active = _slots_active, bool(on)
    return prev


def _slot_for(t):
    if t is None or not _grad_slots or not _slots_active:
        return None
    s = _grad_slots.get(t.data_ptr())
    if s is None:
        return None
    owner = s.param() if s.param is not None else None
    if owner is None or owner.data_ptr() != t.data_ptr():       # the parameter is gone (its address may have been reused)
        _grad_slots.pop(t.data_ptr(), None)
        return None
    return s if s.grad.shape == t.shape and s.grad.device == t.device else None


def _slot_done(slot):
    slot.writes += 1
    if slot.on_write is not None:
        slot.on_write(slot)


# ------------------------------------------------------------------------------ packing cache
_epoch_n = 0
_cache_epoch = (str(L.h16_dtype()), 0)       # (active 16-bit format, invalidation count): compared for equality inside the cache tags
SHAPED_WEIGHTS = _os.environ.get("HESIC_SHAPED_WEIGHTS", "1") != "0"      # A/B switch for PackedWeight(shaped=True)


def invalidate_weight_cache():
    """Forget every packed inference weight (call after changing parameters through an API that does not bump
    the tensor version counter, e.g. a fused optimiser step issued under ``torch.no_grad()`` outside training)."""
    global _cache_epoch, _epoch_n
    _epoch_n += 1
    _cache_epoch = (_cache_epoch[0], _epoch_n)


# Training-step pack registry (used by train.Trainer): with ``train_pack_cache(True)`` the packers keep one persistent
# packed buffer per (layer, layout) also in grad mode and register it here; ``repack_all()`` then refreshes every
# registered buffer with ONE launch (hesic_pack_conv_weights_batched) right after the optimiser update, instead of ~70
# separate 7 us launches spread over the next forward/backward.  Outside that mode training re-packs on every call.
_train_pack_cache = False
_pack_registry = []            # entries: dict(weight, mask, wp, dims..., owner, key)
_pack_table = None             # (signature, device job table, n_jobs, total_blocks)
_gdn_registry = []             # PackedGdn objects holding a persistent training pack
_gdn_table = None              # (signature, device job table, n_jobs)


def train_pack_cache(on):
    global _train_pack_cache
    prev, _train_pack_cache = _train_pack_cache, bool(on)
    return prev


def _repack_gdns():
    """The registered GDN parameter packs (PackedGdn, Trainer step) refreshed in one launch."""
    global _gdn_table
    ents = [g for g in _gdn_registry if g._train is not None and g._train["beta"]() is not None and g._train["gamma"]() is not None]
    if len(ents) != len(_gdn_registry):
        _gdn_registry[:] = ents
    if not ents:
        return 0
    import numpy as np
    sig = tuple((g._train["beta"]().data_ptr(), g._train["gamma"]().data_ptr(), g._train["gp"].data_ptr()) for g in ents)
    if _gdn_table is None or _gdn_table[0] != sig:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("repack_all: the set of packed GDN parameters changed during HIP-graph capture; run a warm-up step first")
        jobs = np.zeros(len(ents), dtype=np.dtype([("beta", "<u8"), ("gamma", "<u8"), ("gp", "<u8"), ("bp", "<u8"), ("beta_min", "<f4"), ("pad", "<i4")]))
        for i, g in enumerate(ents):
            t = g._train
            jobs[i] = (t["beta"]().data_ptr(), t["gamma"]().data_ptr(), t["gp"].data_ptr(), t["bp"].data_ptr(), t["beta_min"], 0)
        _gdn_table = (sig, torch.from_numpy(jobs.view(np.uint8).copy()).to(ents[0]._train["gp"].device), len(ents))
    _, table, n = _gdn_table
    L.call("hesic_gdn_pack_params_batched", L.ptr(table), n, L.stream())
    for g in ents:
        t = g._train
        t["tag"] = (t["beta"]().data_ptr(), t["gamma"]().data_ptr(), t["beta"]()._version, t["gamma"]()._version, _cache_epoch)
    return n


def repack_all():
    """Refresh every registered packed weight (and GDN parameter pack) from its parameter in one launch each and mark it current."""
    global _pack_table
    _repack_gdns()
    ents = [e for e in _pack_registry if e["weight"]() is not None]
    if len(ents) != len(_pack_registry):
        _pack_registry[:] = ents
    if not ents:
        return 0
    import numpy as np
    sig = tuple((e["weight"]().data_ptr(), 0 if e["mask"] is None else e["mask"].data_ptr(), e["wp"].data_ptr()) for e in ents)
    if _pack_table is None or _pack_table[0] != sig:
        if torch.cuda.is_current_stream_capturing():
            # the job table is uploaded from host memory: never from inside a graph capture (warm-up steps build it)
            raise RuntimeError("repack_all: the set of packed weights changed during HIP-graph capture; run a warm-up step first")
        jobs = np.zeros(len(ents), dtype=np.dtype([("w", "<u8"), ("mask", "<u8"), ("wp", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("KH", "<i4"),
                                                      ("KW", "<i4"), ("transposed", "<i4"), ("flip", "<i4"), ("dtype", "<i4"), ("block0", "<i4")]))
        blk = 0
        for i, e in enumerate(ents):
            w = e["weight"]()
            jobs[i] = (w.data_ptr(), 0 if e["mask"] is None else e["mask"].data_ptr(), e["wp"].data_ptr(), e["cout"], e["cin"], e["kh"], e["kw"],
                       int(e["transposed"]), int(e["flip"]), L.dt(e["dtype"]), blk)
            blk += ((e["cout"] + 7) // 8) * ((e["cin"] + 31) // 32)          # one block per 8-cout x 32-cin tile
        dev = ents[0]["wp"].device
        table = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
        _pack_table = (sig, table, len(ents), blk)
    _, table, n, blk = _pack_table
    L.call("hesic_pack_conv_weights_batched", L.ptr(table), n, blk, L.stream())
    for e in ents:
        w = e["weight"]()
        e["owner"]._cache[e["key"]] = ((w.data_ptr(), w._version, None if e["mask"] is None else e["mask"]._version, _cache_epoch), e["wp"])
    return n


class PackedWeight:
    """Device copy of a conv weight in the kernels' [tap][Cout][Cin] layout.

    Inference (grad mode off): cached, refreshed when the parameter's version counter or storage moves.
    Training (grad mode on): re-packed on every call and the inference cache is dropped -- fused optimisers
    update parameters without touching the version counter, and a 35 M-parameter repack is ~50 us.  Under
    ``train_pack_cache(True)`` (the Trainer's step) the packed buffer is persistent, registered for ``repack_all()`` and
    reused while its (storage, version, epoch) tag is current."""

    def __init__(self, shaped=False, tr_stride=0):
        self._cache = {}
        # ``shaped``: at 16-bit INFERENCE a Conv2d weight is rounded with error feedback over the taps of each (cout, cin) pair
        # (``hesic_pack_conv_weight_shaped``) -- for layers whose input is a spatially smooth feature map (g_a_conv2..4); a ConvTranspose2d
        # weight (``tr_stride`` = its stride; round 5: the synthesis stacks) inside each output phase's tap class
        self.shaped, self.tr_stride = shaped, int(tr_stride)

    def get(self, weight, mask, cout, cin, kh, kw, transposed, flip, dtype):
        # autograd.Function bodies run with grad mode off, so a training step also takes the cached branch (one repack
        # per layout and step, when the version tag moves)
        caching = not torch.is_grad_enabled()
        # an inference pack with error-feedback rounding and the plain pack of a training forward / backward are different images of one weight
        key = (transposed, flip, dtype, cout, cin, bool(self.shaped and SHAPED_WEIGHTS and inference_call() and not _train_pack_cache))
        tag = (weight.data_ptr(), weight._version, None if mask is None else mask._version, _cache_epoch)
        hit = None
        if caching:
            hit = self._cache.get(key)
            if hit is not None and hit[0] == tag:
                return hit[1]
        elif self._cache:
            self._cache.clear()
        wref = None
        if caching and _train_pack_cache:
            # the parameter behind this tensor: the tensor itself in a forward, the forward's registration in a backward
            # (saved tensors come back as fresh wrappers around the parameter's storage)
            if isinstance(weight, torch.nn.Parameter):
                wref = __import__("weakref").ref(weight)
            else:
                for e in _pack_registry:
                    w0 = e["weight"]()
                    if e["owner"] is self and w0 is not None and w0.data_ptr() == weight.data_ptr() and w0.shape == weight.shape:
                        wref = e["weight"]
                        break
        if wref is not None:
            # Trainer step: persistent buffer, registered for the batched repack that follows the optimiser update
            reuse = hit is not None and hit[1].device == weight.device and hit[1].dtype == dtype
            wp = hit[1] if reuse else torch.empty(kh * kw * cout * cin, dtype=dtype, device=weight.device)
            L.call("hesic_pack_conv_weight", L.ptr(weight.detach()), L.ptr(mask), L.ptr(wp), cout, cin, kh, kw,
                   int(transposed), int(flip), L.dt(dtype), L.stream())
            for e in _pack_registry:
                if e["owner"] is self and e["key"] == key:
                    e["wp"], e["weight"], e["mask"] = wp, wref, mask
                    break
            else:
                _pack_registry.append({"weight": wref, "mask": mask, "wp": wp, "cout": cout, "cin": cin, "kh": kh, "kw": kw,
                                       "transposed": transposed, "flip": flip, "dtype": dtype, "owner": self, "key": key})
            self._cache[key] = (tag, wp)
            return wp
        wp = torch.empty(kh * kw * cout * cin, dtype=dtype, device=weight.device)
        if (self.shaped and SHAPED_WEIGHTS and caching and inference_call() and not _train_pack_cache and dtype != torch.float32 and mask is None and not transposed
                and not flip and kh * kw > 1 and weight.dtype == torch.float32 and weight.is_contiguous()):
            L.call("hesic_pack_conv_weight_shaped", L.ptr(weight.detach()), L.ptr(wp), cout, cin, kh, kw, L.stream())
        elif (self.shaped and self.tr_stride and SHAPED_WEIGHTS and caching and inference_call() and not _train_pack_cache and dtype != torch.float32 and mask is None
                and transposed and not flip and kh * kw > 1 and weight.dtype == torch.float32 and weight.is_contiguous()):
            L.call("hesic_pack_conv_weight_shaped_tr", L.ptr(weight.detach()), L.ptr(wp), cout, cin, kh, kw, self.tr_stride, L.stream())
        else:
            L.call("hesic_pack_conv_weight", L.ptr(weight.detach()), L.ptr(mask), L.ptr(wp), cout, cin, kh, kw,
                   int(transposed), int(flip), L.dt(dtype), L.stream())
        if caching:
            self._cache[key] = (tag, wp)
        return wp


_img_cache = {}


def _weight_image(kind, weight, gp=None, gtag=None):
    """LDS weight image of an image-side MFMA kernel (``hesic_sconv_pack_weight_image``), cached per weight (+ GDN parameters)
    version: kind 0 = g_a_conv1 + GDN (64 KB, needs the packed gamma'), kind 1 = g_s_conv4 (24 KB).  The entry remembers the
    tensor OBJECT (weak reference): an address / version pair alone would also match a new tensor allocated where a dead one was."""
    import weakref
    key = (kind, weight.data_ptr())
    tag = (weight._version, gtag, _cache_epoch)
    hit = _img_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is weight:
        return hit[1]
    if kind == 1 and SHAPED_WEIGHTS and inference_call() and not _train_pack_cache:
        kind = 2          # inference: g_s_conv4's panel rounded with error feedback per output phase (csrc/sconv.hip, round 5)
        key = (kind, weight.data_ptr())
        hit = _img_cache.get(key)
        if hit is not None and hit[0] == tag and hit[2]() is weight:
            return hit[1]
    img = torch.empty(65536 if kind == 0 else 24576, dtype=torch.uint8, device=weight.device)
    L.call("hesic_sconv_pack_weight_image", kind, L.ptr(_c(weight)), L.ptr(gp), L.ptr(img), L.stream())
    if len(_img_cache) > 64:
        _img_cache.clear()
    try:
        _img_cache[key] = (tag, img, weakref.ref(weight))
    except TypeError:
        pass
    return img


def _wide_conv(x, wp, bias, B, H, W, Cin, Ho, Wo, Cout, k, stride, pad, transposed, act=0, in_abs=0, tap_mask=0,
               out=None, out_c_off=0, x_c_off=0, f32_out=None):
    """Launch the implicit-GEMM kernel. ``x`` NHWC (may be wider than Cin), returns / fills NHWC ``out``.
    ``f32_out`` ("only" | "both", bf16 storage): also / only store the output as fp32 from the fp32 accumulators
    (``hesic_conv2d_forward_f32out``); returns ``(out | None, out_f32)`` then."""
    dtype = x.dtype
    if out is None and f32_out != "only":
        out = _empty_nhwc(B, Cout, Ho, Wo, dtype, x.device)
    y_ps, y_co = (out.shape[1], out_c_off) if out is not None else (Cout, 0)
    d = L.ConvDesc(B, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, int(transposed), L.dt(dtype), act, in_abs,
                   x.shape[1], x_c_off, y_ps, y_co, tap_mask)
    key = (B, H, W, Cin, Ho, Wo, Cout, k, stride, pad, int(transposed), L.dt(dtype), tap_mask, bool(f32_out))
    need = _ws_bytes.get(key)
    if need is None:      # a launch with an fp32 latent output decides its K split per image: its own query
        need = _ws_bytes[key] = int((L.lib().hesic_conv2d_f32out_ws_bytes if f32_out else L.lib().hesic_conv2d_ws_bytes)(C.byref(d)))
    ws = None
    if need and SPLIT_K:      # low-resolution layer: split-K launch, fp32 partial tiles in a scratch buffer
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    if f32_out:
        out32 = _empty_nhwc(B, Cout, Ho, Wo, torch.float32, x.device)
        L.call("hesic_conv2d_forward_f32out", C.byref(d), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(out), L.ptr(out32), Cout, 0,
               L.ptr(ws), need if ws is not None else 0, L.stream())
        return out, out32
    if ws is not None:
        L.call("hesic_conv2d_forward_ws", C.byref(d), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(out), L.ptr(ws), need, L.stream())
    else:
        L.call("hesic_conv2d_forward", C.byref(d), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(out), L.stream())
    return out


_ws_bytes = {}
SPLIT_K = True


class no_split_k:
    """Context manager: run the convs inside with one block per output tile (no split-K).  A split-K launch sums its K
    slices in a different order than the plain launch, so results depend (in the last bit) on which launch a SHAPE gets;
    the bit-stream codec, whose decoder re-evaluates the encoder's full-map convolutions one pixel at a time, needs
    shape-independent numbers."""

    def __enter__(self):
        global SPLIT_K
        self._prev, SPLIT_K = SPLIT_K, False

    def __exit__(self, *exc):
        global SPLIT_K
        SPLIT_K = self._prev


def _sdesc(x, y, Cin, Cout, k, stride, pad, transposed, act=0):
    B, _, H, W = x.shape
    _, _, Ho, Wo = y.shape
    xs, ys = x.stride(), y.stride()
    return L.SConvDesc(B, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, int(transposed), L.dt(x), L.dt(y), act, 0,
                       xs[0], xs[1], xs[2], xs[3], ys[0], ys[1], ys[2], ys[3])


def _out_hw(H, W, k, stride, pad, transposed):
    if transposed:
        f = lambda n: (n - 1) * stride - 2 * pad + k + stride - 1
    else:
        f = lambda n: (n + 2 * pad - k) // stride + 1
    return f(H), f(W)


def _narrow_conv_grads(x, weight, gy, cfg, dims, has_bias, need_dx, need_dw, bias=None):
    """dx / dw / dbias of an image-side conv (few channels on one side: strided kernels)."""
    k, stride, pad, transposed, act, in_abs, tap_mask, packer, mask = cfg
    B, H, W, Cin, Ho, Wo, Cout = dims
    dx = dw = db = None
    ydt = torch.float32 if _is_narrow(Cout) else (_compute_dtype if _is_narrow(Cin) else x.dtype)
    gy = gy.to(ydt)
    gy = gy.contiguous() if _is_narrow(Cout) else _nhwc(gy)
    w = (weight.detach() if mask is None else weight.detach() * mask).contiguous()
    d = _sdesc(x, gy, Cin, Cout, k, stride, pad, transposed)
    if need_dx:
        dx = torch.empty_like(x)
        L.call("hesic_sconv2d_dgrad", C.byref(d), L.ptr(gy), L.ptr(w), L.ptr(dx), L.stream())
        if in_abs:
            dx = dx * torch.sign(x)
    if need_dw:
        # flat gradient slots: the kernel overwrites, so it may write the slot itself only while the (cleared) slot is untouched
        ws_, bs_ = (_slot_for(weight) if mask is None else None), _slot_for(bias) if has_bias else None
        w_direct = ws_ is not None and ws_.writes == 0 and weight.dtype == torch.float32
        b_direct = bs_ is not None and bs_.writes == 0
        dw = ws_.grad if w_direct else torch.empty_like(weight, dtype=torch.float32)
        db = (bs_.grad if b_direct else torch.empty(Cout, dtype=torch.float32, device=x.device)) if has_bias else None
        nws = L.lib().hesic_sconv2d_wgrad_ws_bytes(C.byref(d))
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
        L.call("hesic_sconv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(db), L.ptr(ws), nws, L.stream())
        if mask is not None:
            dw = dw * mask
        if ws_ is not None:
            if not w_direct:
                ws_.grad.add_(dw)
            _slot_done(ws_)
            dw = None
        if bs_ is not None:
            if not b_direct:
                bs_.grad.add_(db)
            _slot_done(bs_)
            db = None
    return dx, dw, db


def _wide_conv_grads(x, weight, gy, cfg, dims, has_bias, need_dx, need_dw, bias=None):
    """dx / dw / dbias of a wide conv (implicit-GEMM kernels): shared by _ConvFn and _ConvGdnFn.  The weight gradient is one
    split-K MFMA launch + one finishing launch that reduces the K slices, writes the PyTorch layout and sums dY's columns
    for the bias (``hesic_conv2d_wgrad_direct``)."""
    k, stride, pad, transposed, act, in_abs, tap_mask, packer, mask = cfg
    B, H, W, Cin, Ho, Wo, Cout = dims
    dx = dw = db = None
    gy = _nhwc(gy.to(x.dtype))
    if need_dx:
        # data gradient = the opposite op with the same weight tensor read in the other layout
        if Cout % 32 == 0 and Cin % 8 == 0:
            wp = packer.get(weight, mask, Cin, Cout, k, k, not transposed, False, x.dtype)
            dx = _wide_conv(gy, wp, None, B, Ho, Wo, Cout, H, W, Cin, k, stride, pad, not transposed)
        else:   # odd channel counts (parity tests only): strided VALU kernel
            w = (weight.detach() if mask is None else weight.detach() * mask).contiguous()
            dx = torch.empty_like(x, memory_format=_CL)
            d = _sdesc(x, gy, Cin, Cout, k, stride, pad, transposed)
            L.call("hesic_sconv2d_dgrad", C.byref(d), L.ptr(gy), L.ptr(w), L.ptr(dx), L.stream())
        if in_abs:
            dx = dx * torch.sign(x)
    if need_dw:
        db = torch.empty(Cout, dtype=torch.float32, device=x.device) if has_bias else None
        d = L.ConvDesc(B, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, int(transposed), L.dt(x), 0, in_abs,
                       x.shape[1], 0, gy.shape[1], 0, tap_mask)
        nws = L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(d))
        ws_, bs_ = _slot_for(weight), (_slot_for(bias) if has_bias else None)
        if ws_ is not None and weight.dtype == torch.float32 and (not has_bias or bs_ is not None) and (mask is None or tap_mask):
            # flat gradient buffer: the finishing kernel adds dW (PyTorch layout) and dbias into the slots.  Nothing downstream in
            # the backward pass reads a weight gradient, so with a weight-gradient stream set (train.Trainer) the two launches go
            # there: the chain of data gradients -- the critical path of the step -- does not wait for them, and they fill the gaps
            # between its launches.  ONE such stream: the two gradients of a twice-used weight (encoder1) add into their slot in order.
            side = _wgrad_stream if x.is_cuda else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                x.record_stream(side)
                gy.record_stream(side)
                with torch.cuda.stream(side):
                    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x.device)
                    L.call("hesic_conv2d_wgrad_direct", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(ws_.grad), L.ptr(bs_.grad if has_bias else None), 1,
                           L.ptr(ws), nws, L.stream())
                    _slot_done(ws_)
                    if has_bias:
                        _slot_done(bs_)
                return dx, None, None
            nsp = 0
            if _finish_queue is not None and WGRAD_PARTIAL_BATCH:
                # the shared-grid route: its own (smaller) K-slice count, named to both batched calls; the workspace is sized for it
                nsp = L.lib().hesic_conv2d_wgrad_nsplit(C.byref(d), 1)
                nws = L.lib().hesic_conv2d_wgrad_ws_bytes_n(C.byref(d), nsp)
            ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x.device)
            if _finish_queue is not None:
                # deferred finishing pass (train.Trainer.step): only the split-K MFMA launch now; the K-slice reduce + layout change +
                # bias column sums of up to 8 layers share one launch (flush_wgrad_finish).  The job keeps ws and gy alive until then.
                dwp = ws_.grad.data_ptr()
                if len(_finish_queue) >= WGRAD_FINISH_BATCH or any(j.dwp == dwp for j in _finish_queue) or (_finish_queue and _finish_queue[0].desc.dtype != d.dtype):
                    flush_wgrad_finish()
                if WGRAD_PARTIAL_BATCH:
                    _finish_queue.append(_WgJob(d, ws, gy, ws_, bs_ if has_bias else None, dwp, x, nws, nsp))      # x stays alive until the flush
                else:
                    L.call("hesic_conv2d_wgrad_partial", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(ws), nws, L.stream())
                    _finish_queue.append(_WgJob(d, ws, gy, ws_, bs_ if has_bias else None, dwp, None, nws, 0))
                return dx, None, None
            L.call("hesic_conv2d_wgrad_direct", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(ws_.grad), L.ptr(bs_.grad if has_bias else None), 1,
                   L.ptr(ws), nws, L.stream())
            _slot_done(ws_)
            if has_bias:
                _slot_done(bs_)
            return dx, None, None
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x.device)
        dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)      # dead taps of a masked conv are zero-filled by the call
        L.call("hesic_conv2d_wgrad_direct", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(db), 0, L.ptr(ws), nws, L.stream())
        if mask is not None and not tap_mask:
            dw = dw * mask
    return dx, dw, db


class _ConvFn(torch.autograd.Function):
    """conv()/deconv() of compressai/models/utils.py:104-118 (+ MaskedConv2d, layers.py:21-45)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        L.require_cuda(x, weight)
        k, stride, pad, transposed, act, in_abs, tap_mask, packer, mask = cfg
        if transposed:
            Cin, Cout = weight.shape[0], weight.shape[1]
        else:
            Cout, Cin = weight.shape[0], weight.shape[1]
        B, Cx, H, W = x.shape
        if Cx != Cin:
            raise RuntimeError(f"conv: expected {Cin} input channels, got {Cx}")
        Ho, Wo = _out_hw(H, W, k, stride, pad, transposed)
        narrow = _is_narrow(Cin) or _is_narrow(Cout) or Cin % 32 or Cout % 8 or (transposed and (Ho != H * stride))
        ctx.cfg, ctx.narrow, ctx.dims = cfg, narrow, (B, H, W, Cin, Ho, Wo, Cout)
        if narrow:
            ydt = torch.float32 if _is_narrow(Cout) else (_compute_dtype if _is_narrow(Cin) else x.dtype)
            if not _is_narrow(Cin):
                x = _nhwc(x)
            y = _empty_nhwc(B, Cout, Ho, Wo, ydt, x.device) if not _is_narrow(Cout) else \
                torch.empty((B, Cout, Ho, Wo), dtype=ydt, device=x.device)
            w = weight.detach() if mask is None else (weight.detach() * mask)
            d = _sdesc(x, y, Cin, Cout, k, stride, pad, transposed, act)
            if transposed and Cin == 128 and Cout == 3 and k == 5 and stride == 2 and mask is None and x.dtype == _h16() and weight.dtype == torch.float32:
                # g_s_conv4: the kernel's LDS weight panel is pre-packed once per weight update
                L.call("hesic_sconv2d_forward_prepacked", C.byref(d), L.ptr(x), L.ptr(w.contiguous()), L.ptr(_weight_image(1, weight)), L.ptr(bias),
                       L.ptr(y), L.stream())
            else:
                L.call("hesic_sconv2d_forward", C.byref(d), L.ptr(x), L.ptr(w.contiguous()), L.ptr(bias), L.ptr(y), L.stream())
        else:
            x = _nhwc(x)
            wp = packer.get(weight, mask, Cout, Cin, k, k, transposed, False, x.dtype)
            y = _wide_conv(x, wp, bias, B, H, W, Cin, Ho, Wo, Cout, k, stride, pad, transposed, act, in_abs, tap_mask)
        ctx.save_for_backward(x, weight, y if act else None)
        ctx.has_bias, ctx.bias = bias is not None, bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y_act = ctx.saved_tensors
        k, stride, pad, transposed, act, in_abs, tap_mask, packer, mask = ctx.cfg
        B, H, W, Cin, Ho, Wo, Cout = ctx.dims
        if act:
            g2 = torch.empty_like(y_act)
            gyc = gy.to(y_act.dtype)
            gyc = _nhwc(gyc) if y_act.is_contiguous(memory_format=_CL) and y_act.dim() == 4 and not _is_narrow(Cout) else gyc.contiguous()
            L.call("hesic_act_backward", L.ptr(y_act), L.ptr(gyc), L.ptr(g2), y_act.numel(), act, L.dt(y_act), L.stream())
            gy = g2
        dx = dw = db = None
        if ctx.narrow:
            dx, dw, db = _narrow_conv_grads(x, weight, gy, ctx.cfg, ctx.dims, ctx.has_bias, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.bias)
        else:
            dx, dw, db = _wide_conv_grads(x, weight, gy, ctx.cfg, ctx.dims, ctx.has_bias, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.bias)
        if not ctx.has_bias:
            db = None
        return dx, dw, db, None


import os as _os
FUSE_CONV_GDN = True      # module switch for profiling


class PackedGdn:
    """gamma' (bf16, LDS-image + MFMA-fragment copies) / beta' (fp32) for the fused conv+GDN epilogue; inference-only cache."""

    def __init__(self):
        self._hit = None
        self._train = None

    def get(self, beta, gamma, beta_min):
        tag = (beta.data_ptr(), gamma.data_ptr(), beta._version, gamma._version, _cache_epoch)
        if self._hit is not None and self._hit[0] == tag and not torch.is_grad_enabled():
            return self._hit[1], self._hit[2]
        # (grad mode is OFF inside an autograd.Function.forward -- where the fused conv + GDN stages call this: rounds 3-4 therefore re-packed 12
        # of the 15 GDNs one by one every step on top of the batched launch; the Trainer's step flag alone decides)
        if (_train_pack_cache and isinstance(beta, torch.nn.Parameter) and isinstance(gamma, torch.nn.Parameter)
                and gamma.dtype == torch.float32 and gamma.is_contiguous() and beta.is_contiguous()):
            # Trainer step: a persistent pack, refreshed with every other GDN's by ONE launch behind the optimiser update (repack_all) --
            # 15 pack launches a step otherwise
            t = self._train
            if t is not None and t["tag"] == tag and t["gp"].dtype == _h16() and t["gp"].device == gamma.device:
                return t["gp"], t["bp"]
            import weakref
            gp = torch.empty(2 * 128 * 128, dtype=_h16(), device=gamma.device)
            bp = torch.empty(128, dtype=torch.float32, device=gamma.device)
            L.call("hesic_gdn_pack_params", L.ptr(beta), L.ptr(gamma), float(beta_min), L.ptr(gp), L.ptr(bp), 128, L.stream())
            self._train = {"tag": tag, "gp": gp, "bp": bp, "beta": weakref.ref(beta), "gamma": weakref.ref(gamma), "beta_min": float(beta_min)}
            if self not in _gdn_registry:
                _gdn_registry.append(self)
            return gp, bp
        gp = torch.empty(2 * 128 * 128, dtype=_h16(), device=gamma.device)
        bp = torch.empty(128, dtype=torch.float32, device=gamma.device)
        L.call("hesic_gdn_pack_params", L.ptr(_c(beta)), L.ptr(_c(gamma)), float(beta_min), L.ptr(gp),
               L.ptr(bp), 128, L.stream())
        self._hit = (tag, gp, bp)
        return gp, bp


FUSE_CONV_GDN_TRAIN = True      # module switch for profiling


def conv2d_gdn_fusable(x, weight, gdn_channels, transposed):
    """The fused epilogue exists for bf16 storage, 128 output channels and a wide input; with autograd on, the wide
    stages keep the fusion through ``_ConvGdnFn`` (the kernel then also stores the conv output for GDN's backward), the
    same goes for the image-side 3 -> 128 stage (``_SConvGdnFn``)."""
    cout = weight.shape[1] if transposed else weight.shape[0]
    cin = weight.shape[0] if transposed else weight.shape[1]
    if not x.is_cuda or cout != 128 or gdn_channels != 128 or not FUSE_CONV_GDN:
        return False
    if cin == 3:      # g_a_conv1 + g_a_gdn1: image in (any float dtype), bf16 storage out
        if torch.is_grad_enabled() and not FUSE_CONV_GDN_TRAIN:
            return False
        return (not transposed and _is16() and weight.shape[-1] == 5
                and x.dtype in (torch.float32, _h16()))
    if torch.is_grad_enabled() and not FUSE_CONV_GDN_TRAIN:
        return False
    return x.dtype == _h16() and cin % 32 == 0


def _gdn_backward(v, gy, beta, gamma, inverse, beta_min):
    """GDN.backward on (input v, dy): returns (dv, dbeta, dgamma); with flat gradient slots the parameter gradients are
    added in place and come back as None."""
    B, Cc, H, W = v.shape
    P = B * H * W
    gv = torch.empty_like(v, memory_format=_CL)
    sb, sg = _slot_for(beta), _slot_for(gamma)
    direct = sb is not None and sg is not None
    dbeta = sb.grad if direct else torch.empty_like(beta, dtype=torch.float32)
    dgamma = sg.grad if direct else torch.empty_like(gamma, dtype=torch.float32)
    ws = torch.empty(max(1, L.lib().hesic_gdn_backward_ws_bytes(P, Cc)), dtype=torch.uint8, device=v.device)
    if (direct and _finish_queue is not None and GDN_FINISH_BATCH and beta.is_contiguous() and gamma.is_contiguous()
            and L.lib().hesic_gdn_backward_partial_ok(P, Cc, L.dt(v))):
        # Trainer step: dx now, the parameter-gradient finish with the other GDNs' in one launch (flush_gdn_finish); a module used twice in a
        # step (encoder1) must not have two jobs in one launch
        if any(j.slot_gamma is sg for j in _gdn_finish_queue) or len(_gdn_finish_queue) >= 16:
            flush_gdn_finish()
        L.call("hesic_gdn_backward_partial", L.ptr(v), L.ptr(gy), L.ptr(beta), L.ptr(gamma), L.ptr(gv), L.ptr(ws), P, Cc, int(inverse),
               float(beta_min), L.dt(v), L.stream())
        _gdn_finish_queue.append(_GdnJob(ws, P, beta, gamma, sb, sg, float(beta_min)))
        return gv, None, None
    L.call("hesic_gdn_backward_acc", L.ptr(v), L.ptr(gy), L.ptr(_c(beta)), L.ptr(_c(gamma)), L.ptr(gv), L.ptr(dbeta),
           L.ptr(dgamma), int(direct), L.ptr(ws), P, Cc, int(inverse), float(beta_min), L.dt(v), L.stream())
    if direct:
        _slot_done(sb)
        _slot_done(sg)
        return gv, None, None
    return gv, dbeta, dgamma


def _n2w_image(weight, beta, gamma, gp, x):
    """LDS image of g_a_conv1's weights + the K-permuted gamma' for the fused 3 -> 128 kernel's fast form (fp32 planar image,
    5x5 stride 2); None -> the kernel gathers in its prologue (other layouts)."""
    if tuple(weight.shape) != (128, 3, 5, 5) or weight.dtype != torch.float32 or x.dtype != torch.float32:
        return None
    return _weight_image(0, weight, gp, (gamma.data_ptr(), gamma._version, beta._version, gp.data_ptr()))


class _SConvGdnFn(torch.autograd.Function):
    """g_a_gdn1(g_a_conv1(image)) fused under autograd (the 3 -> 128 stage): v = conv output is stored next to y."""

    @staticmethod
    def forward(ctx, x, weight, bias, beta, gamma, cfg):
        k, stride, pad, inverse, beta_min, gdn_packer = cfg
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        Ho, Wo = _out_hw(H, W, k, stride, pad, False)
        gp, bp = gdn_packer.get(beta, gamma, beta_min)
        y = _empty_nhwc(B, Cout, Ho, Wo, _h16(), x.device)
        v = _empty_nhwc(B, Cout, Ho, Wo, _h16(), x.device)
        d = _sdesc(x, y, Cin, Cout, k, stride, pad, False)
        L.call("hesic_sconv2d_gdn_forward_prepacked", C.byref(d), L.ptr(x), L.ptr(_c(weight)),
               L.ptr(_n2w_image(weight, beta, gamma, gp, x)), L.ptr(bias), L.ptr(gp), L.ptr(bp), int(inverse), L.ptr(y), L.ptr(v), L.stream())
        ctx.save_for_backward(x, weight, v, beta, gamma)
        ctx.cfg, ctx.dims, ctx.has_bias, ctx.bias = cfg, (B, H, W, Cin, Ho, Wo, Cout), bias is not None, bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, v, beta, gamma = ctx.saved_tensors
        k, stride, pad, inverse, beta_min, _ = ctx.cfg
        gv, dbeta, dgamma = _gdn_backward(v, _nhwc(gy.to(v.dtype)), beta, gamma, inverse, beta_min)
        ccfg = (k, stride, pad, False, 0, 0, 0, None, None)
        dx, dw, db = _narrow_conv_grads(x, weight, gv, ccfg, ctx.dims, ctx.has_bias, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.bias)
        return dx, dw, (db if ctx.has_bias else None), dbeta, dgamma, None


class _ConvGdnFn(torch.autograd.Function):
    """(I)GDN(conv(x)) with the fused forward kernel kept under autograd: the kernel stores the conv output v next to y,
    the backward is GDN's backward on (v, gy) followed by the conv's data / weight gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, beta, gamma, cfg):
        k, stride, pad, transposed, inverse, beta_min, packer, gdn_packer = cfg
        Cin, Cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
        B, _, H, W = x.shape
        Ho, Wo = _out_hw(H, W, k, stride, pad, transposed)
        x = _nhwc(x)
        wp = packer.get(weight, None, Cout, Cin, k, k, transposed, False, x.dtype)
        gp, bp = gdn_packer.get(beta, gamma, beta_min)
        y = _empty_nhwc(B, Cout, Ho, Wo, x.dtype, x.device)
        v = _empty_nhwc(B, Cout, Ho, Wo, x.dtype, x.device)
        d = L.ConvDesc(B, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, int(transposed), L.dt(x), 0, 0, Cin, 0, Cout, 0, 0)
        L.call("hesic_conv2d_gdn_forward_train", C.byref(d), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(gp), L.ptr(bp), int(inverse),
               L.ptr(y), L.ptr(v), L.stream())
        ctx.save_for_backward(x, weight, v, beta, gamma)
        ctx.cfg, ctx.dims, ctx.has_bias, ctx.bias = cfg, (B, H, W, Cin, Ho, Wo, Cout), bias is not None, bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, v, beta, gamma = ctx.saved_tensors
        k, stride, pad, transposed, inverse, beta_min, packer, _ = ctx.cfg
        gv, dbeta, dgamma = _gdn_backward(v, _nhwc(gy.to(v.dtype)), beta, gamma, inverse, beta_min)
        ccfg = (k, stride, pad, transposed, 0, 0, 0, packer, None)
        dx, dw, db = _wide_conv_grads(x, weight, gv, ccfg, ctx.dims, ctx.has_bias, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.bias)
        return dx, dw, (db if ctx.has_bias else None), dbeta, dgamma, None


def conv2d_gdn(x, weight, bias, beta, gamma, *, kernel_size, stride, padding, transposed, inverse, beta_min, packer, gdn_packer):
    """(I)GDN(conv(x)) in ONE kernel (inference only; training keeps the two autograd ops)."""
    L.require_cuda(x, weight)
    k = kernel_size
    Cin, Cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    B, _, H, W = x.shape
    Ho, Wo = _out_hw(H, W, k, stride, padding, transposed)
    if Cin == 3 and torch.is_grad_enabled():
        return _apply(_SConvGdnFn, x, weight, bias, beta, gamma, (k, stride, padding, inverse, beta_min, gdn_packer))
    if Cin == 3:      # image-side stage: strided fp32/bf16 image in, bf16 NHWC out
        gp, bp = gdn_packer.get(beta, gamma, beta_min)
        out = _empty_nhwc(B, Cout, Ho, Wo, _h16(), x.device)
        d = _sdesc(x, out, Cin, Cout, k, stride, padding, False)
        L.call("hesic_sconv2d_gdn_forward_prepacked", C.byref(d), L.ptr(x), L.ptr(_c(weight)),
               L.ptr(_n2w_image(weight, beta, gamma, gp, x)), L.ptr(bias), L.ptr(gp), L.ptr(bp), int(inverse), L.ptr(out), None, L.stream())
        return out
    if torch.is_grad_enabled():
        return _apply(_ConvGdnFn, x, weight, bias, beta, gamma, (k, stride, padding, transposed, inverse, beta_min, packer, gdn_packer))
    x = _nhwc(x)
    wp = packer.get(weight, None, Cout, Cin, k, k, transposed, False, x.dtype)
    gp, bp = gdn_packer.get(beta, gamma, beta_min)
    out = _empty_nhwc(B, Cout, Ho, Wo, x.dtype, x.device)
    d = L.ConvDesc(B, H, W, Cin, Ho, Wo, Cout, k, k, stride, padding, int(transposed), L.dt(x), 0, 0, Cin, 0, Cout, 0, 0)
    L.call("hesic_conv2d_gdn_forward", C.byref(d), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(gp), L.ptr(bp), int(inverse),
           L.ptr(out), L.stream())
    return out


def conv2d(x, weight, bias, *, kernel_size, stride, padding, transposed=False, act=L.ACT_NONE, in_abs=False,
           tap_mask=0, packer=None, mask=None):
    packer = packer if packer is not None else PackedWeight()
    return _apply(_ConvFn, x, weight, bias, (kernel_size, stride, padding, transposed, act, int(in_abs), tap_mask, packer, mask))


FP32_LATENTS = _os.environ.get("HESIC_BF16_LATENTS") is None      # A/B switch: set to store latents / sigma / mu as bf16 again


def fp32_latents():
    """True when the inference forward keeps what feeds round() and the likelihoods in fp32 although the feature maps are
    bf16: the latents y, z and the sigma / mu maps are written from the convs' fp32 accumulators (round 1 stored them as
    bf16: one ulp = 1/16 at |y| ~ 10, so latents near .5 flipped and bpp / PSNR sat just outside the 1e-3 target)."""
    return FP32_LATENTS and _is16() and not torch.is_grad_enabled()


def conv2d_latent(x, weight, bias, *, kernel_size, stride, padding, transposed=False, act=L.ACT_NONE, in_abs=False,
                  tap_mask=0, packer=None, mask=None, want_lo=True):
    """A conv whose output feeds an entropy model: returns ``(lo, hi)`` -- ``lo`` in the storage dtype for the next conv
    (None unless ``want_lo``), ``hi`` for round() / the likelihood.  bf16 inference: ``hi`` is fp32 straight from the
    accumulators; otherwise both are the ordinary output."""
    k = kernel_size
    Cin, Cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    ok = (fp32_latents() and x.is_cuda and x.dtype == _h16() and not _is_narrow(Cin) and not _is_narrow(Cout)
          and Cin % 32 == 0 and Cout % 8 == 0)
    if not ok:
        y = conv2d(x, weight, bias, kernel_size=k, stride=stride, padding=padding, transposed=transposed, act=act, in_abs=in_abs,
                   tap_mask=tap_mask, packer=packer, mask=mask)
        return (y if want_lo else None), y
    L.require_cuda(x, weight)
    packer = packer if packer is not None else PackedWeight()
    B, _, H, W = x.shape
    Ho, Wo = _out_hw(H, W, k, stride, padding, transposed)
    x = _nhwc(x)
    wp = packer.get(weight, mask, Cout, Cin, k, k, transposed, False, x.dtype)
    return _wide_conv(x, wp, bias, B, H, W, Cin, Ho, Wo, Cout, k, stride, padding, transposed, act, int(in_abs), tap_mask,
                      f32_out="both" if want_lo else "only")


# ------------------------------------------------------------------------------ bf16x3 ("hi/lo") analysis path
# The reference computes g_a in fp32 and round()s its output; single-bf16 operands move y by ~3e-3 relative and flip ~1 % of
# the latents.  In the "bf16x3" analysis mode every value on the way to y is a PAIR of bf16 (hi = bf16(v), lo = bf16(v - hi)):
# activations [hi(C) | lo(C)] per pixel, weights [w_hi | w_lo] along Cin, three MFMA products per staged operand pair
# (x_hi w_hi + x_lo w_hi + x_hi w_lo; x_lo w_lo, 2^-18, is dropped) -- bf16 arithmetic on the matrix cores at ~2^-17 relative per operand.  Inference only.
# Modes (16-bit inference; the 16-bit format itself -- bfloat16 or float16 -- is the compute dtype):
#   "x3"    every analysis layer and the hyper-analysis on pairs, three products per MAC;
#   "x3c2"  as "x3", but g_a_conv2 (128 -> 128 5x5 s2 on the largest map: 70 % of g_a's MACs) multiplies SINGLE operands -- one product per
#           MAC; its input leaves the conv1 + GDN kernel as one 16-bit value per channel, its GDN epilogue runs on pairs and hands pairs on.
#           Meant for float16 (11-bit significand: ~6e-4 of the latents flip against the fp32 reference; with bfloat16 it is ~5e-3);
#   "x2"    pairs everywhere, TWO products per MAC: x pairs x single error-feedback weights (no w_lo term; round 5, measured and not the default);
#   "x1"    single operands everywhere (round 2's path; float16: ~1.3e-3 flips, bfloat16: ~1e-2).
# "bf16x3" / "bf16" are the round-3 names of "x3" / "x1".
ANALYSIS_MODES = ("x1", "x3", "x3c2", "x2")
_ANALYSIS_ALIASES = {"bf16": "x1", "bf16x3": "x3"}


def _canon_analysis(mode):
    mode = _ANALYSIS_ALIASES.get(mode, mode)
    if mode != "auto" and mode not in ANALYSIS_MODES:
        raise ValueError(f"analysis precision must be 'auto' or one of {ANALYSIS_MODES} (or {tuple(_ANALYSIS_ALIASES)})")
    return mode


_analysis_mode = _canon_analysis(_os.environ.get("HESIC_ANALYSIS", "auto"))


def set_analysis_precision(mode):
    """Operand precision of the analysis transforms + hyper-analysis of a 16-bit inference forward: "x3", "x3c2", "x2", "x1" (see above) or
    "auto" (the default) = "x3" for both 16-bit formats.  Round 4's default for float16 maps was "x3c2" (1.25x the speed of "x3"); round 5
    measured it at TRAINED operating points (piecewise-smooth pairs, 30 - 33 dB, ``profiles/scripts/parity_smooth.py``): 2.6 - 8e-4 of the
    latents flip there and every flip moves the reconstruction -- |dPSNR| 2 - 6e-3 dB, outside north_star's 1e-3 -- where "x3" stays at
    <= 5e-6 flips / 1.6 - 3.6e-4 dB.  "x3c2" (and "x2" = pairs x single error-feedback weights, 1.5 - 2.1e-4 flips, 1.3 - 1.5e-3 dB) stay
    available as explicit fast modes.  Returns the previous setting."""
    global _analysis_mode
    prev, _analysis_mode = _analysis_mode, _canon_analysis(mode)
    return prev


def analysis_precision():
    """The mode in effect for the current compute dtype ("auto" resolved)."""
    if _analysis_mode == "auto":
        return "x3"
    return _analysis_mode


def analysis_hilo(x):
    """True when the analysis stack should take the hi/lo route for input ``x``: 16-bit inference on the GPU."""
    return (analysis_precision() != "x1" and _is16() and not torch.is_grad_enabled() and x.is_cuda)


def analysis_conv2_single():
    """True in the "x3c2" mode: g_a_conv2 on single operands between the pair layers."""
    return analysis_precision() == "x3c2"


PAIR_WEIGHT_SCALING = True      # part of the pair arithmetic since round 5 (no run-time switch: it changes the last bit of y)


def _pair_weight_shift(w):
    """Power-of-two exponent s for packing a weight as the pair (w 2^s)_hi | (w 2^s)_lo.  binary16 only: the lo half of an unscaled weight of
    0.02 is 5e-6 -- a SUBNORMAL half (step 6e-8), so the pair carries 2^-20 instead of 2^-22 and a 128 -> 128 5x5 layer's output is
    ~1e-6 relative off, the largest single term of the pair mode's flipped latents (profiles/scripts/gdn_pair_precision.py).  With the
    largest |w| brought to [2^13, 2^14) every weight down to 2^-17 of it keeps a normal lo half; the kernel multiplies its accumulators by
    2^-s once (exact).  bfloat16 has fp32's range: s = 0.  One device -> host sync per repack (inference packs are cached)."""
    if not PAIR_WEIGHT_SCALING or _h16() != torch.float16 or not w.is_cuda:
        return 0
    m = float(w.abs().max())
    if not (m > 0.0) or m != m or m == float("inf"):
        return 0
    import math
    return max(0, min(24, 13 - math.frexp(m)[1] + 1))


class PackedWeightHiLo:
    """[w_hi | w_lo] (bf16, the kernels' [tap][Cout][2 Cin] layout) of a conv weight; inference cache keyed like ``PackedWeight``.
    ``as_1x1``: flatten (Cout, Cin, k, k) to a 1x1 weight over the im2col columns, zero-padded to ``kp`` columns."""

    def __init__(self):
        self._hit = None

    def get(self, weight, as_1x1=False, kp=0, single=False):
        """``single``: the rows of the two-product form (``hesic_conv2d_forward_hilo_w1``): [w rounded with error feedback over the taps | 0]."""
        tag = (weight.data_ptr(), weight._version, _cache_epoch, as_1x1, kp, single)
        if self._hit is not None and self._hit[0] == tag:
            return self._hit[1]
        if single:
            cout, cin, kh, kw = weight.shape
            tmp = torch.empty(kh * kw * cout * cin, dtype=_h16(), device=weight.device)
            L.call("hesic_pack_conv_weight_shaped", L.ptr(_c(weight.detach())), L.ptr(tmp), cout, cin, kh, kw, L.stream())
            wp = torch.empty((kh * kw, cout, 2, cin), dtype=_h16(), device=weight.device).fill_(0)
            wp[:, :, 0] = tmp.view(kh * kw, cout, cin)
            wp = wp.reshape(-1)
            self._hit = (tag, wp)
            return wp
        w = weight.detach().float()
        if as_1x1:
            cout = w.shape[0]
            flat = w.reshape(cout, -1)
            w = torch.cat([flat, flat.new_zeros(cout, kp - flat.shape[1])], 1).reshape(cout, kp, 1, 1)
        sh = _pair_weight_shift(w)
        if sh:
            w = w * float(2 ** sh)
        hi = w.to(_h16()).float()
        lo = (w - hi).to(_h16()).float()
        w2 = torch.cat([hi, lo], 1).contiguous()
        cout, cin2, kh, kw = w2.shape
        wp = torch.empty(kh * kw * cout * cin2, dtype=_h16(), device=w2.device)
        L.call("hesic_pack_conv_weight", L.ptr(w2), None, L.ptr(wp), cout, cin2, kh, kw, 0, 0, L.H16, L.stream())
        wp.hesic_acc_scale = float(2.0 ** -sh)          # conv2d_hilo hands it to the launch (hesic_conv2d_hilo_set_acc_scale)
        self._hit = (tag, wp)
        return wp


class PackedGdnLo:
    """Fragment-order lo half of gamma' for the hi/lo GDN epilogue (``hesic_gdn_pack_params_lo``); inference cache."""

    def __init__(self):
        self._hit = None

    def get(self, gamma):
        tag = (gamma.data_ptr(), gamma._version, _cache_epoch)
        if self._hit is not None and self._hit[0] == tag:
            return self._hit[1]
        glo = torch.empty(128 * 128, dtype=_h16(), device=gamma.device)
        L.call("hesic_gdn_pack_params_lo", L.ptr(_c(gamma)), L.ptr(glo), 128, L.stream())
        self._hit = (tag, glo)
        return glo


class PackedN2wHiLo:
    """LDS images [w_hi | w_lo | gamma'_hi | gamma'_lo] (128 KB) of the fused hi/lo g_a_conv1 + GDN kernel; inference cache."""

    def __init__(self):
        self._hit = {}

    def get(self, weight, gamma, out1=False):
        """``out1``: the image of the kernel's single-output form (w rounded with error feedback over the taps, lo halves unused)."""
        tag = (weight.data_ptr(), weight._version, gamma.data_ptr(), gamma._version, _cache_epoch)
        hit = self._hit.get(out1)
        if hit is not None and hit[0] == tag:
            return hit[1]
        img = torch.empty(128 * 1024, dtype=torch.uint8, device=weight.device)
        sh = 0 if out1 else _pair_weight_shift(weight.detach().float())
        if sh:
            # pair form, binary16: weights times 2^s in the image (normal lo halves); sconv_gdn_hilo then passes bias 2^s and beta' 4^s
            L.call("hesic_sconv_pack_weight_image_hilo_scaled", L.ptr(_c(weight)), L.ptr(_c(gamma)), float(2 ** sh), L.ptr(img), L.stream())
        else:
            L.call("hesic_sconv_pack_weight_image_hilo_out1" if out1 else "hesic_sconv_pack_weight_image_hilo", L.ptr(_c(weight)), L.ptr(_c(gamma)),
                   L.ptr(img), L.stream())
        img.hesic_wscale = float(2 ** sh)
        img.hesic_scaled = {}                              # (tensor address, version) -> the scaled copy of bias / beta'
        self._hit[out1] = (tag, img)
        return img


def sconv_gdn_hilo_ok(x, weight):
    """The fused hi/lo 3 -> 128 kernel takes an fp32 image with unit pixel stride and an even width."""
    return (x.dtype == torch.float32 and x.dim() == 4 and x.stride(3) == 1 and x.shape[3] % 2 == 0 and tuple(weight.shape) == (128, 3, 5, 5)
            and weight.dtype == torch.float32 and min(x.stride()) >= 0)


def sconv_gdn_hilo(x, image, bias, beta_packed, inverse, out1=False):
    """GDN(conv(x)) of the 3 -> 128 5x5 stride-2 stage on hi/lo pairs in one kernel: (B, 256, H/2, W/2) [hi | lo] 16-bit NHWC;
    ``out1``: the same arithmetic, but the output is ONE 16-bit value per channel, (B, 128, H/2, W/2)."""
    L.require_cuda(x)
    wsc = getattr(image, "hesic_wscale", 1.0)
    if wsc != 1.0:
        def scaled(t, f):
            if t is None:
                return None
            key = (t.data_ptr(), t._version, f)
            hit = image.hesic_scaled.get(key)
            if hit is None:
                if len(image.hesic_scaled) > 8:
                    image.hesic_scaled.clear()
                hit = image.hesic_scaled[key] = (t.detach().float() * f).contiguous()
            return hit
        bias, beta_packed = scaled(bias, wsc), scaled(beta_packed, wsc * wsc)
    B, Cc, H, W = x.shape
    Ho, Wo = _out_hw(H, W, 5, 2, 2, False)
    y = _empty_nhwc(B, 128 if out1 else 256, Ho, Wo, _h16(), x.device)
    d = _sdesc(x, y, 3, 128, 5, 2, 2, False)
    L.call("hesic_sconv2d_gdn_forward_hilo_out1" if out1 else "hesic_sconv2d_gdn_forward_hilo", C.byref(d), L.ptr(x), L.ptr(image), L.ptr(bias),
           L.ptr(beta_packed), int(inverse), L.ptr(y), L.stream())
    return y


def conv2d_gdn_hilo_out(x, wp, bias, cin, *, kernel_size, stride, padding, gdn):
    """(I)GDN(conv(x)) with SINGLE 16-bit operands in the conv (one product per MAC), the GDN on pairs from the fp32 accumulators and a
    hi/lo output map (B, 256, Ho, Wo) -- ``hesic_conv2d_gdn_forward_hilo_out``; ``gdn`` = (gamma_packed, gamma_lo_packed, beta_packed, inverse)."""
    L.require_cuda(x)
    k = kernel_size
    B, cx, H, W = x.shape
    Ho, Wo = _out_hw(H, W, k, stride, padding, False)
    x = _nhwc(x)
    gp, glo, bp, inverse = gdn
    y = _empty_nhwc(B, 256, Ho, Wo, _h16(), x.device)
    d = L.ConvDesc(B, H, W, cin, Ho, Wo, 128, k, k, stride, padding, 0, L.H16, 0, 0, cx, 0, 256, 0, 0)
    L.call("hesic_conv2d_gdn_forward_hilo_out", C.byref(d), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(gp), L.ptr(glo), L.ptr(bp), int(inverse),
           L.ptr(y), L.stream())
    return y


def im2col_hilo(x, k, stride, padding, kp):
    """Column matrix of a few-channel fp32 image as hi/lo bf16: (B, 2*kp, Ho, Wo) NHWC (``hesic_im2col_hilo``)."""
    L.require_cuda(x)
    if x.dtype != torch.float32:
        x = x.float()
    B, Cc, H, W = x.shape
    Ho, Wo = _out_hw(H, W, k, stride, padding, False)
    cols = _empty_nhwc(B, 2 * kp, Ho, Wo, _h16(), x.device)
    st = (C.c_int64 * 4)(*x.stride())
    L.call("hesic_im2col_hilo", L.ptr(x), st, B, Cc, H, W, k, k, stride, padding, Ho, Wo, kp, L.ptr(cols), L.stream())
    return cols


class HiLo(tuple):
    """A feature map stored as [hi(C) | lo(C)] bf16 pairs per pixel: ``HiLo((tensor (B, 2C, H, W) NHWC bf16, C))``.  A tuple, so the
    stream bookkeeping of the inference schedule (``models._tensors``) sees the tensor inside."""
    __slots__ = ()
    t = property(lambda self: self[0])
    c = property(lambda self: self[1])


def conv2d_hilo(x_hilo, wp3, bias, cin, cout, *, kernel_size, stride, padding, gdn=None, act=L.ACT_NONE, out="f32", out_abs=False, products=3):
    """Implicit GEMM on hi/lo operands (``hesic_conv2d_forward_hilo``).  ``gdn`` = (gamma_packed, gamma_lo_packed, beta_packed,
    inverse): fused hi/lo (I)GDN, returns the (B, 2*cout, Ho, Wo) hi/lo map.  Otherwise ``out``: "f32" -> act(conv + bias) as fp32
    (B, cout, Ho, Wo); "hilo" -> the hi/lo map (of |.| with ``out_abs``); "both" -> (hilo, f32)."""
    L.require_cuda(x_hilo)
    k = kernel_size
    B, c2, H, W = x_hilo.shape
    Ho, Wo = _out_hw(H, W, k, stride, padding, False)
    x_hilo = _nhwc(x_hilo)
    entry = "hesic_conv2d_forward_hilo_w1" if products == 2 else "hesic_conv2d_forward_hilo"      # ``products`` = 2: ``wp3`` from PackedWeightHiLo.get(single=True)
    acc_scale = getattr(wp3, "hesic_acc_scale", 1.0)            # weights packed times 2^s (PackedWeightHiLo): the launch unscales its sums
    if gdn is not None:
        gp, glo, bp, inverse = gdn
        y = _empty_nhwc(B, 2 * cout, Ho, Wo, _h16(), x_hilo.device)
        d = L.ConvDesc(B, H, W, cin, Ho, Wo, cout, k, k, stride, padding, 0, L.H16, 0, 0, c2, 0, 2 * cout, 0, 0)
        if acc_scale != 1.0:
            L.call("hesic_conv2d_hilo_set_acc_scale", acc_scale)
        L.call(entry, C.byref(d), L.ptr(x_hilo), L.ptr(wp3), L.ptr(bias), L.ptr(gp), L.ptr(glo), L.ptr(bp), int(inverse),
               L.ptr(y), 0, None, 0, 0, None, 0, L.stream())
        return y
    y = _empty_nhwc(B, 2 * cout, Ho, Wo, _h16(), x_hilo.device) if out in ("hilo", "both") else None
    y32 = _empty_nhwc(B, cout, Ho, Wo, torch.float32, x_hilo.device) if out in ("f32", "both") else None
    d = L.ConvDesc(B, H, W, cin, Ho, Wo, cout, k, k, stride, padding, 0, L.H16, act, 0, c2, 0, 2 * cout if y is not None else cout, 0, 0)
    key = ("hilo", B, H, W, cin, cout, k, stride, padding)
    need = _ws_bytes.get(key)
    if need is None:
        need = _ws_bytes[key] = int(L.lib().hesic_conv2d_hilo_ws_bytes(C.byref(d)))
    ws = torch.empty(need, dtype=torch.uint8, device=x_hilo.device) if (need and SPLIT_K) else None
    if acc_scale != 1.0:
        L.call("hesic_conv2d_hilo_set_acc_scale", acc_scale)
    L.call(entry, C.byref(d), L.ptr(x_hilo), L.ptr(wp3), L.ptr(bias), None, None, None, 0, L.ptr(y), int(out_abs),
           L.ptr(y32), cout, 0, L.ptr(ws), need if ws is not None else 0, L.stream())
    return (y, y32) if out == "both" else (y if out == "hilo" else y32)


class PackedGroup:
    """Packed bf16 weights (+ concatenated fp32 bias) of several same-geometry convs side by side along Cout, for
    ``conv2d_grouped``; inference cache with the tag policy of ``PackedWeight``."""

    def __init__(self):
        self._hit = None

    def get(self, weights, biases, k, transposed):
        """Returns (packed weights, bias, per-branch cout offsets, padded total): every branch starts on a multiple of 128 couts (the
        cout tile of the kernel; pad rows are zero weights / zero bias -- 960 -> 1024 for the mixture-parameter layers)."""
        tag = tuple((w.data_ptr(), w._version) for w in weights) + tuple((b.data_ptr(), b._version) for b in biases if b is not None) + (_cache_epoch,)
        if self._hit is not None and self._hit[0] == tag:
            return self._hit[1:]
        couts = [(w.shape[1] if transposed else w.shape[0]) for w in weights]
        cin = weights[0].shape[0] if transposed else weights[0].shape[1]
        offs, total = [], 0
        for co in couts:
            offs.append(total)
            total += -(-co // 128) * 128
        dev = weights[0].device
        wp = torch.empty(k * k * total * cin, dtype=_h16(), device=dev).fill_(0)
        bias = torch.empty(total, dtype=torch.float32, device=dev).fill_(0)
        for w, b, co, off in zip(weights, biases, couts, offs):
            L.call("hesic_pack_conv_weight_slice", L.ptr(w.detach().contiguous()), L.ptr(wp), co, cin, k, k, int(transposed), total, off, L.stream())
            if b is not None:
                bias[off:off + co].copy_(b.detach())
        self._hit = (tag, wp, bias, offs, total)
        return wp, bias, offs, total


def grouped_ok(x):
    """The grouped hyper-synthesis launches exist for bf16 inference."""
    return GROUP_HYPER and not torch.is_grad_enabled() and x.is_cuda and x.dtype == _h16()


GROUP_HYPER = _os.environ.get("HESIC_NO_GROUP_HYPER") is None      # A/B switch


def conv2d_grouped(x, weights, biases, packer, *, kernel_size, stride, padding, transposed=False, shared_input=True, x_c_off=0,
                   x_group_step=None, acts=None, f32_out=None):
    """Several convs of one geometry as ONE implicit-GEMM launch (``hesic_conv2d_forward_grouped``): ``weights`` / ``biases`` are the
    branches' tensors; with ``shared_input`` every branch reads channels [x_c_off, x_c_off + Cin) of ``x``, otherwise branch g reads
    [x_c_off + g * x_group_step, ...) (default step: Cin).  ``acts``: one activation per branch (at most two distinct values,
    switching once).  Returns ``(out, offsets)``: the concatenated output (B, padded sum of Cout, Ho, Wo) -- bf16, or with
    ``f32_out="only"`` fp32 from the accumulators -- and each branch's first channel in it (branches start on multiples of 128)."""
    L.require_cuda(x, *weights)
    k = kernel_size
    couts = [(w.shape[1] if transposed else w.shape[0]) for w in weights]
    cin = weights[0].shape[0] if transposed else weights[0].shape[1]
    G = len(weights)
    if not shared_input and len(set(couts)) != 1:
        raise ValueError("conv2d_grouped: branches with their own input slices need equal Cout")
    wp, bias, offs, total = packer.get(weights, biases, k, transposed)
    acts = list(acts) if acts is not None else [L.ACT_NONE] * G
    act, act2, split = acts[0], acts[0], 0
    for g in range(1, G):
        if acts[g] != acts[g - 1]:
            if split:
                raise ValueError("conv2d_grouped: the activation may switch once along the branches")
            act2, split = acts[g], offs[g]
    B, _, H, W = x.shape
    Ho, Wo = _out_hw(H, W, k, stride, padding, transposed)
    x = _nhwc(x)
    out = None if f32_out == "only" else _empty_nhwc(B, total, Ho, Wo, _h16(), x.device)
    out32 = _empty_nhwc(B, total, Ho, Wo, torch.float32, x.device) if f32_out else None
    groups = 1 if shared_input else G
    step = 0 if shared_input else (cin if x_group_step is None else int(x_group_step))
    d = L.ConvDesc(B, H, W, cin, Ho, Wo, total, k, k, stride, padding, int(transposed), L.H16, act, 0, x.shape[1], x_c_off, total, 0, 0)
    L.call("hesic_conv2d_forward_grouped", C.byref(d), groups, step, act2, split, L.ptr(x), L.ptr(wp), L.ptr(bias),
           L.ptr(out), L.ptr(out32), total, 0, L.stream())
    return (out32 if f32_out == "only" else (out if not f32_out else (out, out32))), offs


def conv2d_slice(x, x_c_off, weight, bias, *, kernel_size, stride, padding, transposed=False, act=L.ACT_NONE, packer=None):
    """Inference conv on the channel slice [x_c_off, x_c_off + Cin) of a wider NHWC tensor, read in place (no slice copy)."""
    L.require_cuda(x, weight)
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        raise RuntimeError("conv2d_slice is an inference form")
    k = kernel_size
    Cin, Cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    B, _, H, W = x.shape
    Ho, Wo = _out_hw(H, W, k, stride, padding, transposed)
    x = _nhwc(x)
    packer = packer if packer is not None else PackedWeight()
    wp = packer.get(weight, None, Cout, Cin, k, k, transposed, False, x.dtype)
    return _wide_conv(x, wp, bias, B, H, W, Cin, Ho, Wo, Cout, k, stride, padding, transposed, act, 0, 0, x_c_off=x_c_off)


def conv2d_into(x, weight, bias, out, out_c_off, *, kernel_size, stride, padding, transposed=False, act=L.ACT_NONE, packer=None,
                mask=None, tap_mask=0):
    """Inference conv whose output lands in channels [out_c_off, out_c_off + Cout) of the wider NHWC tensor ``out`` (the kernels
    write a channel slice in place): what a ``torch.cat`` of conv outputs would otherwise copy together."""
    L.require_cuda(x, weight, out)
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        raise RuntimeError("conv2d_into is an inference form")
    k = kernel_size
    Cin, Cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    B, _, H, W = x.shape
    Ho, Wo = _out_hw(H, W, k, stride, padding, transposed)
    if out.shape[0] != B or tuple(out.shape[2:]) != (Ho, Wo) or out_c_off + Cout > out.shape[1] or not out.is_contiguous(memory_format=_CL) \
            or out.dtype != x.dtype:
        raise ValueError("conv2d_into: `out` must be an NHWC tensor of the conv's dtype and output size with room for the channel slice")
    x = _nhwc(x)
    packer = packer if packer is not None else PackedWeight()
    wp = packer.get(weight, mask, Cout, Cin, k, k, transposed, False, x.dtype)
    _wide_conv(x, wp, bias, B, H, W, Cin, Ho, Wo, Cout, k, stride, padding, transposed, act, 0, tap_mask, out=out, out_c_off=out_c_off)
    return out


def copy_into(x, out, out_c_off):
    """``out[:, out_c_off:out_c_off + C] = x`` for NHWC tensors of one dtype (``hesic_copy_channels``)."""
    L.require_cuda(x, out)
    B, Cx, H, W = x.shape
    x = _nhwc(x)
    L.call("hesic_copy_channels", L.ptr(x), L.ptr(out), B * H * W, Cx, Cx, 0, out.shape[1], out_c_off, L.dt(x), L.stream())
    return out


def round_to(x, dtype):
    """round-half-even(x) stored as ``dtype``: ``_quantize(x, "dequantize")`` without means on an fp32 latent."""
    L.require_cuda(x)
    x = x.detach()
    if not (x.is_contiguous() or x.is_contiguous(memory_format=_CL)):
        x = x.contiguous(memory_format=_CL)
    y = torch.empty_like(x, dtype=dtype)
    L.call("hesic_round", L.ptr(x), L.dt(x), L.ptr(y), L.dt(dtype), x.numel(), L.stream())
    return y


# ------------------------------------------------------------------------------------ GDN
def conv3x3_c32_ok(x, weight):
    """Inference-only fast path of the enhancement net's 32-channel 3x3 convs (``hesic_conv3x3_c32_forward``)."""
    return (not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and x.shape[1] == 32 and x.dtype == _h16()
            and weight.shape[1] == 32 and tuple(weight.shape[2:]) == (3, 3) and (weight.shape[0] == 32 or weight.shape[0] <= 4)
            and weight.dtype == torch.float32 and _is16())


def pack_images_c32(xa, xb):
    """cat((xa, xb), 1) of two (B,3,H,W) images as channels 0..5 of a zero-padded (B,32,H,W) bf16 NHWC tensor."""
    L.require_cuda(xa, xb)
    B, _, H, W = xa.shape
    out = _empty_nhwc(B, 32, H, W, _h16(), xa.device)
    L.call("hesic_pack_images_c32", L.ptr(xa.float().contiguous()), L.ptr(xb.float().contiguous()), L.ptr(out), B, H, W, L.stream())
    return out


def conv3x3_c32_img6(xa, xb, weight, bias, act=L.ACT_NONE):
    """act(conv3x3(cat((xa, xb), 1)) + bias) of the enhancement net's input layer (newnet1.py:300-301) in ONE launch (round 6,
    ``hesic_conv3x3_c32_forward_img6``): the two fp32 planar (B,3,H,W) images are read as they are, (B,32,H,W) 16-bit NHWC out; bit-identical to
    ``pack_images_c32`` + ``conv3x3_c32`` with the weight zero-padded along Cin."""
    L.require_cuda(xa, xb, weight)
    B, _, H, W = xa.shape
    y = _empty_nhwc(B, 32, H, W, _h16(), xa.device)
    L.call("hesic_conv3x3_c32_forward_img6", L.ptr(xa.float().contiguous()), L.ptr(xb.float().contiguous()), L.ptr(weight.detach().float().contiguous()),
           L.ptr(None if bias is None else bias.detach().float()), int(act), L.ptr(y), B, H, W, L.stream())
    return y


def conv3x3_c32(x, weight, bias, act=L.ACT_NONE, res1=None, res2=None):
    """act(conv3x3(x) + bias) + res1 + res2 in one launch: x (B,32,H,W) bf16 (any layout, made NHWC); 32 couts -> bf16 NHWC
    with bf16 residuals, <= 4 couts -> fp32 planar with one fp32 planar residual (the 32 -> 3 output conv + the image)."""
    L.require_cuda(x, weight)
    B, _, H, W = x.shape
    cout = weight.shape[0]
    x = _nhwc(x)
    w = weight.detach().contiguous()
    if cout == 32:
        y = _empty_nhwc(B, 32, H, W, _h16(), x.device)
        r1 = None if res1 is None else _nhwc(res1.to(_h16()))
        r2 = None if res2 is None else _nhwc(res2.to(_h16()))
    else:
        y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
        r1 = None if res1 is None else res1.to(torch.float32).contiguous()
        r2 = None
        if res2 is not None:
            raise RuntimeError("conv3x3_c32: the planar form takes one residual")
    L.call("hesic_conv3x3_c32_forward", L.ptr(x), L.ptr(w), L.ptr(None if bias is None else bias.detach().float()), cout, int(act),
           L.ptr(r1), L.ptr(r2), L.ptr(y), B, H, W, L.stream())
    return y


RESBLOCK_FUSED = True      # module switch: a ResidualBlock is ONE launch at inference


def resblock_c32(x, w1, b1, w2, b2, act=L.ACT_LEAKY, res2=None):
    """act(conv3x3(act(conv3x3(x, w1) + b1), w2) + b2) + x + res2 in one launch (``hesic_resblock_c32_forward``): the ResidualBlock of
    the enhancement stage at inference (layers.py:125-147), x (B,32,H,W) in the 16-bit format; round 6: agrees with two ``conv3x3_c32`` calls to the last bits, not bit for bit (another summation order:
    include/hesic_hip.h)."""
    L.require_cuda(x, w1, w2)
    B, _, H, W = x.shape
    x = _nhwc(x)
    y = _empty_nhwc(B, 32, H, W, _h16(), x.device)
    r2 = None if res2 is None else _nhwc(res2.to(_h16()))
    f32 = lambda t: None if t is None else t.detach().float().contiguous()
    L.call("hesic_resblock_c32_forward", L.ptr(x), L.ptr(f32(w1)), L.ptr(f32(b1)), L.ptr(f32(w2)), L.ptr(f32(b2)), int(act), L.ptr(r2), L.ptr(y),
           B, H, W, L.stream())
    return y


EN_TRAIN_FAST = True      # module switch


def conv3x3_c32_train_ok(x, weight):
    """Training form of the 32 -> 32 fast path: autograd on, bf16 storage (stage 2 trains the enhancement net with HSIC frozen,
    newnet1.py:272-311, newtrain6_real.py)."""
    return (EN_TRAIN_FAST and torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and x.shape[1] == 32 and x.dtype == _h16()
            and weight.shape[0] == 32 and weight.shape[1] <= 32 and tuple(weight.shape[2:]) == (3, 3) and weight.dtype == torch.float32
            and _is16())


def conv3x3_c32_wgrad(x, g, weight, bias):
    """(dw, dbias) of a 3x3 conv over 32-channel NHWC bf16 maps (``hesic_conv3x3_c32_wgrad``): ``x`` the layer input, ``g`` the gradient
    w.r.t. conv + bias; ``weight`` (Cout <= 32, Cin <= 32, 3, 3) and ``bias`` only give shapes / flat gradient slots (with slots the
    gradients are added in place and come back as None)."""
    B, _, H, W = x.shape
    cout, cin = weight.shape[0], weight.shape[1]
    nws = int(L.lib().hesic_conv3x3_c32_wgrad_ws_bytes())
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    sw, sb = _slot_for(weight), (_slot_for(bias) if bias is not None else None)
    if sw is not None and (bias is None or sb is not None):
        L.call("hesic_conv3x3_c32_wgrad", L.ptr(x), L.ptr(g), L.ptr(sw.grad), L.ptr(None if bias is None else sb.grad), cout, cin, 1,
               L.ptr(ws), nws, B, H, W, L.stream())
        _slot_done(sw)
        if bias is not None:
            _slot_done(sb)
        return None, None
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if bias is not None else None
    L.call("hesic_conv3x3_c32_wgrad", L.ptr(x), L.ptr(g), L.ptr(dw), L.ptr(db), cout, cin, 0, L.ptr(ws), nws, B, H, W, L.stream())
    return dw, db


def _mirror_t(weight, pad_to=None):
    """(Cout, Cin, 3, 3) -> the weight of the data-gradient conv: (Cin, Cout, 3, 3) with the taps mirrored, Cout zero-padded to ``pad_to``."""
    wt = weight.detach().flip(2, 3).transpose(0, 1)
    if pad_to is not None and wt.shape[1] < pad_to:
        wt = torch.nn.functional.pad(wt, (0, 0, 0, 0, 0, pad_to - wt.shape[1]))
    return wt.contiguous()


class _Conv3x3C32Fn(torch.autograd.Function):
    """act(conv3x3(x) + bias) of the 32-channel enhancement layers under autograd, forward, data gradient AND weight gradient on kernels
    built for the shape: dx = conv3x3(act'(y) * dy, W^T with the taps mirrored) is ``hesic_conv3x3_c32_forward`` on another weight tensor,
    dW / dbias come from ``hesic_conv3x3_c32_wgrad`` (flat slots included).  ``weight`` may have fewer than 32 input channels (the
    6 -> 32 input layer on the zero-padded image map: the kernel then sees the zero-padded weight).  Residual adds stay outside (the
    activation's derivative needs the sign of the un-added output)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = _nhwc(x)
        w = weight.detach()
        if w.shape[1] < 32:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 32 - w.shape[1]))
        y = conv3x3_c32(x, w, bias, act=act)
        ctx.save_for_backward(x, weight, y if act else None)
        ctx.act, ctx.bias = act, bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y_act = ctx.saved_tensors
        gy = _nhwc(gy.to(_h16()))
        if ctx.act:
            g = torch.empty_like(y_act)
            L.call("hesic_act_backward", L.ptr(y_act), L.ptr(gy), L.ptr(g), y_act.numel(), ctx.act, L.dt(y_act), L.stream())
        else:
            g = gy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv3x3_c32(g, _mirror_t(weight, 32), None)          # (a narrower conv's extra input channels get zero gradient)
        if ctx.needs_input_grad[1]:
            dw, db = conv3x3_c32_wgrad(x, g, weight, ctx.bias)
        return dx, dw, db, None


class _Conv3x3C32OutFn(torch.autograd.Function):
    """The 32 -> 3 output conv of ``Enhancement`` plus the image it refines (fp32 planar out) under autograd: the incoming gradient
    goes into channels 0..2 of a zero-padded 32-channel map, after which data and weight gradient are the 32-channel kernels'."""

    @staticmethod
    def forward(ctx, t, weight, bias, image):
        t = _nhwc(t)
        y = conv3x3_c32(t, weight, bias, res1=image)
        ctx.save_for_backward(t, weight)
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, gy):
        t, weight = ctx.saved_tensors
        gy = gy.float().contiguous()
        g32 = pack_images_c32(gy, torch.zeros_like(gy))
        if weight.shape[0] != 3:
            raise NotImplementedError("conv3x3_c32 output layer: 3 output channels")
        dt = dw = db = None
        if ctx.needs_input_grad[0]:
            dt = conv3x3_c32(g32, _mirror_t(weight, 32), None)
        if ctx.needs_input_grad[1]:
            dw, db = conv3x3_c32_wgrad(t, g32, weight, ctx.bias)
        return dt, dw, db, (gy if ctx.needs_input_grad[3] else None)


def conv3x3_c32_train(x, weight, bias, act=L.ACT_NONE, packer=None):
    return _Conv3x3C32Fn.apply(x, weight, bias, act)


def conv3x3_c32_out_train(t, weight, bias, image):
    return _Conv3x3C32OutFn.apply(t, weight, bias, image)


def conv2d_cat(xa, xb, weight, bias, *, kernel_size, stride, padding, transposed=False, packer=None, gdn=None, gdn_on_input=False):
    """conv(torch.cat((xa, xb), 1)) (newnet1.py:643,686).  At inference the 6 -> 3 image-side stages read their two
    3-channel halves straight from the two tensors (``hesic_sconv2d_forward_cat``: no concatenated copy); otherwise the
    ordinary cat + conv2d (autograd) path runs.  ``gdn``: the 3-channel GDN module next to the conv -- applied to the conv's
    output (pre_conv -> GDN, :643-644) or, with ``gdn_on_input``, to ``xa`` before the cat (IGDN -> cat -> after_conv, :684-686);
    fused into the same launch on the fast path."""
    cin = weight.shape[0] if transposed else weight.shape[1]
    cout = weight.shape[1] if transposed else weight.shape[0]
    ok = (not torch.is_grad_enabled() and xa.is_cuda and xb.is_cuda and cin == 6 and cout == 3 and kernel_size == 5 and stride == 1
          and padding == 2 and xa.shape[1] + xb.shape[1] == 6 and xa.shape[-1] >= 128 and xa.shape[0] == xb.shape[0]
          and xa.shape[2:] == xb.shape[2:] and xa.dtype in (torch.float32, _h16()) and xb.dtype in (torch.float32, _h16()))
    fuse = ok and gdn is not None and FUSE_GDN3 and (not gdn_on_input or xa.shape[1] == 3)
    if not ok or (gdn is not None and not fuse):
        gdn_now = None if gdn is None else (lambda t: _gdn_op(t, gdn.beta, gdn.gamma, gdn.inverse, gdn.beta_min))      # the operator, not the module's (deferring) __call__
        if gdn is not None and gdn_on_input:
            xa = gdn_now(xa)
        if ok:
            y = conv2d_cat(xa, xb, weight, bias, kernel_size=kernel_size, stride=stride, padding=padding, transposed=transposed, packer=packer)
        else:
            y = conv2d(torch.cat((xa.float(), xb.float()), 1), weight, bias, kernel_size=kernel_size, stride=stride,
                       padding=padding, transposed=transposed, packer=packer)
        return gdn_now(y) if (gdn is not None and not gdn_on_input) else y
    B, _, H, W = xa.shape
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=xa.device)
    d = _sdesc(xa, y, cin, cout, kernel_size, stride, padding, transposed)
    xbs = (C.c_int64 * 4)(*xb.stride())
    if fuse:
        L.call("hesic_sconv2d_forward_cat_gdn", C.byref(d), L.ptr(xa), L.ptr(xb), xbs, L.dt(xb), int(xa.shape[1]), L.ptr(_c(weight)),
               L.ptr(bias), L.ptr(_c(gdn.beta)), L.ptr(_c(gdn.gamma)), float(gdn.beta_min), int(bool(gdn.inverse)), int(bool(gdn_on_input)),
               L.ptr(y), L.stream())
    else:
        L.call("hesic_sconv2d_forward_cat", C.byref(d), L.ptr(xa), L.ptr(xb), xbs, L.dt(xb), int(xa.shape[1]),
               L.ptr(_c(weight)), L.ptr(bias), L.ptr(y), L.stream())
    return y


class _GdnFn(torch.autograd.Function):
    """GDN.forward (compressai/layers/gdn.py:55-70), reparametrisation included."""

    @staticmethod
    def forward(ctx, x, beta, gamma, inverse, beta_min):
        L.require_cuda(x, beta, gamma)
        B, Cc, H, W = x.shape
        x = _nhwc(x)
        y = torch.empty_like(x, memory_format=_CL)
        L.call("hesic_gdn_forward", L.ptr(x), L.ptr(_c(beta)), L.ptr(_c(gamma)), L.ptr(y),
               B * H * W, Cc, int(inverse), float(beta_min), L.dt(x), L.stream())
        ctx.save_for_backward(x, beta, gamma)
        ctx.inverse, ctx.beta_min = inverse, beta_min
        return y

    @staticmethod
    def backward(ctx, gy):
        x, beta, gamma = ctx.saved_tensors
        dx, dbeta, dgamma = _gdn_backward(x, _nhwc(gy.to(x.dtype)), beta, gamma, ctx.inverse, ctx.beta_min)
        return dx, dbeta, dgamma, None, None


class _Gdn3PlanarFn(torch.autograd.Function):
    """The 3-channel image-side GDN / IGDN (pre_gdn / after_gdn, newnet1.py:630,669) on a planar tensor under autograd: forward
    ``hesic_gdn_forward_planar``, backward one fused pass (``hesic_gdn_backward_planar_acc``) -- no NHWC copies, planar in and out."""

    @staticmethod
    def forward(ctx, x, beta, gamma, inverse, beta_min):
        L.require_cuda(x, beta, gamma)
        B, Cc, H, W = x.shape
        y = torch.empty_like(x)
        L.call("hesic_gdn_forward_planar", L.ptr(x), L.ptr(_c(beta)), L.ptr(_c(gamma)), L.ptr(y), B, Cc, H * W,
               int(inverse), float(beta_min), L.dt(x), L.stream())
        ctx.save_for_backward(x, beta, gamma)
        ctx.inverse, ctx.beta_min = inverse, beta_min
        return y

    @staticmethod
    def backward(ctx, gy):
        x, beta, gamma = ctx.saved_tensors
        B, Cc, H, W = x.shape
        gy = gy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        sb, sg = _slot_for(beta), _slot_for(gamma)
        direct = sb is not None and sg is not None
        dbeta = sb.grad if direct else torch.empty_like(beta, dtype=torch.float32)
        dgamma = sg.grad if direct else torch.empty_like(gamma, dtype=torch.float32)
        ws = torch.empty(64, dtype=torch.uint8, device=x.device)
        L.call("hesic_gdn_backward_planar_acc", L.ptr(x), L.ptr(gy), L.ptr(_c(beta)), L.ptr(_c(gamma)), L.ptr(dx), L.ptr(dbeta),
               L.ptr(dgamma), int(direct), L.ptr(ws), B, H * W, Cc, int(ctx.inverse), float(ctx.beta_min), L.dt(x), L.stream())
        if direct:
            _slot_done(sb)
            _slot_done(sg)
            return dx, None, None, None, None
        return dx, dbeta, dgamma, None, None


GDN3_PLANAR_TRAIN = True      # module switch: False = the NHWC route of rounds 1-3 under autograd


def gdn(x, beta, gamma, inverse=False, beta_min=1e-6):
    if x.shape[1] == 3 and x.is_cuda and x.is_contiguous() and not torch.is_grad_enabled() and x.dtype in (torch.float32, _h16()):
        # image-side GDN on a planar tensor at inference: no NHWC round trip
        B, Cc, H, W = x.shape
        y = torch.empty_like(x)
        L.call("hesic_gdn_forward_planar", L.ptr(x), L.ptr(_c(beta)), L.ptr(_c(gamma)), L.ptr(y), B, Cc, H * W,
               int(inverse), float(beta_min), L.dt(x), L.stream())
        return y
    if (GDN3_PLANAR_TRAIN and x.shape[1] == 3 and x.is_cuda and x.is_contiguous() and x.dtype in (torch.float32, _h16())
            and beta.dtype == torch.float32 and gamma.dtype == torch.float32):
        return _apply(_Gdn3PlanarFn, x, beta, gamma, inverse, beta_min)
    return _apply(_GdnFn, x, beta, gamma, inverse, beta_min)


_gdn_op = gdn          # conv2d_cat's parameter ``gdn`` is the MODULE


# ----------------------------------------------------------------------------------- warp
class _WarpFn(torch.autograd.Function):
    """kornia.warp_perspective(src, M, dsize) (third party; call sites newnet1.py:746,753,767)."""

    @staticmethod
    def forward(ctx, src, M, dsize, align_corners, inverse_map=False):
        L.require_cuda(src, M)
        B, Cc, H, W = src.shape
        Ho, Wo = int(dsize[0]), int(dsize[1])
        if src.dtype not in (torch.float32, _h16()):
            raise TypeError("warp_perspective: float32 or the active 16-bit format only")
        Mf = M.detach().to(torch.float32).contiguous()
        dst = torch.empty((B, Cc, Ho, Wo), dtype=src.dtype, device=src.device)
        ss, ds = src.stride(), dst.stride()
        d = L.WarpDesc(B, Cc, H, W, Ho, Wo, int(align_corners), L.dt(src), L.dt(dst), int(inverse_map), ss[0], ss[1], ss[2], ss[3],
                       ds[0], ds[1], ds[2], ds[3])
        L.call("hesic_warp_perspective_forward", C.byref(d), L.ptr(src), L.ptr(Mf), L.ptr(dst), L.stream())
        ctx.save_for_backward(Mf)
        ctx.meta = (src.shape, src.dtype, Ho, Wo, int(align_corners), int(inverse_map))
        return dst

    @staticmethod
    def backward(ctx, g):
        (Mf,) = ctx.saved_tensors
        shape, sdt, Ho, Wo, ac, inv = ctx.meta
        B, Cc, H, W = shape
        g = g.contiguous()
        dsrc = _zeros(shape, torch.float32, g.device)
        ss, ds = dsrc.stride(), g.stride()
        d = L.WarpDesc(B, Cc, H, W, Ho, Wo, ac, L.F32, L.dt(g), inv, ss[0], ss[1], ss[2], ss[3], ds[0], ds[1], ds[2], ds[3])
        L.call("hesic_warp_perspective_backward", C.byref(d), L.ptr(g), L.ptr(Mf), L.ptr(dsrc), L.stream())
        return dsrc.to(sdt), None, None, None, None


def warp_perspective(src, M, dsize, align_corners=True, inverse_map=False):
    """``inverse_map``: ``M`` already maps destination pixels to source pixels, i.e. this is the warp by ``M^-1``
    (Independent_EN warps view 2 by ``torch.inverse(h_matrix)``, newnet1.py:1290-1291: no 3x3 inversion launches)."""
    return _apply(_WarpFn, src, M, dsize, align_corners, inverse_map)


# ---------------------------------------------------------------------- entropy bottleneck
EB_BOUND_SLOT = 60


def eb_pack_params(matrices, biases, factors, quantiles, lik_bound=1e-9):
    """[C][64] fp32 table read by hesic_eb_forward/backward (layout in csrc/entropy.hip); slot 60 = likelihood lower bound."""
    Cc = quantiles.shape[0]
    cols = [m.reshape(Cc, -1) for m in matrices] + [b.reshape(Cc, -1) for b in biases] + \
           [f.reshape(Cc, -1) for f in factors] + [quantiles[:, 0, 1:2]]
    t = torch.cat(cols, 1).to(torch.float32)
    assert t.shape[1] == 59, "EntropyBottleneck kernels are specialised for filters=(3,3,3,3)"
    t = torch.nn.functional.pad(t, (0, L.EB_PARAM_STRIDE - 59)).contiguous()
    t[:, EB_BOUND_SLOT] = float(lik_bound)
    return t


def eb_unpack_grads(dparams, matrices, biases, factors, quantiles):
    out, o = [], 0
    for group in (matrices, biases, factors):
        for p in group:
            n = p[0].numel()
            out.append(dparams[:, o:o + n].reshape(p.shape).to(p.dtype))
            o += n
    dq = _zeros(quantiles.shape, quantiles.dtype, quantiles.device)
    dq[:, 0, 1] = dparams[:, 58]
    return out, dq


def _eb_layout(matrices, biases, factors, quantiles, lik_bound=0.0, grads=False):
    """``EbLayout`` of a bottleneck's 14 parameter tensors (or, ``grads``: of their flat gradient slots; None if any has no
    slot).  Table columns: matrices, then biases, then factors, each (C, rows, cols) flattened per channel; column 58 = the
    median = quantiles[:, 0, 1]."""
    lay = L.EbLayout()
    col, j = 0, 0
    for group in (matrices, biases, factors):
        for p in group:
            t = p
            if grads:
                sl = _slot_for(p)
                if sl is None:
                    return None
                t = sl.grad
            n = p[0].numel()
            lay.ptr[j], lay.width[j], lay.stride[j], lay.first[j], lay.col[j] = t.data_ptr(), n, n, 0, col
            col += n
            j += 1
    if col != 58:
        raise NotImplementedError("EntropyBottleneck kernels are specialised for filters=(3,3,3,3)")
    t = quantiles
    if grads:
        sl = _slot_for(quantiles)
        if sl is None:
            return None
        t = sl.grad
    lay.ptr[j], lay.width[j], lay.stride[j], lay.first[j], lay.col[j] = t.data_ptr(), 1, 3, 1, 58
    lay.n, lay.lik_bound = j + 1, float(lik_bound)
    return lay


def _eb_table(matrices, biases, factors, quantiles, lik_bound):
    """The raw [C][64] parameter table in one launch (``hesic_eb_pack_table``)."""
    ps = (*matrices, *biases, *factors, quantiles)
    if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
        return eb_pack_params([m.detach() for m in matrices], [b.detach() for b in biases], [f.detach() for f in factors],
                              quantiles.detach(), lik_bound)
    Cc = quantiles.shape[0]
    table = torch.empty((Cc, L.EB_PARAM_STRIDE), dtype=torch.float32, device=quantiles.device)
    lay = _eb_layout(matrices, biases, factors, quantiles, lik_bound)
    L.call("hesic_eb_pack_table", C.byref(lay), L.ptr(table), Cc, L.stream())
    return table


class _EbFn(torch.autograd.Function):
    """EntropyBottleneck.forward (entropy_models.py:384-411): returns (z_hat, likelihood)."""

    @staticmethod
    def forward(ctx, z, noise, quantiles, n_mat, lik_bound, *params):
        L.require_cuda(z)
        matrices, biases, factors = params[:n_mat], params[n_mat:2 * n_mat], params[2 * n_mat:]
        B, Cc, H, W = z.shape
        z = _nhwc(z)
        table = _eb_table(matrices, biases, factors, quantiles, lik_bound)
        zh = torch.empty_like(z, memory_format=_CL)
        lik = _empty_nhwc(B, Cc, H, W, torch.float32, z.device)
        nz = None if noise is None else _nhwc(noise.to(z.dtype))
        L.call("hesic_eb_forward", L.ptr(z), L.ptr(table), L.ptr(nz), L.ptr(zh), L.ptr(lik), None, B * H * W, Cc,
               L.dt(z), L.stream())
        ctx.save_for_backward(z, table, nz, quantiles, *params)
        ctx.n_mat = n_mat
        return zh, lik

    @staticmethod
    def backward(ctx, g_zh, g_lik):
        z, table, nz, quantiles, *params = ctx.saved_tensors
        n_mat = ctx.n_mat
        B, Cc, H, W = z.shape
        dz = torch.empty_like(z, memory_format=_CL)
        dpar = _zeros(table.shape, table.dtype, table.device)
        g_lik = _nhwc(g_lik.to(torch.float32))
        g_zh = None if g_zh is None else _nhwc(g_zh.to(z.dtype))
        L.call("hesic_eb_backward", L.ptr(z), L.ptr(table), L.ptr(nz), L.ptr(g_lik), L.ptr(g_zh), L.ptr(dz), L.ptr(dpar),
               B * H * W, Cc, L.dt(z), L.stream())
        glay = _eb_layout(params[:n_mat], params[n_mat:2 * n_mat], params[2 * n_mat:], quantiles, grads=True)
        if glay is not None:          # flat gradient slots: one scatter-add launch instead of 14 slice copies + AccumulateGrad
            L.call("hesic_eb_scatter_grads", C.byref(glay), L.ptr(dpar), Cc, 1, L.stream())
            for p in (*params, quantiles):
                _slot_done(_slot_for(p))
            return (dz, None, None, None, None, *([None] * len(params)))
        grads, dq = eb_unpack_grads(dpar, params[:n_mat], params[n_mat:2 * n_mat], params[2 * n_mat:], quantiles)
        return (dz, None, dq, None, None, *grads)


class PackedEb:
    """The [C][64] parameter table of an EntropyBottleneck; inference-only cache (same policy as PackedWeight)."""

    def __init__(self):
        self._hit = None

    def get(self, matrices, biases, factors, quantiles, lik_bound=1e-9):
        ps = (*matrices, *biases, *factors, quantiles)
        tag = tuple((p.data_ptr(), p._version) for p in ps) + (_cache_epoch, float(lik_bound))
        if self._hit is None or self._hit[0] != tag:
            raw = _eb_table(matrices, biases, factors, quantiles, lik_bound)
            ready = torch.empty_like(raw)           # softplus / tanh applied once here, not per thread and launch
            L.call("hesic_eb_prepare_params", L.ptr(raw), L.ptr(ready), raw.shape[0], L.stream())
            self._hit = (tag, ready)
        return self._hit[1]


def entropy_bottleneck(z, matrices, biases, factors, quantiles, noise=None, packer=None, out_dtype=None, lik_bound=1e-9):
    """``out_dtype`` (inference, fp32 ``z`` only): storage of z_hat when it differs from z's (bf16 mode with fp32 latents)."""
    if packer is not None and noise is None and not torch.is_grad_enabled():
        # inference: cached parameter table, no autograd bookkeeping -- one launch
        L.require_cuda(z)
        B, Cc, H, W = z.shape
        z = _nhwc(z)
        table = packer.get(matrices, biases, factors, quantiles, lik_bound)
        if out_dtype is not None and out_dtype != z.dtype and z.dtype == torch.float32:
            zh = _empty_nhwc(B, Cc, H, W, out_dtype, z.device)
            lik = _empty_nhwc(B, Cc, H, W, torch.float32, z.device)
            L.call("hesic_eb_forward_f32in", L.ptr(z), L.ptr(table), L.ptr(zh), L.dt(out_dtype), L.ptr(lik), None, B * H * W, Cc, L.stream())
            return zh, lik
        zh = torch.empty_like(z, memory_format=_CL)
        lik = _empty_nhwc(B, Cc, H, W, torch.float32, z.device)
        L.call("hesic_eb_forward", L.ptr(z), L.ptr(table), None, L.ptr(zh), L.ptr(lik), None, B * H * W, Cc, L.dt(z), L.stream())
        return zh, lik
    return _apply(_EbFn, z, noise, quantiles, len(matrices), float(lik_bound), *matrices, *biases, *factors)


class _EbAuxFn(torch.autograd.Function):
    """EntropyBottleneck.loss (entropy_models.py:345-348): sum |c(quantiles) - target| with the cumulative's parameters
    detached -- forward AND the quantile gradient in one launch (the tensor-op form is ~35 launches per bottleneck)."""

    @staticmethod
    def forward(ctx, quantiles, tail_mass, n_mat, *params):
        L.require_cuda(quantiles)
        matrices, biases, factors = params[:n_mat], params[n_mat:2 * n_mat], params[2 * n_mat:]
        Cc = quantiles.shape[0]
        table = _eb_table(matrices, biases, factors, quantiles, 0.0)
        loss = _zeros(1, torch.float32, quantiles.device)
        dq = torch.empty_like(quantiles, dtype=torch.float32)
        L.call("hesic_eb_aux_loss", L.ptr(table), L.ptr(_c(quantiles)), float(tail_mass), L.ptr(loss), L.ptr(dq), Cc, 0, L.stream())
        ctx.save_for_backward(dq)
        ctx.n_params = len(params)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dq,) = ctx.saved_tensors
        return (dq * g, None, None, *([None] * ctx.n_params))


def eb_aux_loss(matrices, biases, factors, quantiles, tail_mass=1e-9):
    return _EbAuxFn.apply(quantiles, float(tail_mass), len(matrices), *matrices, *biases, *factors)


# ------------------------------------------------------------ Gaussian (mixture) conditional
class _GmmFn(torch.autograd.Function):
    """GaussianMixtureConditional.forward (entropy_models.py:661-702) for K>1 with weights, and
    GaussianConditional.forward (:546-554) for K==1 (means optional, used in the quantiser)."""

    @staticmethod
    def forward(ctx, y, scales, means, weights, noise, K, use_means_in_quant, scale_bound, lik_bound):
        L.require_cuda(y, scales, means)
        B, M, H, W = y.shape
        y = _nhwc(y)
        if means is None:
            means = _zeros(scales.shape, scales.dtype, scales.device).contiguous(memory_format=_CL) if scales.dim() == 4 else _zeros(scales.shape, scales.dtype, scales.device)
        # scales / means may be channel slices of one tensor (chunk(2,1)): read in place
        ps, sptr, mptr, _keep = _sm_pointers(_nd(scales), _nd(means), y, B, K, M)
        if ps == K * M:
            scales, means = _keep
        wts = None if weights is None else weights.detach().reshape(B, K * M).to(torch.float32).contiguous()
        yh = torch.empty_like(y, memory_format=_CL)
        lik = _empty_nhwc(B, M, H, W, torch.float32, y.device)
        nz = None if noise is None else _nhwc(noise.to(y.dtype))
        d = L.GmmDesc(B, H * W, M, K, L.dt(y), int(use_means_in_quant), ps, 0, 0, float(scale_bound), float(lik_bound))
        L.call("hesic_gmm_forward", C.byref(d), L.ptr(y), sptr, mptr, L.ptr(wts), L.ptr(nz), L.ptr(yh), L.ptr(lik), None,
               L.stream())
        ctx.save_for_backward(y, scales, means, wts, nz)
        ctx.meta = (K, int(use_means_in_quant), float(scale_bound), float(lik_bound), weights is not None)
        return yh, lik

    @staticmethod
    def backward(ctx, g_yh, g_lik):
        y, scales, means, wts, nz = ctx.saved_tensors
        K, umq, sb, lb, has_w = ctx.meta
        B, M, H, W = y.shape
        scales, means = _nhwc(scales.to(y.dtype)), _nhwc(means.to(y.dtype))
        dy = torch.empty_like(y, memory_format=_CL)
        dsc = torch.empty_like(scales, memory_format=_CL)
        dmu = torch.empty_like(means, memory_format=_CL)
        dw = _zeros((B, K * M), torch.float32, y.device) if has_w else None
        g_lik = _nhwc(g_lik.to(torch.float32))
        g_yh = None if g_yh is None else _nhwc(g_yh.to(y.dtype))
        d = L.GmmDesc(B, H * W, M, K, L.dt(y), umq, K * M, 0, 0, sb, lb)
        L.call("hesic_gmm_backward", C.byref(d), L.ptr(y), L.ptr(scales), L.ptr(means), L.ptr(wts), L.ptr(nz), L.ptr(g_lik),
               L.ptr(g_yh), L.ptr(dy), L.ptr(dsc), L.ptr(dmu), L.ptr(dw), L.stream())
        if dw is not None:
            dw = dw.reshape(B, K * M, 1, 1)
        return dy, dsc, dmu, dw, None, None, None, None, None


def _sm_pointers(scales, means, y, B, K, M):
    """(pixel stride, scales ptr, means ptr, keep-alive) of the two parameter maps in ``y``'s dtype.  Channel slices of wider
    channels_last tensors (``chunk(2, 1)`` of HESIC+'s entropy parameters, the sigma | means halves of the grouped hyper-synthesis
    output) are read in place: all the kernels need is NHWC with one common pixel stride."""
    def nhwc_stride(t):
        if t.dim() != 4 or t.dtype != y.dtype or t.shape[0] != B or t.shape[2:] != y.shape[2:]:
            return None
        sb, sc, sh, sw = t.stride()
        H, W = t.shape[2], t.shape[3]
        ok = sc == 1 and sw >= t.shape[1] and (W == 1 or sh == W * sw) and (B == 1 or sb == H * W * sw) and (H == 1 or sh == W * sw)
        return sw if ok else None
    ps_s, ps_m = nhwc_stride(scales), nhwc_stride(means)
    if ps_s is not None and ps_s == ps_m and scales.shape[1] == K * M and means.shape[1] == K * M:
        return ps_s, L.ptr(scales), L.ptr(means), (scales, means)
    scales, means = _nhwc(scales.to(y.dtype)), _nhwc(means.to(y.dtype))
    return K * M, L.ptr(scales), L.ptr(means), (scales, means)


def _gmm_f32in(y, scales, means, weights, K, use_means_in_quant, scale_bound, lik_bound, out_dtype):
    """Inference with fp32 latents / parameters and y_hat stored as ``out_dtype`` (``hesic_gmm_forward_f32in``)."""
    L.require_cuda(y, scales, means)
    B, M, H, W = y.shape
    y = _nhwc(y.detach())
    ps, sptr, mptr, keep = _sm_pointers(scales.detach(), means.detach(), y, B, K, M)
    wts = None if weights is None else weights.detach().reshape(B, K * M).to(torch.float32).contiguous()
    yh = _empty_nhwc(B, M, H, W, out_dtype, y.device)
    lik = _empty_nhwc(B, M, H, W, torch.float32, y.device)
    d = L.GmmDesc(B, H * W, M, K, L.F32, int(use_means_in_quant), ps, 0, 0, float(scale_bound), float(lik_bound))
    L.call("hesic_gmm_forward_f32in", C.byref(d), L.ptr(y), sptr, mptr, L.ptr(wts), L.ptr(yh), L.dt(out_dtype), L.ptr(lik), None, L.stream())
    del keep
    return yh, lik


def _use_f32in(y, noise, out_dtype, K, M):
    return (out_dtype is not None and out_dtype != y.dtype and y.dtype == torch.float32 and noise is None and not torch.is_grad_enabled()
            and K in (1, 5) and M % 2 == 0 and y.is_cuda)


def gaussian_mixture(y, scales, means, weights, K, noise=None, scale_bound=0.11, lik_bound=1e-9, out_dtype=None):
    if _use_f32in(y, noise, out_dtype, K, y.shape[1]):
        return _gmm_f32in(y, scales, means, weights, K, False, scale_bound, lik_bound, out_dtype)
    return _apply(_GmmFn, y, scales, means, weights, noise, K, False, scale_bound, lik_bound)


def gaussian_conditional(y, scales, means=None, noise=None, scale_bound=0.11, lik_bound=1e-9, out_dtype=None):
    if means is not None and _use_f32in(y, noise, out_dtype, 1, y.shape[1]):
        return _gmm_f32in(y, scales, means, None, 1, True, scale_bound, lik_bound, out_dtype)
    return _apply(_GmmFn, y, scales, means, None, noise, 1, means is not None, scale_bound, lik_bound)


def gmm_cdf_tables(scales, means, weights, channels, minmax, K, b=0, scale_bound=0.11):
    """Per-element cumulative-frequency tables of the real bit-stream (HSIC.compress / decompress,
    ywz/mywork/newnet1.py:925-978): returns an int32 tensor (len(channels), H, W, 2*minmax+2) on the device whose rows
    are the uint32 tables [0, cumsum(round(clip(pmf)/sum*65536))] of image ``b``."""
    L.require_cuda(scales, means)
    B, KM, H, W = scales.shape
    M = KM // K
    dt = scales.dtype
    scales, means = _nhwc(scales), _nhwc(means.to(dt))
    wts = None if weights is None else weights.detach().reshape(B, K * M).to(torch.float32).contiguous()
    ch = torch.as_tensor(channels, dtype=torch.int32, device=scales.device).contiguous()
    out = torch.empty((ch.numel(), H, W, 2 * int(minmax) + 2), dtype=torch.int32, device=scales.device)
    d = L.GmmDesc(B, H * W, M, K, L.dt(scales), 0, K * M, 0, 0, float(scale_bound), 0.0)
    L.call("hesic_gmm_cdf", C.byref(d), int(b), L.ptr(scales), L.ptr(means), L.ptr(wts), L.ptr(ch), ch.numel(), int(minmax),
           L.ptr(out), L.stream())
    return out


def quantize_symbols(y, means=None):
    """EntropyModel._quantize(x, 'symbols', means) (entropy_models.py:98-125): int32 indices."""
    L.require_cuda(y)
    B, M, H, W = y.shape
    y = _nhwc(y)
    sc = torch.ones_like(y, memory_format=_CL)
    mu = torch.empty_like(y, memory_format=_CL).fill_(0) if means is None else _nhwc(means.to(y.dtype).expand_as(y))
    yh = torch.empty_like(y, memory_format=_CL)
    lik = _empty_nhwc(B, M, H, W, torch.float32, y.device)
    sym = torch.empty((B, M, H, W), dtype=torch.int32, device=y.device).contiguous(memory_format=_CL)
    d = L.GmmDesc(B, H * W, M, 1, L.dt(y), int(means is not None), M, 0, 0, 0.11, 1e-9)
    L.call("hesic_gmm_forward", C.byref(d), L.ptr(y), L.ptr(sc), L.ptr(mu), None, None, L.ptr(yh), L.ptr(lik), L.ptr(sym),
           L.stream())
    return sym


# ----------------------------------------------------------------------------------- glue
class _Upsample4CatFn(torch.autograd.Function):
    """cat(UpsamplingBilinear2d(x4)(z), y1, dim=1) (newnet1.py:524,556-557) in one buffer."""

    @staticmethod
    def forward(ctx, z, y1):
        L.require_cuda(z, y1)
        B, Cz, H, W = z.shape
        _, Cy, Hy, Wy = y1.shape
        if (Hy, Wy) != (4 * H, 4 * W):
            raise RuntimeError("upsample4_cat: y1 must be 4x the size of z")
        z, y1 = _nhwc(z), _nhwc(y1.to(z.dtype))
        out = _empty_nhwc(B, Cz + Cy, Hy, Wy, z.dtype, z.device)
        L.call("hesic_upsample4_forward", L.ptr(z), L.ptr(out), B, H, W, Cz, Cz + Cy, 0, L.dt(z), L.stream())
        L.call("hesic_copy_channels", L.ptr(y1), L.ptr(out), B * Hy * Wy, Cy, Cy, 0, Cz + Cy, Cz, L.dt(z), L.stream())
        ctx.dims = (B, Cz, H, W, Cy)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Cz, H, W, Cy = ctx.dims
        g = _nhwc(g)
        dz = _empty_nhwc(B, Cz, H, W, g.dtype, g.device)
        dy1 = _empty_nhwc(B, Cy, 4 * H, 4 * W, g.dtype, g.device)
        L.call("hesic_upsample4_backward", L.ptr(g), L.ptr(dz), B, H, W, Cz, Cz + Cy, 0, L.dt(g), L.stream())
        L.call("hesic_copy_channels", L.ptr(g), L.ptr(dy1), B * 16 * H * W, Cy, Cz + Cy, Cz, Cy, 0, L.dt(g), L.stream())
        return dz, dy1


def upsample4_cat(z, y1):
    return _apply(_Upsample4CatFn, z, y1)


class _Upsample4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        L.require_cuda(z)
        B, Cz, H, W = z.shape
        z = _nhwc(z)
        out = _empty_nhwc(B, Cz, 4 * H, 4 * W, z.dtype, z.device)
        L.call("hesic_upsample4_forward", L.ptr(z), L.ptr(out), B, H, W, Cz, Cz, 0, L.dt(z), L.stream())
        ctx.dims = (B, Cz, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Cz, H, W = ctx.dims
        g = _nhwc(g)
        dz = _empty_nhwc(B, Cz, H, W, g.dtype, g.device)
        L.call("hesic_upsample4_backward", L.ptr(g), L.ptr(dz), B, H, W, Cz, Cz, 0, L.dt(g), L.stream())
        return dz


def upsample4(z):
    return _apply(_Upsample4Fn, z)


class _SpatialMaxFn(torch.autograd.Function):
    """spatial_pool2d (newnet1.py:441-453) [+ LeakyReLU :497]: (B,C,H,W) -> (B,C,1,1) fp32."""

    @staticmethod
    def forward(ctx, x, leaky):
        L.require_cuda(x)
        B, Cc, H, W = x.shape
        x = _nhwc(x)
        out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        need_arg = ctx.needs_input_grad[0]                          # the arg-max is only needed by the backward
        arg = torch.empty((B, Cc), dtype=torch.int32, device=x.device) if need_arg else None
        L.call("hesic_spatial_max", L.ptr(x), L.ptr(out), L.ptr(arg), B, H * W, Cc, L.dt(x), int(leaky), L.stream())
        ctx.save_for_backward(arg, out)
        ctx.meta = (x.shape, x.dtype, leaky)
        return out.reshape(B, Cc, 1, 1)

    @staticmethod
    def backward(ctx, g):
        arg, out = ctx.saved_tensors
        shape, dtype, leaky = ctx.meta
        B, Cc, H, W = shape
        g = g.reshape(B, Cc)
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        dx = torch.empty((B, H * W, Cc), dtype=dtype, device=g.device)
        L.call("hesic_spatial_max_backward", L.ptr(g), L.ptr(out), L.ptr(arg), L.ptr(dx), B, H * W, Cc, L.dt(dtype), int(leaky), L.stream())
        return dx.reshape(B, H, W, Cc).permute(0, 3, 1, 2), None


def spatial_max(x, leaky=False):
    return _apply(_SpatialMaxFn, x, leaky)


class _SoftmaxKFn(torch.autograd.Function):
    """softmax over K of a (B, K*M, 1, 1) tensor with channel k*M+m (newnet1.py:510-512)."""

    @staticmethod
    def forward(ctx, logits, K, M):
        L.require_cuda(logits)
        B = logits.shape[0]
        lg = logits.reshape(B, K * M).to(torch.float32).contiguous()
        w = torch.empty_like(lg)
        L.call("hesic_softmax_k_forward", L.ptr(lg), L.ptr(w), B, K, M, L.stream())
        ctx.save_for_backward(w)
        ctx.meta = (K, M, logits.dtype, logits.shape)
        return w.reshape(B, K * M, 1, 1)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        K, M, dtype, shape = ctx.meta
        B = w.shape[0]
        g = g.reshape(B, K * M).to(torch.float32).contiguous()
        dl = torch.empty_like(w)
        L.call("hesic_softmax_k_backward", L.ptr(w), L.ptr(g), L.ptr(dl), B, K, M, L.stream())
        return dl.reshape(shape).to(dtype), None, None


def softmax_k(logits, K, M):
    return _apply(_SoftmaxKFn, logits, K, M)


class _PooledLinearFn(torch.autograd.Function):
    """conv1x1(N -> N) on a pooled (B, N, 1, 1) vector (newnet1.py:500) under autograd: three small launches instead of
    the conv tile pipeline (forward + dgrad + wgrad + colsum + reduce on 8 "pixels" cost ~0.4 ms per training step)."""

    @staticmethod
    def forward(ctx, pooled, weight, bias):
        L.require_cuda(pooled, weight)
        B, N = pooled.shape[0], pooled.shape[1]
        p = pooled.reshape(B, N).to(torch.float32).contiguous()
        w = weight.detach().reshape(N, N).to(torch.float32).contiguous()
        out = torch.empty((B, N), dtype=torch.float32, device=p.device)
        L.call("hesic_pooled_linear_forward", L.ptr(p), L.ptr(w), L.ptr(None if bias is None else bias.detach().float()), L.ptr(out),
               B, N, L.stream())
        ctx.save_for_backward(p, w)
        ctx.weight, ctx.bias = weight, bias
        ctx.meta = (pooled.dtype, weight.dtype, weight.shape, bias is not None, None if bias is None else bias.dtype)
        return out.reshape(B, N, 1, 1)

    @staticmethod
    def backward(ctx, g):
        p, w = ctx.saved_tensors
        pdt, wdt, wshape, has_b, bdt = ctx.meta
        B, N = p.shape
        g = g.reshape(B, N).to(torch.float32).contiguous()
        need_p, need_w, need_b = ctx.needs_input_grad
        dp = torch.empty_like(p) if need_p else None
        # flat gradient slots: the kernel overwrites, so it writes the (cleared, still untouched) slots themselves
        ws_, bs_ = _slot_for(ctx.weight), (_slot_for(ctx.bias) if has_b else None)
        direct = need_w and ws_ is not None and ws_.writes == 0 and (not has_b or (bs_ is not None and bs_.writes == 0))
        if direct:
            dw, db = ws_.grad, (bs_.grad if has_b else None)
        else:
            dw = torch.empty((N, N), dtype=torch.float32, device=p.device) if (need_w or (need_b and has_b)) else None
            db = torch.empty(N, dtype=torch.float32, device=p.device) if (need_b and has_b) else None
        L.call("hesic_pooled_linear_backward", L.ptr(p), L.ptr(w), L.ptr(g), L.ptr(dp), L.ptr(dw), L.ptr(db), B, N, L.stream())
        if direct:
            _slot_done(ws_)
            if has_b:
                _slot_done(bs_)
            return (None if dp is None else dp.reshape(B, N, 1, 1).to(pdt), None, None)
        return (None if dp is None else dp.reshape(B, N, 1, 1).to(pdt), dw.reshape(wshape).to(wdt) if need_w else None,
                None if db is None else db.to(bdt))


def pooled_linear(pooled, weight, bias):
    return _apply(_PooledLinearFn, pooled, weight, bias)


def mix_weights(pooled, weight, bias, K, M):
    """conv1x1(K*M -> K*M) on the pooled vector + softmax over K (newnet1.py:500,510-512).
    Forward-only HIP kernel; under autograd the caller uses the differentiable torch fallback
    below because the op is 0.001 GMAC."""
    L.require_cuda(pooled, weight)
    B = pooled.shape[0]
    p = pooled.reshape(B, K * M).to(torch.float32).contiguous()
    w = weight.detach().reshape(K * M, K * M).to(torch.float32).contiguous()
    logits = torch.empty((B, K * M), dtype=torch.float32, device=p.device)
    out = torch.empty((B, K * M), dtype=torch.float32, device=p.device)
    L.call("hesic_mix_weights_forward", L.ptr(p), L.ptr(w), L.ptr(None if bias is None else bias.detach()), L.ptr(logits),
           L.ptr(out), B, K, M, L.stream())
    return out.reshape(B, K * M, 1, 1)


# ----------------------------------------------------------------------------- reductions
RD_SUMS_FUSED = True       # module switch: False = one launch per likelihood map / image pair (rounds 1-4)


def sum_log2(lik, out=None):
    """sum(log2(lik)) as an fp64 device scalar (bits = -sum)."""
    L.require_cuda(lik)
    lik = lik.contiguous() if not lik.is_contiguous(memory_format=_CL) else lik
    if out is None:
        out = _zeros(1, torch.float64, lik.device)
    L.call("hesic_sum_log2", L.ptr(lik), lik.numel(), L.ptr(out), L.stream())
    return out


def sum_sq_diff(a, b, out=None):
    L.require_cuda(a, b)
    B, Cc, H, W = a.shape
    if out is None:
        out = _zeros(1, torch.float64, a.device)
    sa = (C.c_int64 * 4)(*a.stride())
    sb = (C.c_int64 * 4)(*b.stride())
    L.call("hesic_sum_sq_diff", L.ptr(a), L.dt(a), sa, L.ptr(b), L.dt(b), sb, B, Cc, H, W, L.ptr(out), L.stream())
    return out


def rd_sums(liks, lik_outs, pairs, sq_outs):
    """sum(log2(lik_i)) into ``lik_outs[i]`` and sum((a - b)^2) of the image pairs into ``sq_outs[i]`` -- every reduction behind one forward's
    bpp / PSNR in ONE launch (``hesic_rd_sums``; four + two launches before).  The accumulators are added to (zero-fill them first)."""
    L.require_cuda(*liks, *[t for p in pairs for t in p])
    liks = [l if (l.is_contiguous() or l.is_contiguous(memory_format=_CL)) else l.contiguous() for l in liks]
    nl, ns = len(liks), len(pairs)
    vpl, i64l = C.c_void_p * max(nl, 1), C.c_int64 * max(nl, 1)
    vps, i32s, i64s, i32d = C.c_void_p * max(ns, 1), C.c_int32 * max(ns, 1), C.c_int64 * (4 * max(ns, 1)), C.c_int32 * (4 * max(ns, 1))
    sa, sb, dims = [], [], []
    for a, b in pairs:
        sa += list(a.stride()); sb += list(b.stride()); dims += list(a.shape)
    L.call("hesic_rd_sums", nl, vpl(*[l.data_ptr() for l in liks]), i64l(*[l.numel() for l in liks]), vpl(*[o.data_ptr() for o in lik_outs]),
           ns, vps(*[a.data_ptr() for a, _ in pairs]), i32s(*[L.dt(a) for a, _ in pairs]), i64s(*sa), vps(*[b.data_ptr() for _, b in pairs]),
           i32s(*[L.dt(b) for _, b in pairs]), i64s(*sb), i32d(*dims), vps(*[o.data_ptr() for o in sq_outs]), L.stream())


class _RdLossFn(torch.autograd.Function):
    """RateDistortionLoss (ywz/mywork/newtrain1.py:37-56) as two kinds of HIP reductions:
    loss = lmbda*255^2*(MSE1+MSE2) + sum_t sum(log lik_t)/(-ln2 * B*H*W).  Returns (loss, bpp, mse)."""

    @staticmethod
    def forward(ctx, lmbda, x1, x2, x1_hat, x2_hat, *liks):
        import math
        B, Cc, H, W = x1.shape
        npix = B * H * W
        acc = _zeros(3, torch.float64, x1.device)
        liks = tuple(l if (l.is_contiguous() or l.is_contiguous(memory_format=_CL)) else l.contiguous() for l in liks)
        if x1.is_cuda and RD_SUMS_FUSED and len(liks) <= 8:
            rd_sums(liks, [acc[0:1]] * len(liks), [(x1_hat, x1), (x2_hat, x2)], [acc[1:2], acc[2:3]])
        else:
            for l in liks:
                sum_log2(l, acc[0:1])
            sum_sq_diff(x1_hat, x1, acc[1:2])
            sum_sq_diff(x2_hat, x2, acc[2:3])
        if x1.is_cuda:
            out3 = torch.empty(3, dtype=torch.float32, device=x1.device)
            L.call("hesic_rd_loss_combine", L.ptr(acc), float(lmbda) * 255.0 ** 2, npix, B * Cc * H * W, L.ptr(out3), L.stream())
            loss, bpp, mse = out3[0], out3[1], out3[2]
        else:
            bpp = -acc[0] / npix
            mse = (acc[1] + acc[2]) / (B * Cc * H * W)
            loss = (lmbda * 255.0 ** 2 * mse + bpp).float()
            bpp, mse = bpp.float(), mse.float()
        ctx.save_for_backward(x1, x2, x1_hat, x2_hat, *liks)
        ctx.meta = (float(lmbda), npix, B * Cc * H * W, -1.0 / (math.log(2.0) * npix))
        ctx.mark_non_differentiable(bpp, mse)        # reported values (the reference logs them); only `loss` carries a gradient
        return loss, bpp, mse

    @staticmethod
    def backward(ctx, g_loss, g_bpp, g_mse):
        x1, x2, x1_hat, x2_hat, *liks = ctx.saved_tensors
        lmbda, npix, numel, lik_scale = ctx.meta
        # g_loss stays on the device (reading it would be a host sync): the kernels compute the gradient of the unscaled loss
        # and, unless SCALED_LOSS is off, one multiply per gradient applies g_loss (loss / accum_steps, loss scaling, ...).
        # Trainer.step calls loss.backward() on the unscaled root and switches the multiplies off.
        g = 1.0
        B, Cc, H, W = x1.shape
        grads = []
        for xh, x in ((x1_hat, x1), (x2_hat, x2)):
            gx = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
            sa, sb = (C.c_int64 * 4)(*xh.stride()), (C.c_int64 * 4)(*x.stride())
            L.call("hesic_sq_diff_backward", L.ptr(xh), L.dt(xh), sa, L.ptr(x), L.dt(x), sb, B, Cc, H, W,
                   g * lmbda * 255.0 ** 2 * 2.0 / numel, L.ptr(gx), L.stream())
            grads.append(gx)
        gl = []
        for l in liks:
            o = torch.empty_like(l)
            L.call("hesic_log_backward", L.ptr(l), g * lik_scale, L.ptr(o), l.numel(), L.stream())
            gl.append(o)
        if SCALED_LOSS:
            grads = [t.mul_(g_loss) for t in grads]
            gl = [t.mul_(g_loss) for t in gl]
        return (None, None, None, grads[0], grads[1], *gl)


SCALED_LOSS = True      # False (Trainer.step): loss.backward() is called on the unscaled loss, skip the six g_loss multiplies


def rd_loss(out, x1, x2, lmbda):
    """dict(loss, bpp_loss, mse_loss) like the reference's criterion; differentiable through ``loss``."""
    lk = out["likelihoods"]
    loss, bpp, mse = _RdLossFn.apply(lmbda, x1, x2, out["x1_hat"], out["x2_hat"], lk["y1"], lk["y2"], lk["z1"], lk["z2"])
    return {"loss": loss, "bpp_loss": bpp.detach(), "mse_loss": mse.detach()}


def set_phase_fusion(mode):
    """How transposed stride-2 layers run on the implicit-GEMM kernels (``hesic_conv2d_set_phase_fusion``): 0 = one block per (tile,
    output phase), 1 = auto (the four phases of a tile in one block when the launch fills the chip that way; default), 2 = whenever the
    shape is eligible.  Results are bit-identical in every mode.  Returns the previous mode."""
    prev = int(L.lib().hesic_conv2d_set_phase_fusion(int(mode)))
    if prev < 0:
        raise ValueError(f"set_phase_fusion: bad mode {mode}: {L.lib().hesic_last_error().decode()}")
    return prev
