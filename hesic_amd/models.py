"""HESIC (``HSIC``) and HESIC+ (``HSICJoint``): the build's counterparts of the reference model files
``ywz/mywork/newnet1.py`` and ``ywz/mywork/newnet1_joint.py``.

Same module tree, parameter names and shapes (reference checkpoints load with ``strict=True``), same
forward semantics including its quirks, but the forward is written against the fused
operators of ``hesic_amd.functional``: activations and ``abs`` ride in the conv kernels, the
``Upsample x4 + cat`` is one buffer, the Python double loop of ``spatial_pool2d`` is one reduction, the
1x1 conv + softmax head is one kernel, and the duplicated ``warp(x1_hat)`` (:753 == :767) runs once.
"""
import math

import torch
import torch.nn as nn

from compressai.entropy_models import EntropyBottleneck, GaussianConditional, GaussianMixtureConditional
from compressai.layers import GDN, MaskedConv2d, ResidualBlock, conv3x3
from compressai.models.utils import HipConv2d, conv, deconv

from . import _lib as L
from . import functional as Fn
from . import handover as _ho
from .geometry import warp_perspective

RELU, LEAKY, NONE = L.ACT_RELU, L.ACT_LEAKY, L.ACT_NONE

# inference runs the two views' independent front ends on two HIP streams (set False for a single-stream schedule)
import os as _os
import threading as _threading
OVERLAP_STREAMS = _os.environ.get("HESIC_NO_OVERLAP") is None
CAT_FREE_EP = True          # module switch (HESIC+): False = torch.cat in front of entropy_parameters
_side_streams = {}


_RECORD = True


def _rec(t, stream):
    if _RECORD:
        t.record_stream(stream)


def _side_stream(device, idx=0):
    """Side stream ``idx`` of the schedule: a fixed small set per device.  torch hands out streams from a pool of 32 per device
    round-robin, so a NEW stream object may alias one that already exists (another side stream or the caller's current / capture
    stream); a fork onto an alias is a stream waiting for itself or a nested fork (which crashes hipStreamEndCapture) -- checked here."""
    key = (device.type, device.index, idx)
    cur = torch.cuda.current_stream(device).cuda_stream
    taken = {cur} | {st.cuda_stream for k, st in _side_streams.items() if k[:2] == key[:2] and k != key}
    st = _side_streams.get(key)
    tries = 0
    while st is None or st.cuda_stream in taken:
        st = torch.cuda.Stream(device=device)
        tries += 1
        if tries > 64:
            raise RuntimeError("no free HIP stream for the inference schedule")
    _side_streams[key] = st
    return st


def metrics_stream(device):
    """A stream for work that follows a forward without holding the main stream (``bench.py`` reduces bits / squared error on it):
    the schedule's view-1 rate stream, idle from the middle of a forward to ~0.3 ms into the next one -- a fifth stream would share
    one of the runtime's four hardware queues with the schedule."""
    return _side_stream(device, 12)


def hand_over(tensors, stream):
    """Tell the caching allocator that ``tensors`` (an iterable) are also in use on ``stream`` (``Tensor.record_stream``); with
    HESIC_NO_RECORD_STREAM set -- an UNSAFE debugging switch: a side-stream tensor can then be reused while a kernel on another
    stream still reads it -- nothing is recorded."""
    for t in tensors:
        _rec(t, stream)


# ---- cross-stream synchronisation of the inference schedule.  Every event record / wait of the schedule goes through these three
# helpers: issued eagerly they are the torch calls; while a ``SegmentedForward`` is being built they cut the streams' work into
# single-stream SEGMENTS (one HIP graph each) and write the events into its replay plan (see SegmentedForward).
_seg_tls = _threading.local()         # .rec: the recorder of a SegmentedForward build on THIS thread (another thread's forward stays eager)


def _seg_rec_get():
    return getattr(_seg_tls, "rec", None)


def _ev_record(stream):
    rec = _seg_rec_get()
    if rec is not None:
        return rec.record(stream)
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


def _wait_event(stream, ev):
    rec = _seg_rec_get()
    if rec is not None:
        return rec.wait(stream, ev)
    stream.wait_event(ev)


def _wait_stream(stream, other):
    rec = _seg_rec_get()
    if rec is not None:
        return rec.wait(stream, rec.record(other))
    stream.wait_stream(other)


class _SegmentRecorder:
    """Build pass of ``SegmentedForward``: a stream's launches between two synchronisation points are captured into ONE graph (a
    single chain of kernel nodes: the runtime has no branches to order), the synchronisation points themselves become ``record`` /
    ``wait`` entries of the replay plan.  A segment's launch entry stands where the segment was OPENED, so the plan keeps the eager
    issue order of the schedule."""

    def __init__(self):
        self.plan, self.open, self.touched, self.empty = [], {}, set(), []

    def _close(self, st):
        ent = self.open.pop(st.cuda_stream, None)
        if ent is not None:
            _, g, holder = ent
            import warnings
            with torch.cuda.stream(st), warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                g.capture_end()
            # Whether the segment holds work is read off the capture itself: torch warns "The CUDA Graph is empty" when capture_end found no
            # node (and builds no executable graph).  The count of ``L.call`` launches (``touched``) alone would drop a segment made only of
            # torch-native kernels (copy_, cat, .float()) and leave its outputs uninitialised; it stays as a cross-check.
            empty = any("Graph is empty" in str(w.message) for w in caught)
            if empty and st.cuda_stream in self.touched:
                raise RuntimeError("SegmentedForward: a segment with recorded launches captured as an empty graph")
            if not empty:
                holder.append(g)
            else:
                self.empty.append(g)                    # destroyed after the build: a graph must not be released while another stream captures
            self.touched.discard(st.cuda_stream)

    def _open(self, st):
        if st.cuda_stream not in self.open:
            g, holder = torch.cuda.CUDAGraph(), []
            with torch.cuda.stream(st):
                g.capture_begin(capture_error_mode="thread_local")
            self.open[st.cuda_stream] = (st, g, holder)
            self.plan.append(("launch", st, holder))

    def touch(self):
        self.touched.add(torch.cuda.current_stream().cuda_stream)

    def record(self, st):
        self._close(st)
        ev = torch.cuda.Event()
        self.plan.append(("record", st, ev))
        self._open(st)
        return ev

    def wait(self, st, ev):
        self._close(st)
        self.plan.append(("wait", st, ev))
        self._open(st)

    def finish(self):
        for st, _, _ in list(self.open.values()):
            self._close(st)
        self.empty.clear()
        return [e for e in self.plan if e[0] != "launch" or e[2]]


def _branches(ref, *fns):
    """Run independent branches; at inference each extra branch gets its own HIP stream (forked from / joined to the
    current one) so their small kernels overlap.  Returns the branch results in order."""
    if not (OVERLAP_STREAMS and ref.is_cuda and not torch.is_grad_enabled()) or _Fork.depth:
        return [f() for f in fns]          # inside a fork: in order (a fork joined back into a forked stream crashes hipStreamEndCapture, ROCm 7.2)
    cur = torch.cuda.current_stream()
    outs = [None] * len(fns)
    streams = [_side_stream(ref.device, 1 + i) for i in range(len(fns) - 1)]
    for st in streams:
        st.wait_stream(cur)
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            outs[i + 1] = fns[i + 1]()
    outs[0] = fns[0]()
    for i, st in enumerate(streams):
        cur.wait_stream(st)
        o = outs[i + 1]
        for t in (o if isinstance(o, (tuple, list)) else (o,)):
            _rec(t, cur)
    return outs


def _tensors(o):
    if torch.is_tensor(o):
        yield o
    elif isinstance(o, (tuple, list)):
        for t in o:
            yield from _tensors(t)


class _Fork:
    """One branch of the inference schedule on its own HIP stream: starts where the forking stream stands (plus the branches it
    needs), ``join`` makes a stream wait for it and hands its tensors over to that stream's allocator bookkeeping."""

    depth = 0

    def __init__(self, stream, fn, after=(), start=None):
        self.stream = stream
        if start is not None:
            _wait_event(stream, start)                  # fork from an EARLIER point of the forking stream
        else:
            _wait_stream(stream, torch.cuda.current_stream())
        for f in after:
            _wait_stream(stream, f.stream)
        with torch.cuda.stream(stream):
            for f in after:
                for t in _tensors(f.out):
                    _rec(t, stream)
            _Fork.depth += 1
            try:
                self.out = fn()
            finally:
                _Fork.depth -= 1

    def then(self, fn, start=None):
        """More work on the same stream, after everything the forking stream has issued so far (or up to the event ``start``)."""
        if start is not None:
            _wait_event(self.stream, start)
        else:
            _wait_stream(self.stream, torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            _Fork.depth += 1
            try:
                out = fn()
            finally:
                _Fork.depth -= 1
        self.out = (self.out, out)
        return self

    def join(self, stream=None):
        stream = stream or torch.cuda.current_stream()
        _wait_stream(stream, self.stream)
        for t in _tensors(self.out):
            _rec(t, stream)
        return self.out


class StereoCompressionModel(nn.Module):
    """Base with TWO entropy bottlenecks (reference newnet1.py:36-104): ``parameters()`` skips them,
    ``aux_parameters()`` yields only them; conv weights get kaiming init only if built before this ctor
    runs (they are not, as in the reference, so PyTorch default init stays -- SURVEY.md 8a)."""

    def __init__(self, entropy_bottleneck_channels, init_weights=True):
        super().__init__()
        self.entropy_bottleneck1 = EntropyBottleneck(entropy_bottleneck_channels)
        self.entropy_bottleneck2 = EntropyBottleneck(entropy_bottleneck_channels)
        if init_weights:
            self._initialize_weights()

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def parameters(self, recurse=True):
        for m in self.children():
            if not isinstance(m, EntropyBottleneck):
                yield from m.parameters()

    def aux_parameters(self):
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                yield from m.parameters()

    def update(self, force=False):
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                m.update(force=force)

    _warned_warp = False

    def load_state_dict(self, state_dict, *args, **kwargs):
        """``nn.Module.load_state_dict`` + one note, once per process, while no warp convention has been chosen: a checkpoint carries
        no trace of the kornia release it was trained with, and the two conventions differ (``geometry.use_reference_era_warp``)."""
        from . import geometry
        if not geometry._CONVENTION_CHOSEN and not StereoCompressionModel._warned_warp:
            import warnings
            StereoCompressionModel._warned_warp = True
            warnings.warn("hesic_amd: loading a checkpoint with the default warp convention " + geometry.warp_convention() + ". Checkpoints "
                          "trained in the reference's pinned environment (torch 1.6 / kornia 0.4.x) expect align_corners=False: call "
                          "hesic_amd.geometry.use_reference_era_warp() (or set HESIC_WARP_ALIGN_CORNERS=0) for them; "
                          "use_reference_era_warp(False) keeps the default and silences this note.", stacklevel=2)
        return super().load_state_dict(state_dict, *args, **kwargs)


# ------------------------------------------------------------------------------ analysis / synthesis
class Encoder1(nn.Module):
    """g_a: conv5s2 -> GDN x3 -> conv5s2 (newnet1.py:580-601)."""

    def __init__(self, N, M):
        super().__init__()
        self.g_a_conv1, self.g_a_gdn1 = conv(3, N), GDN(N)
        self.g_a_conv2, self.g_a_gdn2 = conv(N, N), GDN(N)
        self.g_a_conv3, self.g_a_gdn3 = conv(N, N), GDN(N)
        self.g_a_conv4 = conv(N, M)
        self._mark_shaped()

    def _mark_shaped(self):
        # single-operand inference launches of these layers (the third analysis pass, g_a_conv2 in the "x3c2" mode) take weights rounded
        # with error feedback over the taps (Fn.PackedWeight(shaped=True)): their inputs are spatially smooth GDN outputs
        for c in (self.g_a_conv2, self.g_a_conv3, self.g_a_conv4):
            c.shaped_weights = True

    def trunk(self, x):
        x = self.g_a_conv1.run_gdn(x, self.g_a_gdn1)        # conv + GDN in one kernel at inference
        x = self.g_a_conv2.run_gdn(x, self.g_a_gdn2)
        return self.g_a_conv3.run_gdn(x, self.g_a_gdn3)

    def stack(self, x):
        return self.g_a_conv4(self.trunk(x))

    def forward(self, x):
        return self.stack(x)

    def latent(self, x, want_lo=True, exact=False, lo_abs=False):
        """(lo, hi): the latent in the storage dtype for the hyper-analysis convs (None unless ``want_lo``) and as it feeds
        round() / the likelihood (fp32 from the accumulators at bf16 inference, see ``Fn.conv2d_latent``).  ``exact`` (the passes
        whose rounded output the reference transmits: y1 and y2) takes the hi/lo route in the bf16x3 analysis mode: ``hi`` then has
        fp32-grade accuracy and ``lo`` is a ``Fn.HiLo`` map of y (of |y| with ``lo_abs``: HESIC's hyper-analysis reads |y|)."""
        if exact and Fn.analysis_hilo(x) and self._hilo_ok():
            return self.latent_hilo(x, want_lo, lo_abs)
        return self.g_a_conv4.run_latent(self.trunk(x), want_lo=want_lo)

    def _hilo_ok(self):
        c1, c4 = self.g_a_conv1, self.g_a_conv4
        return (tuple(c1.weight.shape[1:]) == (3, 5, 5) and c1.stride[0] == 2 and c1.weight.shape[0] == 128 and c4.weight.shape[1] == 128
                and c4.weight.shape[0] % 8 == 0)

    def latent_hilo(self, x, want_lo=True, lo_abs=False):
        """g_a on hi/lo bf16 pairs (``Fn`` bf16x3 section): im2col of the image -> 1x1 implicit GEMM + GDN -> two 5x5 stride-2
        layers + GDN -> the 5x5 stride-2 output layer, written as fp32 from the accumulators (newnet1.py:590-601 in ~fp32 accuracy)."""
        c1, g1 = self.g_a_conv1, self.g_a_gdn1
        if not hasattr(self, "_hl1"):
            self._hl1 = Fn.PackedWeightHiLo(), Fn.PackedGdnLo(), Fn.PackedN2wHiLo()
        gp, bp = g1.packer().get(g1.beta, g1.gamma, g1.beta_min)
        fused1 = Fn.sconv_gdn_hilo_ok(x, c1.weight)
        if fused1 and Fn.analysis_conv2_single() and self.g_a_conv2.weight.shape[:2] == (128, 128):
            # "x3c2": conv1 + GDN on pairs inside the kernel, ONE 16-bit value per channel out; g_a_conv2 multiplies single operands,
            # its GDN runs on pairs again and hands pairs to g_a_conv3
            t = Fn.sconv_gdn_hilo(x, self._hl1[2].get(c1.weight, g1.gamma, out1=True), c1.bias, bp, g1.inverse, out1=True)
            t = self.g_a_conv2.run_gdn_hilo_out(t, self.g_a_gdn2)
            np_ = 2 if X3C2_TWO_PRODUCT_TAIL else 3          # g_a_conv3 / conv4: pairs x single error-feedback weights (two products per pair)
            t = self.g_a_conv3.run_hilo(t, gdn=self.g_a_gdn3, products=np_)
            if not want_lo:
                return None, self.g_a_conv4.run_hilo(t, out="f32", products=np_)
            lo, y = self.g_a_conv4.run_hilo(t, out="both", out_abs=lo_abs, products=np_)
            return Fn.HiLo((lo, self.g_a_conv4.weight.shape[0])), y
        if fused1:
            t = Fn.sconv_gdn_hilo(x, self._hl1[2].get(c1.weight, g1.gamma), c1.bias, bp, g1.inverse)       # conv + GDN in one kernel
            if Fn.analysis_precision() == "x2":
                # pairs everywhere, TWO products per MAC: x pairs x single error-feedback weights (no w_lo term)
                p2, p3, p4 = X2_LAYERS
                t = self.g_a_conv2.run_hilo(t, gdn=self.g_a_gdn2, products=p2)
                t = self.g_a_conv3.run_hilo(t, gdn=self.g_a_gdn3, products=p3)
                if not want_lo:
                    return None, self.g_a_conv4.run_hilo(t, out="f32", products=p4)
                lo, y = self.g_a_conv4.run_hilo(t, out="both", out_abs=lo_abs, products=p4)
                return Fn.HiLo((lo, self.g_a_conv4.weight.shape[0])), y
        else:
            KP = 96                                        # other layouts: 3 * 25 = 75 im2col columns (padded) -> 1x1 implicit GEMM + GDN
            t = Fn.im2col_hilo(x, 5, 2, 2, KP)
            t = Fn.conv2d_hilo(t, self._hl1[0].get(c1.weight, as_1x1=True, kp=KP), c1.bias, KP, 128, kernel_size=1, stride=1, padding=0,
                               gdn=(gp, self._hl1[1].get(g1.gamma), bp, g1.inverse))
        t = self.g_a_conv2.run_hilo(t, gdn=self.g_a_gdn2)
        t = self.g_a_conv3.run_hilo(t, gdn=self.g_a_gdn3)
        if not want_lo:
            return None, self.g_a_conv4.run_hilo(t, out="f32")
        lo, y = self.g_a_conv4.run_hilo(t, out="both", out_abs=lo_abs)
        return Fn.HiLo((lo, self.g_a_conv4.weight.shape[0])), y


class Encoder2(Encoder1):
    """cat(x1_warp, x2) -> conv5s1(6->3) -> GDN(3) -> g_a (newnet1.py:626-655)."""

    def __init__(self, N, M):
        nn.Module.__init__(self)
        self.pre_conv = conv(6, 3, stride=1)
        self.pre_gdn = GDN(3)
        self.g_a_conv1, self.g_a_gdn1 = conv(3, N), GDN(N)
        self.g_a_conv2, self.g_a_gdn2 = conv(N, N), GDN(N)
        self.g_a_conv3, self.g_a_gdn3 = conv(N, N), GDN(N)
        self.g_a_conv4 = conv(N, M)
        self._mark_shaped()

    def forward(self, x1_warp, x2):
        t = self.pre_conv.run_cat(x1_warp, x2, gdn=self.pre_gdn)
        return self.stack(t)

    def latent(self, x1_warp, x2, want_lo=True, exact=False, lo_abs=False):
        t = self.pre_conv.run_cat(x1_warp, x2, gdn=self.pre_gdn)
        if exact and Fn.analysis_hilo(t) and self._hilo_ok():
            return self.latent_hilo(t, want_lo, lo_abs)
        return self.g_a_conv4.run_latent(self.trunk(t), want_lo=want_lo)


class Decoder1(nn.Module):
    """g_s: deconv5s2 -> IGDN x3 -> deconv5s2 (newnet1.py:603-624)."""

    def __init__(self, N, M):
        super().__init__()
        self.g_s_conv1, self.g_s_gdn1 = deconv(M, N), GDN(N, inverse=True)
        self.g_s_conv2, self.g_s_gdn2 = deconv(N, N), GDN(N, inverse=True)
        self.g_s_conv3, self.g_s_gdn3 = deconv(N, N), GDN(N, inverse=True)
        self.g_s_conv4 = deconv(N, 3)
        # round 5: at 16-bit inference the weights of the layers whose input is a spatially smooth IGDN output are rounded with error feedback
        # inside each output phase's tap class (Fn.PackedWeight(shaped=True, tr_stride=2); g_s_conv4's LDS panel likewise, Fn._weight_image) --
        # their plain rounding was the default mode's whole residual PSNR deviation at a trained point.  g_s_conv1 reads the integer latents
        # (not smooth: error feedback would cost sqrt(2) there) and keeps plain rounding.
        for c in (self.g_s_conv2, self.g_s_conv3):
            c.shaped_weights = True

    def stack(self, y):
        y = self.g_s_conv1.run_gdn(y, self.g_s_gdn1)        # deconv + IGDN in one kernel at inference
        y = self.g_s_conv2.run_gdn(y, self.g_s_gdn2)
        y = self.g_s_conv3.run_gdn(y, self.g_s_gdn3)
        return self.g_s_conv4(y)

    def forward(self, y_hat):
        return self.stack(y_hat)


class Decoder2(Decoder1):
    """g_s -> IGDN(3) -> cat(., x1_hat_warp) -> deconv5s1(6->3) (newnet1.py:657-692)."""

    def __init__(self, N, M):
        super().__init__(N, M)
        self.after_gdn = GDN(3, inverse=True)
        self.after_conv = deconv(6, 3, stride=1)

    def forward(self, y_hat, x1_hat_warp):
        return self.after_conv.run_cat(self.stack(y_hat), x1_hat_warp, gdn=self.after_gdn, gdn_on_input=True)


# ----------------------------------------------------------------------------------- hyper networks
class encode_hyper(nn.Module):
    """|y| -> conv5s1+ReLU -> conv5s2+ReLU -> conv5s2 (newnet1.py:420-437); abs and ReLUs are fused."""

    def __init__(self, N, M):
        super().__init__()
        self.encode_hyper = nn.Sequential(conv(M, N, kernel_size=5, stride=1), nn.ReLU(), conv(N, N, kernel_size=5),
                                          nn.ReLU(), conv(N, N, kernel_size=5))

    def forward(self, y):
        s = self.encode_hyper
        t = s[0].run(y, act=RELU, in_abs=True)
        t = s[2].run(t, act=RELU)
        return s[4].run(t)

    def latent(self, y):
        """z as it feeds the bottleneck (fp32 at bf16 inference).  A ``Fn.HiLo`` map of |y| (the bf16x3 analysis route) runs the three
        layers on hi/lo pairs: z then matches the reference's fp32 hyper-analysis like y does."""
        s = self.encode_hyper
        if isinstance(y, Fn.HiLo):
            t = s[0].run_hilo(y.t, act=RELU)
            t = s[2].run_hilo(t, act=RELU)
            return s[4].run_hilo(t, out="f32")
        t = s[0].run(y, act=RELU, in_abs=True)
        t = s[2].run(t, act=RELU)
        return s[4].run_latent(t, want_lo=False)[1]


class spatial_pool2d(nn.Module):
    """Global spatial max per (sample, channel) (newnet1.py:441-453) as one reduction kernel."""

    def forward(self, X):
        return Fn.spatial_max(_ho.plain(X), leaky=False)


def _mixture_weights(head, feat, K, M):
    """spatial max -> LeakyReLU -> conv1x1 -> softmax over K (newnet1.py:496-512), channel = k*M+m."""
    pooled = Fn.spatial_max(feat, leaky=True)
    c1 = head[5]
    if torch.is_grad_enabled() and (pooled.requires_grad or c1.weight.requires_grad):
        if pooled.is_cuda and c1.weight.shape[-1] == 1 and c1.weight.shape[0] == c1.weight.shape[1] == pooled.shape[1]:
            return Fn.softmax_k(Fn.pooled_linear(pooled, c1.weight, c1.bias), K, M)
        return Fn.softmax_k(c1.run(pooled), K, M)
    return Fn.mix_weights(pooled, c1.weight, c1.bias, K, M)


class gmm_hyper_y1(nn.Module):
    """z1_hat -> (sigma, means, weights), each K*M channels at 4x the resolution (newnet1.py:456-514)."""

    def __init__(self, N, M, K):
        super().__init__()
        self.N, self.M, self.K = N, M, K
        self.gmm_sigma = nn.Sequential(deconv(N, N, kernel_size=5), nn.ReLU(), deconv(N, N, kernel_size=5), nn.ReLU(),
                                       conv(N, M * K, kernel_size=5, stride=1), nn.ReLU())
        self.gmm_means = nn.Sequential(deconv(N, N, kernel_size=5), nn.LeakyReLU(), deconv(N, N, kernel_size=5),
                                       nn.LeakyReLU(), conv(N, M * K, kernel_size=5, stride=1))
        self.gmm_weights = nn.Sequential(deconv(N, N, kernel_size=5), nn.LeakyReLU(), deconv(N, M * K, kernel_size=5),
                                         spatial_pool2d(), nn.LeakyReLU(), conv(M * K, M * K, kernel_size=1, stride=1))

    def _forward_grouped(self, z):
        """bf16 inference: 4 implicit-GEMM launches instead of 8 (+ their split-K reduces).  The three first layers share their
        input -> one transposed conv with Cout = 3 x 128 (ReLU on the sigma third, LeakyReLU on the rest); the sigma / mean second
        layers -> one launch of two groups; the two 128 -> 960 output convs -> one launch of two groups writing sigma | means as
        one fp32 tensor straight from the accumulators.  The weights branch's 128 -> 960 layer runs beside them on a side stream."""
        s, m, w = self.gmm_sigma, self.gmm_means, self.gmm_weights
        if not hasattr(self, "_pg"):
            self._pg = [Fn.PackedGroup() for _ in range(3)]
        t1, o1 = Fn.conv2d_grouped(z, [s[0].weight, m[0].weight, w[0].weight], [s[0].bias, m[0].bias, w[0].bias], self._pg[0], kernel_size=5,
                                   stride=2, padding=2, transposed=True, shared_input=True, acts=[RELU, LEAKY, LEAKY])

        def sigma_means():
            t2, o2 = Fn.conv2d_grouped(t1, [s[2].weight, m[2].weight], [s[2].bias, m[2].bias], self._pg[1], kernel_size=5, stride=2, padding=2,
                                       transposed=True, shared_input=False, x_c_off=o1[0], x_group_step=o1[1] - o1[0], acts=[RELU, LEAKY])
            return Fn.conv2d_grouped(t2, [s[4].weight, m[4].weight], [s[4].bias, m[4].bias], self._pg[2], kernel_size=5, stride=1, padding=2,
                                     shared_input=False, x_c_off=o2[0], x_group_step=o2[1] - o2[0], acts=[RELU, NONE],
                                     f32_out="only" if Fn.fp32_latents() else None)

        (sm, o3), weights = _branches(z, sigma_means, lambda: _mixture_weights(w, w[2].run_slice(t1, o1[2]), self.K, self.M))
        MK = self.M * self.K
        return sm[:, o3[0]:o3[0] + MK], sm[:, o3[1]:o3[1] + MK], weights

    def forward(self, z, hi=False):
        """``hi``: sigma / means as they feed the likelihood (fp32 from the accumulators at bf16 inference)."""
        if hi and Fn.grouped_ok(z):
            return self._forward_grouped(z)
        s, m, w = self.gmm_sigma, self.gmm_means, self.gmm_weights
        last = (lambda c, t, act=NONE: c.run_latent(t, act=act, want_lo=False)[1]) if hi else (lambda c, t, act=NONE: c.run(t, act=act))
        sigma, means, weights = _branches(
            z,
            lambda: last(s[4], s[2].run(s[0].run(z, act=RELU), act=RELU), RELU),
            lambda: last(m[4], m[2].run(m[0].run(z, act=LEAKY), act=LEAKY)),
            lambda: _mixture_weights(w, w[2].run(w[0].run(z, act=LEAKY)), self.K, self.M))
        return sigma, means, weights


class gmm_hyper_y2(nn.Module):
    """(z2_hat, y1_hat_warp) -> (sigma, means, weights) (newnet1.py:517-577); upsample + cat fused."""

    def __init__(self, N, M, K):
        super().__init__()
        self.N, self.M, self.K = N, M, K
        self.upsample_layer = nn.UpsamplingBilinear2d(scale_factor=4)   # no parameters; kept for the module tree
        self.gmm_sigma = nn.Sequential(conv(N + M, N, kernel_size=5, stride=1), nn.ReLU(), conv(N, N, kernel_size=5, stride=1),
                                       nn.ReLU(), conv(N, M * K, kernel_size=5, stride=1), nn.ReLU())
        self.gmm_means = nn.Sequential(conv(N + M, N, kernel_size=5, stride=1), nn.LeakyReLU(),
                                       conv(N, N, kernel_size=5, stride=1), nn.LeakyReLU(),
                                       conv(N, M * K, kernel_size=5, stride=1))
        self.gmm_weights = nn.Sequential(conv(N + M, N, kernel_size=5, stride=1), nn.LeakyReLU(),
                                         conv(N, M * K, kernel_size=5, stride=1), spatial_pool2d(), nn.LeakyReLU(),
                                         conv(M * K, M * K, kernel_size=1, stride=1))

    def _forward_grouped(self, c):
        """As ``gmm_hyper_y1._forward_grouped``: 320 -> 3 x 128 on the shared concat buffer, sigma / mean 128 -> 128 as two groups,
        128 -> 960 twice as two groups with fp32 output; the weights branch's 128 -> 960 on a side stream."""
        s, m, w = self.gmm_sigma, self.gmm_means, self.gmm_weights
        if not hasattr(self, "_pg"):
            self._pg = [Fn.PackedGroup() for _ in range(3)]
        t1, o1 = Fn.conv2d_grouped(c, [s[0].weight, m[0].weight, w[0].weight], [s[0].bias, m[0].bias, w[0].bias], self._pg[0], kernel_size=5,
                                   stride=1, padding=2, shared_input=True, acts=[RELU, LEAKY, LEAKY])

        def sigma_means():
            t2, o2 = Fn.conv2d_grouped(t1, [s[2].weight, m[2].weight], [s[2].bias, m[2].bias], self._pg[1], kernel_size=5, stride=1, padding=2,
                                       shared_input=False, x_c_off=o1[0], x_group_step=o1[1] - o1[0], acts=[RELU, LEAKY])
            return Fn.conv2d_grouped(t2, [s[4].weight, m[4].weight], [s[4].bias, m[4].bias], self._pg[2], kernel_size=5, stride=1, padding=2,
                                     shared_input=False, x_c_off=o2[0], x_group_step=o2[1] - o2[0], acts=[RELU, NONE],
                                     f32_out="only" if Fn.fp32_latents() else None)

        (sm, o3), weights = _branches(c, sigma_means, lambda: _mixture_weights(w, w[2].run_slice(t1, o1[2]), self.K, self.M))
        MK = self.M * self.K
        return sm[:, o3[0]:o3[0] + MK], sm[:, o3[1]:o3[1] + MK], weights

    def forward(self, z2, y1, hi=False):
        c = Fn.upsample4_cat(z2, y1)
        if hi and Fn.grouped_ok(c):
            return self._forward_grouped(c)
        s, m, w = self.gmm_sigma, self.gmm_means, self.gmm_weights
        last = (lambda cv, t, act=NONE: cv.run_latent(t, act=act, want_lo=False)[1]) if hi else (lambda cv, t, act=NONE: cv.run(t, act=act))
        sigma, means, weights = _branches(
            c,
            lambda: last(s[4], s[2].run(s[0].run(c, act=RELU), act=RELU), RELU),
            lambda: last(m[4], m[2].run(m[0].run(c, act=LEAKY), act=LEAKY)),
            lambda: _mixture_weights(w, w[2].run(w[0].run(c, act=LEAKY)), self.K, self.M))
        return sigma, means, weights


def _check_pair(x1, x2, h_matrix):
    """Argument errors up front, with a usable message (the reference fails deep inside with a size mismatch: SURVEY.md 5)."""
    if x1.dim() != 4 or x1.shape != x2.shape or x1.shape[1] != 3:
        raise RuntimeError(f"HSIC.forward: x1 and x2 must be (B, 3, H, W) tensors of one shape, got {tuple(x1.shape)} and {tuple(x2.shape)}")
    B, _, H, W = x1.shape
    if B == 0:
        raise RuntimeError("HSIC.forward: empty batch")
    if H % 64 or W % 64:
        raise RuntimeError(f"HSIC.forward: H and W must be multiples of 64 (the hyper path down-samples by 64), got {H}x{W}; "
                           "zero-pad with hesic_amd.models.pad_to_multiple and crop the reconstructions")
    if h_matrix.shape[-2:] != (3, 3) or h_matrix.dim() != 3 or h_matrix.shape[0] not in (1, B):
        raise RuntimeError(f"HSIC.forward: h_matrix must be (B, 3, 3) (or (1, 3, 3)), got {tuple(h_matrix.shape)}")


def _noise(nz, key, like, training):
    if not training:
        return None
    if nz is not None and key in nz:
        return nz[key].to(like.device)
    return torch.empty_like(like).uniform_(-0.5, 0.5)


def _quant(model, y, nz, key, training):
    """EntropyModel._quantize(y, 'noise' | 'dequantize') without means (newnet1.py:755)."""
    if training:
        return y + _noise(nz, key, y, True).to(y.dtype)
    return model._quantize(y, "dequantize")


def _round_latent(model, y_hi):
    """``_quantize(y, "dequantize")`` without means at inference: an fp32 latent of the bf16 mode is rounded in fp32 and
    stored in the storage dtype (integers: exact) for the convs that read it."""
    cdt = Fn.compute_dtype()
    if y_hi.is_cuda and y_hi.dtype == torch.float32 and cdt != torch.float32:
        return Fn.round_to(y_hi, cdt)
    return model._quantize(y_hi, "dequantize")


# --------------------------------------------------------------------------------------------- HESIC
class HSIC(StereoCompressionModel):
    """HESIC (reference ``HSIC``, ywz/mywork/newnet1.py:696-783)."""
    _LO_ABS = True            # encode_hyper reads |y| (newnet1.py:434)

    def __init__(self, N=128, M=192, K=5, **kwargs):
        super().__init__(entropy_bottleneck_channels=N, **kwargs)
        self.gaussian1 = GaussianMixtureConditional(K=K)
        self.gaussian2 = GaussianMixtureConditional(K=K)
        self.N, self.M, self.K = int(N), int(M), int(K)
        self.encoder1, self.encoder2 = Encoder1(N, M), Encoder2(N, M)
        self.decoder1, self.decoder2 = Decoder1(N, M), Decoder2(N, M)
        self._h_a1, self._h_a2 = encode_hyper(N=N, M=M), encode_hyper(N=N, M=M)
        self._h_s1, self._h_s2 = gmm_hyper_y1(N=N, M=M, K=K), gmm_hyper_y2(N=N, M=M, K=K)

    def forward(self, x1, x2, h_matrix, noise=None):
        """``noise`` (training only, optional): dict z1,y1,y1w,z2,y2 of U(-1/2,1/2) draws, in the order the
        reference makes them; absent keys are drawn on the device."""
        _check_pair(x1, x2, h_matrix)
        if not self.training and not torch.is_grad_enabled() and x1.is_cuda:
            return self._forward_eval(x1, x2, h_matrix, two_streams=OVERLAP_STREAMS)
        tr = self.training
        size = (x1.shape[-2], x1.shape[-1])
        y1 = self.encoder1(x1)
        z1 = self._h_a1(y1)
        z1_hat, z1_lik = self.entropy_bottleneck1.forward_with_noise(z1, _noise(noise, "z1", z1, tr))
        s1, m1, w1 = self._h_s1(z1_hat)
        y1_hat, y1_lik = self.gaussian1(y1, s1, m1, w1, noise=_noise(noise, "y1", y1, tr))
        x1_hat = self.decoder1(y1_hat)

        x1_warp = warp_perspective(x1, h_matrix, size)
        y2 = self.encoder2(x1_warp, x2)
        x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)          # :753 and :767 are the same tensor
        y1_hat_w = _quant(self.gaussian1, self.encoder1(x1_hat_warp), noise, "y1w", tr)

        z2 = self._h_a2(y2)
        z2_hat, z2_lik = self.entropy_bottleneck2.forward_with_noise(z2, _noise(noise, "z2", z2, tr))
        s2, m2, w2 = self._h_s2(z2_hat, y1_hat_w)
        y2_hat, y2_lik = self.gaussian2(y2, s2, m2, w2, noise=_noise(noise, "y2", y2, tr))
        x2_hat = self.decoder2(y2_hat, x1_hat_warp)
        return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat,
                "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}


    # ---------------------------------------------------------------------------------------- real bit-stream
    # HSIC.compress / decompress of the reference (ywz/mywork/newnet1.py:823-1073, :1076-1273).  Same
    # flow and the same header file; the per-pixel Python loops of the reference are one HIP launch per view for the
    # cumulative-frequency tables (hesic_gmm_cdf) + one pass of the host range coder.  The .bin payload is coded by
    # libhesic_host's own carry-less range coder: the reference's `range_coder` package is third party, absent and
    # unpinned, so byte-compatibility of that file is not claimed (round trips are exact).  Batch size 1, like the reference.
    _CDF_CHUNK_BYTES = 256 << 20

    def _cdf_chunks(self, gmm, channels, minmax, scale_bound, y_hat=None):
        """(channels, symbols | None, cdf) numpy chunks in the reference's coding order: channel-major, rows, columns."""
        import numpy as np
        scales, means, weights = gmm
        H, W = scales.shape[-2:]
        per_ch = H * W * (2 * minmax + 2) * 4
        step = max(1, self._CDF_CHUNK_BYTES // per_ch)
        for i in range(0, len(channels), step):
            ch = channels[i:i + step]
            cdf = Fn.gmm_cdf_tables(scales, means, weights, ch, minmax, self.K, scale_bound=scale_bound)
            cdf = cdf.cpu().numpy().view(np.uint32).reshape(-1, 2 * minmax + 2)
            sym = None
            if y_hat is not None:
                sym = (y_hat[0, ch].float().cpu().numpy().astype(np.int64) + minmax).reshape(-1).astype(np.int32)
            yield ch, sym, cdf

    def _analysis(self, x1, x2, h_matrix):
        size = (x1.shape[-2], x1.shape[-1])
        cdt = Fn.compute_dtype()
        y1_lo, y1 = self.encoder1.latent(x1, exact=True, lo_abs=self._LO_ABS)
        z1 = self._h_a1.latent(y1_lo)
        z1_strings = self.entropy_bottleneck1.compress(z1)
        z1_hat = self.entropy_bottleneck1.decompress(z1_strings, z1.size()[-2:]).to(cdt)
        gmm1 = self._h_s1(z1_hat, hi=True)
        y1_hat = _round_latent(self.gaussian1, y1)
        x1_hat = self.decoder1(y1_hat)
        x1_warp = warp_perspective(x1, h_matrix, size)
        y2_lo, y2 = self.encoder2.latent(x1_warp, x2, exact=True, lo_abs=self._LO_ABS)
        z2 = self._h_a2.latent(y2_lo)
        z2_strings = self.entropy_bottleneck2.compress(z2)
        z2_hat = self.entropy_bottleneck2.decompress(z2_strings, z2.size()[-2:]).to(cdt)
        x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)
        y1_hat_w = _round_latent(self.gaussian1, self.encoder1.latent(x1_hat_warp, want_lo=False)[1])
        gmm2 = self._h_s2(z2_hat, y1_hat_w, hi=True)
        y2_hat = _round_latent(self.gaussian2, y2)
        return (y1_hat, z1_hat, z1_strings, gmm1), (y2_hat, z2_hat, z2_strings, gmm2)

    def compress(self, x1, x2, h_matrix, output_name, output_path="", device=None):
        import os
        import time
        import numpy as np
        from ._host import RangeEncoder
        if x1.shape[0] != 1:
            raise ValueError("HSIC.compress codes one stereo pair per call (batch size 1, as the reference)")
        if self.entropy_bottleneck1._offset.numel() == 0:
            self.update()
        with torch.no_grad():
            v1, v2 = self._analysis(x1, x2, h_matrix)
        out1 = os.path.join(output_path, str(output_name) + ".npz")      # the reference's name for its raw header file
        out2 = os.path.join(output_path, str(output_name) + ".bin")
        head = bytearray(np.array(x1.shape[2:], dtype=np.uint16).tobytes())
        enc = RangeEncoder()
        start = time.time()
        for (y_hat, _z_hat, z_strings, gmm), gauss in ((v1, self.gaussian1), (v2, self.gaussian2)):
            yi = y_hat[0].float()
            flag = (yi.abs().sum(dim=(1, 2)) > 0).cpu().numpy().astype(np.uint8)
            minmax = int(max(float(yi.abs().max()), 1.0))
            if len(z_strings[0]) > 65535 or minmax > 32767:
                raise ValueError("HSIC.compress: z string longer than the uint16 header field or latent range beyond the table kernel's 32767")
            head += np.array([len(z_strings[0]), minmax], dtype=np.uint16).tobytes()
            head += np.packbits(flag).tobytes()
            head += z_strings[0]
            channels = [int(c) for c in np.nonzero(flag)[0]]
            for _ch, sym, cdf in self._cdf_chunks(gmm, channels, minmax, gauss._bound(), y_hat):
                enc.encode(sym, cdf)
        payload = payload_header() + enc.finish()
        with open(out1, "wb") as f:
            f.write(bytes(head))
        with open(out2, "wb") as f:
            f.write(payload)
        num_pixels = x1.shape[2] * x1.shape[3] * 2
        return {"bpp_real": (len(head) + len(payload)) * 8 / num_pixels, "bpp_side": len(head) * 8 / num_pixels,
                "enctime": time.time() - start, "y1_hat": v1[0], "y2_hat": v2[0], "z1_hat": v1[1], "z2_hat": v2[1]}

    def decompress(self, x1, x2, h_matrix, output_name, output_path="", device=None):
        """x1 / x2 are unused (the reference only reads their size, which the header carries); kept for its signature."""
        import os
        import time
        import numpy as np
        from ._host import RangeDecoder
        if self.entropy_bottleneck1._offset.numel() == 0:
            self.update()
        dev = h_matrix.device
        with open(os.path.join(output_path, str(output_name) + ".npz"), "rb") as f:
            blob = f.read()
        pos = 4
        x_shape = np.frombuffer(blob[:4], dtype=np.uint16).astype(int)
        views = []
        for _ in range(2):
            length, minmax = (int(v) for v in np.frombuffer(blob[pos:pos + 4], dtype=np.uint16))
            pos += 4
            flag = np.unpackbits(np.frombuffer(blob[pos:pos + (self.M + 7) // 8], dtype=np.uint8))[:self.M]
            pos += (self.M + 7) // 8
            views.append((minmax, [int(c) for c in np.nonzero(flag)[0]], blob[pos:pos + length]))
            pos += length
        y_shape = x_shape // 16
        z_shape = y_shape // 4
        with open(os.path.join(output_path, str(output_name) + ".bin"), "rb") as f:
            payload = f.read()
        _, off = check_payload(payload)
        dec = RangeDecoder(payload[off:])
        start = time.time()
        cdt = Fn.compute_dtype()
        size = (int(x_shape[0]), int(x_shape[1]))

        def decode_view(gmm, minmax, channels, gauss):
            y_hat = torch.zeros((1, self.M, int(y_shape[0]), int(y_shape[1])), dtype=torch.float32)
            for ch, _sym, cdf in self._cdf_chunks(gmm, channels, minmax, gauss._bound()):
                sym = dec.decode(cdf).reshape(len(ch), int(y_shape[0]), int(y_shape[1]))
                y_hat[0, ch] = torch.from_numpy(sym.astype(np.float32) - minmax)
            return y_hat.to(dev, cdt).contiguous(memory_format=torch.channels_last)

        with torch.no_grad():
            zs = (int(z_shape[0]), int(z_shape[1]))
            z1_hat = self.entropy_bottleneck1.decompress([views[0][2]], zs).to(dev, cdt)
            z2_hat = self.entropy_bottleneck2.decompress([views[1][2]], zs).to(dev, cdt)
            gmm1 = self._h_s1(z1_hat, hi=True)
            y1_hat = decode_view(gmm1, views[0][0], views[0][1], self.gaussian1)
            x1_hat = self.decoder1(y1_hat)
            x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)
            y1_hat_w = _round_latent(self.gaussian1, self.encoder1.latent(x1_hat_warp, want_lo=False)[1])
            gmm2 = self._h_s2(z2_hat, y1_hat_w, hi=True)
            y2_hat = decode_view(gmm2, views[1][0], views[1][1], self.gaussian2)
            x2_hat = self.decoder2(y2_hat, x1_hat_warp)
        return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat, "z1_hat": z1_hat, "z2_hat": z2_hat,
                "dectime": time.time() - start}

    def _forward_eval(self, x1, x2, h_matrix, two_streams=True):
        """Inference schedule.  The chain that bounds the step is encoder1 -> round -> decoder1 -> warp -> {decoder2 | encoder1 ->
        h_s2 -> likelihood}: the reconstructions need the ROUNDED latents only (the mixture's quantiser takes no means,
        entropy_models.py:661-702), the hyper paths only price them.  With ``two_streams`` the main stream walks that chain and
        everything else is forked beside it: view 2's analysis (warp -> encoder2 -> round; h_a2 -> bottleneck), which depends only
        on the inputs; view 1's rate (h_a1 -> bottleneck -> h_s1 -> likelihood), as soon as y1 exists; decoder2, as soon as the
        warped reconstruction and y2_hat exist.  The small-grid kernels of one branch fill the CUs the others leave idle.  Results
        are identical to the single-stream order (``HESIC_NO_OVERLAP=1``; the tests run both).

        What feeds round() and the likelihoods -- y, z, sigma, mu -- is fp32 even in the bf16 mode (``Fn.fp32_latents``):
        those convs also / only write their fp32 accumulators, the entropy kernels read fp32 and store the integer-valued
        y_hat / z_hat in the storage dtype for the synthesis convs."""
        size = (x1.shape[-2], x1.shape[-1])
        cdt = Fn.compute_dtype()

        def view2_latents():
            x1_warp = warp_perspective(x1, h_matrix, size)
            y2_lo, y2 = self.encoder2.latent(x1_warp, x2, exact=True, lo_abs=self._LO_ABS)
            return y2_lo, y2, _round_latent(self.gaussian2, y2)

        def view2_hyper(y2_lo):
            return self.entropy_bottleneck2.forward_with_noise(self._h_a2.latent(y2_lo), None, out_dtype=cdt)

        def view1_rate(y1_lo, y1):
            z1_hat, z1_lik = self.entropy_bottleneck1.forward_with_noise(self._h_a1.latent(y1_lo), None, out_dtype=cdt)
            s1, m1, w1 = self._h_s1(z1_hat, hi=True)
            return self.gaussian1(y1, s1, m1, w1, out_dtype=cdt)[1], z1_lik

        if not (two_streams and x1.is_cuda):
            y1_lo, y1 = self.encoder1.latent(x1, exact=True, lo_abs=self._LO_ABS)
            y1_hat = _round_latent(self.gaussian1, y1)
            y1_lik, z1_lik = view1_rate(y1_lo, y1)
            x1_hat = self.decoder1(y1_hat)
            x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)          # :753 and :767 are the same tensor
            y1_hat_w = _round_latent(self.gaussian1, self.encoder1.latent(x1_hat_warp, want_lo=False)[1])
            y2_lo, y2, y2_hat = view2_latents()
            z2_hat, z2_lik = view2_hyper(y2_lo)
            s2, m2, w2 = self._h_s2(z2_hat, y1_hat_w, hi=True)
            y2_lik = self.gaussian2(y2, s2, m2, w2, out_dtype=cdt)[1]
            x2_hat = self.decoder2(y2_hat, x1_hat_warp)
        else:
            main, dev = torch.cuda.current_stream(), x1.device
            # Four streams in all: main, view 2 (its analysis, then decoder2), view 1's rate, h_s2's side branch.  Measured on one
            # box, 512^2 B=8, ms per step: the round-1/2 schedule (decoders behind the likelihoods) 2.19; this one 2.06-2.09; decoder2
            # and h_a2 on streams of their own 2.05 in five runs of six and 2.19 in the sixth (five streams share the runtime's four
            # hardware queues; which two share changes the result); the same with GPU_MAX_HW_QUEUES=8: 2.80 (two full-grid kernels
            # side by side slow each other more than the overlap buys); everything off the chain on ONE side stream: 2.14-2.17.
            def view2_front():
                y2_lo, y2, y2_hat = view2_latents()
                return y2, y2_hat, view2_hyper(y2_lo)

            # The chain is ISSUED first and the side branches fork from events on it: kernels reach a stream -- and nodes a captured
            # graph -- in issue order, and a branch issued in front of the chain's next kernel was seen to run in front of it.
            here = _ev_record

            v2 = _Fork(_side_stream(dev, 10), view2_front)
            y1_lo, y1 = self.encoder1.latent(x1, exact=True, lo_abs=self._LO_ABS)
            y1_hat = _round_latent(self.gaussian1, y1)
            ev_y1 = here(main)
            x1_hat = self.decoder1(y1_hat)
            x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)
            ev_xw = here(main)
            y1_hat_w = _round_latent(self.gaussian1, self.encoder1.latent(x1_hat_warp, want_lo=False)[1])
            ev2 = here(v2.stream)                       # view 2's latents and hyper-latents exist from here on
            v2.then(lambda: self.decoder2(v2.out[1], x1_hat_warp), start=ev_xw)
            r1 = _Fork(_side_stream(dev, 12), lambda: view1_rate(y1_lo, y1), start=ev_y1)
            _wait_event(main, ev2)
            y2, y2_hat, (z2_hat, z2_lik) = v2.out[0]
            for t in (y2, z2_hat, z2_lik):
                _rec(t, main)
            s2, m2, w2 = self._h_s2(z2_hat, y1_hat_w, hi=True)
            y2_lik = self.gaussian2(y2, s2, m2, w2, out_dtype=cdt)[1]
            y1_lik, z1_lik = r1.join()
            x2_hat = v2.join()[1]
        return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat,
                "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}


# -------------------------------------------------------------------------------------------- HESIC+
def _nhwc_rows(t):
    """(1, C, H, W) map -> (H * W, C) rows, one per pixel (a view when the map is channels_last)."""
    return t.permute(0, 2, 3, 1).reshape(t.shape[2] * t.shape[3], t.shape[1])


# ---- the .bin payload container (this repository's own file: the reference's third-party `range_coder` is absent).
# A payload can only be decoded by a decoder that forms the SAME cumulative-frequency tables, bit for bit: they come out of the
# hyper-synthesis (and, for view 2, the decoder1 -> warp -> encoder1 pass) run in the encoder's storage format.  Since round 4 every payload
# starts with 4 magic bytes + one MODE byte naming what the tables depend on; a decoder in another mode raises instead of desynchronising.
# A/B switch, OFF: g_a_conv3 / conv4 of "x3c2" on pairs x single error-feedback weights (hesic_conv2d_forward_hilo_w1, two products per pair).
# Measured (round 4, same box, alternating runs): 3617 / 3586 pairs/s against 3582 / 3567 at three products (+0.7 %), flips 5.4e-4 vs 5.3e-4 at
# 512^2 but 7.7e-4 vs 6.3e-4 on the 256^2 golden -- not worth a quarter of the margin to the 1e-3 bar
X3C2_TWO_PRODUCT_TAIL = False
X2_LAYERS = (2, 2, 2)          # "x2" mode (measured and not the default): products per MAC of g_a_conv2 / conv3 / conv4
WAVEFRONT_GRAPHS = True      # module switch: False = round 3's per-group launches from Python
# A/B switch: 1 = the device reads the decoded symbols from, and writes the tables into, pinned host memory itself (no copy nodes)
WAVEFRONT_ZEROCOPY = True
# A/B switch, off: 1 = the table launch of a group is the sixth node of its captured step (hesic_gmm_cdf_dyn: channel list and alphabet read on
# the device).  Measured neutral (per view: 1.0 vs 1.6 ms of launch calls, 8.3 vs 7.6 ms of waiting -- a graph node costs what a launch costs)
WAVEFRONT_TABLE_IN_GRAPH = False
_CDF_WAVE_MAX = 1024          # csrc/entropy.hip: alphabets the wave-per-row table kernel takes
# A/B switch: 1 = the C loop replays each group's recorded launches one by one instead of launching its graph
WAVEFRONT_TAPE = True
WAVEFRONT_C_LOOP = True      # module switch: False = the group loop in Python (six C calls per group)
PAYLOAD_MAGIC = b"HSC\x03"               # format 3 (round 5: two mode bytes).  Format 2 (round 4) had one; format 1 (rounds 2-3) no header
TABLE_KERNEL_VERSION = 2                  # bump when hesic_gmm_cdf / the table-producing launches change their arithmetic
_MODE_BYTES = 2


def payload_mode_bytes():
    """Everything the cumulative-frequency tables depend on besides the weights (round 4's single byte left out the warp convention
    and the summation-order switches, so a decoder differing in one of them passed the check and desynchronised on view 2).
    byte 0 -- bits 0-1: storage format of the maps (0 fp32, 1 bfloat16, 2 float16); bit 2: error-feedback weight rounding on the
    single-operand analysis launches (the third analysis pass feeds view 2's tables); bit 3: fp32 latents; bits 4-7: table-kernel version.
    byte 1 -- bit 0: warp convention (``geometry.DEFAULT_ALIGN_CORNERS``: x1_hat_warp, hence round(encoder1(warp(x1_hat))), hence view 2's
    tables); bit 1: split-K launches allowed (``Fn.SPLIT_K``); bit 2: grouped hyper-synthesis launches (``Fn.GROUP_HYPER``) -- both reorder
    the fp32 sums of the hyper-synthesis; bits 3-4: analysis precision of the third pass' producer chain (0 x3, 1 x3c2, 2 x1, 3 x2: x1_hat is
    decoded from y1_hat alone, but the mode is recorded so that a mismatch is reported rather than argued about)."""
    from . import geometry as _geo
    dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[Fn.compute_dtype()]
    b0 = dt | (int(bool(Fn.SHAPED_WEIGHTS) and dt != 0) << 2) | (int(bool(Fn.FP32_LATENTS)) << 3) | (TABLE_KERNEL_VERSION << 4)
    an = 0 if dt == 0 else {"x3": 0, "x3c2": 1, "x1": 2, "x2": 3}[Fn.analysis_precision()]
    b1 = int(bool(_geo.DEFAULT_ALIGN_CORNERS)) | (int(bool(Fn.SPLIT_K)) << 1) | (int(bool(Fn.GROUP_HYPER)) << 2) | (an << 3)
    return bytes([b0, b1])


def payload_mode_byte():
    """Byte 0 of ``payload_mode_bytes()`` (the round-4 name; tests and tools that look at the storage format use it)."""
    return payload_mode_bytes()[0]


def describe_mode_bytes(b):
    b0, b1 = b[0], b[1]
    return (f"{('float32', 'bfloat16', 'float16', '?')[b0 & 3]} maps, error-feedback analysis weights {'on' if b0 & 4 else 'off'}, "
            f"fp32 latents {'on' if b0 & 8 else 'off'}, table kernels v{b0 >> 4}, warp align_corners={'True' if b1 & 1 else 'False'}, "
            f"split-K {'on' if b1 & 2 else 'off'}, grouped hyper-synthesis {'on' if b1 & 4 else 'off'}, analysis {('x3', 'x3c2', 'x1', 'x2')[(b1 >> 3) & 3]}")


def payload_header(extra=b""):
    return PAYLOAD_MAGIC + payload_mode_bytes() + extra


def check_payload(payload, n_extra=0):
    """Validate the container header; returns (extra bytes, offset of the range-coder stream)."""
    n = len(PAYLOAD_MAGIC)
    if len(payload) < n + _MODE_BYTES + n_extra or payload[:n] != PAYLOAD_MAGIC:
        raise ValueError("decompress: the .bin payload does not start with this coder's format-3 header (a file written by rounds 2-4 of "
                         "this package, or not one of its payloads): re-encode it -- the header names the mode the tables were formed in")
    mode, here = bytes(payload[n:n + _MODE_BYTES]), payload_mode_bytes()
    if mode != here:
        raise ValueError(f"decompress: the payload was written with [{describe_mode_bytes(mode)}] but this process decodes with "
                         f"[{describe_mode_bytes(here)}]; the cumulative-frequency tables would differ in the last count and the range decoder "
                         "would desynchronise.  Select the writer's mode (hesic_amd.set_compute_dtype, set_analysis_precision, "
                         "geometry.use_reference_era_warp / HESIC_WARP_ALIGN_CORNERS, HESIC_SHAPED_WEIGHTS, HESIC_BF16_LATENTS, HESIC_NO_GROUP_HYPER)")
    return payload[n + _MODE_BYTES:n + _MODE_BYTES + n_extra], n + _MODE_BYTES + n_extra


def _seq3(seq, x, last_act=NONE):
    """conv/deconv -> LeakyReLU -> conv/deconv -> LeakyReLU -> conv (the h_a / h_s / entropy_parameters
    Sequentials of newnet1_joint.py:611-665) with the activations fused."""
    return seq[4].run(seq[2].run(seq[0].run(x, act=LEAKY), act=LEAKY), act=last_act)


def _seq3_hi(seq, x):
    """``_seq3`` whose last conv feeds an entropy model: returns its ``hi`` output (``run_latent``); a ``Fn.HiLo`` input runs the
    three layers on hi/lo pairs (bf16x3 analysis route)."""
    if isinstance(x, Fn.HiLo):
        return seq[4].run_hilo(seq[2].run_hilo(seq[0].run_hilo(x.t, act=LEAKY), act=LEAKY), out="f32")
    return seq[4].run_latent(seq[2].run(seq[0].run(x, act=LEAKY), act=LEAKY), want_lo=False)[1]


class HSICJoint(StereoCompressionModel):
    """HESIC+ (reference ``HSIC`` of ywz/mywork/newnet1_joint.py:585-753): Minnen-style hyperprior +
    masked-conv context model + 1x1 entropy-parameter nets + single Gaussian."""
    _LO_ABS = False           # h_a reads y itself (newnet1_joint.py:611-617)

    def __init__(self, N=128, M=192, K=5, **kwargs):
        super().__init__(entropy_bottleneck_channels=N, **kwargs)
        self.gaussian1 = GaussianMixtureConditional(K=K)      # only its _quantize is used (:715)
        self.gaussian2 = GaussianMixtureConditional(K=K)
        self.N, self.M, self.K = int(N), int(M), int(K)
        self.encoder1, self.encoder2 = Encoder1(N, M), Encoder2(N, M)
        self.decoder1, self.decoder2 = Decoder1(N, M), Decoder2(N, M)

        def h_a():
            return nn.Sequential(conv(M, N, stride=1, kernel_size=3), nn.LeakyReLU(inplace=True),
                                 conv(N, N, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                 conv(N, N, stride=2, kernel_size=5))

        def h_s():
            return nn.Sequential(deconv(N, M, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                 deconv(M, M * 3 // 2, stride=2, kernel_size=5), nn.LeakyReLU(inplace=True),
                                 conv(M * 3 // 2, M * 2, stride=1, kernel_size=3))

        def ep(cin):
            return nn.Sequential(HipConv2d(cin, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
                                 HipConv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
                                 HipConv2d(M * 8 // 3, M * 6 // 3, 1))

        self.h_a1, self.h_s1 = h_a(), h_s()
        self.entropy_parameters1 = ep(M * 12 // 3)
        self.context_prediction1 = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)
        self.gaussian_conditional1 = GaussianConditional(None)
        self.h_a2, self.h_s2 = h_a(), h_s()
        self.entropy_parameters2 = ep(5 * M)
        self.context_prediction2 = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)
        self.gaussian_conditional2 = GaussianConditional(None)

    def forward(self, x1, x2, h_matrix, noise=None):
        """noise keys (training): z1, y1, y1b, z2, y1w, y2, y2b (reference draw order)."""
        _check_pair(x1, x2, h_matrix)
        if not self.training and not torch.is_grad_enabled() and x1.is_cuda:
            return self._forward_eval(x1, x2, h_matrix)
        tr = self.training
        size = (x1.shape[-2], x1.shape[-1])
        y1 = self.encoder1(x1)
        z1 = _seq3(self.h_a1, y1)
        z1_hat, z1_lik = self.entropy_bottleneck1.forward_with_noise(z1, _noise(noise, "z1", z1, tr))
        params1 = _seq3(self.h_s1, z1_hat)
        y1_hat = _quant(self.gaussian_conditional1, y1, noise, "y1", tr)
        ctx1 = self.context_prediction1(y1_hat)
        gp1 = _seq3(self.entropy_parameters1, torch.cat((params1, ctx1), 1))
        sc1, mu1 = gp1.chunk(2, 1)
        _, y1_lik = self.gaussian_conditional1(y1, sc1, means=mu1, noise=_noise(noise, "y1b", y1, tr))
        x1_hat = self.decoder1(y1_hat)

        x1_warp = warp_perspective(x1, h_matrix, size)
        y2 = self.encoder2(x1_warp, x2)
        z2 = _seq3(self.h_a2, y2)
        z2_hat, z2_lik = self.entropy_bottleneck2.forward_with_noise(z2, _noise(noise, "z2", z2, tr))
        x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)
        y1_hat_w = _quant(self.gaussian1, self.encoder1(x1_hat_warp), noise, "y1w", tr)
        params2 = _seq3(self.h_s2, z2_hat)
        y2_hat = _quant(self.gaussian_conditional2, y2, noise, "y2", tr)
        ctx2 = self.context_prediction2(y2_hat)
        gp2 = _seq3(self.entropy_parameters2, torch.cat((params2, ctx2, y1_hat_w), 1))
        sc2, mu2 = gp2.chunk(2, 1)
        # the reference evaluates view 2 with gaussian_conditional1 as well (:725); no learnable state, harmless
        _, y2_lik = self.gaussian_conditional1(y2, sc2, means=mu2, noise=_noise(noise, "y2b", y2, tr))
        x2_hat = self.decoder2(y2_hat, x1_hat_warp)
        return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat,
                "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}


    def _forward_eval(self, x1, x2, h_matrix):
        """Inference: y, z and the (scale, mean) maps are fp32 even in the bf16 mode, and the same schedule as ``HSIC._forward_eval``:
        the main stream walks encoder1 -> round -> decoder1 -> warp -> encoder1 -> view 2's context model; view 2's analysis with
        its whole hyper path (inputs only) and then decoder2 run on a second stream, view 1's rate on a third."""
        size = (x1.shape[-2], x1.shape[-1])
        cdt = Fn.compute_dtype()
        catfree = x1.is_cuda and CAT_FREE_EP      # the convs feeding entropy_parameters write their channel slices of ONE buffer (no torch.cat)

        def view2_front():
            x1_warp = warp_perspective(x1, h_matrix, size)
            y2_lo, y2 = self.encoder2.latent(x1_warp, x2, exact=True, lo_abs=self._LO_ABS)
            y2_hat = _round_latent(self.gaussian_conditional2, y2)
            z2_hat, z2_lik = self.entropy_bottleneck2.forward_with_noise(_seq3_hi(self.h_a2, y2_lo), None, out_dtype=cdt)
            params2 = self._params_buffer(self.h_s2, z2_hat, y2_hat, self.M) if catfree else _seq3(self.h_s2, z2_hat)
            return y2, y2_hat, params2, z2_lik

        def view1_rate(y1_lo, y1, y1_hat):
            z1_hat, z1_lik = self.entropy_bottleneck1.forward_with_noise(_seq3_hi(self.h_a1, y1_lo), None, out_dtype=cdt)
            if catfree:
                buf1 = self._params_buffer(self.h_s1, z1_hat, y1_hat)
                sc1, mu1 = self._gauss_full(1, None, y1_hat, buf=buf1)
            else:
                sc1, mu1 = self._gauss_full(1, _seq3(self.h_s1, z1_hat), y1_hat)
            return self.gaussian_conditional1(y1, sc1, means=mu1, out_dtype=cdt)[1], z1_lik

        overlap = OVERLAP_STREAMS and x1.is_cuda
        if overlap:
            main, dev = torch.cuda.current_stream(), x1.device
            v2 = _Fork(_side_stream(dev, 10), view2_front)
        y1_lo, y1 = self.encoder1.latent(x1, exact=True, lo_abs=self._LO_ABS)
        y1_hat = _round_latent(self.gaussian_conditional1, y1)
        if overlap:
            r1 = _Fork(_side_stream(dev, 12), lambda: view1_rate(y1_lo, y1, y1_hat))
        else:
            y1_lik, z1_lik = view1_rate(y1_lo, y1, y1_hat)
        x1_hat = self.decoder1(y1_hat)
        x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)
        if overlap:
            ev2 = _ev_record(v2.stream)
            y2, y2_hat, params2, z2_lik = v2.out
            v2.then(lambda: self.decoder2(y2_hat, x1_hat_warp))
        else:
            y2, y2_hat, params2, z2_lik = view2_front()
        y1_hat_w = _round_latent(self.gaussian1, self.encoder1.latent(x1_hat_warp, want_lo=False)[1])
        if overlap:
            _wait_event(main, ev2)
            for t in (y2, y2_hat, params2, z2_lik):
                _rec(t, main)
        if catfree:
            sc2, mu2 = self._gauss_full(2, None, y2_hat, y1_hat_w, buf=params2)
        else:
            sc2, mu2 = self._gauss_full(2, params2, y2_hat, y1_hat_w)
        # the reference evaluates view 2 with gaussian_conditional1 as well (:725); no learnable state, harmless
        _, y2_lik = self.gaussian_conditional1(y2, sc2, means=mu2, out_dtype=cdt)
        if overlap:
            y1_lik, z1_lik = r1.join()
            x2_hat = v2.join()[1]
        else:
            x2_hat = self.decoder2(y2_hat, x1_hat_warp)
        return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat,
                "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}

    # ---------------------------------------------------------------------------------------- real bit-stream
    # compress / decompress of the reference's HESIC+ (ywz/mywork/newnet1_joint.py:793-1079, :1081-1321): same header
    # file as HESIC, y coded pixel by pixel in raster order (all non-zero channels of a pixel together) under a single
    # Gaussian whose (scale, mean) come from the hyper-decoder and the masked-conv context of the ALREADY CODED latents.
    # The encoder knows every latent, so it evaluates the context model for the whole map in one pass; the decoder walks
    # the map like the reference does (5x5 crop -> masked conv -> 1x1 entropy-parameter net -> tables -> decode -> write
    # back).  Both run their convs without split-K: the kernels then accumulate every output element in the same order
    # whatever the map size, which makes the decoder's per-pixel numbers bit-identical to the encoder's (tested).
    def _params_view(self, which, z_hat):
        h_s = self.h_s1 if which == 1 else self.h_s2
        return _seq3(h_s, z_hat)

    def _gauss_full(self, which, params, y_hat, extra=None, buf=None):
        """(scales, means) = entropy_parameters(cat(params, context(y_hat)[, extra])) (newnet1_joint.py:703-707, :741-746).
        ``buf``: inference form without the cat -- an NHWC buffer whose first channels already hold ``params`` (the hyper-synthesis
        wrote them there); the masked conv writes its slice in place and ``extra`` is copied behind it."""
        ctx_m = self.context_prediction1 if which == 1 else self.context_prediction2
        ep = self.entropy_parameters1 if which == 1 else self.entropy_parameters2
        if buf is not None:
            c0 = buf.shape[1] - ctx_m.out_channels - (0 if extra is None else extra.shape[1])
            ctx_m.forward_into(y_hat, buf, c0)
            if extra is not None:
                Fn.copy_into(extra, buf, c0 + ctx_m.out_channels)
            return _seq3_hi(ep, buf).chunk(2, 1)
        ctx = ctx_m(y_hat)
        cat = (params, ctx) if extra is None else (params, ctx, extra)
        return _seq3_hi(ep, torch.cat(cat, 1)).chunk(2, 1)

    def _params_buffer(self, seq, z_hat, y_like, extra_channels=0):
        """h_s(z_hat) written straight into the first channels of the entropy-parameter net's input buffer."""
        B, _, H, W = y_like.shape
        c_par = seq[4].out_channels
        buf = Fn._empty_nhwc(B, c_par + 2 * self.M + extra_channels, H, W, Fn.compute_dtype(), z_hat.device)
        seq[4].run_into(seq[2].run(seq[0].run(z_hat, act=LEAKY), act=LEAKY), buf, 0)
        return buf

    def _gauss_pixel(self, which, params, y_pad, h, w, extra=None):
        """(scales, means) of pixel (h, w) from the 5x5 crop of the padded, partially decoded map (:903-911)."""
        ctx_m = self.context_prediction1 if which == 1 else self.context_prediction2
        ep = self.entropy_parameters1 if which == 1 else self.entropy_parameters2
        crop = y_pad[:, :, h:h + 5, w:w + 5].contiguous(memory_format=torch.channels_last)
        if not hasattr(ctx_m, "_packer"):
            ctx_m._packer = Fn.PackedWeight()
        ctx = Fn.conv2d(crop, ctx_m.weight, ctx_m.bias, kernel_size=5, stride=1, padding=0, mask=ctx_m.mask,
                        tap_mask=ctx_m._tap_mask, packer=ctx_m._packer)                       # (1, 2M, 1, 1)
        parts = [params[:, :, h:h + 1, w:w + 1], ctx]
        if extra is not None:
            parts.append(extra[:, :, h:h + 1, w:w + 1])
        return _seq3_hi(ep, torch.cat(parts, 1).contiguous(memory_format=torch.channels_last)).chunk(2, 1)

    @staticmethod
    def _header_view(y_hat, z_strings):
        import numpy as np
        yi = y_hat[0].float()
        flag = (yi.abs().sum(dim=(1, 2)) > 0).cpu().numpy().astype(np.uint8)
        minmax = int(max(float(yi.abs().max()), 1.0))
        if len(z_strings[0]) > 65535 or minmax > 32767:
            raise ValueError("compress: z string longer than the uint16 header field or latent range beyond the table kernel's 32767")
        head = np.array([len(z_strings[0]), minmax], dtype=np.uint16).tobytes() + np.packbits(flag).tobytes() + z_strings[0]
        return head, minmax, [int(c) for c in np.nonzero(flag)[0]]

    @staticmethod
    def _wavefronts(H, W):
        """Pixels grouped by t = w + 3 h.  The 5x5 mask-'A' context of pixel (h, w) is rows h-2, h-1 (columns w-2 .. w+2) and
        (h, w-2), (h, w-1): every one of them has a smaller t, so the pixels of one group only depend on earlier groups and
        can be decoded together -- W + 3 (H - 1) device steps instead of H * W.  Returns a list of int64 arrays of raster
        indices h * W + w (ascending inside a group)."""
        import numpy as np
        hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        t = (ww + 3 * hh).reshape(-1)
        order = np.argsort(t, kind="stable")
        cuts = np.flatnonzero(np.diff(t[order])) + 1
        return np.split(order.astype(np.int64), cuts)

    ORDER_RASTER, ORDER_WAVEFRONT = 0, 1
    _TABLE_BYTES = 256 << 20           # bound of one batch of cumulative-frequency tables (device buffer + its host copy)

    def compress(self, x1, x2, h_matrix, output_name, output_path="", device=None, order="wavefront"):
        """``order``: the sequence in which the pixels' symbols enter the range coder.  "raster" is the reference's (rows, then
        columns; newnet1_joint.py:903-1040) and forces a decoder to walk the map pixel by pixel; "wavefront" (default) codes the
        pixels group by group of ``_wavefronts`` so the decoder evaluates every group in one batch.  Same symbols, same
        tables; the first payload byte says which (the payload is this repository's own range coder either way)."""
        import os
        import time
        import numpy as np
        from ._host import RangeEncoder
        if x1.shape[0] != 1:
            raise ValueError("compress codes one stereo pair per call (batch size 1, as the reference)")
        if order not in ("raster", "wavefront"):
            raise ValueError('compress: order must be "raster" or "wavefront"')
        if self.entropy_bottleneck1._offset.numel() == 0:
            self.update()
        size = (x1.shape[-2], x1.shape[-1])
        start = time.time()
        with torch.no_grad(), Fn.no_split_k():
            self.context_prediction1.weight.data *= self.context_prediction1.mask
            self.context_prediction2.weight.data *= self.context_prediction2.mask
            cdt = Fn.compute_dtype()
            y1_lo, y1 = self.encoder1.latent(x1, exact=True, lo_abs=self._LO_ABS)
            z1 = _seq3_hi(self.h_a1, y1_lo)
            z1_strings = self.entropy_bottleneck1.compress(z1)
            z1_hat = self.entropy_bottleneck1.decompress(z1_strings, z1.size()[-2:]).to(cdt)
            y1_hat = _round_latent(self.gaussian_conditional1, y1)
            sc1, mu1 = self._gauss_full(1, self._params_view(1, z1_hat), y1_hat)
            x1_hat = self.decoder1(y1_hat)
            x1_warp = warp_perspective(x1, h_matrix, size)
            y2_lo, y2 = self.encoder2.latent(x1_warp, x2, exact=True, lo_abs=self._LO_ABS)
            z2 = _seq3_hi(self.h_a2, y2_lo)
            z2_strings = self.entropy_bottleneck2.compress(z2)
            z2_hat = self.entropy_bottleneck2.decompress(z2_strings, z2.size()[-2:]).to(cdt)
            x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)
            y1_hat_w = _round_latent(self.gaussian1, self.encoder1.latent(x1_hat_warp, want_lo=False)[1])
            y2_hat = _round_latent(self.gaussian_conditional2, y2)
            sc2, mu2 = self._gauss_full(2, self._params_view(2, z2_hat), y2_hat, y1_hat_w)
            head = bytearray(np.array(x1.shape[2:], dtype=np.uint16).tobytes())
            enc = RangeEncoder()
            bound = self.gaussian_conditional1._bound()
            for y_hat, z_strings, sc, mu in ((y1_hat, z1_strings, sc1, mu1), (y2_hat, z2_strings, sc2, mu2)):
                hv, minmax, channels = self._header_view(y_hat, z_strings)
                head += hv
                if not channels:
                    continue
                H, W = y_hat.shape[-2:]
                n_tab = 2 * minmax + 2
                if order == "raster":
                    rows = max(1, self._TABLE_BYTES // (len(channels) * W * n_tab * 4))
                    for r0 in range(0, H, rows):                     # row blocks keep the table buffers bounded
                        r1 = min(H, r0 + rows)
                        cdf = Fn.gmm_cdf_tables(sc[:, :, r0:r1], mu[:, :, r0:r1], None, channels, minmax, 1, scale_bound=bound)
                        cdf = cdf.cpu().numpy().view(np.uint32).transpose(1, 2, 0, 3).reshape(-1, len(channels), n_tab)
                        sym = y_hat[0, channels, r0:r1].float().cpu().numpy().astype(np.int64) + minmax
                        sym = sym.transpose(1, 2, 0).reshape(-1, len(channels)).astype(np.int32)
                        enc.encode(sym.reshape(-1), np.ascontiguousarray(cdf.reshape(-1, n_tab)))
                else:
                    # wavefront order: the pixels enter the coder group by group of equal t = w + 3 h.  Bands of whole groups, each at most
                    # _TABLE_BYTES of tables: their (scale, mean) rows are gathered on the device, ONE table launch per band (the table
                    # kernel is element-wise, so the numbers are those of the whole-map evaluation), one host copy -- host memory is
                    # bounded by the band, whatever the image size or the latent range (round 3 built the whole map's tables at once)
                    groups = self._wavefronts(H, W)
                    cap = max(1, self._TABLE_BYTES // (len(channels) * n_tab * 4))
                    sc_rows, mu_rows, y_rows = _nhwc_rows(sc), _nhwc_rows(mu), _nhwc_rows(y_hat)
                    ch_t = torch.as_tensor(channels, device=y_hat.device)
                    g0 = 0
                    while g0 < len(groups):
                        g1, P = g0, 0
                        while g1 < len(groups) and (P == 0 or P + len(groups[g1]) <= cap):
                            P += len(groups[g1])
                            g1 += 1
                        pix = torch.from_numpy(np.concatenate(groups[g0:g1])).to(y_hat.device)
                        sc_r, mu_r = (t[pix].contiguous().t().reshape(1, self.M, 1, P) for t in (sc_rows, mu_rows))
                        cdf = Fn.gmm_cdf_tables(sc_r, mu_r, None, channels, minmax, 1, scale_bound=bound)               # (C, 1, P, n)
                        cdf = cdf.cpu().numpy().view(np.uint32).reshape(len(channels), P, n_tab).transpose(1, 0, 2)       # pixel-major
                        sym = (y_rows[pix][:, ch_t].float().cpu().numpy().astype(np.int64) + minmax).astype(np.int32)      # (P, C)
                        enc.encode(sym.reshape(-1), np.ascontiguousarray(cdf).reshape(-1, n_tab))
                        g0 = g1
        payload = payload_header(bytes([self.ORDER_WAVEFRONT if order == "wavefront" else self.ORDER_RASTER])) + enc.finish()
        with open(os.path.join(output_path, str(output_name) + ".npz"), "wb") as f:
            f.write(bytes(head))
        with open(os.path.join(output_path, str(output_name) + ".bin"), "wb") as f:
            f.write(payload)
        num_pixels = x1.shape[2] * x1.shape[3] * 2
        return {"bpp_real": (len(head) + len(payload)) * 8 / num_pixels, "bpp_side": len(head) * 8 / num_pixels,
                "enctime": time.time() - start, "y1_hat": y1_hat, "y2_hat": y2_hat, "z1_hat": z1_hat, "z2_hat": z2_hat}

    # ---- wavefront decode with the device work of a group as a replayed HIP graph (round 4).  Round 3 issued every group's ~15 device
    # operations from Python (gathers, masked conv, cat, 1x1 net, layout copies, table kernel: ~270 us per group, 250 groups per 512^2
    # pair = 0.067 s).  Here the state of the walk lives on the device (csrc/glue.hip: hesic_joint_step -- position in the index tables,
    # the previous group's symbols and rows, the crop / feature gathers in one block), so ONE five-node graph per group size (joint_step,
    # masked conv into its slice of the feature rows, three 1x1 layers) serves every step of every image of that size; the host's share of
    # a step is: copy the previous symbols up, replay, launch the table kernel (its width depends on the image's latent range; it reads
    # (scale, mean) straight from the net's fp32 output rows), copy the tables down, range-decode.  Graphs and their static buffers are
    # cached on the module per (view, map size, format, weights).
    def _wavefront_state(self, which, yh, yw, dev):
        import numpy as np
        cdt = Fn.compute_dtype()
        ctx_m = self.context_prediction1 if which == 1 else self.context_prediction2
        ep = self.entropy_parameters1 if which == 1 else self.entropy_parameters2
        tag = (Fn._cache_epoch, cdt) + tuple((p_.data_ptr(), p_._version) for p_ in list(ctx_m.parameters()) + list(ep.parameters()))
        cache = self.__dict__.setdefault("_wf_cache", {})
        key = (which, yh, yw, dev.index)
        st = cache.get(key)
        if st is not None and st["tag"] == tag:
            return st
        if not hasattr(ctx_m, "_packer"):
            ctx_m._packer = Fn.PackedWeight()
        M, Wp = self.M, yw + 4
        groups = self._wavefronts(yh, yw)
        pmax = max(len(g) for g in groups)
        all_pix = np.concatenate(groups)
        centre = (all_pix // yw + 2) * Wp + (all_pix % yw) + 2
        c_par = (self.h_s1 if which == 1 else self.h_s2)[4].out_channels
        c_feat = c_par + ctx_m.out_channels + (M if which == 2 else 0)
        st = {"tag": tag, "groups": [len(g) for g in groups], "pmax": pmax, "ctx_m": ctx_m, "ep": ep, "graphs": {}, "Wp": Wp,
              "c_par": c_par, "c_feat": c_feat, "e_off": c_par + ctx_m.out_channels,
              "y_pad": torch.zeros((1, M, yh + 4, Wp), dtype=cdt, device=dev).contiguous(memory_format=torch.channels_last),
              "par": torch.zeros((yh * yw, c_par), dtype=cdt, device=dev),
              "ext": torch.zeros((yh * yw, M), dtype=cdt, device=dev) if which == 2 else None,
              "all_centre": torch.from_numpy(centre.astype(np.int64)).to(dev), "all_rows": torch.from_numpy(all_pix.astype(np.int64)).to(dev),
              "pos": torch.zeros(1, dtype=torch.int64, device=dev), "cfg": torch.zeros(4 + M, dtype=torch.int32, device=dev),
              "cfg_pin": torch.zeros(4 + M, dtype=torch.int32).pin_memory(), "tab": None, "tab_in_graph": False,
              "sym": torch.zeros(pmax * M, dtype=torch.int32, device=dev),
              "prev_centre": torch.zeros(pmax, dtype=torch.int64, device=dev),
              "crops": torch.zeros((pmax, 5, 5, M), dtype=cdt, device=dev), "feat": torch.zeros((pmax, c_feat), dtype=cdt, device=dev),
              "sym_pin": torch.zeros(pmax * M, dtype=torch.int32).pin_memory()}
        st["y_flat"] = st["y_pad"].permute(0, 2, 3, 1).reshape((yh + 4) * Wp, M)
        st["state"], st["channels"] = st["cfg"][:3], st["cfg"][4:]          # {nprev, C, minmax} and the coded channels: one upload per view
        cache[key] = st
        return st

    def _wavefront_kernel(self, st, P):
        import ctypes as C
        sym = C.c_void_p(st["sym_pin"].data_ptr()) if WAVEFRONT_ZEROCOPY else L.ptr(st["sym"])      # pinned host memory is device-addressable
        L.call("hesic_joint_step", L.ptr(st["y_flat"]), L.dt(st["y_flat"]), self.M, st["Wp"], sym, L.ptr(st["prev_centre"]), L.ptr(st["state"]),
               L.ptr(st["channels"]), L.ptr(st["all_centre"]), L.ptr(st["all_rows"]), L.ptr(st["pos"]), int(P), L.ptr(st["crops"]), L.ptr(st["par"]),
               st["c_par"], L.ptr(st["ext"]), st["e_off"], L.ptr(st["feat"]), st["c_feat"], L.stream())

    def _wavefront_step_body(self, st, P):
        """The device work of one group of P pixels up to the fp32 (scale | mean) rows of the entropy-parameter net -- five launches,
        what a graph of size P replays."""
        self._wavefront_kernel(st, P)
        ctx_m = st["ctx_m"]
        crops = st["crops"][:P].permute(0, 3, 1, 2)                                                        # (P, M, 5, 5), NHWC in memory
        feat = st["feat"][:P].view(P, st["c_feat"], 1, 1)
        Fn.conv2d_into(crops, ctx_m.weight, ctx_m.bias, feat, st["c_par"], kernel_size=5, stride=1, padding=0, mask=ctx_m.mask,
                       tap_mask=ctx_m._tap_mask, packer=ctx_m._packer)
        sm = _seq3_hi(st["ep"], feat)                                                                      # (P, 2M, 1, 1) fp32: row p = [scales(M) | means(M)]
        if st["tab_in_graph"]:
            # sixth node: the group's tables, channel list / alphabet read from the device state (any image replays the same graph)
            import ctypes as C
            M = self.M
            d = L.GmmDesc(1, P, M, 1, L.F32, 0, 2 * M, 0, M, float(st["bound"]), 0.0)
            tab = st["tab"][1] if WAVEFRONT_ZEROCOPY else st["tab"][0]
            L.call("hesic_gmm_cdf_dyn", C.byref(d), 0, L.ptr(sm), L.ptr(sm), None, L.ptr(st["channels"]), M, L.ptr(st["state"]), 1,
                   C.c_void_p(tab.data_ptr()), L.stream())
        return sm

    def _wavefront_graph(self, st, P):
        ent = st["graphs"].get(P)
        if ent is None:
            keep = {k: st[k].clone() for k in ("pos", "state", "prev_centre", "y_pad")}
            side = torch.cuda.Stream(device=st["pos"].device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # warm-up off the default stream (packs weights, loads kernels); state restored below
                for _ in range(2):
                    self._wavefront_step_body(st, P)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # the launches are also RECORDED while they are captured (entry point + arguments): the graph's private pool keeps every
            # buffer they name alive, so the same launches can be replayed one by one from C (hesic_joint_decode_groups_tape)
            rec = []
            with L.call_hook(lambda name, a: rec.append((name, a))), torch.cuda.graph(g, capture_error_mode="thread_local"):     # this thread's launches only
                out = self._wavefront_step_body(st, P)
            for k, v in keep.items():
                st[k].copy_(v)
            n_step = len(rec) - (1 if st["tab_in_graph"] else 0)          # the table launch (if captured) is issued by the walk itself on the tape path
            try:
                tape = L.tape_from_calls(rec[:n_step])
            except (KeyError, TypeError):
                tape = None                                               # an entry point the tape does not know: graph replay only
            ent = st["graphs"][P] = (g, out, tape)
        return ent

    def _decode_view_graphed(self, dec, which, params, minmax, channels, extra, yh, yw, bound):
        import ctypes as C
        import numpy as np
        dev = params.device
        st = self._wavefront_state(which, yh, yw, dev)
        M, Cn, n_tab = self.M, len(channels), 2 * minmax + 2
        if not st["graphs"]:
            st["tab_in_graph"], st["bound"] = bool(WAVEFRONT_TABLE_IN_GRAPH), float(bound)
            if st["tab_in_graph"]:                              # the graphs hold the table buffer's address: sized once for the largest alphabet they take
                rows = M * st["pmax"]
                st["tab"] = (torch.empty((rows, _CDF_WAVE_MAX + 1), dtype=torch.int32, device=dev),
                             torch.empty((rows, _CDF_WAVE_MAX + 1), dtype=torch.int32).pin_memory())
        for P in sorted(set(st["groups"])):                    # first use of a map size: capture (cached on the module)
            self._wavefront_graph(st, P)
        st["y_pad"].zero_()
        st["par"].copy_(_nhwc_rows(params))
        if extra is not None:
            st["ext"].copy_(_nhwc_rows(extra))
        st["pos"].zero_()
        cfg = st["cfg_pin"]
        cfg[0], cfg[1], cfg[2] = 0, Cn, int(minmax)
        cfg[4:4 + Cn] = torch.as_tensor(channels, dtype=torch.int32)
        st["cfg"].copy_(cfg, non_blocking=True)
        ch_dev = st["channels"][:Cn]
        pmax = st["pmax"]
        in_graph = st["tab_in_graph"] and 2 * minmax + 1 <= _CDF_WAVE_MAX and float(bound) == st["bound"]
        if in_graph:
            tab_dev, tab_pin = st["tab"]
        else:
            # table buffers (device + pinned) kept with the state: pinning costs ~0.1 ms a call.  Not the ones a captured table launch
            # writes: an alphabet beyond its 1024 entries (random-weight regimes) goes through launches issued per group
            big = st.get("tab_big")
            if big is None or big[0].shape[1] < n_tab:
                big = st["tab_big"] = (torch.empty((M * pmax, n_tab), dtype=torch.int32, device=dev),
                                       torch.empty((M * pmax, n_tab), dtype=torch.int32).pin_memory())
            tab_dev, tab_pin = big
        descs = {P: L.GmmDesc(1, P, M, 1, L.F32, 0, 2 * M, 0, M, float(bound), 0.0) for P in set(st["groups"])}
        stream = L.stream()
        sym_dev, sym_host = L.ptr(st["sym"]), C.c_void_p(st["sym_pin"].data_ptr())
        tab_d, tab_h = L.ptr(tab_dev), C.c_void_p(tab_pin.data_ptr())
        zc = WAVEFRONT_ZEROCOPY
        if zc:                                  # same address on both sides: the copies fall away (hesic_joint_decode_groups skips them too)
            sym_dev, tab_d = sym_host, tab_h
        ch_p = L.ptr(ch_dev)
        groups = st["groups"]
        if WAVEFRONT_C_LOOP:
            # the whole walk in one C call (hesic_joint_decode_groups): per group the captured device step, the table launch, the two copies,
            # a polled wait and the host range decoder through its C entry point -- no Python between the groups
            n = len(groups)
            fn, handle = dec.grid_callback()
            outs = (C.c_void_p * n)(*[st["graphs"][P][1].data_ptr() for P in groups])
            sizes = (C.c_int32 * n)(*groups)
            use_tape = WAVEFRONT_TAPE and all(st["graphs"][P][2] is not None for P in set(groups))
            try:
                if use_tape:
                    # recorded launches replayed one by one; the table launch is then always issued by the walk (never the captured one)
                    tapes = (C.c_void_p * n)(*[C.addressof(st["graphs"][P][2][0]) for P in groups])
                    lens = (C.c_int32 * n)(*[len(st["graphs"][P][2][0]) for P in groups])
                    dsc = (L.GmmDesc * n)(*[descs[P] for P in groups])
                    L.call("hesic_joint_decode_groups_tape", n, sizes, tapes, lens, dsc, outs, ch_p, Cn, int(minmax), tab_d, tab_h, sym_dev, sym_host,
                           fn, handle, 1, stream)
                else:
                    execs = (C.c_void_p * n)(*[st["graphs"][P][0].raw_cuda_graph_exec() for P in groups])
                    dsc = None if in_graph else (L.GmmDesc * n)(*[descs[P] for P in groups])
                    L.call("hesic_joint_decode_groups", n, sizes, execs, dsc, outs, ch_p, Cn, int(minmax), tab_d, tab_h, sym_dev, sym_host, fn, handle,
                           1, stream)
            except RuntimeError as e:
                if "range decoder failed" in str(e):
                    raise ValueError("RangeDecoder.decode_grid: bad table") from e
                raise
        else:
            # per step: copy the previous symbols up, replay, table launch, copy the tables down, wait, decode
            call, raw = L.call, dec.decode_grid_raw
            nprev = 0
            for P in groups:
                if nprev and not zc:
                    call("hesic_memcpy_async", sym_dev, sym_host, nprev * Cn * 4, 1, stream)
                g, sm = st["graphs"][P][:2]
                g.replay()
                smp = L.ptr(sm)
                if not in_graph:
                    call("hesic_gmm_cdf_rows", C.byref(descs[P]), 0, smp, smp, None, ch_p, Cn, int(minmax), 1, tab_d, stream)
                if not zc:
                    call("hesic_memcpy_async", tab_h, tab_d, Cn * P * n_tab * 4, 2, stream)
                call("hesic_stream_synchronize", stream)
                raw(tab_h, n_tab, P, Cn, Cn, 1, sym_host)                                            # pixel-major rows, (P, Cn) symbols into the pinned buffer
                nprev = P
            if not zc:
                call("hesic_memcpy_async", sym_dev, sym_host, nprev * Cn * 4, 1, stream)
        self._wavefront_kernel(st, 0)                              # the last group's symbols
        return st["y_pad"][:, :, 2:-2, 2:-2].contiguous(memory_format=torch.channels_last)

    def decompress(self, x1, x2, h_matrix, output_name, output_path="", device=None):
        """x1 / x2 are unused (the header carries the size); kept for the reference's signature."""
        import os
        import time
        import numpy as np
        from ._host import RangeDecoder
        if self.entropy_bottleneck1._offset.numel() == 0:
            self.update()
        dev = h_matrix.device
        with open(os.path.join(output_path, str(output_name) + ".npz"), "rb") as f:
            blob = f.read()
        x_shape = np.frombuffer(blob[:4], dtype=np.uint16).astype(int)
        pos, views = 4, []
        for _ in range(2):
            length, minmax = (int(v) for v in np.frombuffer(blob[pos:pos + 4], dtype=np.uint16))
            pos += 4
            flag = np.unpackbits(np.frombuffer(blob[pos:pos + (self.M + 7) // 8], dtype=np.uint8))[:self.M]
            pos += (self.M + 7) // 8
            views.append((minmax, [int(c) for c in np.nonzero(flag)[0]], blob[pos:pos + length]))
            pos += length
        yh, yw = int(x_shape[0]) // 16, int(x_shape[1]) // 16
        size = (int(x_shape[0]), int(x_shape[1]))
        with open(os.path.join(output_path, str(output_name) + ".bin"), "rb") as f:
            payload = f.read()
        extra, off = check_payload(payload, 1)
        if extra[0] not in (self.ORDER_RASTER, self.ORDER_WAVEFRONT):
            raise ValueError("decompress: not a HESIC+ payload of this coder (unknown pixel-order byte)")
        wavefront = extra[0] == self.ORDER_WAVEFRONT
        dec = RangeDecoder(payload[off:])
        cdt = Fn.compute_dtype()
        bound = self.gaussian_conditional1._bound()
        start = time.time()

        def decode_view(which, params, minmax, channels, extra=None):
            """Raster payload: the reference's walk, one pixel per step (newnet1_joint.py:1190-1260).  Wavefront payload: one step
            per group of mutually independent pixels -- their 5x5 crops gathered into one batch, ONE masked-conv launch, ONE pass
            of the 1x1 entropy-parameter net, ONE table launch and ONE device -> host copy for the whole group."""
            y_pad = torch.zeros((1, self.M, yh + 4, yw + 4), dtype=cdt, device=dev).contiguous(memory_format=torch.channels_last)
            if not channels:
                return y_pad[:, :, 2:-2, 2:-2].contiguous(memory_format=torch.channels_last)
            ch_t = torch.as_tensor(channels, device=dev)
            if not wavefront:
                for h in range(yh):
                    for w in range(yw):
                        sc, mu = self._gauss_pixel(which, params, y_pad, h, w, extra)
                        cdf = Fn.gmm_cdf_tables(sc, mu, None, channels, minmax, 1, scale_bound=bound)
                        sym = dec.decode(cdf.cpu().numpy().view(np.uint32).reshape(len(channels), -1))
                        y_pad[0, ch_t, h + 2, w + 2] = torch.from_numpy(sym.astype(np.float32) - minmax).to(dev, cdt)
                return y_pad[:, :, 2:-2, 2:-2].contiguous(memory_format=torch.channels_last)
            if WAVEFRONT_GRAPHS and dev.type == "cuda":
                return self._decode_view_graphed(dec, which, params, minmax, channels, extra, yh, yw, bound)
            ctx_m = self.context_prediction1 if which == 1 else self.context_prediction2
            ep = self.entropy_parameters1 if which == 1 else self.entropy_parameters2
            if not hasattr(ctx_m, "_packer"):
                ctx_m._packer = Fn.PackedWeight()
            Wp = yw + 4
            y_flat = y_pad.permute(0, 2, 3, 1).reshape((yh + 4) * Wp, self.M)            # views of the NHWC storage: one row per padded pixel
            par_flat = _nhwc_rows(params)
            ext_flat = None if extra is None else _nhwc_rows(extra)
            groups = self._wavefronts(yh, yw)
            # gather indices of every group, uploaded once: the 25 padded-map rows of each pixel's crop, its own padded row, its raster row
            win = (np.arange(5)[:, None] * Wp + np.arange(5)[None, :]).reshape(-1)
            all_pix = np.concatenate(groups)
            top_left = (all_pix // yw) * Wp + (all_pix % yw)
            crop_idx = torch.from_numpy((top_left[:, None] + win[None, :]).reshape(-1)).to(dev)
            centre_idx = torch.from_numpy(top_left + 2 * Wp + 2).to(dev)
            pix_idx = torch.from_numpy(all_pix).to(dev)
            pos = 0
            for grp in groups:
                P = len(grp)
                crops = y_flat[crop_idx[pos * 25:(pos + P) * 25]].view(P, 5, 5, self.M).permute(0, 3, 1, 2)      # (P, M, 5, 5), NHWC in memory
                ctx = Fn.conv2d(crops, ctx_m.weight, ctx_m.bias, kernel_size=5, stride=1, padding=0, mask=ctx_m.mask,
                                tap_mask=ctx_m._tap_mask, packer=ctx_m._packer)                                    # (P, 2M, 1, 1)
                rows = pix_idx[pos:pos + P]
                parts = [par_flat[rows], ctx.reshape(P, -1)]
                if ext_flat is not None:
                    parts.append(ext_flat[rows])
                feat = torch.cat(parts, 1)
                sc, mu = _seq3_hi(ep, feat.view(P, feat.shape[1], 1, 1).contiguous(memory_format=torch.channels_last)).chunk(2, 1)
                # (P, M, 1, 1) -> one image row of P pixels for the table kernel: the same P x M memory
                sc_r, mu_r = (t.reshape(P, self.M).contiguous().t().reshape(1, self.M, 1, P) for t in (sc, mu))
                cdf = Fn.gmm_cdf_tables(sc_r, mu_r, None, channels, minmax, 1, scale_bound=bound)               # (C, 1, P, n)
                tab = cdf.cpu().numpy().view(np.uint32).reshape(len(channels) * P, -1)                 # channel-major rows, as the kernel wrote them
                sym = dec.decode_grid(tab, P, len(channels), 1, P)                                       # pixel-major stream order, no host transpose
                vals = torch.from_numpy(sym.astype(np.float32) - minmax).to(dev, cdt)
                y_flat[centre_idx[pos:pos + P].unsqueeze(1), ch_t.unsqueeze(0)] = vals
                pos += P
            return y_pad[:, :, 2:-2, 2:-2].contiguous(memory_format=torch.channels_last)

        with torch.no_grad(), Fn.no_split_k():
            self.context_prediction1.weight.data *= self.context_prediction1.mask
            self.context_prediction2.weight.data *= self.context_prediction2.mask
            zs = (yh // 4, yw // 4)
            z1_hat = self.entropy_bottleneck1.decompress([views[0][2]], zs).to(dev, cdt)
            z2_hat = self.entropy_bottleneck2.decompress([views[1][2]], zs).to(dev, cdt)
            y1_hat = decode_view(1, self._params_view(1, z1_hat), views[0][0], views[0][1])
            x1_hat = self.decoder1(y1_hat)
            x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)
            y1_hat_w = _round_latent(self.gaussian1, self.encoder1.latent(x1_hat_warp, want_lo=False)[1])
            y2_hat = decode_view(2, self._params_view(2, z2_hat), views[1][0], views[1][1], y1_hat_w)
            x2_hat = self.decoder2(y2_hat, x1_hat_warp)
        return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat, "z1_hat": z1_hat, "z2_hat": z2_hat,
                "dectime": time.time() - start}


# -------------------------------------------------------------------- enhancement
class Enhancement_Block(nn.Module):
    """Three residual blocks with an outer skip (newnet1.py:272-286)."""

    def __init__(self):
        super().__init__()
        self.RB1, self.RB2, self.RB3 = ResidualBlock(32, 32), ResidualBlock(32, 32), ResidualBlock(32, 32)

    def forward(self, x):
        return self.RB3(self.RB2(self.RB1(x)), outer_skip=x)


class Enhancement(nn.Module):
    """conv3x3(6->32) -> 3 Enhancement_Blocks -> conv3x3(32->3) -> + x (newnet1.py:288-311)."""

    def __init__(self):
        super().__init__()
        self.conv1 = conv3x3(6, 32)
        self.EB1, self.EB2, self.EB3 = Enhancement_Block(), Enhancement_Block(), Enhancement_Block()
        self.conv2 = conv3x3(32, 3)

    def forward(self, x, x_another_warp):
        if (Fn.conv3x3_c32_ok(x.new_empty((1, 32, 1, 1), dtype=Fn._h16()), self.EB1.RB1.conv1.weight) and x.is_cuda
                and tuple(self.conv1.weight.shape) == (32, 6, 3, 3)):
            # inference: the 6 -> 32 input conv reads the two planar images directly (round 6: one launch instead of pack_images_c32 + the
            # 32-channel kernel on a zero-padded map and weight; same numbers)
            t = Fn.conv3x3_c32_img6(x, x_another_warp, self.conv1.weight, self.conv1.bias)
        elif (x.is_cuda and not x.requires_grad and not x_another_warp.requires_grad and tuple(self.conv1.weight.shape) == (32, 6, 3, 3)
              and Fn.conv3x3_c32_train_ok(x.new_empty((1, 32, 1, 1), dtype=Fn._h16()), self.conv1.weight)):
            # stage-2 training (HSIC frozen, so the images carry no gradient): the same packed 32-channel input map; the 6-input-channel
            # weight is zero-padded inside the op, its gradient comes back for the six real channels
            t = Fn.conv3x3_c32_train(Fn.pack_images_c32(x, x_another_warp), self.conv1.weight, self.conv1.bias)
        else:
            t = _ho.plain(self.conv1(torch.cat((x.float(), x_another_warp.float()), 1)))
        t = self.EB3(self.EB2(self.EB1(t)))
        if Fn.conv3x3_c32_ok(t, self.conv2.weight):          # 32 -> 3 output conv + the image it refines, fp32 planar out
            return Fn.conv3x3_c32(t, self.conv2.weight, self.conv2.bias, res1=x)
        if (Fn.EN_TRAIN_FAST and torch.is_grad_enabled() and t.is_cuda and t.dtype == Fn._h16() and t.shape[1] == 32
                and tuple(self.conv2.weight.shape) == (3, 32, 3, 3)):
            return Fn.conv3x3_c32_out_train(t, self.conv2.weight, self.conv2.bias, x.float())
        return self.conv2(t) + x


class Independent_EN(nn.Module):
    """Stage-2 cross-view enhancement (newnet1.py:1278-1300): view 1 is refined with view 2 warped by H^-1,
    view 2 with view 1 warped by H.  Warping by H^-1 needs no inversion at all: the warp kernel wants the destination ->
    source map, which for H^-1 is H itself (the reference's torch.inverse + kornia's own inverse cancel)."""

    def __init__(self):
        super().__init__()
        self.EH1, self.EH2 = Enhancement(), Enhancement()

    def forward(self, x1_hat, x2_hat, h_matrix):
        size = (x1_hat.shape[-2], x1_hat.shape[-1])
        x1_hat_warp = warp_perspective(x1_hat, h_matrix, size)
        x2_hat_warp = warp_perspective(x2_hat, h_matrix, size, inverse_map=True)   # warp by inverse(H) (:1290): H is its inverse map
        return {"x1_hat": self.EH1(x1_hat, x2_hat_warp), "x2_hat": self.EH2(x2_hat, x1_hat_warp)}


class GMM_together(nn.Module):
    """HESIC followed by the enhancement stage (newnet1.py:1304-1321) -- what the README's test script evaluates."""

    def __init__(self, N=128, M=192, K=5, **kwargs):
        super().__init__()
        self.m1 = HSIC(N, M, K)
        self.m2 = Independent_EN()

    def forward(self, x1, x2, h):
        out1 = self.m1(x1, x2, h)
        out2 = self.m2(out1["x1_hat"], out1["x2_hat"], h)
        return {"x1_hat": out2["x1_hat"], "x2_hat": out2["x2_hat"], "likelihoods": out1["likelihoods"]}


# ------------------------------------------------------------------------------------------ metrics
def pad_to_multiple(x, multiple=64):
    """Zero-pad H, W (bottom / right) up to a multiple of 64: the hyper path needs it (the reference raises on e.g.
    860x1080); pixel coordinates -- hence the homography -- are unchanged."""
    h, w = x.shape[-2:]
    ph, pw = (-h) % multiple, (-w) % multiple
    return x if ph == 0 and pw == 0 else torch.nn.functional.pad(x, (0, pw, 0, ph))


class GraphedForward:
    """The eval forward (+ the bits / squared-error reductions) captured once into a HIP graph and replayed.

    The whole path is capturable: every launch goes to the current stream, the side streams of the multi-stream schedule
    fork from / join to the capture stream, nothing reads back to the host (the scale bound is cached on the host side) and
    all scratch comes from the caching allocator.  Replay removes the per-launch host work (~95 launches per forward); on
    a host that already keeps the GPU fed it changes nothing (2.57 vs 2.58 ms at B=8, 512x512), on a slower or busier
    host it is the difference between a launch-bound and a GPU-bound step.  Inputs are copied into static buffers; the
    returned tensors are the graph's static outputs (overwritten by the next call)."""

    def __init__(self, net, x1, x2, h_matrix, with_metrics=True, warmup=3):
        if net.training:
            raise RuntimeError("GraphedForward captures the inference schedule: call net.eval() first")
        self.net, self.with_metrics = net, with_metrics
        self.x1, self.x2, self.h = x1.clone(), x2.clone(), h_matrix.clone()
        side = torch.cuda.Stream(device=x1.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):      # warm-up off the default stream: packs weights, fills caches
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # "thread_local": a process group's watchdog thread may poll events while this thread captures (see train.GraphedTrainer)
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = self._run()

    def _run(self):
        out = self.net(self.x1, self.x2, self.h)
        return (out, rate_distortion(out, self.x1, self.x2)) if self.with_metrics else (out, None)

    def __call__(self, x1=None, x2=None, h_matrix=None):
        for dst, src in ((self.x1, x1), (self.x2, x2), (self.h, h_matrix)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.out


class SegmentedForward:
    """The eval forward as a PLAN of single-stream HIP graphs -- the eager schedule with its host cost removed.

    ``GraphedForward`` captures the whole multi-stream forward into one graph; the runtime then orders that graph's parallel branches
    its own way, which costs 5 - 8 % against eager issue at 8 x 512^2.  Here every stretch of a stream between
    two synchronisation points of the schedule is its own graph (a plain chain of kernel nodes: nothing left to reorder) and the
    synchronisation points are replayed as what they are -- event records and waits on the schedule's own streams, in the schedule's
    own issue order.  A replay is ~20 graph launches + ~15 event operations from the host (~0.15 ms) instead of ~75 ctypes launches
    (~1.1 ms), and the GPU sees the eager timeline.  Same call signature and result convention as ``GraphedForward``: static input
    buffers, static outputs (overwritten by the next call), bit-identical to the eager forward (tested)."""

    def __init__(self, net, x1, x2, h_matrix, with_metrics=False, warmup=3):
        if net.training:
            raise RuntimeError("SegmentedForward captures the inference schedule: call net.eval() first")
        if with_metrics:
            raise ValueError("SegmentedForward: reduce the metrics behind the replay (models.rate_distortion on the static outputs)")
        if _seg_rec_get() is not None:
            raise RuntimeError("SegmentedForward: another build is in progress on this thread")
        self.net = net
        self.x1, self.x2, self.h = x1.clone(), x2.clone(), h_matrix.clone()
        dev = x1.device
        self.main = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream()
        self.main.wait_stream(cur)
        with torch.no_grad(), torch.cuda.stream(self.main):      # eager warm-up: packs weights, fills caches, settles the side streams
            for _ in range(warmup):
                net(self.x1, self.x2, self.h)
        torch.cuda.synchronize(dev)
        rec = _SegmentRecorder()
        _seg_tls.rec = rec
        try:
            with torch.no_grad(), torch.cuda.stream(self.main), L.call_hook(lambda name, a: rec.touch()):      # thread-local: see _lib.call_hook
                rec._open(self.main)
                self.out = net(self.x1, self.x2, self.h)
                self.plan = rec.finish()
        except BaseException:
            for st, g, _ in list(rec.open.values()):             # leave no stream in capture mode behind
                try:
                    with torch.cuda.stream(st):
                        g.capture_end()
                except Exception:
                    pass
            raise
        finally:
            _seg_tls.rec = None
        cur.wait_stream(self.main)
        self.n_graphs = sum(1 for e in self.plan if e[0] == "launch")

    def __call__(self, x1=None, x2=None, h_matrix=None):
        for dst, src in ((self.x1, x1), (self.x2, x2), (self.h, h_matrix)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        cur = torch.cuda.current_stream()
        self.main.wait_stream(cur)
        for op, st, arg in self.plan:
            if op == "launch":
                with torch.cuda.stream(st):
                    arg[0].replay()
            elif op == "record":
                arg.record(st)
            else:
                st.wait_event(arg)
        cur.wait_stream(self.main)
        return self.out, None


def _clone_out(o):
    if torch.is_tensor(o):
        return o.clone()
    if isinstance(o, dict):
        return {k: _clone_out(v) for k, v in o.items()}
    if isinstance(o, (tuple, list)):
        return type(o)(_clone_out(v) for v in o)
    return o


class AutoForward:
    """``net(x1, x2, h)`` issued the faster way for THIS host, model and input size: eager launches (the host pays ~1.1 ms per
    forward: fine when the GPU needs longer) or a ``GraphedForward`` replay (no host cost, but the runtime orders a graph's parallel
    branches its own way: ~7 % slower than eager issue for HESIC at 8 x 512^2, faster for small batches).  The choice is made once,
    by timing ``trial`` forwards of each on the example inputs; inputs of other shapes fall back to eager.  ``mode`` reports the
    choice ("eager" | "graph"), ``timings`` both per-forward times in ms.

    A drop-in for ``net(x1, x2, h)``: the result is the caller's own in both modes -- in graph mode the graph's static output
    buffers are CLONED before they are returned (``static_outputs=True`` hands out the buffers themselves, which the next call
    overwrites: for loops that consume a result before asking for the next).  The graph holds the packed weights of the moment
    of capture: every call checks the parameters' (storage, version) tags and ``Fn``'s cache epoch and re-captures when a
    ``load_state_dict`` / optimiser step / ``invalidate_weight_cache()`` moved them.  When eager wins the trial the graph and its
    memory pool are dropped."""

    def __init__(self, net, x1, x2, h_matrix, trial=30, static_outputs=False):
        import time
        if net.training:
            raise RuntimeError("AutoForward wraps the inference schedule: call net.eval() first")
        self.net, self.static_outputs = net, static_outputs
        self._slots, self._calls = None, 0
        self._graph = GraphedForward(net, x1, x2, h_matrix, with_metrics=False)
        self._tag = self._weights_tag()

        def eager():
            with torch.no_grad():
                return net(x1, x2, h_matrix)

        def timed(fn):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(trial):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / trial * 1e3

        self.timings = {"eager": timed(eager), "graph": timed(self._graph)}
        self.mode = "graph" if self.timings["graph"] < self.timings["eager"] else "eager"
        self._shape = (tuple(x1.shape), tuple(x2.shape), tuple(h_matrix.shape))
        if self.mode == "eager":
            self._graph = None                     # frees the captured graph and its private memory pool

    def _weights_tag(self):
        """(cache epoch, storage format, analysis mode, (identity, storage, version) of every parameter / buffer).  What is cached is the list of
        SLOTS -- (a module's ``_parameters`` / ``_buffers`` dict, name) -- not the tensors: every call looks the current tensor up, so a replaced
        Parameter object (``load_state_dict(assign=True)``, ``mod.weight = nn.Parameter(...)``) is seen on the very next call and no old
        parameter is kept alive (the round-4 cache of the tensor list replayed stale packed weights for up to 255 calls).  The slot
        list itself (modules added or removed) is rebuilt every 256 calls; rebuilding ``state_dict`` per call cost ~0.3 ms."""
        self._calls += 1
        if self._slots is None or self._calls % 256 == 0:
            self._slots = [(d, n) for m in self.net.modules() for d in (m._parameters, m._buffers) for n in d]
        tag = []
        for d, n in self._slots:
            t = d.get(n)
            if t is not None:
                tag.append((id(t), t.data_ptr(), t._version))
        return (Fn._cache_epoch, Fn.compute_dtype(), Fn.analysis_precision(), tuple(tag))

    def __call__(self, x1, x2, h_matrix):
        if self.mode == "graph" and (tuple(x1.shape), tuple(x2.shape), tuple(h_matrix.shape)) == self._shape:
            tag = self._weights_tag()
            if self._graph is None or tag != self._tag:      # parameters changed since capture (the graph would replay stale packed weights), or no graph yet
                self._graph = GraphedForward(self.net, x1, x2, h_matrix, with_metrics=False)
                self._tag = tag
            out = self._graph(x1, x2, h_matrix)[0]
            return out if self.static_outputs else _clone_out(out)
        with torch.no_grad():
            return self.net(x1, x2, h_matrix)


MS_SSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def ms_ssim(x_hat, x, data_range=1.0):
    """Multi-scale SSIM per image, (N,) fp64 tensor on the device -- ``pytorch_msssim.ms_ssim(x_hat, x, data_range, size_average=False)``
    as the reference's evaluation calls it (ywz/mywork/test3real.py:107-109; the package is third party and absent: its published
    algorithm is restated in ``csrc/msssim.hip``, pinned against the oracle and an independent numpy route).  Five launches of
    ``hesic_ssim_scale`` with four 2 x 2 pools in between; no host synchronisation.  fp32 images of any strides; the smaller side must
    exceed 160 pixels (five scales of an 11-tap window)."""
    L.require_cuda(x_hat, x)
    if x_hat.shape != x.shape or x.dim() != 4:
        raise ValueError("ms_ssim: two (N, C, H, W) tensors of the same shape")
    if min(x.shape[-2:]) <= 160:
        raise ValueError("ms_ssim: the smaller image side must exceed (11 - 1) * 2^4 = 160")
    a, b = (t if t.dtype == torch.float32 else t.float() for t in (x_hat, x))
    B, Cc, H, W = a.shape
    sums = torch.empty((5, B, Cc, 2), dtype=torch.float64, device=a.device).fill_(0)
    counts = []
    import ctypes as C
    for lvl in range(5):
        sa, sb = (C.c_int64 * 4)(*a.stride()), (C.c_int64 * 4)(*b.stride())
        L.call("hesic_ssim_scale", L.ptr(a), sa, L.ptr(b), sb, B, Cc, H, W, float(data_range), L.ptr(sums[lvl]), L.stream())
        counts.append((H - 10) * (W - 10))
        if lvl < 4:
            Ho, Wo = (H + 2 * (H % 2) - 2) // 2 + 1, (W + 2 * (W % 2) - 2) // 2 + 1
            na, nb = (torch.empty((B, Cc, Ho, Wo), dtype=torch.float32, device=a.device) for _ in range(2))
            L.call("hesic_avgpool2_pad", L.ptr(a), sa, L.ptr(na), B, Cc, H, W, L.stream())
            L.call("hesic_avgpool2_pad", L.ptr(b), sb, L.ptr(nb), B, Cc, H, W, L.stream())
            a, b, H, W = na, nb, Ho, Wo
    means = sums / torch.tensor(counts, dtype=torch.float64, device=sums.device).reshape(5, 1, 1, 1)
    v = torch.cat((means[:4, :, :, 1], means[4:, :, :, 0]), 0).clamp_min(0)                 # cs of scales 1-4, ssim of scale 5
    w = torch.tensor(MS_SSIM_WEIGHTS, dtype=torch.float64, device=sums.device).reshape(5, 1, 1)
    return torch.prod(v ** w, 0).mean(1)


def ms_ssim_db(v):
    """-10 log10(1 - MS-SSIM): the scale of the reference's rate-distortion plots (Readme.md:46, cvpr-fix.png)."""
    return -10.0 * torch.log10((1.0 - v).clamp_min(1e-12))


def rate_distortion(out, x1, x2):
    """bits / squared error of one forward as fp64 device scalars (HIP reductions, no host sync):
    returns dict(bits_{y1,y2,z1,z2}, sse1, sse2, num_pixels)."""
    keys = list(out["likelihoods"].keys())
    acc = Fn._zeros(len(keys) + 2, torch.float64, x1.device)      # one zero-fill (a kernel, not a memset: graph-safe) for all six accumulators
    h, w = x1.shape[-2:]                      # x1/x2 are the ORIGINAL images: padded reconstructions are cropped (views)
    n = len(keys)

    def bits_sums():
        for i, k in enumerate(keys):
            Fn.sum_log2(out["likelihoods"][k], out=acc[i:i + 1])
        return acc

    def sq_sums():
        Fn.sum_sq_diff(out["x1_hat"][..., :h, :w], x1, out=acc[n:n + 1])
        Fn.sum_sq_diff(out["x2_hat"][..., :h, :w], x2, out=acc[n + 1:n + 2])
        return acc

    if x1.is_cuda and Fn.RD_SUMS_FUSED and n <= 8:
        # round 5: all six reductions in ONE launch (hesic_rd_sums)
        Fn.rd_sums([out["likelihoods"][k] for k in keys], [acc[i:i + 1] for i in range(n)],
                   [(out["x1_hat"][..., :h, :w], x1), (out["x2_hat"][..., :h, :w], x2)], [acc[n:n + 1], acc[n + 1:n + 2]])
    else:
        _branches(x1, bits_sums, sq_sums)         # six small-grid reductions: the two families side by side (disjoint accumulators)
    bits = -acc[:n]
    res = {"bits_" + k: bits[i:i + 1] for i, k in enumerate(keys)}
    res["sse1"], res["sse2"] = acc[n:n + 1], acc[n + 1:n + 2]
    res["num_pixels"] = x1.shape[0] * h * w    # bpp over the original pixel count
    return res


def metrics_from(rd, channels=3):
    """Reference conventions (ywz/mywork/test3real.py:69-72,110-122; newtrain1.py:141-142)."""
    n = rd["num_pixels"]
    bits = {k[5:]: float(v) for k, v in rd.items() if k.startswith("bits_")}
    mse1, mse2 = float(rd["sse1"]) / (n * channels), float(rd["sse2"]) / (n * channels)
    p1, p2 = 10 * math.log10(1 / mse1), 10 * math.log10(1 / mse2)
    bpp_loss = sum(bits.values()) / n
    return {"bits": bits, "bpp_loss": bpp_loss, "bpp": bpp_loss / 2, "mse1": mse1, "mse2": mse2, "psnr1": p1,
            "psnr2": p2, "psnr": (p1 + p2) / 2}
