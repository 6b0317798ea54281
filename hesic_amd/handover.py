"""Hand-over between consecutive modules of the drop-in package at inference (round 6).

The reference's model files (ywz/mywork/newnet1.py:590-601, :615-624, :641-655, :676-692, :433-437, :496-512, :562-577, :724-783;
newnet1_joint.py:611-665, :675-753) call one small module after the other -- ``conv`` then ``GDN`` then ``conv`` ..., ``torch.abs`` in
front of the hyper-analysis, ``nn.ReLU`` / ``nn.LeakyReLU`` between the hyper convs, ``torch.cat`` in front of the 6 -> 3 stages and the
entropy-parameter nets, ``chunk`` behind them.  Run literally, every one of those is a launch with a 16-bit tensor at its boundary: no
conv -> GDN fusion, no hi/lo pairs between the analysis layers, no fp32 latents for round() and the likelihoods -- three times slower
than ``hesic_amd.models``' own schedule and outside the 1e-3 tolerance in 16-bit storage.

So, with grad mode off on the GPU, a wide module of this package does not launch when it is called: it returns a ``Carrier`` -- a tensor
subclass with the logical NCHW shape, the storage dtype and channels_last strides the real output would have, and a node that says how to
produce it (``ConvNode``: this conv on that input with that activation).  The NEXT module of the package that receives it decides the
launch: ``GDN`` issues the fused conv + (I)GDN kernel, a conv asks its input for hi/lo pairs when the input is on the pair route of the
"x3" analysis mode, an entropy model asks for the fp32 accumulators.  ``torch.abs`` / ``relu`` / ``leaky_relu`` / ``cat`` / ``chunk`` /
bilinear x4 up-sampling on a Carrier are recorded as nodes too (they become ``in_abs`` / ``out_abs`` / ``act`` flags, in-place channel-slice
writes and the ``upsample4_cat`` kernel).  ANY other access -- an ATen operator this file does not know, printing, ``.cpu()`` --
materialises the plain 16-bit tensor first (``__torch_dispatch__``), so foreign code sees an ordinary tensor with the values the unfused
launch would have stored.  Nothing here computes: every number still comes from a kernel of ``libhesic_hip*.so`` through
``hesic_amd.functional``.

Which analysis passes run on pairs: the stack that starts from an image the caller made (``x1``, ``pre_gdn``'s output).  The output of the
image-side synthesis layers (``g_s_conv4``, ``after_conv``) carries a tag that ``warp_perspective`` hands on; an analysis stack that starts
from a tagged tensor -- the third pass ``encoder1(warp(x1_hat))`` (newnet1.py:753-757), whose rounded output is not transmitted -- runs on
single operands, as in ``models.HSIC._forward_eval``.  A tensor that lost its tag in foreign code takes the pair route (slower, never less
accurate).

``HESIC_NO_HANDOVER=1`` switches all of this off (every module launches when called: round 5's path A)."""
from __future__ import annotations

import os as _os
import weakref

import torch
from torch.utils._pytree import tree_map

from . import _lib as L
from . import functional as Fn

ENABLED = _os.environ.get("HESIC_NO_HANDOVER") is None
_FORCE_CPU = False          # tests/test_handover_cpu.py: exercise the node logic on CPU tensors with stubbed module launches
NONE, RELU, LEAKY = L.ACT_NONE, L.ACT_RELU, L.ACT_LEAKY
DECODED_TAG = "_hesic_decoded"
C2_TAG = "_hesic_c2"
SMOOTH_TAG = "_hesic_smooth"          # the output of a fused conv + (I)GDN launch: the conv that reads it packs error-feedback weights
aten = torch.ops.aten


def active(x):
    """True when module calls on ``x`` are deferred: inference (grad mode off) on the GPU."""
    return ENABLED and not torch.is_grad_enabled() and torch.is_tensor(x) and (x.is_cuda or _FORCE_CPU)


def is_carrier(x):
    return type(x) is Carrier


def tag_decoded(t):
    """Mark an image-side synthesis output (see the module docstring); returns ``t``."""
    try:
        setattr(t, DECODED_TAG, True)
    except Exception:
        pass
    return t


def inherit_tags(dst, src):
    if getattr(src, DECODED_TAG, False):
        tag_decoded(dst)
    return dst


# ------------------------------------------------------------------------------------------------ nodes
class _Node:
    """What a Carrier stands for.  ``plain``: the 16-bit (or fp32-mode) NHWC tensor; ``pairs``: the (B, 2C, H, W) [hi | lo] map of the pair
    route; ``f32``: the fp32 accumulators.  Whatever has been produced is kept; the getters produce the cheapest missing form."""
    __slots__ = ("plain", "pairs", "f32", "C", "__weakref__")

    def __init__(self, C):
        self.plain = self.pairs = self.f32 = None
        self.C = C

    def resolved(self):
        return self.plain is not None or self.pairs is not None or self.f32 is not None

    def pair_route(self):
        return self.pairs is not None

    def produce(self, want):
        raise RuntimeError("hesic_amd.handover: a value node with nothing in it")

    def get_plain(self):
        if self.plain is None:
            if not self.resolved():
                self.produce("plain")
            if self.plain is None:
                if self.pairs is not None:
                    self.plain = self.pairs[:, :self.C]              # hi = h16(v): what the single-operand launch would have stored
                else:
                    self.plain = self.f32 if self.f32.dtype == Fn.compute_dtype() else self.f32.to(Fn.compute_dtype())
        return self.plain

    def get_pairs(self):
        if self.pairs is None:
            if not self.resolved():
                self.produce("pairs")
            if self.pairs is None and self.f32 is not None:
                hi = self.f32.to(Fn._h16())
                self.pairs = torch.cat((hi, (self.f32 - hi.float()).to(Fn._h16())), 1).contiguous(memory_format=torch.channels_last)
        return self.pairs

    def get_f32(self):
        if self.f32 is None:
            if not self.resolved():
                self.produce("f32")
            if self.f32 is None:
                if self.pairs is not None:
                    self.f32 = self.pairs[:, :self.C].float() + self.pairs[:, self.C:].float()
                else:
                    self.f32 = self.plain if self.plain.dtype == torch.float32 else self.plain.float()
        return self.f32

    def conv_input(self):
        """(tensor, kind, in_abs) for a conv that reads this value: kind "pairs" -> the hi/lo map, "plain" -> an NHWC tensor."""
        if self.pair_route():
            return self.get_pairs(), "pairs", False
        return self.get_plain(), "plain", False


class ValueNode(_Node):
    __slots__ = ()

    def __init__(self, C, plain=None, pairs=None, f32=None):
        super().__init__(C)
        self.plain, self.pairs, self.f32 = plain, pairs, f32


def _src_node(src):
    return src._node if type(src) is Carrier else None


def _hilo_capable(mod, transposed):
    w = mod.weight
    return (not transposed and w.shape[1] % 32 == 0 and w.shape[0] % 8 == 0 and w.shape[1] >= 32 and getattr(mod, "mask", None) is None
            and mod.kernel_size[0] in (1, 3, 5) and mod.stride[0] in (1, 2))


class ConvNode(_Node):
    """``act(mod(src))`` not launched yet.  ``src``: a tensor or a Carrier."""
    __slots__ = ("mod", "src", "act", "transposed", "abs_child", "via_pairs")

    def __init__(self, mod, src, transposed, C):
        super().__init__(C)
        self.mod, self.src, self.act, self.transposed = mod, src, NONE, transposed
        self.abs_child, self.via_pairs = None, False

    def pair_route(self):
        if self.resolved():
            return self.pairs is not None or self.via_pairs
        n = _src_node(self.src)
        return n is not None and n.pair_route() and _hilo_capable(self.mod, self.transposed)

    def with_act(self, act):
        n = ConvNode(self.mod, self.src, self.transposed, self.C)
        n.act = act
        return n

    def get_f32(self):
        # a 16-bit copy was produced first (single-operand modes: the hyper-analysis reads |y| before the likelihood reads y): the fp32
        # accumulators come from a second launch of the same conv rather than from the rounded copy
        if self.f32 is None and self.plain is not None and self.pairs is None and Fn.fp32_latents() and self.plain.dtype != torch.float32:
            keep, self.plain = self.plain, None
            self.produce("f32")
            self.plain = keep
        return super().get_f32()

    def _abs_child(self):
        c = self.abs_child() if self.abs_child is not None else None
        return c if (c is not None and not c.resolved()) else None

    def produce(self, want):
        mod, src = self.mod, self.src
        sn = _src_node(src)
        if self.pair_route():
            p = sn.get_pairs()
            child = self._abs_child()
            prod = 3
            self.via_pairs = True
            if want == "f32" and child is None:
                self.f32 = mod.run_hilo(p, act=self.act, out="f32", products=prod)
            elif self.act != NONE and child is None:
                self.pairs = mod.run_hilo(p, act=self.act, out="hilo", products=prod)
            else:
                lo, y32 = mod.run_hilo(p, act=self.act, out="both", out_abs=child is not None, products=prod)
                self.f32 = y32
                if child is not None:
                    child.pairs = lo             # the hi/lo map of |y|: the hyper-analysis reads it (newnet1.py:434)
                else:
                    self.pairs = lo
            return
        if sn is not None:
            x, _, in_abs = sn.conv_input()
        else:
            x, in_abs = src, False
        w = mod.weight
        cin = w.shape[0] if self.transposed else w.shape[1]
        cout = w.shape[1] if self.transposed else w.shape[0]
        if (not self.transposed and mod.kernel_size[0] == 1 and x.shape[2] == 1 and x.shape[3] == 1 and cin == cout and self.act == NONE
                and not in_abs and x.dtype == torch.float32 and getattr(mod, "mask", None) is None):
            # the 1x1 conv behind spatial_pool2d (newnet1.py:500): a (B, N) x (N, N) product in fp32, as models._mixture_weights runs it
            self.f32 = Fn.pooled_linear(x, w, mod.bias)
            self.plain = self.f32
            return
        kw = {}
        if getattr(mod, "mask", None) is not None:
            mod._fold_mask()
            kw = dict(mask=mod.mask, tap_mask=mod._tap_mask)
        # what feeds round() and the likelihoods is taken from the fp32 accumulators (Fn.conv2d_latent)
        if want == "f32" and Fn.fp32_latents() and not kw and x.dtype == Fn._h16() and cin % 32 == 0 and cout % 8 == 0:
            a = {} if self.transposed else {"in_abs": in_abs}
            self.f32 = mod.run_latent(x, act=self.act, want_lo=False, **a)[1]
        elif self.transposed:
            self.plain = mod.run(x, act=self.act)
        else:
            self.plain = mod.run(x, act=self.act, in_abs=in_abs, **kw)
        if self.plain is not None and self.plain.shape[1] <= 8:
            tag_decoded(self.plain)


class AbsNode(_Node):
    """``torch.abs`` of a Carrier.  On the pair route the producing conv writes the hi/lo map of |y| itself (``out_abs``); on single operands
    the consuming conv takes |x| on load (``in_abs``)."""
    __slots__ = ("parent",)

    def __init__(self, parent_node):
        super().__init__(parent_node.C)
        self.parent = parent_node

    def pair_route(self):
        return self.pairs is not None or (not self.resolved() and self.parent.pair_route())

    def produce(self, want):
        par = self.parent
        if par.pair_route():
            if not par.resolved():
                par.produce("pairs")             # a ConvNode fills self.pairs (abs_child) on the way
            if self.pairs is None:
                p = par.get_pairs()
                neg = (p[:, :self.C] < 0).repeat(1, 2, 1, 1)
                self.pairs = torch.where(neg, -p, p)
            return
        self.plain = torch.abs(par.get_plain())

    def conv_input(self):
        if self.resolved():
            return super().conv_input()
        if self.parent.pair_route():
            self.produce("pairs")
            return self.pairs, "pairs", False
        return self.parent.get_plain(), "plain", True          # |x| on load


class CatNode(_Node):
    """``torch.cat(parts, 1)``: the parts that are pending wide convs write their channel slice of ONE buffer in place, the others are
    copied in (``hesic_copy_channels``); 3 + 3 channel image cats stay two tensors for the cat-free 6 -> 3 kernels."""
    __slots__ = ("parts",)

    def __init__(self, parts, C):
        super().__init__(C)
        self.parts = parts

    def produce(self, want):
        parts = self.parts
        B, _, H, W = parts[0].shape
        if self.C <= 8:            # image-side cat
            self.plain = torch.cat([(_plain_of(p)).float() for p in parts], 1)
            return
        if len(parts) == 2 and type(parts[0]) is Carrier and isinstance(parts[0]._node, UpNode) and not parts[0]._node.resolved():
            z = _plain_of(parts[0]._node.src)
            y1 = _plain_of(parts[1])
            if y1.dtype == z.dtype:
                self.plain = Fn.upsample4_cat(z, y1)           # newnet1.py:556-557 in one kernel
                return
        cdt = Fn.compute_dtype()
        dev = parts[0].device
        buf = Fn._empty_nhwc(B, self.C, H, W, cdt, dev)
        off = 0
        for p in parts:
            c = p.shape[1]
            n = _src_node(p)
            if (isinstance(n, ConvNode) and not n.resolved() and not n.pair_route() and not n.transposed
                    and n.mod.weight.shape[1] % 32 == 0 and c % 8 == 0 and off % 8 == 0):
                sn = _src_node(n.src)
                x, _, in_abs = sn.conv_input() if sn is not None else (n.src, "plain", False)
                if not in_abs and x.dtype == cdt:
                    if getattr(n.mod, "mask", None) is not None:
                        n.mod.forward_into(x, buf, off)
                    else:
                        n.mod.run_into(x, buf, off, act=n.act)
                    n.plain = buf[:, off:off + c]
                    off += c
                    continue
            t = _plain_of(p)
            Fn.copy_into(t if t.dtype == cdt else t.to(cdt), buf, off)
            off += c
        self.plain = buf


class UpNode(_Node):
    """``nn.UpsamplingBilinear2d(scale_factor=4)`` of a Carrier (newnet1.py:524,556)."""
    __slots__ = ("src",)

    def __init__(self, src, C):
        super().__init__(C)
        self.src = src

    def produce(self, want):
        self.plain = Fn.upsample4(_plain_of(self.src))


class SliceNode(_Node):
    """Channels [c0, c1) of a Carrier (``chunk(2, 1)`` of the entropy parameters, newnet1_joint.py:707): views of whatever form is asked for."""
    __slots__ = ("parent", "c0", "c1")

    def __init__(self, parent_node, c0, c1):
        super().__init__(c1 - c0)
        self.parent, self.c0, self.c1 = parent_node, c0, c1

    def produce(self, want):
        if want == "f32":
            self.f32 = self.parent.get_f32()[:, self.c0:self.c1]
        else:
            self.plain = self.parent.get_plain()[:, self.c0:self.c1]


class Gdn3Node(_Node):
    """The 3-channel (I)GDN of the image side (``after_gdn``, newnet1.py:684) waiting for the cat + 6 -> 3 conv behind it."""
    __slots__ = ("gdn", "src")

    def __init__(self, gdn, src):
        super().__init__(3)
        self.gdn, self.src = gdn, src

    def produce(self, want):
        g = self.gdn
        self.plain = inherit_tags(Fn.gdn(self.src, g.beta, g.gamma, g.inverse, g.beta_min), self.src)


# ------------------------------------------------------------------------------------------------ the tensor subclass
class Carrier(torch.Tensor):
    """See the module docstring.  Metadata only (shape, dtype, device, strides); ``_node`` holds the recipe / the produced forms."""

    @staticmethod
    def __new__(cls, node, shape, dtype, device):
        B, C, H, W = shape
        strides = (C * H * W, H * W, W, 1) if C <= 8 else (C * H * W, 1, W * C, C)
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), strides=strides, dtype=dtype, device=device, requires_grad=False)
        r._node = node
        return r

    __torch_function__ = torch._C._disabled_torch_function_impl

    def __repr__(self):
        n = self._node
        return f"Carrier({type(n).__name__}, shape={tuple(self.shape)}, dtype={self.dtype}, resolved={n.resolved()})"

    # what the modules of the package ask for
    def plain(self):
        return self._node.get_plain()

    def f32(self):
        return self._node.get_f32()

    def pairs(self):
        return self._node.get_pairs()

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        h = _HANDLERS.get(func)
        if h is not None:
            r = h(*args, **kwargs)
            if r is not NotImplemented:
                return r
        out = func(*tree_map(_unwrap, args), **tree_map(_unwrap, kwargs))
        if func._schema.is_mutable and args and type(args[0]) is Carrier:
            # a foreign in-place operator wrote the plain tensor: the other forms are stale, the Carrier now stands for that tensor
            n = args[0]._node
            args[0]._node = ValueNode(n.C, plain=n.plain)
            return args[0]
        return out


def _unwrap(t):
    return t._node.get_plain() if type(t) is Carrier else t


def _plain_of(t):
    return t._node.get_plain() if type(t) is Carrier else t


def plain(t):
    """The ordinary tensor behind ``t`` (``t`` itself when it is no Carrier): what operators of this package that have no deferred form call."""
    return _plain_of(t)


def f32_of(t):
    return t._node.get_f32() if type(t) is Carrier else t


def value(t, pairs=None, f32=None):
    """A Carrier around a produced tensor, so that the operators behind it (bilinear up-sampling + cat, chunk, abs) can still be recorded."""
    B, C, H, W = t.shape if t is not None else (f32.shape if f32 is not None else (pairs.shape[0], pairs.shape[1] // 2, *pairs.shape[2:]))
    ref = t if t is not None else (f32 if f32 is not None else pairs)
    dtype = t.dtype if t is not None else Fn.compute_dtype()
    return Carrier(ValueNode(C, plain=t, pairs=pairs, f32=f32), (B, C, H, W), dtype, ref.device)


# ------------------------------------------------------------------------------------------------ recorded ATen operators
def _h_abs(x):
    if type(x) is not Carrier:
        return NotImplemented
    n = x._node
    child = AbsNode(n)
    if isinstance(n, ConvNode) and not n.resolved():
        n.abs_child = weakref.ref(child)
    return Carrier(child, x.shape, x.dtype, x.device)


def _act_handler(act, inplace):
    def h(x, *a, **k):
        if type(x) is not Carrier:
            return NotImplemented
        if act == LEAKY:
            slope = a[0] if a else k.get("negative_slope", 0.01)
            if float(slope) != 0.01:
                return NotImplemented
        n = x._node
        if not (isinstance(n, ConvNode) and not n.resolved() and n.act == NONE and n._abs_child() is None):
            return NotImplemented
        cout = n.C
        if cout <= 8:
            return NotImplemented
        if inplace:
            n.act = act                        # the pre-activation value is gone by the operator's own semantics
            return x
        return Carrier(n.with_act(act), x.shape, x.dtype, x.device)
    return h


def _h_cat(tensors, dim=0):
    ts = list(tensors)
    if not ts or any(t.dim() != 4 for t in ts) or (dim % 4) != 1:
        return NotImplemented
    B, _, H, W = ts[0].shape
    if any(t.shape[0] != B or tuple(t.shape[2:]) != (H, W) for t in ts):
        return NotImplemented
    C = sum(t.shape[1] for t in ts)
    first = next(t for t in ts if type(t) is Carrier)
    dtype = torch.float32 if C <= 8 else Fn.compute_dtype()
    return Carrier(CatNode(ts, C), (B, C, H, W), dtype, first.device)


def _h_upsample(x, output_size=None, align_corners=False, *a, **k):
    if type(x) is not Carrier or not align_corners or x.dim() != 4:
        return NotImplemented
    B, C, H, W = x.shape
    osz = output_size
    if osz is None:
        sf = a[0] if a else k.get("scale_factors")
        if sf is None or [float(s) for s in sf] != [4.0, 4.0]:
            return NotImplemented
        osz = (4 * H, 4 * W)
    if tuple(int(s) for s in osz) != (4 * H, 4 * W):
        return NotImplemented
    return Carrier(UpNode(x, C), (B, C, 4 * H, 4 * W), x.dtype, x.device)


def _h_split(x, split_size, dim=0):
    if type(x) is not Carrier or (dim % 4) != 1 or not isinstance(split_size, int):
        return NotImplemented
    B, C, H, W = x.shape
    out = []
    for c0 in range(0, C, split_size):
        c1 = min(C, c0 + split_size)
        out.append(Carrier(SliceNode(x._node, c0, c1), (B, c1 - c0, H, W), x.dtype, x.device))
    return out


def _h_chunk(x, chunks, dim=0):
    if type(x) is not Carrier or (dim % 4) != 1:
        return NotImplemented
    return _h_split(x, -(-x.shape[1] // int(chunks)), dim)


def _h_slice(x, dim=0, start=None, end=None, step=1):
    if type(x) is not Carrier or (dim % 4) != 1 or step != 1:
        return NotImplemented
    B, C, H, W = x.shape
    c0 = 0 if start is None else (start + C if start < 0 else min(start, C))
    c1 = C if end is None else (end + C if end < 0 else min(end, C))
    if not (0 <= c0 < c1 <= C):
        return NotImplemented
    return Carrier(SliceNode(x._node, c0, c1), (B, c1 - c0, H, W), x.dtype, x.device)


def _h_same(x, *a, **k):
    if type(x) is not Carrier:
        return NotImplemented
    return Carrier(x._node, x.shape, x.dtype, x.device)


_HANDLERS = {
    aten.abs.default: _h_abs,
    aten.relu.default: _act_handler(RELU, False), aten.relu_.default: _act_handler(RELU, True),
    aten.leaky_relu.default: _act_handler(LEAKY, False), aten.leaky_relu_.default: _act_handler(LEAKY, True),
    aten.cat.default: _h_cat,
    aten.upsample_bilinear2d.default: _h_upsample, aten.upsample_bilinear2d.vec: _h_upsample,
    aten.split.Tensor: _h_split, aten.chunk.default: _h_chunk, aten.slice.Tensor: _h_slice,
    aten.detach.default: _h_same, aten.alias.default: _h_same,
}


# ------------------------------------------------------------------------------------------------ what the modules call
def _out_hw(mod, H, W, transposed):
    k, s, p = mod.kernel_size[0], mod.stride[0], mod.padding[0]
    return Fn._out_hw(H, W, k, s, p, transposed)


def conv(mod, x, transposed=False):
    """``HipConv2d.forward`` / ``HipConvTranspose2d.forward`` at inference: a Carrier for a wide output, a launch for an image-side one."""
    mod._check()
    w = mod.weight
    cin = w.shape[0] if transposed else w.shape[1]
    cout = w.shape[1] if transposed else w.shape[0]
    if x.dim() != 4 or x.shape[1] != cin:
        raise RuntimeError(f"conv: expected a (B, {cin}, H, W) input, got {tuple(x.shape)}")
    B, _, H, W = x.shape
    Ho, Wo = _out_hw(mod, H, W, transposed)
    node = ConvNode(mod, x, transposed, cout)
    if cout <= 8:
        sn = _src_node(x)
        if cin == 6 and cout == 3 and isinstance(sn, CatNode) and not sn.resolved() and len(sn.parts) == 2:
            a, b = sn.parts
            an = _src_node(a)
            if a.shape[1] == 3 and b.shape[1] == 3:
                if isinstance(an, Gdn3Node) and not an.resolved():
                    # IGDN(3) -> cat -> after_conv (newnet1.py:684-686): one launch, no concatenated copy
                    y = mod.run_cat(an.src, _plain_of(b), gdn=an.gdn, gdn_on_input=True)
                else:
                    y = mod.run_cat(_plain_of(a), _plain_of(b))
                return tag_decoded(y)
        if cin == 6 and cout == 3 and sn is None:
            # pre_conv (newnet1.py:643): wait for the 3-channel GDN behind it
            return Carrier(node, (B, cout, Ho, Wo), torch.float32, x.device)
        return tag_decoded(mod.run(_plain_of(x)))
    if type(x) is not Carrier and getattr(x, SMOOTH_TAG, False) and cin > 8 and not hasattr(mod, "_packer"):
        # models.Encoder1 / Decoder1 mark these layers themselves; the reference's own model files get the same packs this way
        mod.shaped_weights = True
    dtype = Fn.compute_dtype()
    if cin == cout and H == 1 and W == 1 and mod.kernel_size[0] == 1 and type(x) is not Carrier and x.dtype == torch.float32:
        dtype = torch.float32
    return Carrier(node, (B, cout, Ho, Wo), dtype, x.device)


def _smooth(t):
    setattr(t, SMOOTH_TAG, True)
    return t


def gdn(mod, x):
    """``GDN.forward`` at inference: fuses with the pending conv in front of it (newnet1.py:590-600, :615-623, :643-644)."""
    n = _src_node(x)
    if isinstance(n, ConvNode) and not n.resolved() and n.act == NONE and n._abs_child() is None:
        cv, src = n.mod, n.src
        sn = _src_node(src)
        w = cv.weight
        cin = w.shape[0] if n.transposed else w.shape[1]
        C = mod.beta.numel()
        if cin == 6 and C == 3 and sn is None and not n.transposed:
            return cv.run_cat(src[:, :3], src[:, 3:], gdn=mod)                     # pre_conv + pre_gdn, reading the two halves in place
        if C == 128 and n.C == 128:
            if cin == 3 and sn is None and not n.transposed:
                if Fn.analysis_hilo(src) and not getattr(src, DECODED_TAG, False) and Fn.sconv_gdn_hilo_ok(src, w):
                    # the first layer of an analysis stack the caller feeds: hi/lo pairs from here on ("x3"; "x3c2": one value per channel
                    # out of this kernel, pairs again behind g_a_conv2's GDN)
                    if not hasattr(cv, "_hl1"):
                        cv._hl1 = Fn.PackedN2wHiLo()
                    gp, bp = mod.packer().get(mod.beta, mod.gamma, mod.beta_min)
                    if Fn.analysis_conv2_single():
                        t = Fn.sconv_gdn_hilo(src, cv._hl1.get(w, mod.gamma, out1=True), cv.bias, bp, mod.inverse, out1=True)
                        setattr(t, C2_TAG, True)
                        return t
                    t = Fn.sconv_gdn_hilo(src, cv._hl1.get(w, mod.gamma), cv.bias, bp, mod.inverse)
                    return value(None, pairs=t)
                return _smooth(cv.run_gdn(src, mod))
            if sn is not None and sn.pair_route() and _hilo_capable(cv, n.transposed):
                return value(None, pairs=cv.run_hilo(sn.get_pairs(), gdn=mod))
            xin = _plain_of(src)
            if (not n.transposed and getattr(xin, C2_TAG, False) and Fn.analysis_conv2_single() and tuple(w.shape[:2]) == (128, 128)
                    and xin.dtype == Fn._h16()):
                return value(None, pairs=cv.run_gdn_hilo_out(xin, mod))
            return _smooth(cv.run_gdn(xin, mod))
    if type(x) is not Carrier and x.shape[1] == 3 and mod.beta.numel() == 3:
        return Carrier(Gdn3Node(mod, x), tuple(x.shape), x.dtype, x.device)      # after_gdn: wait for cat + after_conv
    t = _plain_of(x)
    return inherit_tags(Fn.gdn(t, mod.beta, mod.gamma, mod.inverse, mod.beta_min), t)
