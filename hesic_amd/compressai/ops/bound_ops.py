"""LowerBound with the reference's gradient rule (compressai/ops/bound_ops.py:19-52).

On the hot path the bound is applied inside the fused HIP kernels (GDN reparametrisation, scale and
likelihood floors -- csrc/gdn.hip, csrc/entropy.hip); this module is the stand-alone operator the
reference also exposes, used by host code such as ``build_indexes``.  Plain tensor plumbing.
"""
import torch
from torch import nn
from torch.autograd import Function


class LowerBoundFunction(Function):
    """y = max(x, b);  dy/dx := 1 where x >= b or the incoming gradient pushes x upwards, else 0."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.maximum(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        moves_up = g < 0
        return torch.where((x >= bound) | moves_up, g, torch.zeros_like(g)), None


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.tensor([float(bound)], dtype=torch.float32))
        self._host = None

    def value(self):
        """The bound as a Python float for kernel descriptors.  Read back from the buffer once per (storage, version) --
        not per forward: a device->host read in the hot path would serialise the host with the GPU (and cannot be
        captured in a HIP graph)."""
        b = self.bound
        tag = (b.data_ptr(), b._version, b.device)
        if self._host is None or self._host[0] != tag:
            self._host = (tag, float(b))
        return self._host[1]

    def forward(self, x):
        if torch.jit.is_scripting():
            return torch.maximum(x, self.bound)
        return LowerBoundFunction.apply(x, self.bound)
