"""NonNegativeParametrizer (compressai/ops/parametrizers.py:21-44).

Holds the ``pedestal`` / ``lower_bound.bound`` buffers (state-dict compatibility) and the ``init``
transform.  GDN's forward does not call ``forward`` here: the reparametrisation
``max(theta, bound)^2 - pedestal`` runs inside the fused GDN kernel (csrc/gdn.hip)."""
import torch
from torch import nn

from .bound_ops import LowerBound


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum, self.reparam_offset = float(minimum), float(reparam_offset)
        ped = self.reparam_offset * self.reparam_offset
        self.register_buffer("pedestal", torch.tensor([ped], dtype=torch.float32))
        self.lower_bound = LowerBound((self.minimum + ped) ** 0.5)

    def init(self, x):
        """parameter value whose forward() gives back max(x, 0)"""
        return (torch.clamp_min(x, 0.0) + self.pedestal).sqrt()

    def forward(self, theta):
        t = self.lower_bound(theta)
        return t * t - self.pedestal
