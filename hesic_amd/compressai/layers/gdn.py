"""GDN / IGDN / GDN1 modules (reference: compressai/layers/gdn.py:22-97)."""
import torch
import torch.nn as nn

from compressai.ops.parametrizers import NonNegativeParametrizer
from hesic_amd import functional as Fn
from hesic_amd import handover as _ho


class GDN(nn.Module):
    r"""y_i = x_i / sqrt(beta_i + sum_j gamma_ij x_j^2)   (inverse: multiply instead of divide).

    Parameters are stored in the reparametrised domain exactly like the reference (same init, same
    ``beta_reparam`` / ``gamma_reparam`` buffers); forward is ONE fused HIP kernel (csrc/gdn.hip)."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_min = float(beta_min)
        self.beta_reparam = NonNegativeParametrizer(minimum=self.beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        if _ho.active(x):
            return _ho.gdn(self, x)           # inference: fuses with the conv that was called in front of it (hesic_amd/handover.py)
        return Fn.gdn(x, self.beta, self.gamma, self.inverse, self.beta_min)

    def packer(self):
        """cache of the pre-transformed parameters used by the fused conv+GDN epilogue (inference)"""
        if not hasattr(self, "_packer"):
            self._packer = Fn.PackedGdn()
        return self._packer


class GDN1(GDN):
    r"""Simplified GDN, y_i = x_i / (beta_i + sum_j gamma_ij |x_j|) (reference gdn.py:73-97).
    Not on the HESIC path (Cheng2020 models only): kept importable, evaluated with tensor ops."""

    def forward(self, x):
        x = _ho.plain(x)
        c = x.shape[1]
        beta = self.beta_reparam(self.beta)
        gamma = self.gamma_reparam(self.gamma).reshape(c, c, 1, 1)
        norm = torch.nn.functional.conv2d(torch.abs(x), gamma, beta)
        return x * norm if self.inverse else x / norm
