"""The SINGLE-operand fused (I)GDN epilogues of the binary16 build (decoders, third analysis pass): squares stored as v^2 * 2^-6 in one half.
An identity conv (centre tap) + GDN / an identity deconv + IGDN against fp64, by input scale and beta'; the yardstick is one half's own
rounding, 2^-11 = 4.9e-4 (p99 of a single-operand path cannot be better than ~2.4e-4).
    python profiles/scripts/gdn_single_precision.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import hesic_amd
from compressai.layers import GDN
from compressai.models.utils import conv, deconv

hesic_amd.set_compute_dtype(torch.float16)
torch.manual_seed(0)
B, C, S = 2, 128, 64
ped = 2.0 ** -36
for inverse in (False, True):
    layer = (deconv if inverse else conv)(C, C).cuda()
    with torch.no_grad():
        layer.weight.zero_(); layer.bias.zero_()
        for c in range(C):
            layer.weight[c, c, 2, 2] = 1.0
    for beta_v in (1.0, 1e-2, 1e-4):
        for sigma in (1.0, 0.1, 0.03, 0.01):
            g = GDN(C, inverse=inverse).cuda()
            with torch.no_grad():
                gam = torch.rand(C, C, device="cuda") * 2e-3 + torch.eye(C, device="cuda") * 0.1
                g.beta.copy_(torch.full((C,), beta_v + ped, device="cuda").sqrt())
                g.gamma.copy_((gam + ped).sqrt())
                x = (torch.randn(B, C, S, S, device="cuda") * sigma).to(torch.float16).contiguous(memory_format=torch.channels_last)
                y = layer.run_gdn(x, g).double()
                xv = x.double()
                if inverse:          # transposed stride 2, centre tap (2, 2), pad 2, output_padding 1: out[2i, 2j] = x[i, j], zeros elsewhere
                    v = torch.zeros(B, C, 2 * S, 2 * S, dtype=torch.float64, device="cuda")
                    v[:, :, ::2, ::2] = xv
                else:
                    v = xv[:, :, ::2, ::2]
                norm = (g.beta.double() ** 2 - ped).view(1, C, 1, 1) + torch.einsum("ij,bjhw->bihw", g.gamma.double() ** 2 - ped, v * v)
                ref = v * norm.sqrt() if inverse else v / norm.sqrt()
            nz = ref.abs() > 0
            rel = ((y - ref).abs() / ref.abs().clamp_min(1e-30))[nz]
            sel = ref.abs()[nz] > ref.abs()[nz].median()
            print(f"{'IGDN' if inverse else 'GDN '} beta' {beta_v:g} sigma {sigma:g}: median rel err {float(rel[sel].median()):.2e}  p99 {float(rel[sel].quantile(0.99)):.2e}   (2^-11 = 4.9e-4)", flush=True)
