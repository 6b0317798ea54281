"""How exact is the fused pair (hi/lo) GDN epilogue for SMALL activations?  A 5x5 stride-2 conv with identity weights (centre tap) feeds the
fused GDN of the pair path (`run_hilo(..., gdn=...)`); the output pair's sum is compared with fp64 on the same inputs, by input scale and beta.
In the binary16 build the squares are stored scaled by 2^-6: for |v| < 0.0625 their hi half is a SUBNORMAL half (absolute step 6e-8).
    python profiles/scripts/gdn_pair_precision.py [f16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import hesic_amd
from compressai.layers import GDN
from compressai.models.utils import conv
from hesic_amd import functional as Fn

h16 = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "f16") else torch.bfloat16
hesic_amd.set_compute_dtype(h16)
Fn.set_analysis_precision("x3")
torch.manual_seed(0)
B, C, S = 2, 128, 64
layer = conv(C, C, stride=2).cuda()
with torch.no_grad():
    layer.weight.zero_(); layer.bias.zero_()
    for c in range(C):
        layer.weight[c, c, 2, 2] = 1.0
for beta_v in (1.0, 1e-2, 1e-4):
    for sigma in (1.0, 0.1, 0.03, 0.01, 0.003):
        g = GDN(C).cuda()
        with torch.no_grad():
            # reparametrised storage: beta = sqrt(beta' + pedestal), gamma = sqrt(gamma' + pedestal); a trained-looking gamma': diagonal 0.1, small positive rest
            ped = 2.0 ** -36
            gam = torch.rand(C, C, device="cuda") * 2e-3 + torch.eye(C, device="cuda") * 0.1
            g.beta.copy_(torch.full((C,), beta_v + ped, device="cuda").sqrt())
            g.gamma.copy_((gam + ped).sqrt())
        x = (torch.randn(B, C, S, S, device="cuda") * sigma)
        hi = x.to(h16)
        lo = (x - hi.float()).to(h16)
        xin = (hi.float() + lo.float())                                    # what the kernel sees, exactly
        xh = torch.cat((hi, lo), 1).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y = layer.run_hilo(xh, gdn=g)                                   # (B, 2C, S/2, S/2) hi | lo
            yk = (y[:, :C].float() + y[:, C:].float()).double()
            v = xin[:, :, ::2, ::2].double()
            # conv output position o reads input 2o + (k - pad) -> centre tap = 2o
            beta_p = g.beta.double().clamp_min((g.beta_min + ped) ** 0.5) ** 2 - ped if hasattr(g, "beta_min") else g.beta.double() ** 2 - ped
            gamma_p = g.gamma.double() ** 2 - ped
            norm = beta_p.view(1, C, 1, 1) + torch.einsum("ij,bjhw->bihw", gamma_p, v * v)
            ref = v / norm.sqrt()
        err = (yk - ref).abs()
        rel = (err / ref.abs().clamp_min(1e-30))
        sel = ref.abs() > ref.abs().median()
        print(f"{str(h16)[6:]} beta' {beta_v:g} sigma {sigma:g}: median rel err {float(rel[sel].median()):.2e}  p99 {float(rel[sel].quantile(0.99)):.2e}  (2^-22 = 2.4e-7; fp32 ulp 6e-8)", flush=True)

# ---- the image-side pair kernel: g_a_conv1 (3 -> 128, 5x5 s2) + GDN on an image, weights scaled so that the conv outputs are small
print("conv1 + GDN pair kernel (n2w_gdn_hilo_kernel):")
for beta_v in (1.0, 1e-2, 1e-4):
    for wscale in (1.0, 0.1, 0.01):
        c1 = conv(3, C, stride=2).cuda()
        g = GDN(C).cuda()
        ped = 2.0 ** -36
        with torch.no_grad():
            c1.weight.mul_(wscale); c1.bias.mul_(wscale)
            gam = torch.rand(C, C, device="cuda") * 2e-3 + torch.eye(C, device="cuda") * 0.1
            g.beta.copy_(torch.full((C,), beta_v + ped, device="cuda").sqrt())
            g.gamma.copy_((gam + ped).sqrt())
        x = torch.rand(2, 3, 128, 128, device="cuda")
        packer = Fn.PackedN2wHiLo()
        gp, bp = g.packer().get(g.beta, g.gamma, g.beta_min)
        with torch.no_grad():
            y = Fn.sconv_gdn_hilo(x, packer.get(c1.weight, g.gamma), c1.bias, bp, g.inverse)
            yk = (y[:, :C].float() + y[:, C:].float()).double()
            v = torch.nn.functional.conv2d(x.double(), c1.weight.double(), c1.bias.double(), stride=2, padding=2)
            beta_p = g.beta.double() ** 2 - ped
            gamma_p = g.gamma.double() ** 2 - ped
            ref = v / (beta_p.view(1, C, 1, 1) + torch.einsum("ij,bjhw->bihw", gamma_p, v * v)).sqrt()
        rel = ((yk - ref).abs() / ref.abs().clamp_min(1e-30))
        sel = ref.abs() > ref.abs().median()
        print(f"{str(h16)[6:]} beta' {beta_v:g} |v| median {float(v.abs().median()):.2g}: median rel err {float(rel[sel].median()):.2e}  p99 {float(rel[sel].quantile(0.99)):.2e}", flush=True)
