#!/usr/bin/env python3
"""Time of the fused hi/lo g_a_conv1 + GDN kernel (B=8, 512x512 image) with back-to-back launches; HESIC_N2W_STAGGER = A/B switch."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd
from hesic_amd import functional as Fn, synthetic
from compressai.layers import GDN

x = synthetic.stereo_batch(0, 8, 512, 512)[0].cuda()
g = torch.Generator().manual_seed(0)
w = ((torch.rand(128, 3, 5, 5, generator=g) - 0.5) * 0.4).cuda()
b = ((torch.rand(128, generator=g) - 0.5) * 0.1).cuda()
gd = GDN(128).cuda()
gp, bp = gd.packer().get(gd.beta, gd.gamma, gd.beta_min)
OUT1 = os.environ.get("OUT1", "1") == "1"
hesic_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("HESIC_DTYPE", "f16")])
img = Fn.PackedN2wHiLo().get(w, gd.gamma, OUT1)
with torch.no_grad():
    for _ in range(5):
        y = Fn.sconv_gdn_hilo(x, img, b, bp, False, OUT1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y = Fn.sconv_gdn_hilo(x, img, b, bp, False, OUT1)
    e1.record()
    torch.cuda.synchronize()
print(f"n2w_gdn_hilo: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us  out1={OUT1} env={ {k: v for k, v in os.environ.items() if k.startswith('HESIC_N2W')} }")
