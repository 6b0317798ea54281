#!/usr/bin/env python3
"""The two image-side convs of Independent_EN (6 -> 32 from the planar images, 32 -> 3 + image) at B=8 512^2, f16: HIP-event time over 20 launches
and a checksum of the outputs (to compare build variants bit for bit)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hesic_amd
from hesic_amd import functional as Fn, _lib as L
hesic_amd.set_compute_dtype(torch.float16)
torch.manual_seed(0)
B, H, W = 8, 512, 512
xa, xb = torch.rand(B, 3, H, W, device="cuda"), torch.rand(B, 3, H, W, device="cuda")
w6, b6 = torch.randn(32, 6, 3, 3, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
w3, b3 = torch.randn(3, 32, 3, 3, device="cuda") * 0.05, torch.randn(3, device="cuda") * 0.1
def timed(f):
    for _ in range(5): o = f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): o = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3, o
with torch.no_grad():
    t6, y6 = timed(lambda: Fn.conv3x3_c32_img6(xa, xb, w6, b6, act=L.ACT_LEAKY))
    t3, y3 = timed(lambda: Fn.conv3x3_c32(y6, w3, b3, act=L.ACT_NONE, res1=xa))
print("conv 6->32 %.1f us   conv 32->3 %.1f us   checksums %.6f %.6f" % (t6, t3, float(y6.float().double().sum()), float(y3.double().sum())))
