#!/usr/bin/env python3
"""Clock / socket power while ONE single-operand layer repeats (graph of 50 launches): is it at the power cap?"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hesic_amd, bench
from compressai.layers import GDN
from compressai.models.utils import conv, deconv
hesic_amd.set_compute_dtype(torch.float16)
torch.manual_seed(0)
def probe(name, f):
    with torch.no_grad():
        for _ in range(3): f()
        torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s): f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50): f()
        n, wall, st = bench.under_load(g.replay, 2.5)
    print(f"{name:34s} {1e6 * wall / n / 50:8.1f} us per launch   {st}")
x128 = (torch.randn(8, 128, 128, 128, device="cuda") * 0.5).half().contiguous(memory_format=torch.channels_last)
x256 = (torch.randn(8, 128, 256, 256, device="cuda") * 0.5).half().contiguous(memory_format=torch.channels_last)
x32 = (torch.randn(8, 128, 32, 32, device="cuda") * 0.5).half().contiguous(memory_format=torch.channels_last)
d3, ig = deconv(128, 128).cuda(), GDN(128, inverse=True).cuda()
c2, g2 = conv(128, 128).cuda(), GDN(128).cuda()
hd = conv(128, 960, stride=1, kernel_size=5).cuda()
from hesic_amd import functional as Fn
c1, g1 = conv(3, 128).cuda(), GDN(128).cuda()
img = torch.rand(8, 3, 512, 512, device="cuda")
pk = Fn.PackedN2wHiLo()
gp, bp = g1.packer().get(g1.beta, g1.gamma, g1.beta_min)
probe("conv1 + GDN on pairs (n2w hilo)", lambda: Fn.sconv_gdn_hilo(img, pk.get(c1.weight, g1.gamma), c1.bias, bp, g1.inverse))
probe("deconv3 + IGDN (tr4, 128^2 -> 256^2)", lambda: d3.run_gdn(x128, ig))
probe("conv2 + GDN single (256^2 -> 128^2)", lambda: c2.run_gdn(x256, g2))
probe("head conv 128 -> 960 s1 (32^2)", lambda: hd.run(x32))
