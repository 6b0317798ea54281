#!/usr/bin/env python3
"""deconv 128 -> 128 (+ IGDN) on the fused-phase kernel, ten launches on the same input: every output must equal the first bit for bit
(and the unfused path's, hesic_conv2d_set_phase_fusion(0))."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hesic_amd
from hesic_amd import _lib as L
from compressai.layers import GDN
from compressai.models.utils import deconv
for dt in (torch.float16, torch.bfloat16):
    hesic_amd.set_compute_dtype(dt)
    torch.manual_seed(0)
    for S in (128, 64, 200):
        x = (torch.randn(8, 128, S, S, device="cuda") * 0.5).to(dt).contiguous(memory_format=torch.channels_last)
        layer, g = deconv(128, 128).cuda(), GDN(128, inverse=True).cuda()
        with torch.no_grad():
            for name, f in (("igdn", lambda: layer.run_gdn(x, g)), ("plain", lambda: layer.run(x))):
                L.lib().hesic_conv2d_set_phase_fusion(1)
                outs = [f().clone() for _ in range(10)]
                torch.cuda.synchronize()
                same = sum(int(torch.equal(outs[0], o)) for o in outs)
                L.lib().hesic_conv2d_set_phase_fusion(0)
                ref = f().clone()
                L.lib().hesic_conv2d_set_phase_fusion(1)
                d = (ref.float() - outs[0].float()).abs()
                for ph in range(4):
                    dp = d[:, :, (ph >> 1)::2, (ph & 1)::2]
                    bad = (dp.amax(dim=1) > 0)
                    print("   phase", ph, "max diff %.3g" % float(dp.max()), "bad pixels", int(bad.sum()), "of", bad.numel(),
                          "| bad q-columns mod 16:", sorted(set((bad.nonzero()[:, 2] % 16).tolist()))[:16], "| bad q-rows mod 8:", sorted(set((bad.nonzero()[:, 1] % 8).tolist())))
                print(dt, S, name, "identical launches:", same, "/ 10   equals unfused:", bool(torch.equal(ref, outs[0])), "max diff", float((ref.float() - outs[0].float()).abs().max()))
