#!/usr/bin/env python3
"""Steady-state eager inference (HESIC, B=8, 512x512, bf16) with two marker kernels around one forward, for
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python profiles/scripts/eager_steady.py
    python profiles/scripts/show_timeline.py <dir>        # start / end / duration / queue / grid / kernel of that forward"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd
from hesic_amd import models, synthetic
hesic_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("HESIC_DTYPE", "f16")])
net = models.HSIC(); synthetic.fill_state_dict_(net.state_dict()); net = net.cuda().eval(); net.update(force=True)
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 8, 512, 512))
with torch.no_grad():
    for _ in range(20):
        o = net(x1, x2, Hm); models.rate_distortion(o, x1, x2)
    torch.cuda.synchronize()
    for i in range(8):
        if i in (4, 5): torch.zeros(7, device="cuda").fill_(3.0)      # markers around forward #4
        o = net(x1, x2, Hm); models.rate_distortion(o, x1, x2)
    torch.cuda.synchronize()
