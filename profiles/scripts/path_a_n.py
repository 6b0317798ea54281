#!/usr/bin/env python3
"""N forwards of the reference's call order over the drop-in modules (hesic_amd/path_a.py, B=8 512^2, f16 x3) for rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hesic_amd
from hesic_amd import models, path_a, synthetic
kind, n = (sys.argv[1] if len(sys.argv) > 1 else "hsic"), int(sys.argv[2]) if len(sys.argv) > 2 else 5
hesic_amd.set_compute_dtype(torch.float16)
net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
synthetic.fill_state_dict_(net.state_dict())
net = net.cuda().eval()
B = 8 if kind == "hsic" else 4
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, B, 512, 512))
with torch.no_grad():
    for _ in range(n):
        out = path_a.FWD[kind](net, x1, x2, Hm)
        models.rate_distortion(out, x1, x2)
torch.cuda.synchronize()
