#!/usr/bin/env python3
"""N forwards of Independent_EN (B=8 512^2, f16) for rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hesic_amd
from hesic_amd import models, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hesic_amd.set_compute_dtype(torch.float16)
net = models.Independent_EN()
synthetic.fill_state_dict_(net.state_dict())
net = net.cuda().eval()
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 8, 512, 512))
with torch.no_grad():
    for _ in range(n):
        net(x1, x2, Hm)
torch.cuda.synchronize()
