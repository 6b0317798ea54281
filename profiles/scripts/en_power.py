#!/usr/bin/env python3
"""Clock and socket power while Independent_EN (B=8 512^2, f16) repeats for a few seconds, for image-like and for all-zero inputs."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import hesic_amd
from hesic_amd import models, synthetic
import bench
hesic_amd.set_compute_dtype(torch.float16)
net = models.Independent_EN()
synthetic.fill_state_dict_(net.state_dict())
net = net.cuda().eval()
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 8, 512, 512))
for label, a, b in (("images", x1, x2), ("zeros", torch.zeros_like(x1), torch.zeros_like(x2)), ("noise", torch.rand_like(x1), torch.rand_like(x2))):
    with torch.no_grad():
        for _ in range(5):
            net(a, b, Hm)
        n, wall, st = bench.under_load(lambda: net(a, b, Hm), 3.0)
    print(label, "ms per forward %.3f" % (1e3 * wall / n), st)
