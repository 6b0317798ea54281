"""warp_perspective forward (8 x 3 x 512 x 512 fp32, the step's shape) by HIP events over a graph of 50 launches, and bit-equality of the
HESIC_WARP_V4 variant against the default kernel:  python profiles/scripts/warp_time.py   (run once with and once without HESIC_WARP_V4=1)"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from hesic_amd import functional as Fn, synthetic

x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 4, 512, 512))
x1, Hm = x1.repeat(2, 1, 1, 1), Hm.repeat(2, 1, 1)
out = Fn.warp_perspective(x1, Hm, (512, 512))
torch.cuda.synchronize()
if len(sys.argv) > 1:
    torch.save(out.cpu(), sys.argv[1])
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        Fn.warp_perspective(x1, Hm, (512, 512))
    with torch.cuda.graph(g):
        for _ in range(50):
            Fn.warp_perspective(x1, Hm, (512, 512))
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(f"warp 8x3x512x512 V4={os.environ.get('HESIC_WARP_V4', '0')}: {us:.2f} us per launch = {2 * x1.numel() * 4 / us / 1e6:.2f} TB/s")
