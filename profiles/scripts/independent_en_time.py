#!/usr/bin/env python3
"""Independent_EN forward time at B=8, 512x512, bf16 -- the figure quoted in DESIGN.md section 7."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hesic_amd
from hesic_amd import models, synthetic

hesic_amd.set_compute_dtype(torch.bfloat16)
net = models.Independent_EN()
synthetic.fill_state_dict_(net.state_dict())
net = net.cuda().eval()
x1, x2, Hm = synthetic.stereo_batch(0, 2, 512, 512)
x1, x2, Hm = x1.repeat(4, 1, 1, 1).cuda(), x2.repeat(4, 1, 1, 1).cuda(), Hm.repeat(4, 1, 1).cuda()
with torch.no_grad():
    for _ in range(3):
        net(x1, x2, Hm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net(x1, x2, Hm)
    e1.record()
    torch.cuda.synchronize()
print("Independent_EN B=8 512^2 bf16: %.2f ms/forward" % (e0.elapsed_time(e1) / 10))


# stage-2 training step (HSIC frozen: only the enhancement net learns; newtrain6_real.py): forward + MSE backward + Adam
net.train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
t1, t2 = x1.clone(), x2.clone()


def step():
    opt.zero_grad(set_to_none=True)
    out = net(x1, x2, Hm)
    loss = ((out["x1_hat"].float() - t1) ** 2).mean() + ((out["x2_hat"].float() - t2) ** 2).mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    step()
e1.record()
torch.cuda.synchronize()
print("Independent_EN stage-2 training step B=8 512^2 bf16: %.2f ms/step (fast 32-channel path: %s)" % (e0.elapsed_time(e1) / 10, os.environ.get("HESIC_EN_GENERIC") is None))
