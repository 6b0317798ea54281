"""Run-to-run determinism of one Trainer.step: two fresh trainers from the same state, the same inputs and the same noise -- compare the
flat gradient buffer (before the optimiser consumes it the step keeps it: lr = 0) parameter by parameter.
    python profiles/scripts/train_determinism.py        (HESIC_WGRAD_PARTIAL_BATCH=0 for the one-launch-per-layer route)"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import hesic_amd
from hesic_amd import models, synthetic
from hesic_amd.train import Trainer

hesic_amd.set_compute_dtype(torch.bfloat16)
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 4, 512, 512))
x1, x2, Hm = (t.repeat(2, *([1] * (t.dim() - 1))) for t in (x1, x2, Hm))
g = torch.Generator(device="cuda").manual_seed(1)
zs, ys = (8, 128, 8, 8), (8, 192, 32, 32)
noise = {k: torch.empty(zs if k[0] == "z" else ys, device="cuda").uniform_(-0.5, 0.5, generator=g) for k in ("z1", "y1", "y1b", "y1w", "z2", "y2", "y2b")}


def run():
    net = models.HSIC(); synthetic.fill_state_dict_(net.state_dict())
    tr = Trainer(net.cuda(), lmbda=0.0067, lr=0.0, aux_lr=0.0)
    for _ in range(2):
        c = tr.step(x1, x2, Hm, noise=noise)
    torch.cuda.synchronize()
    return tr, {k: float(v) for k, v in c.items() if torch.is_tensor(v) and v.numel() == 1}


a, ca = run()
b, cb = run()
print("losses", ca, cb)
ga, gb = a.main_group.flat_g, b.main_group.flat_g
d = (ga - gb).abs()
print("flat grad: max |a| %.3e  max |a-b| %.3e  rel-to-max %.3e  elements differing %d of %d" % (float(ga.abs().max()), float(d.max()), float(d.max() / ga.abs().max()), int((d > 0).sum()), ga.numel()))
worst = []
for n_, p in a.model.named_parameters():
    if p.grad is None:
        continue
    q = dict(b.model.named_parameters())[n_].grad
    dd = float((p.grad - q).abs().max())
    if dd > 0:
        worst.append((dd / (float(p.grad.abs().max()) + 1e-30), n_))
for r, n_ in sorted(worst, reverse=True)[:12]:
    print("  %.3e  %s" % (r, n_))
