import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hesic_amd
from hesic_amd import models, synthetic, functional as Fn
from hesic_amd.train import GraphedTrainer
hesic_amd.set_compute_dtype(torch.bfloat16)
net = models.HSIC(); synthetic.fill_state_dict_(net.state_dict()); net = net.cuda()
def stats(tag):
    ped = 2.0 ** -36
    for n, m in net.named_modules():
        if m.__class__.__name__ == "GDN" and m.beta.numel() == 128 and ("g_a" in n):
            b = (m.beta.detach().double().clamp_min((m.beta_min + ped) ** 0.5) ** 2 - ped)
            g = (m.gamma.detach().double().clamp_min(2.0 ** -18) ** 2 - ped)
            print(tag, n, "beta' min %.2e med %.2e max %.2e | gamma' diag med %.2e offdiag med %.2e" % (float(b.min()), float(b.median()), float(b.max()), float(g.diag().median()), float((g - torch.diag(g.diag())).median())))
stats("init")
torch.manual_seed(5)
tr = GraphedTrainer(net, lr=1e-4, aux_lr=1e-3, lmbda=0.02)
pool = [tuple(t.cuda() for t in synthetic.smooth_stereo_batch(100 + 8 * i, 8, 256, 256)) for i in range(8)]
for s in range(2000):
    tr.step(*pool[s % 8])
torch.cuda.synchronize()
stats("trained2000")
# activation scale at the GDN inputs (fp32 oracle-free: hook the conv outputs in fp32 mode)
del tr
net.eval(); hesic_amd.set_compute_dtype(torch.float32); Fn.invalidate_weight_cache()
x1, x2, Hm = (t.cuda() for t in synthetic.smooth_stereo_batch(0, 2, 512, 512))
acts = {}
import types
for n, m in net.named_modules():
    if n.endswith(("g_a_conv1", "g_a_conv2", "g_a_conv3")) and "encoder1" in n:
        m.register_forward_hook(lambda mod, i, o, n=n: acts.__setitem__(n, o.detach().float().abs()))
with torch.no_grad():
    e = net.encoder1
    t = e.g_a_conv1(x1); t1 = e.g_a_gdn1(t); t2c = e.g_a_conv2(t1); t2 = e.g_a_gdn2(t2c); t3c = e.g_a_conv3(t2)
for n, a in acts.items():
    q = torch.quantile(a.flatten()[:4000000], torch.tensor([0.1, 0.5, 0.9, 0.999], device=a.device))
    print("act |v| at", n, ["%.3g" % float(v) for v in q], "max %.3g" % float(a.max()))
