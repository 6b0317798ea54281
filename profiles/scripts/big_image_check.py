"""Sanity run on large images (2 pairs of 2048 x 1536, HESIC and HESIC+): the f16 default against the fp32 mode on the same weights --
bpp / PSNR / flipped latents; exercises the 32-bit offset paths of every inference kernel at ~12x the benchmark's image area.
    python profiles/scripts/big_image_check.py"""
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hesic_amd
from hesic_amd import models, synthetic, functional as Fn
res = {}
for kind in ("hsic", "joint"):
    for dt in (torch.float16, torch.float32):
        hesic_amd.set_compute_dtype(dt)
        net = (models.HSIC if kind == "hsic" else models.HSICJoint)(); synthetic.fill_state_dict_(net.state_dict()); net = net.cuda().eval(); net.update(force=True)
        x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(3, 2, 2048, 1536))
        with torch.no_grad():
            o = net(x1, x2, Hm)
            m = models.metrics_from(models.rate_distortion(o, x1, x2))
        res[(kind, str(dt))] = (m["bpp"], m["psnr"], o["y1_hat"].float().cpu())
        print(kind, dt, "bpp %.6f psnr %.5f" % (m["bpp"], m["psnr"]), flush=True)
    a, b = res[(kind, "torch.float16")], res[(kind, "torch.float32")]
    print(kind, "f16 vs f32: dbpp %.2e dpsnr %.2e flips %.2e" % (a[0] - b[0], a[1] - b[1], float((a[2] != b[2]).float().mean())))
