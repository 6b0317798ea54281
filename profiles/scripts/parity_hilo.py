#!/usr/bin/env python3
"""bf16x3 ("hi/lo") analysis path: operator-level error against an fp64 CPU evaluation, and whole-model flips / |dbpp| / |dPSNR|
against the reference golden (256x256) for analysis = bf16 and bf16x3.

    python profiles/scripts/parity_hilo.py [--oracle-512]
One JSON line per case."""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def ops():
    import hesic_amd
    from hesic_amd import functional as Fn
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(1)
    B, H, W = 2, 40, 56
    x = torch.rand(B, 3, H, W, generator=g)
    w1 = (torch.rand(128, 3, 5, 5, generator=g) - 0.5) * 0.4
    b1 = (torch.rand(128, generator=g) - 0.5) * 0.1
    beta = torch.rand(128, generator=g) + 0.5
    gamma = torch.rand(128, 128, generator=g) * 0.02 + 0.1 * torch.eye(128)
    w2 = (torch.rand(128, 128, 5, 5, generator=g) - 0.5) * 0.05
    b2 = (torch.rand(128, generator=g) - 0.5) * 0.1
    w4 = (torch.rand(192, 128, 5, 5, generator=g) - 0.5) * 0.05

    def gdn_ref(v, inverse=False):
        ped = 2.0 ** -36
        be = torch.clamp(beta.double(), min=math.sqrt(1e-6 + ped)) ** 2 - ped
        ga = torch.clamp(gamma.double(), min=2.0 ** -18) ** 2 - ped
        n = F.conv2d(v * v, ga.reshape(128, 128, 1, 1), be)
        return v * torch.sqrt(n) if inverse else v / torch.sqrt(n)

    xd = x.double()
    r1 = gdn_ref(F.conv2d(xd, w1.double(), b1.double(), stride=2, padding=2))
    r2 = gdn_ref(F.conv2d(r1, w2.double(), b2.double(), stride=2, padding=2))
    r4 = F.conv2d(r2, w4.double(), None, stride=2, padding=2)

    dev = "cuda"
    from compressai.layers import GDN
    gd = GDN(128).to(dev)
    with torch.no_grad():
        gd.beta.copy_(beta); gd.gamma.copy_(gamma)
    gp, bp = gd.packer().get(gd.beta, gd.gamma, gd.beta_min)
    glo = Fn.PackedGdnLo().get(gd.gamma)
    pw = [Fn.PackedWeightHiLo() for _ in range(3)]
    with torch.no_grad():
        cols = Fn.im2col_hilo(x.to(dev), 5, 2, 2, 96)
        t1 = Fn.conv2d_hilo(cols, pw[0].get(w1.to(dev), as_1x1=True, kp=96), b1.to(dev), 96, 128, kernel_size=1, stride=1, padding=0, gdn=(gp, glo, bp, False))
        w1d, b1d = w1.to(dev), b1.to(dev)
        t1f = Fn.sconv_gdn_hilo(x.to(dev), Fn.PackedN2wHiLo().get(w1d, gd.gamma), b1d, bp, False)
        t2 = Fn.conv2d_hilo(t1, pw[1].get(w2.to(dev)), b2.to(dev), 128, 128, kernel_size=5, stride=2, padding=2, gdn=(gp, glo, bp, False))
        y = Fn.conv2d_hilo(t2, pw[2].get(w4.to(dev)), None, 128, 192, kernel_size=5, stride=2, padding=2)
    torch.cuda.synchronize()

    def val(t):
        t = t.float().cpu()
        c = t.shape[1] // 2
        return (t[:, :c] + t[:, c:]).double()

    def err(a, r):
        return float((a - r).abs().max() / r.abs().max()), float(((a - r) ** 2).mean().sqrt() / (r ** 2).mean().sqrt())
    for name, a, r in (("conv1+gdn (im2col 1x1)", val(t1), r1), ("conv1+gdn (fused hi/lo kernel)", val(t1f), r1), ("conv2+gdn", val(t2), r2), ("conv4 fp32 out", y.double().cpu(), r4)):
        mx, rms = err(a, r)
        print(json.dumps({"op": name, "max_rel_to_peak": mx, "rms_rel": rms}), flush=True)
    # single-bf16 reference point for the same chain
    hesic_amd.set_compute_dtype(torch.bfloat16)


def model(args):
    import hesic_amd
    from hesic_amd import functional as Fn, models, synthetic
    hesic_amd.set_compute_dtype(torch.bfloat16)
    for kind in ("hsic", "joint"):
        net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
        synthetic.fill_state_dict_(net.state_dict())
        P_cpu = {k: v.clone() for k, v in net.state_dict().items()}
        net = net.cuda().eval()
        cases = [(256, "golden")] + ([(512, "oracle")] if args.oracle_512 else [])
        for size, src in cases:
            x1, x2, Hm = synthetic.stereo_batch(0, 1, size, size)
            if src == "golden":
                g = np.load(os.path.join(ROOT, "tests", "golden", f"{kind}_{size}.npz"))
                ref = {"y1_hat": torch.from_numpy(g["y1_hat"]).to(torch.int16), "y2_hat": torch.from_numpy(g["y2_hat"]).to(torch.int16)}
                n = size * size
                ref_bits = {k: float(g["bits_" + k]) for k in ("y1", "y2", "z1", "z2")}
                ref_bpp = sum(ref_bits.values()) / n / 2
                ref_psnr = (10 * math.log10(1 / float(g["mse1"])) + 10 * math.log10(1 / float(g["mse2"]))) / 2
            else:
                from oracle import hesic_oracle as O
                torch.set_num_threads(min(16, os.cpu_count() or 1))
                with torch.no_grad():
                    o = (O.hsic_forward if kind == "hsic" else O.hsic_joint_forward)(P_cpu, x1, x2, Hm)
                m = O.metrics(o, x1, x2)
                ref = {k: o[k].to(torch.int16) for k in ("y1_hat", "y2_hat")}
                ref_bpp, ref_psnr = m["bpp"], m["psnr"]
                ref_bits = None
            for mode in ("bf16", "bf16x3"):
                Fn.set_analysis_precision(mode)
                with torch.no_grad():
                    out = net(x1.cuda(), x2.cuda(), Hm.cuda())
                    mg = models.metrics_from(models.rate_distortion(out, x1.cuda(), x2.cuda()))
                flips = {k: float((out[k].float().cpu().to(torch.int16) != ref[k]).float().mean()) for k in ("y1_hat", "y2_hat")}
                rec = {"model": kind, "size": size, "ref": src, "analysis": mode,
                       "flips_y1": round(flips["y1_hat"], 6), "flips_y2": round(flips["y2_hat"], 6),
                       "abs_dbpp": round(abs(mg["bpp"] - ref_bpp), 6), "abs_dpsnr_db": round(abs(mg["psnr"] - ref_psnr), 6),
                       "bpp_ref": round(ref_bpp, 5), "psnr_ref": round(ref_psnr, 4)}
                if ref_bits:
                    rec["dbits"] = {k: round(mg["bits"][k] - ref_bits[k], 2) for k in ref_bits}
                print(json.dumps(rec), flush=True)
            Fn.set_analysis_precision("bf16x3")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle-512", action="store_true")
    args = ap.parse_args()
    ops()
    model(args)
