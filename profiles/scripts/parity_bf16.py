#!/usr/bin/env python3
"""bf16-mode parity against the reference golden (256x256) and the CPU oracle (512x512): latent flips, |dbpp|, |dPSNR|,
with the latents / sigma / mu stored as fp32 (default) and as bf16 (HESIC_BF16_LATENTS=1, the round-1 behaviour).

    python profiles/scripts/parity_bf16.py [--oracle-512]
Prints one JSON line per (model, size, latent storage)."""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle-512", action="store_true", help="also compare pair 0 at 512x512 against the CPU oracle (seconds of CPU time)")
    args = ap.parse_args()
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
                ref_bpp = sum(float(g["bits_" + k]) for k in ("y1", "y2", "z1", "z2")) / n / 2
                ref_psnr = (10 * math.log10(1 / float(g["mse1"])) + 10 * math.log10(1 / float(g["mse2"]))) / 2
            else:
                from oracle import hesic_oracle as O
                torch.set_num_threads(min(16, os.cpu_count() or 1))
                with torch.no_grad():
                    o = (O.hsic_forward if kind == "hsic" else O.hsic_joint_forward)(P_cpu, x1, x2, Hm)
                m = O.metrics(o, x1, x2)
                ref = {k: o[k].to(torch.int16) for k in ("y1_hat", "y2_hat")}
                ref_bpp, ref_psnr = m["bpp"], m["psnr"]
            for fp32_lat in (True, False):
                Fn.FP32_LATENTS = fp32_lat
                with torch.no_grad():
                    out = net(x1.cuda(), x2.cuda(), Hm.cuda())
                    mg = models.metrics_from(models.rate_distortion(out, x1.cuda(), x2.cuda()))
                flips = {k: float((out[k].float().cpu().to(torch.int16) != ref[k]).float().mean()) for k in ("y1_hat", "y2_hat")}
                print(json.dumps({"model": kind, "size": size, "ref": src, "latents": "fp32" if fp32_lat else "bf16",
                                  "flips_y1": round(flips["y1_hat"], 6), "flips_y2": round(flips["y2_hat"], 6),
                                  "abs_dbpp": round(abs(mg["bpp"] - ref_bpp), 6), "abs_dpsnr_db": round(abs(mg["psnr"] - ref_psnr), 6),
                                  "bpp_ref": round(ref_bpp, 5), "psnr_ref": round(ref_psnr, 4)}), flush=True)
            Fn.FP32_LATENTS = True


if __name__ == "__main__":
    main()
