#!/usr/bin/env python3
"""Round 4: 16-bit format x analysis precision -> flips / |dbpp| / |dPSNR| against the fp32 CPU oracle on the bench workload's pairs
(512 x 512, random-init-shaped weights) and against the reference golden (256 x 256), plus the step time at B=8.

    python profiles/scripts/parity_f16.py [--pairs 4] [--model hsic|joint]
One JSON line per (dtype, analysis)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--model", default="hsic")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--modes", default="")
    args = ap.parse_args()
    import hesic_amd
    from hesic_amd import functional as Fn, models, synthetic
    from oracle import hesic_oracle as O
    kind = args.model
    net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict())
    P_cpu = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    refs = []
    with torch.no_grad():
        for p in range(args.pairs):
            x1, x2, Hm = synthetic.stereo_batch(p, 1, args.size, args.size)
            o = (O.hsic_forward if kind == "hsic" else O.hsic_joint_forward)(P_cpu, x1, x2, Hm)
            refs.append((x1, x2, Hm, {k: o[k].to(torch.int16) for k in ("y1_hat", "y2_hat")}, O.metrics(o, x1, x2)))
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{kind}_256.npz"))
    gx = synthetic.stereo_batch(0, 1, 256, 256)
    g_ref = {k: torch.from_numpy(g[k]).to(torch.int16) for k in ("y1_hat", "y2_hat")}
    g_bpp = sum(float(g["bits_" + k]) for k in ("y1", "y2", "z1", "z2")) / 256 / 256 / 2
    g_psnr = (10 * math.log10(1 / float(g["mse1"])) + 10 * math.log10(1 / float(g["mse2"]))) / 2
    xb = [t.cuda() for t in synthetic.stereo_batch(0, 8, args.size, args.size)]
    modes = [m.split(":") for m in args.modes.split(",")] if args.modes else \
        [(d, a) for d in ("bf16", "f16") for a in Fn.ANALYSIS_MODES]
    for dname, mode in modes:
        hesic_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dname])
        Fn.set_analysis_precision(mode)
        rec = {"model": kind, "dtype": dname, "analysis": mode}
        with torch.no_grad():
            fl, dbpp, dpsnr = [], [], []
            for x1, x2, Hm, ref, m in refs:
                out = net(x1.cuda(), x2.cuda(), Hm.cuda())
                mg = models.metrics_from(models.rate_distortion(out, x1.cuda(), x2.cuda()))
                fl.append(max(float((out[k].float().cpu().to(torch.int16) != ref[k]).float().mean()) for k in ("y1_hat", "y2_hat")))
                dbpp.append(mg["bpp"] - m["bpp"])
                dpsnr.append(mg["psnr"] - m["psnr"])
            rec.update(flips_worst=round(max(fl), 7), flips=[round(v, 7) for v in fl], dbpp=[round(v, 6) for v in dbpp],
                       dpsnr_db=[round(v, 6) for v in dpsnr], abs_mean_dbpp=round(abs(sum(dbpp) / len(dbpp)), 6),
                       abs_mean_dpsnr_db=round(abs(sum(dpsnr) / len(dpsnr)), 6))
            out = net(*[t.cuda() for t in gx])
            mg = models.metrics_from(models.rate_distortion(out, gx[0].cuda(), gx[1].cuda()))
            rec["golden256"] = {"flips": round(max(float((out[k].float().cpu().to(torch.int16) != g_ref[k]).float().mean()) for k in g_ref), 7),
                                "dbpp": round(mg["bpp"] - g_bpp, 6), "dpsnr_db": round(mg["psnr"] - g_psnr, 6)}
            for _ in range(5):
                net(*xb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                net(*xb)
            torch.cuda.synchronize()
            rec["ms_per_step_b8"] = round((time.perf_counter() - t0) / 30 * 1e3, 3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
