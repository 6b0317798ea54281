#!/usr/bin/env python3
"""Parity at a TRAINED operating point instead of the random-weight regime (bpp ~5.5, PSNR ~5.6 dB, likelihoods in the tails).

No checkpoints exist offline, so the model is trained here for a few thousand graph-replayed steps on synthetic stereo pairs
(fresh pairs every step), which takes it to a low-rate / moderate-quality regime; then one eval forward in bf16 (fp32
latents) and one in fp32 are compared against the CPU oracle run on the SAME trained weights.

    python profiles/scripts/parity_trained.py [--steps 3000] [--lr 1e-4] [--size 512]
Prints one JSON line per storage mode."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--aux-lr", type=float, default=1e-3)
    ap.add_argument("--lmbda", type=float, default=0.0067)
    ap.add_argument("--size", type=int, default=512, help="evaluation size")
    ap.add_argument("--model", choices=["hsic", "joint"], default="hsic")
    args = ap.parse_args()
    import hesic_amd
    from hesic_amd import models, synthetic
    from hesic_amd.train import GraphedTrainer
    from oracle import hesic_oracle as O
    hesic_amd.set_compute_dtype(torch.bfloat16)
    net = (models.HSIC if args.model == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.cuda()
    tr = GraphedTrainer(net, lr=args.lr, aux_lr=args.aux_lr, lmbda=args.lmbda)
    pool = [tuple(t.cuda() for t in synthetic.stereo_batch(100 + 8 * i, 8, 256, 256)) for i in range(16)]      # 128 distinct pairs
    t0 = time.perf_counter()
    for s in range(args.steps):
        c = tr.step(*pool[s % len(pool)])
        if s % 500 == 0 or s == args.steps - 1:
            print(f"# step {s}: loss {float(c['loss']):.3f} bpp {float(c['bpp_loss']):.3f} mse {float(c['mse_loss']):.5f} aux {float(c['aux_loss']):.1f}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    net.eval()
    P = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}
    x1, x2, Hm = synthetic.stereo_batch(0, 1, args.size, args.size)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = (O.hsic_forward if args.model == "hsic" else O.hsic_joint_forward)(P, x1, x2, Hm)
    mr = O.metrics(ref, x1, x2)
    for dt in (torch.bfloat16, torch.float32):
        hesic_amd.set_compute_dtype(dt)
        with torch.no_grad():
            out = net(x1.cuda(), x2.cuda(), Hm.cuda())
            m = models.metrics_from(models.rate_distortion(out, x1.cuda(), x2.cuda()))
        flips = {k: float((out[k].float().cpu() != ref[k]).float().mean()) for k in ("y1_hat", "y2_hat")}
        print(json.dumps({"model": args.model, "trained_steps": args.steps, "train_seconds": round(train_s, 1), "eval": f"{args.size}x{args.size} pair 0",
                          "storage": "bf16 maps + fp32 latents" if dt == torch.bfloat16 else "fp32",
                          "bpp_oracle": round(mr["bpp"], 5), "psnr_oracle": round(mr["psnr"], 4), "abs_dbpp": round(abs(m["bpp"] - mr["bpp"]), 6),
                          "abs_dpsnr_db": round(abs(m["psnr"] - mr["psnr"]), 6), "latent_flips": {k: round(v, 6) for k, v in flips.items()},
                          "nonzero_latents": round(float((ref["y1_hat"] != 0).float().mean()), 4),
                          "met_1e-3": bool(abs(m["bpp"] - mr["bpp"]) < 1e-3 * max(1.0, mr["bpp"]) and abs(m["psnr"] - mr["psnr"]) < 1e-3)}), flush=True)


if __name__ == "__main__":
    main()
