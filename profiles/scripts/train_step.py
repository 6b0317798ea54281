#!/usr/bin/env python3
"""Training-step timing: eager Trainer vs GraphedTrainer (HIP graph replay), HESIC or HESIC+.

    python profiles/scripts/train_step.py [--size 256] [--batch 8] [--model hsic|joint] [--steps 10] [--only g|e]
rocprofv3 --kernel-trace --stats -- python profiles/scripts/train_step.py --only g    # kernel census of the graphed step
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--model", choices=["hsic", "joint"], default="hsic")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only", choices=["g", "e", "both"], default="both")
    args = ap.parse_args()
    import hesic_amd
    from hesic_amd import models, synthetic
    from hesic_amd.train import GraphedTrainer, Trainer
    hesic_amd.set_compute_dtype(torch.bfloat16)
    x1, x2, Hm = synthetic.stereo_batch(0, min(args.batch, 4), args.size, args.size)
    reps = -(-args.batch // x1.shape[0])
    x1, x2, Hm = (t.repeat(reps, *([1] * (t.dim() - 1)))[:args.batch].cuda() for t in (x1, x2, Hm))
    order = {"g": (GraphedTrainer,), "e": (Trainer,), "both": (Trainer, GraphedTrainer)}[args.only]
    for cls in order:
        net = (models.HSIC if args.model == "hsic" else models.HSICJoint)()
        synthetic.fill_state_dict_(net.state_dict())
        tr = cls(net.cuda(), lmbda=0.0067)
        for _ in range(5):
            c = tr.step(x1, x2, Hm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            c = tr.step(x1, x2, Hm)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(f"{cls.__name__}: {args.model} B={args.batch} {args.size}x{args.size} bf16: {dt * 1e3:.2f} ms/step -> {args.batch / dt:.1f} pairs/s, "
              f"loss {float(c['loss']):.3f}", flush=True)


if __name__ == "__main__":
    main()
