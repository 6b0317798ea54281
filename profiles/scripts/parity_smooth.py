#!/usr/bin/env python3
"""How far does a short training run on the piecewise-smooth synthetic pairs (synthetic.smooth_stereo_pair) take HESIC, and what is the
16-bit parity there?  Trains with the graph-replayed step, then evaluates one 512 x 512 smooth pair in fp32 and the 16-bit modes ON THE GPU
(the -m gpu test compares with the CPU oracle; this is the quick look that picks steps / lr / lambda).

    python profiles/scripts/parity_smooth.py [--steps 1500] [--lr 1e-4] [--lmbda 0.0067] [--size 256]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--lmbda", type=float, default=0.0067)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--pool", type=int, default=8)
    args = ap.parse_args()
    import hesic_amd
    from hesic_amd import functional as Fn, models, synthetic
    from hesic_amd.train import GraphedTrainer
    dev = "cuda"
    hesic_amd.set_compute_dtype(torch.bfloat16)
    net = models.HSIC()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.to(dev)
    tr = GraphedTrainer(net, lr=args.lr, aux_lr=1e-3, lmbda=args.lmbda)
    pool = [tuple(t.to(dev) for t in synthetic.smooth_stereo_batch(100 + 8 * i, 8, args.size, args.size)) for i in range(args.pool)]
    traj = []
    for st in range(args.steps):
        c = tr.step(*pool[st % len(pool)])
        if st % 100 == 0 or st == args.steps - 1:
            traj.append((st, round(float(c["loss"]), 3), round(float(c["bpp_loss"]), 4), round(float(c["mse_loss"]), 6)))
    torch.cuda.synchronize()
    del tr
    net.eval()
    net.update(force=True)
    Fn.invalidate_weight_cache()
    x1, x2, Hm = (t.to(dev) for t in synthetic.smooth_stereo_batch(0, 1, 512, 512))
    recs = {}
    ref = None
    for name, dt, an in (("f32", torch.float32, "auto"), ("f16-auto", torch.float16, "auto"), ("f16-x3", torch.float16, "x3"), ("f16-x2", torch.float16, "x2"), ("f16-x2:233", torch.float16, "x2"), ("f16-x2:322", torch.float16, "x2"), ("bf16-x3", torch.bfloat16, "x3")):
        hesic_amd.set_compute_dtype(dt)
        Fn.set_analysis_precision(an)
        os.environ["HESIC_X2_LAYERS"] = ",".join(name.split(":")[1]) if ":" in name else "2,2,2"
        with torch.no_grad():
            out = net(x1, x2, Hm)
            m = models.metrics_from(models.rate_distortion(out, x1, x2))
        if ref is None:
            ref = (out, m)
            recs[name] = {"bpp": m["bpp"], "psnr": m["psnr"]}
        else:
            flips = max(float((out[k].float() != ref[0][k].float()).float().mean()) for k in ("y1_hat", "y2_hat"))
            recs[name] = {"dbpp": m["bpp"] - ref[1]["bpp"], "dpsnr_db": m["psnr"] - ref[1]["psnr"], "flips": flips}
            xb = [t.expand(8, *t.shape[1:]).contiguous() for t in (x1, x2, Hm)]
            with torch.no_grad():
                for _ in range(3):
                    net(*xb)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    net(*xb)
                e1.record()
            torch.cuda.synchronize()
            recs[name]["ms_b8"] = round(e0.elapsed_time(e1) / 10, 3)
    print(json.dumps({"args": vars(args), "traj": traj, "eval_512": recs}))


if __name__ == "__main__":
    main()
