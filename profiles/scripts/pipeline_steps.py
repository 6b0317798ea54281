#!/usr/bin/env python3
"""Do two eval forwards in flight beat one?  The forward's critical path (enc1 -> dec1 -> third analysis pass -> hyper-synthesis 2 -> dec2) is
about as long as the step itself; consecutive batches are independent, so a serving loop may start batch k+1 on another stream while batch k
is in its tail.  Times N steps issued on ONE outer stream against the same steps alternating over D outer streams.

    python profiles/scripts/pipeline_steps.py [--model hsic] [--batch 8] [--depth 2] [--steps 60]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hsic")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--steps", type=int, default=60)
    args = ap.parse_args()
    import hesic_amd
    from hesic_amd import models, synthetic
    hesic_amd.set_compute_dtype(torch.float16)
    net = (models.HSIC if args.model == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.cuda().eval()
    pool = [tuple(t.cuda() for t in synthetic.stereo_batch(4 * j, args.batch, 512, 512)) for j in range(4)]
    outer = [torch.cuda.Stream() for _ in range(args.depth)]

    def run(depth, n):
        cur = torch.cuda.current_stream()
        for s in outer:
            s.wait_stream(cur)
        keep = []
        for i in range(n):
            a, b, h = pool[i % 4]
            if depth == 1:
                with torch.no_grad():
                    o = net(a, b, h)
                    keep.append(models.rate_distortion(o, a, b))
            else:
                with torch.cuda.stream(outer[i % depth]), torch.no_grad():
                    o = net(a, b, h)
                    keep.append(models.rate_distortion(o, a, b))
            if len(keep) > 4:
                keep.pop(0)
        for s in outer:
            cur.wait_stream(s)
        return keep

    for depth in (1, args.depth, 1, args.depth):
        run(depth, 20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(depth, args.steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print(f"{args.model} B={args.batch} depth {depth}: {ms:.3f} ms/step  {args.batch / ms * 1e3:.0f} pairs/s  (GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')})")


if __name__ == "__main__":
    main()
