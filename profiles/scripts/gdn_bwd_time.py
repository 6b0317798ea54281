#!/usr/bin/env python3
"""GDN backward (bf16 storage, 128 channels) on one map, timed with HIP events over back-to-back calls.

    python profiles/scripts/gdn_bwd_time.py [--batch 8] [--size 256] [--inverse] [--dump grads.pt]
HESIC_GDN_BWD_SPLIT=1 selects the three-launch form of round 2 (dx + dn, 1-tap weight gradient over 256 pixel slices, reduce)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--inverse", action="store_true")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dump", default="")
    args = ap.parse_args()
    import hesic_amd
    from hesic_amd import functional as Fn, synthetic
    hesic_amd.set_compute_dtype(torch.bfloat16)
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=4)
    torch.manual_seed(0)
    B, S = args.batch, args.size
    x = (torch.randn(B, 128, S, S, device="cuda") * 1.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, 128, S, S, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    beta, gamma = sd["g.beta"].cuda(), sd["g.gamma"].cuda()
    f = lambda: Fn._gdn_backward(x, g, beta, gamma, args.inverse, 1e-6)
    for _ in range(3):
        out = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        out = f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    mb = 3 * x.numel() * 2 / 1e6
    print(f"gdn backward B={B} {S}x{S} inverse={int(args.inverse)}: {us:.1f} us  ({mb / us / 1e6 * 1e6:.0f} GB/s on x, gy, dx)  split={os.environ.get('HESIC_GDN_BWD_SPLIT', '0')}")
    if args.dump:
        torch.save([t.float().cpu() for t in out], args.dump)


if __name__ == "__main__":
    main()
