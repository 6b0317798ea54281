#!/usr/bin/env python3
"""hesic_eb_backward on the hyper-latents of a training step (B = 8, 128 x 8 x 8), HIP-event time per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import hesic_amd
    from hesic_amd import _lib as L, functional as Fn
    from compressai.entropy_models import EntropyBottleneck
    hesic_amd.set_compute_dtype(torch.bfloat16)
    eb = EntropyBottleneck(128).cuda().train()
    z = (torch.randn(8, 128, 8, 8, device="cuda") * 3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    for _ in range(3):
        zh, lik = eb(z)
        (torch.log(lik).sum()).backward()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            zh, lik = eb(z)
            (torch.log(lik).sum()).backward()
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if "eb_" in e.key:
            print(f"{e.key[:60]:60s} {e.count:4d} {e.device_time_total / e.count:8.1f} us")


if __name__ == "__main__":
    main()
