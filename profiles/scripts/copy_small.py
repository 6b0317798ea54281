#!/usr/bin/env python3
"""Copy ceiling at the sizes of the small streaming kernels: fp32 copy of N MB in -> N MB out, back-to-back launches."""
import torch
for mb in (25, 50, 80, 134):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e3
    print(f"copy {mb} MB -> {mb} MB: {t:6.1f} us = {2 * mb / t:.2f} TB/s")
