#!/usr/bin/env python3
"""What a plain copy / fill / add reaches on this box (the ceiling the streaming kernels are priced against in DESIGN.md)."""
import torch


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # microseconds


for mb in (134, 268, 536):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device="cuda", dtype=torch.bfloat16)
    y, z = torch.empty_like(x), torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy {mb} MB: {t:6.1f} us -> {2 * mb / t:.2f} TB/s (read + write)")
    t = timeit(lambda: y.fill_(1.0))
    print(f"fill {mb} MB: {t:6.1f} us -> {mb / t:.2f} TB/s (write)")
    t = timeit(lambda: torch.add(x, y, out=z))
    print(f"add  {mb} MB x3: {t:6.1f} us -> {3 * mb / t:.2f} TB/s")
