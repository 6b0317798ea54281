#!/usr/bin/env python3
"""gmm_hyper_y2's shared-input first layers (320 -> 3 x 128, 5x5 s1 at 32x32, B=8) as the grouped implicit-GEMM launch, graph-replayed:
python profiles/scripts/hs2_first_layer_time.py   (A/B: HESIC_IGEMM_BM=32|64|128)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd  # noqa: E402
from hesic_amd import functional as Fn  # noqa: E402

hesic_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
B, S, cin = 8, 32, 320
x = (torch.randn(B, cin, S, S, device="cuda") * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
ws = [torch.randn(128, cin, 5, 5, device="cuda") * 0.02 for _ in range(3)]
bs = [torch.zeros(128, device="cuda") for _ in range(3)]
pg = Fn.PackedGroup()
f = lambda: Fn.conv2d_grouped(x, ws, bs, pg, kernel_size=5, stride=1, padding=2, shared_input=True, acts=[1, 2, 2])
with torch.no_grad():
    for _ in range(3):
        f()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(30):
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 30 * 1e3
fl = 2.0 * B * S * S * cin * 25 * 384
print(f"h_s2 first layers: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s  BM={os.environ.get('HESIC_IGEMM_BM', 'auto')}")
