#!/usr/bin/env python3
"""Cycle stamps of the phases of n2w_gdn_hilo_kernel (tile iteration 3 of block 0, waves 0 and 4): needs the library built with -DN2W_DBG
    (cd hesic_amd/csrc && touch sconv_hilo.hip && make EXTRA=-DN2W_DBG) -- a debug build, rebuild without it afterwards."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd
from hesic_amd import functional as Fn, synthetic, _lib as L
from compressai.layers import GDN

x = synthetic.stereo_batch(0, 8, 512, 512)[0].cuda()
g = torch.Generator().manual_seed(0)
w = ((torch.rand(128, 3, 5, 5, generator=g) - 0.5) * 0.4).cuda()
b = ((torch.rand(128, generator=g) - 0.5) * 0.1).cuda()
gd = GDN(128).cuda()
gp, bp = gd.packer().get(gd.beta, gd.gamma, gd.beta_min)
img = Fn.PackedN2wHiLo().get(w, gd.gamma, True)
with torch.no_grad():
    for _ in range(5):
        y = Fn.sconv_gdn_hilo(x, img, b, bp, False, True)
    torch.cuda.synchronize()
buf = (ctypes.c_longlong * 32)()
lib = L.lib()
lib.hesic_debug_n2w_read.argtypes = [ctypes.c_void_p]
print("rc", lib.hesic_debug_n2w_read(buf))
names = ["top", "split done", "barrier1 passed", "conv done", "gdn done", "barrier2 passed", "rsqrt done", "stores issued"]
for wv in range(2):
    t = [buf[wv * 16 + k] for k in range(8)]
    print(f"wave {wv * 4}: " + "  ".join(f"{names[k]} +{t[k] - t[k - 1]}" for k in range(1, 8)) + f"   | iteration {t[7] - t[0]}")
for wv in range(2):
    t = [buf[wv * 16 + k] for k in range(16)]
    seq = [(5, "barrier2"), (6, "rsqrt"), (8, "pass0 packed+written"), (9, "pass0 read back"), (10, "pass0 stores issued"), (11, "pass1 written"), (12, "pass1 read"), (13, "pass1 stores")]
    print(f"wave {wv * 4} V phase: " + "  ".join(f"{n} +{t[k] - t[seq[i - 1][0]]}" for i, (k, n) in enumerate(seq) if i))
print("wave4.top - wave0.top:", buf[16] - buf[0])
