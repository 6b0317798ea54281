#!/usr/bin/env python3
"""Time build variants of csrc/enh.hip's ResidualBlock kernel back to back (libraries profiles/scripts/micro/en_abl/libenhv_<name>.so, built by hand with
-D flags next to a stub of the error plumbing; never shipped): B=8 512^2, random data, plain and with the outer skip."""
import ctypes as C, glob, os, sys, torch
B, H, W = 8, 512, 512
torch.manual_seed(0)
x = (torch.randn(B, H, W, 32, device="cuda") * 0.5).half()
sk = (torch.randn(B, H, W, 32, device="cuda") * 0.5).half()
y = torch.empty_like(x)
w1, w2 = (torch.randn(32, 32, 3, 3, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.randn(32, device="cuda") * 0.1 for _ in range(2))
names = sys.argv[1:] or sorted(os.path.basename(f)[8:-3] for f in glob.glob("profiles/scripts/micro/en_abl/libenhv_*.so"))
ref = None
for rnd in range(2):
  for name in names:
    lib = C.CDLL(f"profiles/scripts/micro/en_abl/libenhv_{name}.so")
    f = lib.hesic_resblock_c32_forward
    f.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p]
    out = []
    for res in (None, sk):
        def run():
            rc = f(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 2, None if res is None else res.data_ptr(), y.data_ptr(), B, H, W, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
        if res is None:
            if ref is None: ref = y.clone()
            d = float((y.float() - ref.float()).abs().max())
    print(f"{name:24s} plain {out[0]:7.1f} us   outer skip {out[1]:7.1f} us   max |diff| to the first variant {d:.4g}")
