#!/usr/bin/env python3
"""Cycle stamps (clock64) of block 0 of the rolling ResidualBlock kernel, stages 2..5, every wave: where a stage's cycles go.  Needs a build of
csrc/enh.hip with -DR3_STAMP (profiles/scripts/micro/en_abl/libenhv_stamp.so)."""
import ctypes as C, torch
B, H, W = 8, 512, 512
x = (torch.randn(B, H, W, 32, device="cuda") * 0.5).half(); y = torch.empty_like(x)
w1, w2 = (torch.randn(32, 32, 3, 3, device="cuda") * 0.05 for _ in range(2)); b1, b2 = (torch.randn(32, device="cuda") * 0.1 for _ in range(2))
lib = C.CDLL("profiles/scripts/micro/en_abl/libenhv_stamp.so")
st = torch.zeros(4 * 8 * 8, dtype=torch.int64, device="cuda")
lib.hesic_en_stamp_buffer.argtypes = [C.c_void_p]
assert lib.hesic_en_stamp_buffer(st.data_ptr()) == 0
f = lib.hesic_resblock_c32_forward
f.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p]
for _ in range(3):
    assert f(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 2, None, y.data_ptr(), B, H, W, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
t = st.cpu().view(4, 8, 8)
t0 = int(t[0, :, 0].min())
names = ["stage top", "own DMA landed", "barrier 1 passed", "next DMA issued", "rows done", "barrier 2 passed"]
for s in range(4):
    print(f"stage {s + 2}: (cycles since the first stamp)   waves 0-3 producers, 4-7 consumers")
    for k in range(6):
        print(f"  {names[k]:18s} " + " ".join(f"{int(t[s, w, k]) - t0:7d}" for w in range(8)))
