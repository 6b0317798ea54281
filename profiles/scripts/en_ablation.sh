#!/bin/bash
# Where the time of the one-launch ResidualBlock (csrc/enh.hip: c32_resblock_r3_kernel; rounds 3 - 5: c32_resblock_kernel, profiles/experiments/) goes: compile-time ablations (-DRB_ABL=<bits>) of the kernel,
# each built into its own small library next to a stub of the error plumbing, timed back to back at B=8 512^2.  Never in the shipped libraries.
# Build (in the build container, hipcc cross-compiles; the .so files travel with the snapshot):
#   for abl in 0 1 2 3 4 8 16 7; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHESIC_H16_IS_F16=1 -DRB_ABL=$abl -Ihesic_amd/csrc -c hesic_amd/csrc/enh.hip -o /tmp/e.o
#     && hipcc --offload-arch=gfx950 -shared -fPIC /tmp/e.o stub.o -o profiles/scripts/micro/en_abl/libenh_$abl.so; done      (stub.cpp: hesic_set_error)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'P'
import ctypes as C, torch
B, H, W = 8, 512, 512
x = (torch.randn(B, H, W, 32, device="cuda") * 0.5).half()
y = torch.empty_like(x)
w1, w2 = (torch.randn(32, 32, 3, 3, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.randn(32, device="cuda") * 0.1 for _ in range(2))
names = {0: "full kernel", 1: "no MFMAs", 2: "no fragment reads", 3: "no MFMAs, no fragment reads", 4: "no identity loads / output stores",
         8: "no producer work", 16: "no consumer work", 7: "skeleton (epilogues' LDS + VALU, barriers, halo loads)"}
for abl, name in names.items():
    try:
        lib = C.CDLL(f"profiles/scripts/micro/en_abl/libenh_{abl}.so")
    except OSError as e:
        print(abl, "load failed", e); continue
    f = lib.hesic_resblock_c32_forward
    f.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p]
    def run():
        rc = f(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 2, None, y.data_ptr(), B, H, W, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"RB_ABL={abl:2d}  {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us   {name}")
P
