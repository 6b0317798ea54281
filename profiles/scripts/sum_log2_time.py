import torch, sys, os
sys.path.insert(0, "/root/repo")
import hesic_amd
from hesic_amd import functional as Fn, _lib as L
x = torch.rand(8, 192, 32, 32, device="cuda") * 0.9 + 0.05
xs = torch.rand(8, 128, 8, 8, device="cuda") * 0.9 + 0.05
out = torch.zeros(1, dtype=torch.float64, device="cuda")
def run(t):
    L.call("hesic_sum_log2", L.ptr(t), t.numel(), L.ptr(out), L.stream())
for t in (x, xs):
    for _ in range(5): run(t)
    out.zero_(); run(t); torch.cuda.synchronize()
    ref = float(torch.log2(t.double()).sum())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run(t)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("HESIC_SUM_LOG2_BLOCKS", "2048"), t.numel(), "us", e0.elapsed_time(e1) / 50 * 1e3, "rel err", abs(float(out) / 51 - ref) / abs(ref) if False else abs(float(out) - 51 * ref) / abs(51 * ref))
