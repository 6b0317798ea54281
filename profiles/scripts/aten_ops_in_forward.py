import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hesic_amd
from hesic_amd import models, synthetic
hesic_amd.set_compute_dtype(torch.float16)
net = models.HSIC(); synthetic.fill_state_dict_(net.state_dict()); net = net.cuda().eval()
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 8, 512, 512))
with torch.no_grad():
    for _ in range(5):
        o = net(x1, x2, Hm); models.rate_distortion(o, x1, x2)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    with torch.no_grad():
        o = net(x1, x2, Hm); rd = models.rate_distortion(o, x1, x2)
    torch.cuda.synchronize()
import collections
ops = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::neg", "aten::fill_", "aten::zero_", "aten::cat", "aten::add", "aten::mul", "aten::sub", "aten::div", "aten::empty_like", "aten::inverse", "aten::linalg_inv", "aten::sum", "aten::stack", "aten::index", "aten::select", "aten::slice"):
        st = [s for s in (e.stack or []) if "hesic_amd" in s or "bench" in s]
        ops[(e.name, tuple(st[:2]))] += 1
for (name, st), n in ops.most_common(60):
    print(n, name, " <- ", " | ".join(s.split("/")[-1] for s in st))
ker = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        ker[e.name[:60]] += 1
print([ (k,v) for k,v in ker.items() if "copy" in k.lower() or "Memcpy" in k or "at::" in k])
