import os, sys, torch, collections
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
evs = list(prof.events())
names = collections.Counter(e.name for e in evs if e.device_type == torch.autograd.DeviceType.CPU and ("emcpy" in e.name or "emset" in e.name))
print("runtime calls:", names)
# the aten ops that enclose a memcpy
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
mem = [e for e in cpu if "emcpy" in e.name]
ops = collections.Counter()
for m in mem:
    enc = [e for e in cpu if e.name.startswith("aten::") and e.time_range.start <= m.time_range.start and e.time_range.end >= m.time_range.end]
    enc.sort(key=lambda e: e.time_range.end - e.time_range.start)
    outer = enc[-1] if enc else None
    st = [s for s in (outer.stack if outer is not None and outer.stack else []) if "hesic_amd" in s or "compressai" in s][:3]
    ops[(outer.name if outer is not None else "?", tuple(s.split("/")[-1] for s in st), tuple(outer.input_shapes) if outer is not None and outer.input_shapes else ())] += 1
for k, n in ops.most_common(40):
    print(n, k)
