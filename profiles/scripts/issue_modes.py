#!/usr/bin/env python3
"""ms per forward of the three ways to issue the eval forward: eager launches, one whole-forward HIP graph, a plan of single-stream
graphs (models.SegmentedForward).   python profiles/scripts/issue_modes.py [hsic|joint] [batch]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd
from hesic_amd import models, synthetic
kind = sys.argv[1] if len(sys.argv) > 1 else "hsic"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hesic_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("HESIC_DTYPE", "f16")])
net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
synthetic.fill_state_dict_(net.state_dict()); net = net.cuda().eval()
pool = [tuple(t.cuda() for t in synthetic.stereo_batch(4 * j, B, 512, 512)) for j in range(4)]
def timed(fn, n=60):
    for i in range(15): fn(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def eager(i):
    with torch.no_grad(): return net(*pool[i % 4])
res = {"model": kind, "batch": B, "eager_ms": round(timed(eager), 3)}
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(8): eager(i)
res["eager_host_issue_ms"] = round((time.perf_counter() - t) / 8 * 1e3, 3)          # host time to issue a forward (queue not waited for)
torch.cuda.synchronize()
g1 = models.GraphedForward(net, *pool[0], with_metrics=False)
res["graph_ms"] = round(timed(lambda i: g1(*pool[i % 4])), 3)
del g1
g2 = models.SegmentedForward(net, *pool[0])
res["segments"] = g2.n_graphs; res["plan_ops"] = len(g2.plan)
res["segmented_ms"] = round(timed(lambda i: g2(*pool[i % 4])), 3)
print(json.dumps(res))
