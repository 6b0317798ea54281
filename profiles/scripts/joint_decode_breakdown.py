#!/usr/bin/env python3
"""Where a HESIC+ wavefront decode (512 x 512 pair) spends its time: the two wavefront walks (host range decoding inside them timed
separately) against everything around them (side file, bottlenecks, hyper-synthesis, decoder1, warp, third analysis pass, decoder2)."""
import json, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd
from hesic_amd import models, synthetic, _host
hesic_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("HESIC_DTYPE", "f16")])
net = models.HSICJoint(); synthetic.fill_state_dict_(net.state_dict()); net = net.cuda().eval(); net.update(force=True)
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 1, 512, 512))
acc = {"walk_s": 0.0, "host_decode_s": 0.0, "groups": 0}
orig_walk = net._decode_view_graphed
def walk(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig_walk(*a, **k); torch.cuda.synchronize(); acc["walk_s"] += time.perf_counter() - t; return r
net._decode_view_graphed = walk
orig_raw = _host.RangeDecoder.decode_grid_raw
def raw(self, *a):
    t = time.perf_counter(); orig_raw(self, *a); acc["host_decode_s"] += time.perf_counter() - t; acc["groups"] += 1
_host.RangeDecoder.decode_grid_raw = raw
with tempfile.TemporaryDirectory() as td:
    for rep in range(3):
        enc = net.compress(x1, x2, Hm, "p", td)
        for k in acc: acc[k] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec = net.decompress(None, None, Hm, "p", td)
        torch.cuda.synchronize(); total = time.perf_counter() - t0
g = max(acc["groups"], 1)
print(json.dumps({"total_s": round(total, 4), "walks_s": round(acc["walk_s"], 4), "around_s": round(total - acc["walk_s"], 4),
                  "host_decode_s": round(acc["host_decode_s"], 4), "groups": acc["groups"],
                  "per_group_us": round(acc["walk_s"] / g * 1e6, 1), "per_group_host_decode_us": round(acc["host_decode_s"] / g * 1e6, 1)}))
