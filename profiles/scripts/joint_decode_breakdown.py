#!/usr/bin/env python3
"""Where a HESIC+ wavefront decode spends its time: host range decoding vs everything else (graph replay, table launch, copies, sync)."""
import json, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd
from hesic_amd import models, synthetic, _host
hesic_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("HESIC_DTYPE", "f16")])
net = models.HSICJoint(); synthetic.fill_state_dict_(net.state_dict()); net = net.cuda().eval(); net.update(force=True)
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 1, 512, 512))
acc = {"decode_grid_s": 0.0, "calls": 0, "symbols": 0, "table_bytes": 0}
orig = _host.RangeDecoder.decode_grid
def timed(self, cdf, *a):
    t = time.perf_counter(); r = orig(self, cdf, *a); acc["decode_grid_s"] += time.perf_counter() - t; acc["calls"] += 1; acc["symbols"] += r.size; acc["table_bytes"] += cdf.nbytes; return r
_host.RangeDecoder.decode_grid = timed
with tempfile.TemporaryDirectory() as td:
    for rep in range(3):
        enc = net.compress(x1, x2, Hm, "p", td)
        for k in acc: acc[k] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec = net.decompress(None, None, Hm, "p", td)
        torch.cuda.synchronize(); total = time.perf_counter() - t0
print(json.dumps({"total_s": round(total, 4), **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in acc.items()},
                  "per_step_us_other": round((total - acc["decode_grid_s"]) / max(acc["calls"], 1) * 1e6, 1)}))
