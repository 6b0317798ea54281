#!/usr/bin/env python3
"""N eval forwards of HESIC / HESIC+ (B=8 or HESIC_BATCH, 512x512, f16 or HESIC_DTYPE, single stream) -- the workload behind the rocprofv3 --pmc passes
(profiles/make_pmc_json.py) and the kernel traces:  python profiles/scripts/forward_n.py [hsic|joint] [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hesic_amd  # noqa: E402
from hesic_amd import models, synthetic  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "hsic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hesic_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("HESIC_DTYPE", "f16")])     # analysis mode: HESIC_ANALYSIS
net = models.HSIC() if which == "hsic" else models.HSICJoint()
synthetic.fill_state_dict_(net.state_dict())
net = net.cuda().eval()
net.update(force=True)
x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, int(os.environ.get("HESIC_BATCH", "8")), 512, 512))
with torch.no_grad():
    for _ in range(n):
        out = net(x1, x2, Hm)
        models.rate_distortion(out, x1, x2)
torch.cuda.synchronize()
