#!/usr/bin/env python3
"""HESIC+ bit-stream: decode time of one 512 x 512 pair, wavefront payload (one batched device step per group of independent
pixels) against raster payload (the reference's order, one step per pixel).  Prints one JSON line."""
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import hesic_amd
    from hesic_amd import models, synthetic
    hesic_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("HESIC_DTYPE", "f16")])
    net = models.HSICJoint()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.cuda().eval()
    net.update(force=True)
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 1, size, size))
    rec = {"size": size}
    with tempfile.TemporaryDirectory() as td:
        for order in ("wavefront", "raster"):
            for rep in range(2):          # second run: packed weights and table kernels warm
                enc = net.compress(x1, x2, Hm, order, td, order=order)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dec = net.decompress(None, None, Hm, order, td)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            assert torch.equal(dec["y2_hat"].float(), enc["y2_hat"].float())
            rec[order] = {"decode_s": round(dt, 3), "encode_s": round(enc["enctime"], 3), "bpp_real": round(enc["bpp_real"], 4)}
    rec["speedup"] = round(rec["raster"]["decode_s"] / rec["wavefront"]["decode_s"], 2)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
