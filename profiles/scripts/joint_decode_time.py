#!/usr/bin/env python3
"""HESIC+ bit-stream: decode time (median of the warm runs; min and max beside it) of one 512 x 512 pair, wavefront payload (one batched device step per group of independent
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
            times = []
            for rep in range(7 if order == "wavefront" else 2):          # first run: packs weights, captures the group graphs, loads kernels
                enc = net.compress(x1, x2, Hm, order, td, order=order)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dec = net.decompress(None, None, Hm, order, td)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            assert torch.equal(dec["y2_hat"].float(), enc["y2_hat"].float())
            warm = sorted(times[1:])
            rec[order] = {"decode_s": round(warm[len(warm) // 2], 4), "decode_s_min": round(warm[0], 4), "decode_s_max": round(warm[-1], 4),
                          "runs": len(warm), "encode_s": round(enc["enctime"], 3), "bpp_real": round(enc["bpp_real"], 4)}
    rec["speedup"] = round(rec["raster"]["decode_s"] / rec["wavefront"]["decode_s"], 2)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
