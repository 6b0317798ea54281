#!/usr/bin/env python3
"""Weight gradient of one wide 5x5 stride-2 layer (hesic_conv2d_wgrad_partial = the split-K MFMA launch alone, and _direct = with its
finishing pass), timed with HIP events, row kernel (round 5) against the one-tap-per-block kernel.

    python profiles/scripts/wgrad_layer_time.py [--layer conv2|deconv3|conv3|deconv2] [--batch 8] [--iters 20]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SHAPES = {"conv2": (0, 256), "deconv3": (1, 128), "conv3": (0, 128), "deconv2": (1, 64)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="conv2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--blocks", default="", help="comma list of HESIC_WGRAD_ROW_BLOCKS values to try")
    args = ap.parse_args()
    import hesic_amd
    from hesic_amd import _lib as L
    hesic_amd.set_compute_dtype(torch.bfloat16)
    tr, S = SHAPES[args.layer]
    B = args.batch
    Ho = S * 2 if tr else S // 2
    torch.manual_seed(0)
    x = (torch.randn(B, 128, S, S, device="cuda") * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = (torch.randn(B, 128, Ho, Ho, device="cuda") * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    d = L.ConvDesc(B, S, S, 128, Ho, Ho, 128, 5, 5, 2, 2, tr, L.BF16, 0, 0, 128, 0, 128, 0, 0)
    flops = 2.0 * B * (S * S if tr else Ho * Ho) * 128 * 128 * 25
    st = L.stream()
    res = {}
    outs = {}
    variants = [("tap", {"HESIC_WGRAD_ROW": "0"})]
    for nb in ([int(v) for v in args.blocks.split(",")] if args.blocks else [256]):
        variants.append((f"row{nb}", {"HESIC_WGRAD_ROW": "1", "HESIC_WGRAD_ROW_MINQ": "0", "HESIC_WGRAD_ROW_BLOCKS": str(nb)}))
    for name, env in variants:
        for k in ("HESIC_WGRAD_ROW", "HESIC_WGRAD_ROW_MINQ", "HESIC_WGRAD_ROW_BLOCKS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        nws = int(L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(d)))
        ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
        dw = torch.zeros(128, 128, 5, 5, device="cuda")
        db = torch.zeros(128, device="cuda")
        t = {}
        for mode in ("partial", "direct"):
            def f():
                if mode == "partial":
                    L.call("hesic_conv2d_wgrad_partial", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(ws), nws, st)
                else:
                    L.call("hesic_conv2d_wgrad_direct", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(db), 0, L.ptr(ws), nws, st)
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            t[mode] = e0.elapsed_time(e1) / args.iters * 1e3
        outs[name] = (dw.clone(), db.clone())
        res[name] = {"partial_us": round(t["partial"], 1), "direct_us": round(t["direct"], 1), "ws_MB": round(nws / 1e6, 1),
                     "tflops_partial": round(flops / t["partial"] / 1e6), "frac_of_2500": round(flops / t["partial"] / 1e6 / 2500, 3)}
    ref = outs["tap"]
    for name in outs:
        if name != "tap":
            res[name]["max_rel_dw_vs_tap"] = float((outs[name][0] - ref[0]).abs().max() / ref[0].abs().max())
            res[name]["max_rel_db_vs_tap"] = float((outs[name][1] - ref[1]).abs().max() / (ref[1].abs().max() + 1))
    print(json.dumps({"layer": args.layer, "batch": B, "gflop": round(flops / 1e9, 2), "variants": res}))


if __name__ == "__main__":
    main()
