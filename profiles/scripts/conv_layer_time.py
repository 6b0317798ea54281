#!/usr/bin/env python3
"""One layer of the analysis / synthesis stacks on the implicit-GEMM kernel, timed with HIP events.

    python profiles/scripts/conv_layer_time.py [--layer conv2|deconv3|deconv_plain|conv3|plain] [--batch 8] [--size 256]
Environment A/B switches of csrc/conv_igemm.hip apply (HESIC_IGEMM_WS; the HESIC_IGEMM_DBG ablation switches of rounds 1-2 were removed in round 3: their run-time branches split the K loop into basic blocks)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="conv2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256, help="input height = width of the layer")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dump", default="", help="save the layer output here (to compare A/B variants bit for bit)")
    ap.add_argument("--hilo", action="store_true", help="the bf16x3 form of conv2 / conv3 / plain: hi/lo operand pairs, fused hi/lo GDN")
    ap.add_argument("--cin", type=int, default=128, help="--layer plain / deconv_plain: input channels")
    ap.add_argument("--cout", type=int, default=128, help="--layer plain / deconv_plain: output channels")
    ap.add_argument("--stride", type=int, default=2, help="--layer plain: stride (kernel 5x5)")
    ap.add_argument("--dtype", choices=["bf16", "f16"], default="bf16", help="16-bit storage format = which library is bound")
    ap.add_argument("--graph", action="store_true", help="replay the launches from a HIP graph (no host launch cost in the figure)")
    args = ap.parse_args()
    import hesic_amd
    from compressai.layers import GDN
    from compressai.models.utils import conv, deconv
    h16 = torch.float16 if args.dtype == "f16" else torch.bfloat16
    hesic_amd.set_compute_dtype(h16)
    torch.manual_seed(0)
    B, S = args.batch, args.size
    ci, co = (args.cin, args.cout) if args.layer in ("plain", "deconv_plain") else (128, 128)
    x = (torch.randn(B, ci, S, S, device="cuda") * 0.5).to(h16).contiguous(memory_format=torch.channels_last)
    if args.layer in ("conv2", "conv3", "plain"):
        st = args.stride if args.layer == "plain" else 2
        layer, g = conv(ci, co, stride=st).cuda(), GDN(128).cuda()
        flops = 2.0 * B * (S // st) ** 2 * ci * co * 25 + (0 if args.layer == "plain" else 2.0 * B * (S // 2) ** 2 * 128 * 128)
    else:
        layer, g = deconv(ci, co).cuda(), GDN(128, inverse=True).cuda()
        flops = 2.0 * B * S * S * ci * co * 25 + (0 if args.layer == "deconv_plain" else 2.0 * B * (2 * S) ** 2 * 128 * 128)
    f = (lambda: layer.run(x)) if args.layer in ("plain", "deconv_plain") else (lambda: layer.run_gdn(x, g))
    if args.hilo:
        xf = torch.randn(B, 128, S, S, device="cuda") * 0.5
        hi = xf.to(torch.bfloat16)
        xh = torch.cat((hi, (xf - hi.float()).to(torch.bfloat16)), 1).contiguous(memory_format=torch.channels_last)
        f = (lambda: layer.run_hilo(xh, out="hilo")) if args.layer == "plain" else (lambda: layer.run_hilo(xh, gdn=g))
    with torch.no_grad():
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                f()
            torch.cuda.current_stream().wait_stream(s)
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                for _ in range(args.iters):
                    f()
            g_.replay()
            torch.cuda.synchronize()
            e0.record()
            g_.replay()
            e1.record()
        else:
            e0.record()
            for _ in range(args.iters):
                f()
            e1.record()
        torch.cuda.synchronize()
    if args.dump:
        with torch.no_grad():
            torch.save(f().float().cpu(), args.dump)
    us = e0.elapsed_time(e1) / args.iters * 1e3
    print(f"{args.layer} {ci}->{co} B={B} in {S}x{S} BM={os.environ.get('HESIC_IGEMM_BM', 'auto')}: {us:.1f} us  {flops / us / 1e6:.0f} TFLOP/s   PHASE4={os.environ.get('HESIC_IGEMM_PHASE4', '1')} WS={os.environ.get('HESIC_IGEMM_WS', '0')} hilo={int(args.hilo)} BM256_HILO={os.environ.get('HESIC_IGEMM_BM256_HILO', '0')}")


if __name__ == "__main__":
    main()
