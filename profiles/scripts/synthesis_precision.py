#!/usr/bin/env python3
"""Where does the default mode's residual PSNR deviation at a trained point come from?  With pair analysis the latents are the reference's
(<= 1e-5 flips), so what is left (1e-4 ... 9e-4 dB over the round's runs) is the 16-bit SYNTHESIS side.  On one trained weight set:
  A  fp32 mode (the reference's arithmetic)                          -> PSNR_A
  B  fp32 mode with the synthesis / hyper-synthesis WEIGHTS rounded to float16 (and back)  -> weight-rounding share
  C  float16 mode (x3)                                               -> all of it
  D  float16 mode with the decoders' weights ALREADY float16-representable (so packing rounds nothing) -> activation-storage share
    python profiles/scripts/synthesis_precision.py [--steps 2000]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--pairs", type=int, default=4)
    args = ap.parse_args()
    import hesic_amd
    from hesic_amd import functional as Fn, models, synthetic
    from hesic_amd.train import GraphedTrainer
    dev = "cuda"
    hesic_amd.set_compute_dtype(torch.bfloat16)
    net = models.HSIC()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.to(dev)
    torch.manual_seed(5)
    tr = GraphedTrainer(net, lr=1e-4, aux_lr=1e-3, lmbda=0.02)
    pool = [tuple(t.to(dev) for t in synthetic.smooth_stereo_batch(100 + 8 * i, 8, 256, 256)) for i in range(8)]
    for st in range(args.steps):
        tr.step(*pool[st % len(pool)])
    torch.cuda.synchronize()
    del tr
    net.eval()
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    xs = [tuple(t.to(dev) for t in synthetic.smooth_stereo_batch(j, 1, 512, 512)) for j in range(args.pairs)]

    def run(dt, round_keys=None):
        sd = {k: v.clone() for k, v in sd0.items()}
        if round_keys is not None:
            for k in sd:
                if sd[k].dtype == torch.float32 and k.endswith("weight") and any(p in k for p in round_keys) and sd[k].dim() == 4:
                    sd[k] = sd[k].half().float()
        net.load_state_dict(sd)
        hesic_amd.set_compute_dtype(dt)
        Fn.invalidate_weight_cache()
        ps, bs = [], []
        for a, b, h in xs:
            with torch.no_grad():
                o = net(a, b, h)
                m = models.metrics_from(models.rate_distortion(o, a, b))
            ps.append(m["psnr"]); bs.append(m["bpp"])
        return sum(ps) / len(ps), sum(bs) / len(bs), ps

    dec = (".g_s_conv",)          # the wide synthesis layers (the image-side 6 -> 3 stage keeps fp32 weights in every mode)
    hs = ("_h_s1.", "_h_s2.")
    A = run(torch.float32)
    B = run(torch.float32, dec)
    per_layer = {}
    for i in (1, 2, 3, 4):
        r = run(torch.float32, (f".g_s_conv{i}.",))
        per_layer[f"g_s_conv{i}"] = r[0] - A[0]
    B2 = run(torch.float32, dec + hs)
    keep = Fn.SHAPED_WEIGHTS
    Fn.SHAPED_WEIGHTS = False
    Cp = run(torch.float16)                      # plain rounding of every single-operand weight
    Fn.SHAPED_WEIGHTS = keep
    Cc = run(torch.float16)
    D = run(torch.float16, dec)
    out = {"psnr_fp32": A[0], "bpp_fp32": A[1], "fp32_with_ONE_synthesis_layer_f16_rounded": per_layer,
           "B_fp32_with_f16_rounded_decoder_weights": {"dpsnr": B[0] - A[0], "per_pair": [p - q for p, q in zip(B[2], A[2])]},
           "B2_plus_hyper_synthesis_weights": {"dpsnr": B2[0] - A[0], "dbpp": B2[1] - A[1]},
           "C_f16_mode_plain_rounding": {"dpsnr": Cp[0] - A[0], "dbpp": Cp[1] - A[1], "per_pair": [p - q for p, q in zip(Cp[2], A[2])]},
           "C_f16_mode": {"dpsnr": Cc[0] - A[0], "dbpp": Cc[1] - A[1], "per_pair": [p - q for p, q in zip(Cc[2], A[2])]},
           "D_f16_mode_decoder_weights_already_f16": {"dpsnr_vs_B": D[0] - B[0], "per_pair": [p - q for p, q in zip(D[2], B[2])]}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
