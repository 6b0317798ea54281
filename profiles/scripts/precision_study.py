#!/usr/bin/env python3
"""CPU study (round 4): which operand precision does each analysis layer need?

Emulates the matrix-core arithmetic of candidate schemes on the CPU oracle (operands rounded to the storage type, products and sums in
fp32 -- a product of two bf16 / fp16 values is exact in fp32, so only the summation order differs from an MFMA) for the passes whose
rounded output the reference transmits (encoder1(x1), encoder2(x1_warp, x2), both hyper-analyses), everything else in fp32, and counts
flipped latents / bit deltas against the all-fp32 oracle on the bench workload's pairs.

    python profiles/scripts/precision_study.py [--size 256] [--pairs 2] [--schemes a,b,...]

Per-layer modes: f32 | bf16 | bf16x3 | f16 | f16x2 (x as fp16 hi|lo, w single fp16) | f16w2 (x single, w hi|lo) | f16x3
One JSON line per scheme.  Test / study infrastructure only (imports oracle/)."""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hesic_oracle as O  # noqa: E402
from hesic_amd import synthetic  # noqa: E402


# serpentine walk over the 5x5 taps: consecutive taps are spatial neighbours
NS_ORDER = [r * 5 + (c if r % 2 == 0 else 4 - c) for r in range(5) for c in range(5)]


def _r(t, dt):
    return t.to(dt).float()


def split(t, dt):
    hi = _r(t, dt)
    return hi, _r(t - hi, dt)


def conv_mode(x, w, b, stride, mode):
    pad = w.shape[-1] // 2
    c = lambda a, ww: F.conv2d(a, ww, None, stride=stride, padding=pad)
    if mode == "f32":
        y = c(x, w)
    elif mode in ("bf16", "f16"):
        dt = torch.bfloat16 if mode == "bf16" else torch.float16
        y = c(_r(x, dt), _r(w, dt))
    elif mode in ("bf16x3", "f16x3"):
        dt = torch.bfloat16 if mode == "bf16x3" else torch.float16
        xh, xl = split(x, dt)
        wh, wl = split(w, dt)
        y = c(xh, wh) + c(xl, wh) + c(xh, wl)
    elif mode in ("f16ns", "f16x2ns"):       # weights rounded with error feedback over the taps of each (cout, cin): sum of the tap errors ~ 0
        wf = w.reshape(w.shape[0], w.shape[1], -1)
        out = torch.empty_like(wf)
        e = torch.zeros_like(wf[..., 0])
        order = NS_ORDER if wf.shape[-1] == 25 else range(wf.shape[-1])
        for t in order:
            tgt = wf[..., t] + e
            q = _r(tgt, torch.float16)
            e = tgt - q
            out[..., t] = q
        wq = out.reshape(w.shape)
        if mode == "f16ns":
            y = c(_r(x, torch.float16), wq)
        else:
            xh, xl = split(x, torch.float16)
            y = c(xh, wq) + c(xl, wq)
    elif mode == "f16x2":
        xh, xl = split(x, torch.float16)
        wh = _r(w, torch.float16)
        y = c(xh, wh) + c(xl, wh)
    elif mode == "f16w2":
        xh = _r(x, torch.float16)
        wh, wl = split(w, torch.float16)
        y = c(xh, wh) + c(xh, wl)
    else:
        raise ValueError(mode)
    return y if b is None else y + b.reshape(1, -1, 1, 1)


def gdn_mode(x, beta, gamma, mode):
    C = x.shape[1]
    bb = O.nonneg(beta, 1e-6)
    g = O.nonneg(gamma).reshape(C, C, 1, 1)
    sq = x * x
    norm = conv_mode(sq, g, None, 1, mode) + bb.reshape(1, -1, 1, 1)
    return x * torch.rsqrt(norm)


class Scheme:
    """layer-mode table: conv1..4, gdn1..3, hyper (three convs)"""

    def __init__(self, name, conv, gdn, hyper):
        self.name, self.conv, self.gdn, self.hyper = name, conv, gdn, hyper


def g_a(P, pre, x, s):
    stats = {}
    for i in (1, 2, 3):
        x = conv_mode(x, P[f"{pre}g_a_conv{i}.weight"], P[f"{pre}g_a_conv{i}.bias"], 2, s.conv[i - 1])
        stats[f"conv{i}_absmax"] = float(x.abs().max())
        x = gdn_mode(x, P[f"{pre}g_a_gdn{i}.beta"], P[f"{pre}g_a_gdn{i}.gamma"], s.gdn[i - 1])
        stats[f"gdn{i}_absmax"] = float(x.abs().max())
    return conv_mode(x, P[pre + "g_a_conv4.weight"], P[pre + "g_a_conv4.bias"], 2, s.conv[3]), stats


def hyper(P, pre, y, s):
    t = F.relu(conv_mode(torch.abs(y), P[pre + "encode_hyper.0.weight"], P[pre + "encode_hyper.0.bias"], 1, s.hyper[0]))
    t = F.relu(conv_mode(t, P[pre + "encode_hyper.2.weight"], P[pre + "encode_hyper.2.bias"], 2, s.hyper[1]))
    return conv_mode(t, P[pre + "encode_hyper.4.weight"], P[pre + "encode_hyper.4.bias"], 2, s.hyper[2])


def forward(P, x1, x2, Hm, s, K=5, M=192):
    size = x1.shape[-2:]
    y1, st = g_a(P, "encoder1.", x1, s)
    z1 = hyper(P, "_h_a1.", y1, s)
    z1_hat, z1_lik = O.eb_forward(P, "entropy_bottleneck1.", z1)
    s1, m1, w1 = O.gmm_hyper_y1(P, z1_hat, K, M)
    y1_hat, y1_lik = O.gmm_forward(y1, s1, m1, w1, K)
    x1_hat = O.g_s(P, "decoder1.", y1_hat)
    x1_warp = O.warp_perspective(x1, Hm, size, True)
    t = O._cv(P, "encoder2.pre_conv", torch.cat((x1_warp, x2), 1), stride=1)
    t = O._gdn(P, "encoder2.pre_gdn", t)
    y2, _ = g_a(P, "encoder2.", t, s)
    x1_hat_warp = O.warp_perspective(x1_hat, Hm, size, True)
    y1_w = O.g_a(P, "encoder1.", x1_hat_warp)
    y1_hat_w = torch.round(y1_w)
    z2 = hyper(P, "_h_a2.", y2, s)
    z2_hat, z2_lik = O.eb_forward(P, "entropy_bottleneck2.", z2)
    s2, m2, w2 = O.gmm_hyper_y2(P, z2_hat, y1_hat_w, K, M)
    y2_hat, y2_lik = O.gmm_forward(y2, s2, m2, w2, K)
    x2_hat = O.decoder2(P, y2_hat, x1_hat_warp)
    return {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat, "z1_hat": z1_hat, "z2_hat": z2_hat,
            "y1": y1, "y2": y2,
            "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}, st


SCHEMES = {
    "f32": Scheme("f32", ["f32"] * 4, ["f32"] * 3, ["f32"] * 3),
    "bf16": Scheme("bf16", ["bf16"] * 4, ["bf16"] * 3, ["bf16"] * 3),
    "bf16x3": Scheme("bf16x3", ["bf16x3"] * 4, ["bf16x3"] * 3, ["bf16x3"] * 3),
    "f16": Scheme("f16", ["f16"] * 4, ["f16"] * 3, ["f16"] * 3),
    "f16x2": Scheme("f16x2", ["f16x2"] * 4, ["f16x2"] * 3, ["f16x2"] * 3),
    "f16w2": Scheme("f16w2", ["f16w2"] * 4, ["f16w2"] * 3, ["f16w2"] * 3),
    "f16x3": Scheme("f16x3", ["f16x3"] * 4, ["f16x3"] * 3, ["f16x3"] * 3),
    # conv2 (70 % of g_a's MACs) cheap, the rest exact
    "c2_f16": Scheme("c2_f16", ["bf16x3", "f16", "bf16x3", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c2_f16ns": Scheme("c2_f16ns", ["bf16x3", "f16ns", "bf16x3", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c2_f16x2ns": Scheme("c2_f16x2ns", ["bf16x3", "f16x2ns", "bf16x3", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c23_f16x2ns": Scheme("c23_f16x2ns", ["bf16x3", "f16x2ns", "f16x2ns", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c234_f16x2ns": Scheme("c234_f16x2ns", ["bf16x3", "f16x2ns", "f16x2ns", "f16x2ns"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c1234_f16x2ns": Scheme("c1234_f16x2ns", ["f16x2ns", "f16x2ns", "f16x2ns", "f16x2ns"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c23_f16ns": Scheme("c23_f16ns", ["bf16x3", "f16ns", "f16ns", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c2_f16x2": Scheme("c2_f16x2", ["bf16x3", "f16x2", "bf16x3", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c2_f16w2": Scheme("c2_f16w2", ["bf16x3", "f16w2", "bf16x3", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c23_f16": Scheme("c23_f16", ["bf16x3", "f16", "f16", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c23_f16x2": Scheme("c23_f16x2", ["bf16x3", "f16x2", "f16x2", "bf16x3"], ["bf16x3", "bf16x3", "bf16x3"], ["bf16x3"] * 3),
    "c2_f16_g2_f16": Scheme("c2_f16_g2_f16", ["bf16x3", "f16", "bf16x3", "bf16x3"], ["bf16x3", "f16", "bf16x3"], ["bf16x3"] * 3),
    "c234_f16x2": Scheme("c234_f16x2", ["bf16x3", "f16x2", "f16x2", "f16x2"], ["bf16x3", "f16x2", "f16x2"], ["bf16x3"] * 3),
    "all_f16x2_hyp3": Scheme("all_f16x2_hyp3", ["f16x2"] * 4, ["f16x2"] * 3, ["bf16x3"] * 3),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--schemes", default=",".join(k for k in SCHEMES if k != "f32"))
    ap.add_argument("--salt", type=int, default=0)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    from hesic_amd import models
    net = models.HSIC()
    synthetic.fill_state_dict_(net.state_dict(), args.salt)
    P = {k: v.clone() for k, v in net.state_dict().items()}
    refs = []
    with torch.no_grad():
        for p in range(args.pairs):
            x1, x2, Hm = synthetic.stereo_batch(p, 1, args.size, args.size)
            o, st = forward(P, x1, x2, Hm, SCHEMES["f32"])
            refs.append((x1, x2, Hm, o, O.metrics(o, x1, x2)))
        print(json.dumps({"ranges": st}), flush=True)
        for name in args.schemes.split(","):
            s = SCHEMES[name]
            fl1 = fl2 = n = 0
            dbpp, dpsnr, rel_y = [], [], []
            for x1, x2, Hm, o, m in refs:
                a, _ = forward(P, x1, x2, Hm, s)
                ma = O.metrics(a, x1, x2)
                fl1 += int((a["y1_hat"] != o["y1_hat"]).sum())
                fl2 += int((a["y2_hat"] != o["y2_hat"]).sum())
                n += o["y1_hat"].numel()
                dbpp.append(ma["bpp"] - m["bpp"])
                dpsnr.append(ma["psnr"] - m["psnr"])
                rel_y.append(float(((a["y1"] - o["y1"]) ** 2).mean().sqrt() / (o["y1"] ** 2).mean().sqrt()))
            print(json.dumps({"scheme": name, "size": args.size, "pairs": args.pairs,
                              "flips_y1": round(fl1 / n, 7), "flips_y2": round(fl2 / n, 7),
                              "dbpp": [round(v, 6) for v in dbpp], "dpsnr_db": [round(v, 6) for v in dpsnr],
                              "y1_rms_rel": [float(f"{v:.3e}") for v in rel_y]}), flush=True)


if __name__ == "__main__":
    main()
