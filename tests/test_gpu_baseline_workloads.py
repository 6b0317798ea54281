"""The EXACT BASELINE.json workloads on the GPU against fixtures recorded from the reference itself
(tests/golden/make_golden.py models3: the reference's ``HSIC.forward`` / training step run in the build container):

  C2  HESIC, 8 pairs of 512 x 512, bf16 maps with the bf16x3 analysis path (the benchmark's headline configuration)
  C3  one R-D training step on 8 pairs of 512 x 512 (per GPU; the 8-GPU job averages eight of these gradients)
  C4  HESIC+, 4 pairs of 512 x 512
  C5  the four lambda-models on an 860 x 1080 pair zero-padded to 896 x 1088, metrics over the original pixels

Bars (written next to each assert): the integer latents may differ from the reference's in at most 1e-3 of the positions
(measured: <= 2e-5), per-pair bits within 2e-3 relative / squared error within 1e-3 relative, PSNR within 1e-3 dB."""
import math

import numpy as np
import pytest
import torch

from conftest import T, load_golden
from hesic_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(kind, dtype, salt=0):
    import hesic_amd
    from hesic_amd import models
    hesic_amd.set_compute_dtype(dtype)
    net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict(), salt=salt)
    return net.to(DEV).eval()


@pytest.fixture(autouse=True)
def _reset():
    yield
    import hesic_amd
    from hesic_amd import functional as Fn
    hesic_amd.set_compute_dtype(torch.float32)
    Fn.set_analysis_precision("auto")


def per_pair(out, x1, x2):
    B = x1.shape[0]
    bits = {k: (torch.log2(v.double()).reshape(B, -1).sum(1) * -1).cpu().numpy() for k, v in out["likelihoods"].items()}
    h, w = x1.shape[-2:]
    sse = [((out[k][..., :h, :w].double() - x.double()) ** 2).reshape(B, -1).sum(1).cpu().numpy() for k, x in (("x1_hat", x1), ("x2_hat", x2))]
    return bits, sse


def check_against(g, out, x1, x2, flips_max, bits_rel, sse_rel, psnr_db, view_psnr_db=None):
    """Per-pair comparison with a golden recorded from the reference; every measured maximum is in the assertion message.
    ``psnr_db`` bounds the PSNR the reference reports (mean of the two views, test3real.py:110-122), ``view_psnr_db`` each view."""
    B, _, h, w = x1.shape
    bits, sse = per_pair(out, x1, x2)
    meas = {}
    for k in ("y1_hat", "y2_hat"):
        flips = (out[k].float().cpu().to(torch.int16) != T(g[k]).to(torch.int16)).float().reshape(B, -1).mean(1)
        meas["flips_" + k] = float(flips.max())
    for k in ("y1", "y2", "z1", "z2"):
        meas["bits_rel_" + k] = float(np.abs(bits[k] / g["bits_" + k] - 1).max())
    dps = []
    for i, k in enumerate(("sse1", "sse2")):
        meas["sse_rel_" + k] = float(np.abs(sse[i] / g[k] - 1).max())
        dps.append(10 * np.log10(g[k] / sse[i]))
        meas["dpsnr_db_view%d" % (i + 1)] = float(np.abs(dps[-1]).max())
    meas["dpsnr_db"] = float(np.abs((dps[0] + dps[1]) / 2).max())
    tot = sum(bits[k] for k in bits) / (h * w * 2)
    ref = sum(g["bits_" + k] for k in ("y1", "y2", "z1", "z2")) / (h * w * 2)
    meas["dbpp_abs"] = float(np.abs(tot - ref).max())
    meas["dbpp_rel"] = float(np.abs(tot / ref - 1).max())
    meas["dbpp_abs_set_mean"] = float(abs((tot - ref).mean()))          # the set average is what the reference reports (test3real.py:110-122)
    ok = (max(meas["flips_y1_hat"], meas["flips_y2_hat"]) <= flips_max and max(meas["bits_rel_" + k] for k in ("y1", "y2", "z1", "z2")) <= bits_rel
          and max(meas["sse_rel_sse1"], meas["sse_rel_sse2"]) <= sse_rel and meas["dpsnr_db"] < psnr_db
          and max(meas["dpsnr_db_view1"], meas["dpsnr_db_view2"]) < (view_psnr_db or psnr_db))
    assert ok, meas
    print("measured:", {k: float("%.3g" % v) for k, v in meas.items()})
    return meas


# Bars per (16-bit format, analysis mode) = measured on MI355X + margin (round 4; the measured maxima are printed by every run):
#   flips: share of the integer latents that differ from the reference's, worst pair;  bits: per-pair, per-latent-group bits, relative;
#   sse: per-view squared error, relative;  psnr / view: dB of the two-view mean the reference reports / of each view;
#   dbpp: |mean over the pairs of (bpp - bpp_ref)|, ABSOLUTE (north_star: bpp within 1e-3)
MODE_BARS = {
    ("f16", "x3c2"): dict(flips_max=1e-3, bits_rel=1.5e-3, sse_rel=1e-4, psnr_db=4e-4, view_psnr_db=6e-4, dbpp=1e-3),     # round 4's default; an explicit fast mode now
    ("f16", "x3"): dict(flips_max=1e-4, bits_rel=1e-3, sse_rel=5e-5, psnr_db=1e-4, view_psnr_db=2e-4, dbpp=1e-3),          # the benchmark's default (round 5)
    ("bf16", "x3"): dict(flips_max=2e-4, bits_rel=1e-3, sse_rel=1e-3, psnr_db=1e-3, view_psnr_db=2e-3, dbpp=1e-3),        # round 3's headline mode
}


@pytest.mark.parametrize("fmt,analysis", list(MODE_BARS), ids=["-".join(k) for k in MODE_BARS])
@pytest.mark.parametrize("kind,batch", [("hsic", 8), ("joint", 4)], ids=["C2-hesic-b8", "C4-hesicplus-b4"])
def test_16bit_512_batch_matches_the_reference_pair_by_pair(kind, batch, fmt, analysis):
    """BASELINE configs C2 / C4 in the benchmark's own modes (16-bit maps, fp32 accumulation, pair arithmetic on the analysis side): every
    pair of the batch against the reference's fp32 run, and the set-average bpp against the ABSOLUTE 1e-3 bar."""
    from hesic_amd import functional as Fn
    g = load_golden(f"{kind}_512_b{batch}.npz")
    net = build(kind, {"f16": torch.float16, "bf16": torch.bfloat16}[fmt])
    assert Fn.analysis_precision() == "x3"           # "auto": the pair mode for both 16-bit formats (round 5)
    Fn.set_analysis_precision(analysis)
    bars = dict(MODE_BARS[(fmt, analysis)])
    dbpp_bar = bars.pop("dbpp")
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, batch, 512, 512))
    with torch.no_grad():
        out = net(x1, x2, Hm)
        meas = check_against(g, out, x1, x2, **bars)
        assert meas["dbpp_abs_set_mean"] < dbpp_bar, meas
        assert meas["dbpp_rel"] < 1e-3, meas
        if (fmt, analysis, kind) == ("f16", "x3", "hsic"):
            # the default mode holds north_star's ABSOLUTE 1e-3 bpp bar for EVERY pair of C2, not only for the set average the reference
            # reports (round 4's x3c2 default missed it on three of eight pairs: +1.2e-3, -0.8e-3, -1.1e-3; measured now: 5.3e-4 worst pair).
            # C4's random-weight bpp is 12.2: 1e-3 absolute would be 8e-5 RELATIVE there -- its per-pair bar is the relative one above
            # (measured 1.4e-4 = 1.7e-3 absolute worst pair, 3e-4 set average)
            assert meas["dbpp_abs"] < 1e-3, meas
        # pairs are independent: the last pair alone reproduces its slice of the batch bit for bit
        one = net(x1[-1:], x2[-1:], Hm[-1:])
    for k in ("y1_hat", "y2_hat", "x1_hat", "x2_hat"):
        assert torch.equal(out[k][-1:], one[k]), k


@pytest.mark.parametrize("kind,batch", [("hsic", 8), ("joint", 4)], ids=["C2-hesic-b8", "C4-hesicplus-b4"])
def test_single_bf16_analysis_512_batch_stays_inside_its_wider_bars(kind, batch):
    """The same workloads with single-bf16 analysis operands (``set_analysis_precision("bf16")``, the round-2 path, 1.4x faster):
    ~1 % of the latents land on the other side of .5 (bar 2 %), bits within 5e-3, PSNR within 2e-3 dB."""
    from hesic_amd import functional as Fn
    g = load_golden(f"{kind}_512_b{batch}.npz")
    net = build(kind, torch.bfloat16)
    Fn.set_analysis_precision("x1")
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, batch, 512, 512))
    with torch.no_grad():
        out = net(x1, x2, Hm)
    check_against(g, out, x1, x2, flips_max=2e-2, bits_rel=5e-3, sse_rel=1e-3, psnr_db=2e-3, view_psnr_db=3e-3)


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_fp32_512_batch_matches_the_reference(kind):
    """fp32 storage (exact-fp32 MFMA) on two pairs of the 512 x 512 workloads: latents differ only at rounding boundaries."""
    batch = 8 if kind == "hsic" else 4
    g = load_golden(f"{kind}_512_b{batch}.npz")
    g2 = {k: (v[:2] if getattr(v, "ndim", 0) >= 1 and v.shape[0] == batch else v) for k, v in g.items()}
    net = build(kind, torch.float32)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 2, 512, 512))
    with torch.no_grad():
        out = net(x1, x2, Hm)
    check_against(g2, out, x1, x2, flips_max=2e-4, bits_rel=1e-3, sse_rel=1e-3, psnr_db=1e-3)


# C5 bars per mode: (bits of a lambda-model relative, PSNR dB, digest of |y_hat| relative, flipped latents of model 0)
C5_BARS = {("f16", "x3"): (1e-3, 1e-3, 1e-4, 1e-4), ("f16", "x3c2"): (1.5e-3, 1e-3, 1e-3, 1e-3), ("bf16", "x3"): (1e-3, 1e-3, 2e-4, 1e-3)}


@pytest.mark.parametrize("fmt,analysis", list(C5_BARS), ids=["-".join(k) for k in C5_BARS])
@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_c5_lambda_sweep_accumulators_match_the_reference(kind, fmt, analysis):
    """BASELINE config C5 through ``evaluate.LambdaSweep`` (what ``bench.py --sweep`` times): four lambda-models on an 860 x 1080 pair
    padded to 896 x 1088, bits and squared error accumulated on the device, bpp / PSNR over the original pixels -- against the
    reference's run of the same four weight sets on the same padded pair.  Every mode ``bench.py --sweep`` can time: float16 maps with
    the pair analysis (its default), float16 "x3c2" (round 4's timed mode, which had no golden check at this size), bfloat16 pairs."""
    import hesic_amd
    from hesic_amd import functional as Fn, models
    from hesic_amd.evaluate import SWEEP_LAMBDAS, LambdaSweep
    g = load_golden(f"{kind}_c5.npz")
    hesic_amd.set_compute_dtype({"f16": torch.float16, "bf16": torch.bfloat16}[fmt])
    Fn.set_analysis_precision(analysis)
    bits_rel, psnr_db, digest_rel, flips_max = C5_BARS[(fmt, analysis)]
    meas = {}
    sweep = LambdaSweep(kind, torch.device(DEV))
    x1, x2, Hm = synthetic.stereo_batch(0, 1, 860, 1080)
    x1p, x2p = models.pad_to_multiple(x1), models.pad_to_multiple(x2)
    assert x1p.shape[-2:] == (896, 1088)
    x1, x2, x1p, x2p, Hm = (t.to(DEV) for t in (x1, x2, x1p, x2p, Hm))
    outs = [sweep.step(m, x1, x2, x1p, x2p, Hm)[0] for m in range(4)]
    per = sweep.summary(860, 1080)
    for m, lam in enumerate(SWEEP_LAMBDAS):
        ref_bits = sum(float(g[f"bits_{k}_{m}"].sum()) for k in ("y1", "y2", "z1", "z2"))
        s1, s2 = float(g[f"sse1_{m}"].sum()), float(g[f"sse2_{m}"].sum())
        npx = 860 * 1080
        ref_psnr = (10 * math.log10(npx * 3 / s1) + 10 * math.log10(npx * 3 / s2)) / 2
        got = per[str(lam)]
        assert got["pairs"] == 1
        meas[f"bits_rel_{m}"] = abs(got["bits"] / ref_bits - 1)
        meas[f"dpsnr_db_{m}"] = abs(got["psnr"] - ref_psnr)
        assert got["bits"] == pytest.approx(ref_bits, rel=bits_rel), (lam, got["bits"], ref_bits)
        assert abs(got["psnr"] - ref_psnr) < psnr_db, (lam, got["psnr"], ref_psnr)
        for k in ("y1_hat", "y2_hat"):          # digest of every model's latents; the full tensors of model 0
            meas[f"digest_rel_{k}_{m}"] = abs(float(outs[m][k].double().abs().sum()) / float(g[f"{k}_abs_sum_{m}"]) - 1)
            assert meas[f"digest_rel_{k}_{m}"] <= digest_rel, (lam, k, meas)
    for k in ("y1_hat", "y2_hat"):
        assert outs[0][k].shape == (1, 192, 56, 68)
        meas["flips_" + k] = float((outs[0][k].float().cpu().to(torch.int16) != T(g[k]).to(torch.int16)).float().mean())
        assert meas["flips_" + k] <= flips_max, (k, meas)
    print("measured:", {k: float("%.3g" % v) for k, v in meas.items()})


def _train_noise(kind, order, B):
    noise = {}
    for k in order:
        if k[0] == "z":      # reference layout (C, 1, H*W*B) with B fastest -> (B, C, H, W)
            nz = synthetic._uniform(f"noise.{kind}.t512.0.{k}", (128, 1, B * 64), -0.5, 0.5).reshape(128, 8, 8, B).permute(3, 0, 1, 2).contiguous()
        else:
            nz = synthetic._uniform(f"noise.{kind}.t512.0.{k}", (B, 192, 32, 32), -0.5, 0.5)
        noise[k] = nz.to(DEV)
    return noise


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_c3_training_step_512_b8_matches_the_reference(kind, dtype):
    """BASELINE config C3, one rank's share: a training step on 8 pairs of 512 x 512 with the reference's noise draws -- loss terms
    and every parameter's gradient norm of the first backward against the reference's own step, then ``Trainer.step``."""
    import hesic_amd
    from hesic_amd import functional as Fn
    from hesic_amd.train import Trainer
    g = load_golden(f"{kind}_train512.npz")
    net = build(kind, dtype)
    B = 8
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, B, 512, 512))
    noise = _train_noise(kind, [str(s) for s in g["noise_order"]], B)
    rel_loss, rel_gn = (2e-3, 1e-2) if dtype == torch.float32 else (5e-3, 5e-2)
    net.train()
    out = net(x1, x2, Hm, noise=noise)
    crit = Fn.rd_loss(out, x1, x2, 0.0067)
    crit["loss"].backward()
    assert float(crit["loss"]) == pytest.approx(float(g["loss"]), rel=rel_loss)
    assert float(crit["bpp_loss"]) == pytest.approx(float(g["bpp"]), rel=rel_loss)
    assert float(crit["mse_loss"]) == pytest.approx(float(g["mse"]), rel=rel_loss)
    bad = []
    for name, p in net.named_parameters():
        live = "gn_live_" + name
        ref = float(g[live] if live in g else g["gn_" + name])
        got = float(p.grad.double().norm()) if p.grad is not None else 0.0
        if abs(got - ref) > rel_gn * max(ref, 1e-6) + 1e-7:
            bad.append((name, got, ref))
    assert not bad, bad[:8]
    del out, crit
    net.zero_grad(set_to_none=True)
    tr = Trainer(net, lr=1e-4, aux_lr=1e-3, lmbda=0.0067)
    c = tr.step(x1, x2, Hm, noise=noise)
    assert float(c["loss"]) == pytest.approx(float(g["loss"]), rel=rel_loss)
    assert float(c["aux_loss"]) == pytest.approx(float(g["aux"]), rel=5e-3)


def test_trained_operating_point_parity_at_512():
    """Parity where the reference's numbers live (trained models, Readme.md:33-46), not only at the random-weight point: 600
    graph-replayed training steps (bf16, B=8, 256 x 256 synthetic pairs, the deterministic init) take the model to a low-rate operating
    point; then the 16-bit inference modes are compared with the fp32 CPU oracle run on THOSE weights on a 512 x 512 pair.
    Bars (north_star): |dbpp| < 1e-3 absolute, |dPSNR| < 1e-3 dB, <= 1e-3 of the integer latents differ.  The float16 modes are asserted
    (measured: see the printed record); bfloat16 maps with pair analysis (round 3's mode) are reported next to them."""
    import hesic_amd
    from hesic_amd import functional as Fn, models
    from hesic_amd.train import GraphedTrainer
    from oracle import hesic_oracle as O
    hesic_amd.set_compute_dtype(torch.bfloat16)
    net = models.HSIC()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.to(DEV)
    tr = GraphedTrainer(net, lr=1e-4, aux_lr=1e-3, lmbda=0.0067)
    pool = [tuple(t.to(DEV) for t in synthetic.stereo_batch(100 + 8 * i, 8, 256, 256)) for i in range(8)]
    first = last = None
    for st in range(600):
        c = tr.step(*pool[st % len(pool)])
        if st == 0:
            first = float(c["loss"])
    last = float(c["loss"])
    torch.cuda.synchronize()
    assert last < 0.5 * first, (first, last)          # it did train
    del tr
    net.eval()
    net.update(force=True)
    Fn.invalidate_weight_cache()
    P = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}
    x1, x2, Hm = synthetic.stereo_batch(0, 1, 512, 512)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref = O.hsic_forward(P, x1, x2, Hm)
    mr = O.metrics(ref, x1, x2)
    recs = {}
    for name, dt, an in (("f16-x3c2", torch.float16, "x3c2"), ("f16-x3", torch.float16, "auto"), ("bf16-x3", torch.bfloat16, "x3")):
        hesic_amd.set_compute_dtype(dt)
        Fn.set_analysis_precision(an)
        with torch.no_grad():
            out = net(x1.to(DEV), x2.to(DEV), Hm.to(DEV))
            m = models.metrics_from(models.rate_distortion(out, x1.to(DEV), x2.to(DEV)))
        flips = max(float((out[k].float().cpu() != ref[k]).float().mean()) for k in ("y1_hat", "y2_hat"))
        recs[name] = {"dbpp": m["bpp"] - mr["bpp"], "dpsnr_db": m["psnr"] - mr["psnr"], "flips": flips, "mode": Fn.analysis_precision()}
    print("trained point: loss %.2f -> %.2f, bpp %.4f, PSNR %.3f dB (oracle);" % (first, last, mr["bpp"], mr["psnr"]),
          {k: {kk: (float("%.3g" % vv) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in recs.items()})
    assert recs["f16-x3"]["mode"] == "x3" and recs["f16-x3c2"]["mode"] == "x3c2"
    for name in ("f16-x3c2", "f16-x3"):
        r = recs[name]
        assert abs(r["dbpp"]) < 1e-3 and abs(r["dpsnr_db"]) < 1e-3 and r["flips"] <= 1e-3, (name, r)
    r = recs["bf16-x3"]
    assert abs(r["dbpp"]) < 1e-3 and abs(r["dpsnr_db"]) < 2e-3 and r["flips"] <= 1e-3, ("bf16-x3", r)


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_trained_30db_operating_point_parity_at_512(kind):
    """Parity at an operating point like the published ones (Readme.md:33-46: 33 - 37 dB): graph-replayed training steps (bf16, B=8,
    256 x 256 PIECEWISE-SMOOTH synthetic pairs, ``synthetic.smooth_stereo_pair``; lambda 0.02, the reference's lr 1e-4, newtrain1.py:180-185)
    take the deterministic init to >= 30 dB (checked every 500 steps from 1500 on, at most 5000: the loss of this unclipped recipe spikes now
    and then, as the reference's would); then the 16-bit inference modes against the fp32 CPU oracle run on THOSE weights on a SET of four
    512 x 512 smooth pairs (set averages of bpp and PSNR, as the reference's evaluation reports them; flips = the worst pair).
    The default mode (float16 maps, pair analysis) must hold north_star's bars there: |dbpp| < 1e-3, <= 1e-4 flipped latents, and |dPSNR| <
    1e-3 dB + what the counted flips explain (at an MSE of ~6e-4 every flipped latent is visible in the PSNR; see the bar below).  The explicit fast mode "x3c2" (round 4's default) is
    measured next to it (2.6 - 8e-4 flips, |dPSNR| 2 - 6e-3 dB over the round's runs, which is why it stopped being the default; sanity
    bars 2e-3 / 2e-2 dB); bfloat16 pairs likewise (storage noise of the synthesis maps: up to 8e-3 dB)."""
    import hesic_amd
    from hesic_amd import functional as Fn, models
    from hesic_amd.train import GraphedTrainer
    from oracle import hesic_oracle as O
    hesic_amd.set_compute_dtype(torch.bfloat16)
    net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.to(DEV)
    torch.manual_seed(5)
    tr = GraphedTrainer(net, lr=1e-4, aux_lr=1e-3, lmbda=0.02)
    pool = [tuple(t.to(DEV) for t in synthetic.smooth_stereo_batch(100 + 8 * i, 8, 256, 256)) for i in range(8)]
    NP = 4
    x1, x2, Hm = synthetic.smooth_stereo_batch(0, NP, 512, 512)          # a SET of pairs: the reference reports set averages (test3real.py:110-122)
    xd = tuple(t.to(DEV) for t in (x1, x2, Hm))
    steps, psnr_now, best = 0, 0.0, (0.0, None, 0)
    while steps < 6000 and psnr_now < 30.3:
        for st in range(500):
            tr.step(*pool[(steps + st) % len(pool)])
        steps += 500
        if steps >= 1500:
            net.eval()
            Fn.invalidate_weight_cache()
            with torch.no_grad():
                o = net(*xd)
                psnr_now = models.metrics_from(models.rate_distortion(o, xd[0], xd[1]))["psnr"]
            net.train()
            if psnr_now > best[0]:
                best = (psnr_now, {k: v.detach().clone() for k, v in net.state_dict().items()}, steps)
    torch.cuda.synchronize()
    del tr
    if psnr_now < best[0]:                      # a loss spike behind the best point: evaluate the best weights seen
        net.load_state_dict(best[1])
        steps = best[2]
    if best[0] < 30.0:
        # the training itself is pinned elsewhere (two-step traces and the 512^2 step against the reference); this recipe is the reference's,
        # unclipped, and not bit-reproducible (scatter-adds): about one run in ten spikes early and is still recovering at 6000 steps
        pytest.skip("the unclipped recipe did not reach 30 dB in 6000 steps this time (best %.2f dB): no operating point to test" % best[0])
    net.eval()
    net.update(force=True)
    Fn.invalidate_weight_cache()
    P = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    fwd_o = O.hsic_forward if kind == "hsic" else O.hsic_joint_forward
    with torch.no_grad():
        refs = [fwd_o(P, x1[j:j + 1], x2[j:j + 1], Hm[j:j + 1]) for j in range(NP)]
    mrs = [O.metrics(r, x1[j:j + 1], x2[j:j + 1]) for j, r in enumerate(refs)]
    mr = {"bpp": sum(m["bpp"] for m in mrs) / NP, "psnr": sum(m["psnr"] for m in mrs) / NP}
    assert mr["psnr"] >= 30.0, mrs            # the point of this test
    recs = {}
    for name, dt, an in (("f16-x3", torch.float16, "auto"), ("f16-x3c2", torch.float16, "x3c2"), ("bf16-x3", torch.bfloat16, "x3"), ("f32", torch.float32, "auto")):
        hesic_amd.set_compute_dtype(dt)
        Fn.set_analysis_precision(an)
        ms, flips, n_flipped = [], 0.0, 0
        for j in range(NP):
            with torch.no_grad():
                out = net(xd[0][j:j + 1], xd[1][j:j + 1], xd[2][j:j + 1])
                ms.append(models.metrics_from(models.rate_distortion(out, xd[0][j:j + 1], xd[1][j:j + 1])))
            flips = max(flips, max(float((out[k].float().cpu() != refs[j][k]).float().mean()) for k in ("y1_hat", "y2_hat")))
            n_flipped += sum(int((out[k].float().cpu() != refs[j][k]).sum()) for k in ("y1_hat", "y2_hat"))
        recs[name] = {"dbpp": sum(m["bpp"] for m in ms) / NP - mr["bpp"], "dpsnr_db": sum(m["psnr"] for m in ms) / NP - mr["psnr"],
                      "dpsnr_db_worst_pair": max(abs(ms[j]["psnr"] - mrs[j]["psnr"]) for j in range(NP)), "flips": flips,
                      "flipped_latents_in_set": n_flipped, "mode": Fn.analysis_precision() if dt != torch.float32 else "fp32"}
    print("%s trained smooth point after %d steps: bpp %.4f, PSNR %.3f dB (oracle);" % (kind, steps, mr["bpp"], mr["psnr"]),
          {k: {kk: (float("%.3g" % vv) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in recs.items()})
    # PSNR bar at a trained point: at an MSE of ~6e-4 ONE flipped latent (a unit step through the trained synthesis) moves its pair's PSNR by
    # 1 - 1.5e-3 dB -- measured here: 2 flipped latents -> -2.0e-3 dB on the 4-pair set (3.0e-3 on the pair), 4 -> -1.2e-3 -- and a latent at
    # a rounding tie flips under ANY fp32 arithmetic that sums in another order than the reference's (the fp32 mode below: 2 - 10 per set,
    # the default mode 2 - 7 since its pair weights are packed scaled and its GDN squares are scaled per pixel: it sits AT that floor).
    # The analysis precision is therefore held by the flip bar (<= 1e-4 of a map = 19 latents) and the rate bar; the PSNR bar is north_star's
    # 1e-3 dB plus what the counted flips explain (2e-3 dB per flipped latent and pair), i.e. 1e-3 flat when nothing flipped: a synthesis-side
    # error, or a flip worth more than a unit step, still fails.  (A flat 1e-3 held in about two of three trained points of this recipe.)
    psnr_bar = lambda r: 1e-3 + 2e-3 * r["flipped_latents_in_set"] / NP
    r = recs["f16-x3"]
    assert r["mode"] == "x3"
    assert abs(r["dbpp"]) < 1e-3 and abs(r["dpsnr_db"]) < psnr_bar(r) and r["flips"] <= 1e-4 and r["flipped_latents_in_set"] <= 24, ("f16-x3 (default)", r)
    # the non-default modes are MEASURED here (the printed record is the point); their bars only catch a broken path: training is not
    # bit-reproducible, and over the round's runs x3c2 showed 2.6 - 8.1e-4 flips / up to 6.3e-3 dB, bf16 pairs up to 6.6e-4 bpp / 8e-3 dB
    # fp32 storage (exact-fp32 MFMA): what is left there is the summation order alone -- a handful of latents at a rounding tie; each of
    # them moves a pair's PSNR by ~1e-4 dB at this MSE, the floor under every 16-bit figure above
    r = recs["f32"]
    assert abs(r["dbpp"]) < 1e-3 and abs(r["dpsnr_db"]) < psnr_bar(r) and r["flips"] <= 1e-4 and r["flipped_latents_in_set"] <= 24, ("f32", r)
    for name in ("f16-x3c2", "bf16-x3"):
        r = recs[name]
        assert abs(r["dbpp"]) < 2e-3 and abs(r["dpsnr_db"]) < 2e-2 and r["flips"] <= 2e-3, (name, r)


def test_conditioning_latents_of_the_third_analysis_pass_stay_under_their_bar():
    """``round(encoder1(warp(x1_hat)))`` is not transmitted -- it conditions view 2's entropy parameters (newnet1.py:753-757) -- and runs on
    single 16-bit operands in every analysis mode.  Its flips against the oracle had no bar in round 4 (0.88 % in "x3c2", where x1_hat itself
    differs around every flipped latent); pair 0 of the C2 workload: <= 3e-3 in the default mode (measured 1e-3), <= 1e-2 in "x3c2"."""
    import hesic_amd
    from hesic_amd import functional as Fn, geometry
    from oracle import hesic_oracle as O
    net = build("hsic", torch.float16)
    P = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}
    x1, x2, Hm = synthetic.stereo_batch(0, 1, 512, 512)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref = O.hsic_forward(P, x1, x2, Hm)
        want = torch.round(O.g_a(P, "encoder1.", O.warp_perspective(ref["x1_hat"], Hm, x1.shape[-2:], True))).to(torch.int16)
    meas = {}
    for an, bar in (("auto", 3e-3), ("x3c2", 1e-2)):
        Fn.set_analysis_precision(an)
        with torch.no_grad():
            o = net(x1.to(DEV), x2.to(DEV), Hm.to(DEV))
            yw = net.encoder1.latent(geometry.warp_perspective(o["x1_hat"], Hm.to(DEV), tuple(x1.shape[-2:])), want_lo=False)[1]
        meas[Fn.analysis_precision()] = float((torch.round(yw.float()).cpu().to(torch.int16) != want).float().mean())
        assert meas[Fn.analysis_precision()] <= bar, meas
    print("measured:", meas)
