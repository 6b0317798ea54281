"""Whole-model parity on the GPU: HESIC (HSIC) and HESIC+ (HSICJoint) forward against the golden vectors
produced by the reference itself (tests/golden/{hsic,joint}_{64,256}.npz) and against the CPU oracle.

fp32 storage: integer latents may flip only at rounding boundaries (<= 2e-4 of them), bits and MSE within
1e-3 relative -- the BASELINE bar.  bf16 feature maps with the bf16x3 analysis path (default): <= 1e-3 of the latents differ
(measured 4e-5), bits 2e-3, PSNR 1e-3 dB; with single-bf16 analysis operands: bits 4e-3, MSE 1e-3 relative, <= 2 % flipped
latents (measured 1.6e-3 / 1.5e-4 / 1.0 %).  The exact BASELINE workloads (512 x 512 batches, C5, the C3 step) are in
tests/test_gpu_baseline_workloads.py."""
import math
import os

import pytest
import torch

from conftest import GOLDEN, T, load_golden
from hesic_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(kind, dtype=torch.float32):
    import hesic_amd
    from hesic_amd import models
    hesic_amd.set_compute_dtype(dtype)
    net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict())
    return net.to(DEV).eval()


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    import hesic_amd
    hesic_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("size,batch", [(64, 2), (256, 1)])
def test_forward_fp32_matches_reference_golden(kind, size, batch):
    from hesic_amd import models
    g = load_golden(f"{kind}_{size}.npz")
    net = build(kind)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, batch, size, size))
    with torch.no_grad():
        out = net(x1, x2, Hm)
        m = models.metrics_from(models.rate_distortion(out, x1, x2))
    for k in ("y1_hat", "y2_hat"):
        flips = (out[k].cpu().to(torch.int16) != T(g[k])).float().mean()
        assert float(flips) < 2e-4, (k, float(flips))
    for k in ("y1", "y2", "z1", "z2"):
        assert m["bits"][k] == pytest.approx(float(g["bits_" + k]), rel=1e-3), k
    assert m["mse1"] == pytest.approx(float(g["mse1"]), rel=1e-3)
    assert m["mse2"] == pytest.approx(float(g["mse2"]), rel=1e-3)
    n = batch * size * size
    ref_bpp = sum(float(g["bits_" + k]) for k in ("y1", "y2", "z1", "z2")) / n / 2
    ref_psnr = (10 * math.log10(1 / float(g["mse1"])) + 10 * math.log10(1 / float(g["mse2"]))) / 2
    assert abs(m["bpp"] - ref_bpp) < 1e-3 * max(1.0, ref_bpp) and abs(m["psnr"] - ref_psnr) < 1e-3
    if size == 64:
        torch.testing.assert_close(out["x1_hat"].cpu(), T(g["x1_hat"]), rtol=2e-3, atol=2e-4)
        torch.testing.assert_close(out["x2_hat"].cpu(), T(g["x2_hat"]), rtol=2e-3, atol=2e-4)
        torch.testing.assert_close(out["likelihoods"]["z1"].cpu().contiguous(), T(g["lik_z1"]), rtol=2e-3, atol=1e-8)
    else:
        pool = torch.nn.functional.avg_pool2d(out["x2_hat"].cpu(), 8)
        # a latent that flips at a rounding boundary (allowed above) moves its 16x16 footprint slightly
        torch.testing.assert_close(pool, T(g["x2_hat_pool"]), rtol=2e-3, atol=3e-3)


# (16-bit format, analysis mode) -> (max share of flipped latents, total bits relative, PSNR dB) at 256 x 256: measured on MI355X + margin
GOLDEN_256_BARS = {
    ("f16", "x3c2"): (1e-3, 1e-3, 4e-4),      # measured 6.7e-4 / 2e-4 / 4e-5 dB: g_a_conv2 on single fp16 operands (error-feedback weights)
    ("f16", "x3"): (1e-4, 5e-4, 1e-4),        # measured 0 flips / 1e-4 / 7e-6 dB
    ("f16", "x1"): (3e-3, 2e-3, 5e-4),        # measured 1.3e-3
    ("bf16", "x3"): (2e-4, 2e-3, 1e-3),       # round 3's default: measured 2e-5 / 9e-4 / 4e-4 dB
    ("bf16", "x1"): (0.02, 4e-3, 2e-3),       # round 2: 1.0 % / 1.6e-3 / 6e-4 dB
}


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("fmt,analysis", list(GOLDEN_256_BARS), ids=["-".join(k) for k in GOLDEN_256_BARS])
def test_forward_16bit_within_stated_tolerance(kind, fmt, analysis):
    """16-bit feature maps (fp32 accumulation, fp32 latents) against the reference golden at 256x256 (BASELINE config C1's input on the
    GPU) in every (format, analysis precision) the library offers: the integer latents, total bits and PSNR within the bars above."""
    from hesic_amd import functional as Fn, models
    g = load_golden(f"{kind}_256.npz")
    dt = {"f16": torch.float16, "bf16": torch.bfloat16}[fmt]
    net = build(kind, dt)
    flips_max, bits_rel, psnr_db = GOLDEN_256_BARS[(fmt, analysis)]
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 1, 256, 256))
    prev = Fn.set_analysis_precision(analysis)
    try:
        with torch.no_grad():
            assert Fn.fp32_latents()
            out = net(x1, x2, Hm)
            m = models.metrics_from(models.rate_distortion(out, x1, x2))
    finally:
        Fn.set_analysis_precision(prev)
    assert out["y1_hat"].dtype == dt            # integer-valued: exact in the storage dtype of the synthesis convs
    total = sum(float(g["bits_" + k]) for k in ("y1", "y2", "z1", "z2"))
    assert sum(m["bits"].values()) == pytest.approx(total, rel=bits_rel)
    assert m["mse1"] == pytest.approx(float(g["mse1"]), rel=1e-3)
    assert m["mse2"] == pytest.approx(float(g["mse2"]), rel=1e-3)
    ref_psnr = (10 * math.log10(1 / float(g["mse1"])) + 10 * math.log10(1 / float(g["mse2"]))) / 2
    assert abs(m["psnr"] - ref_psnr) < psnr_db, abs(m["psnr"] - ref_psnr)
    for k in ("y1_hat", "y2_hat"):
        flips = (out[k].float().cpu().to(torch.int16) != T(g[k])).float().mean()
        assert float(flips) <= flips_max, (k, float(flips))
    if analysis != "x1":        # the hyper-latents of the pair route are the reference's: z bits to 1e-5 ("x3c2": y differs in 5e-4 of its
        for k in ("z1", "z2"):  # values by one step, which moves z's bits by a little more)
            assert m["bits"][k] == pytest.approx(float(g["bits_" + k]), rel=1e-5 if analysis == "x3" else 2e-3), k


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_fp32_latents_only_change_the_rounding_inputs(kind):
    """HESIC_BF16_LATENTS A/B: with the latents stored as bf16 again (round-1 behaviour) the forward still runs and the two
    modes agree to the bf16 noise level; the fp32-latent mode rounds y from the fp32 accumulator, so re-rounding ITS bf16
    copy reproduces the bf16-latent mode's y_hat wherever the two differ only by storage."""
    from hesic_amd import functional as Fn, models
    net = build(kind, torch.bfloat16)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(2, 1, 128, 128))
    keep = Fn.FP32_LATENTS
    try:
        outs = {}
        for mode in (True, False):
            Fn.FP32_LATENTS = mode
            with torch.no_grad():
                outs[mode] = net(x1, x2, Hm)
        same = (outs[True]["y1_hat"] == outs[False]["y1_hat"]).float().mean()
        assert float(same) > 0.98
        b_hi = float(models.metrics_from(models.rate_distortion(outs[True], x1, x2))["bpp"])
        b_lo = float(models.metrics_from(models.rate_distortion(outs[False], x1, x2))["bpp"])
        assert abs(b_hi - b_lo) < 1e-2 * b_lo
    finally:
        Fn.FP32_LATENTS = keep


def test_forward_512_batch_properties():
    """BASELINE size (512x512): size-independent properties -- pairs are independent (batch-of-2 equals two
    batch-of-1 runs bit for bit), latents are integers, likelihoods are in [1e-9, 1]."""
    net = build("hsic")
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(3, 2, 512, 512))
    with torch.no_grad():
        both = net(x1, x2, Hm)
        one = net(x1[1:], x2[1:], Hm[1:])
    assert torch.equal(both["y2_hat"][1:], one["y2_hat"]) and torch.equal(both["x2_hat"][1:], one["x2_hat"])
    assert torch.equal(both["y1_hat"], both["y1_hat"].round())
    for l in both["likelihoods"].values():
        assert float(l.min()) >= float(torch.tensor(1e-9, dtype=torch.float32)) and float(l.max()) <= 1.0 + 1e-6
    assert both["x1_hat"].shape == (2, 3, 512, 512) and both["y1_hat"].shape == (2, 192, 32, 32)
    assert both["likelihoods"]["z1"].shape == (2, 128, 8, 8)


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_train_trace_matches_reference(kind):
    """Row T: two consecutive optimiser steps at 64x64 with injected noise reproduce the reference's
    (loss, bpp, mse, aux) trace and its step-0 gradient norms."""
    from hesic_amd import models
    from hesic_amd.train import Trainer
    g = load_golden(f"{kind}_train64.npz")
    net = build(kind)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 2, 64, 64))
    order = [str(s) for s in g["noise_order"]]
    shapes = {"z1": (128, 1, 2), "z2": (128, 1, 2)}
    tr = Trainer(net, lr=1e-4, aux_lr=1e-3, lmbda=0.0067)
    trace = []
    for step in range(2):
        noise = {}
        for k in order:
            shp = shapes.get(k, (2, 192, 4, 4))
            nz = synthetic._uniform(f"noise.{kind}.{step}.{k}", shp, -0.5, 0.5)
            if k in ("z1", "z2"):      # reference layout (C, 1, B*H*W) with B fastest -> (B,C,1,1)
                nz = nz.reshape(128, 1, 1, 2).permute(3, 0, 1, 2).contiguous()
            noise[k] = nz.to(DEV)
        if step == 0:
            # gradient norms of the first backward, before any update
            net.train()
            out = net(x1, x2, Hm, noise=noise)
            from hesic_amd import functional as Fn
            Fn.rd_loss(out, x1, x2, 0.0067)["loss"].backward()
            named = dict(net.named_parameters())
            bad = []
            for name, p in named.items():
                # MaskedConv2d: the reference also back-propagates into the taps its mask zeroes on every
                # forward; they never influence the model, so the HIP path skips them -> compare live taps
                live = "gn_live_" + name
                ref = float(g[live] if live in g else g["gn_" + name])
                got = float(p.grad.double().norm()) if p.grad is not None else 0.0
                if abs(got - ref) > 2e-2 * max(ref, 1e-6) + 1e-7:
                    bad.append((name, got, ref))
            assert not bad, bad[:8]
            torch.testing.assert_close(named["encoder1.g_a_conv4.bias"].grad.cpu(), T(g["g_encoder1.g_a_conv4.bias"]), rtol=2e-2, atol=1e-5)
            torch.testing.assert_close(named["decoder2.after_conv.bias"].grad.cpu(), T(g["g_decoder2.after_conv.bias"]), rtol=2e-2, atol=1e-5)
        crit = tr.step(x1, x2, Hm, noise=noise)
        trace.append([float(crit["loss"]), float(crit["bpp_loss"]), float(crit["mse_loss"]), float(crit["aux_loss"])])
    ref = g["trace"]
    for s in range(2):
        for j, nm in enumerate(("loss", "bpp", "mse", "aux")):
            assert trace[s][j] == pytest.approx(float(ref[s][j]), rel=5e-3), (s, nm, trace, ref)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)], ids=["f32", "bf16"])
def test_independent_en_matches_reference(dtype, tol):
    """Independent_EN (9 residual blocks of 3x3 C=32 convs per view + 2 warps) on the HIP path."""
    import hesic_amd
    from hesic_amd import models
    from test_oracle_golden import _en_params
    g = load_golden("en_64.npz")
    hesic_amd.set_compute_dtype(dtype)
    net = models.Independent_EN()
    net.load_state_dict(_en_params(), strict=True)
    net = net.to(DEV).eval()
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(5, 2, 64, 64))
    with torch.no_grad():
        out = net(x1, x2, Hm)
    for k in ("x1_hat", "x2_hat"):
        ref = T(g[k])
        err = float((out[k].float().cpu() - ref).abs().max() / ref.abs().max())
        assert err < tol, (k, err)


def test_stream_overlap_and_fusion_do_not_change_results():
    """The multi-stream inference schedule is bit-identical to the single-stream order; the fused conv+GDN epilogue
    agrees with the two-kernel path to bf16 rounding."""
    from hesic_amd import functional as Fn, models
    net = build("hsic", torch.bfloat16)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(1, 2, 256, 256))
    keep = (models.OVERLAP_STREAMS, Fn.FUSE_CONV_GDN)
    try:
        outs = {}
        for ov in (True, False):
            models.OVERLAP_STREAMS = ov
            with torch.no_grad():
                outs[ov] = net(x1, x2, Hm)
            torch.cuda.synchronize()
        for k in ("x1_hat", "x2_hat", "y1_hat", "y2_hat"):
            assert torch.equal(outs[True][k], outs[False][k]), k
        for k, v in outs[True]["likelihoods"].items():
            assert torch.equal(v, outs[False]["likelihoods"][k]), k
        Fn.FUSE_CONV_GDN = False
        with torch.no_grad():
            ref = net(x1, x2, Hm)
        m_f = models.metrics_from(models.rate_distortion(outs[False], x1, x2))
        m_u = models.metrics_from(models.rate_distortion(ref, x1, x2))
        assert abs(m_f["bpp"] - m_u["bpp"]) < 2e-2 * m_u["bpp"] and abs(m_f["psnr"] - m_u["psnr"]) < 5e-2
    finally:
        models.OVERLAP_STREAMS, Fn.FUSE_CONV_GDN = keep


@pytest.mark.parametrize("which", ["hsic", "joint"])
def test_critical_chain_schedule_is_race_free_back_to_back(which):
    """The forked inference schedule (chain on the main stream; view 2 + decoder2 and view 1's rate on side streams) under the
    conditions that expose a missing cross-stream dependency: forwards back to back without a host sync, every stream's
    allocator pool seeded with NaN / 1e30 garbage of random sizes in between.  Every output of every forward equals the
    single-stream order bit for bit."""
    import random
    from hesic_amd import models
    rng = random.Random(7)
    net = build(which, torch.bfloat16)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(1, 2, 256, 256))

    def perturb():
        junk = []
        for st in [torch.cuda.current_stream()] + [models._side_stream(x1.device, i) for i in (1, 2, 10, 12)]:
            with torch.cuda.stream(st):
                for _ in range(rng.randint(1, 5)):
                    n = rng.choice([1 << 10, 1 << 16, 1 << 20, 3 << 20, 1 << 24])
                    junk.append(torch.full((n,), float("nan") if rng.random() < 0.5 else 1e30, device=DEV))
        torch.cuda.synchronize()
        del junk
        if rng.random() < 0.3:
            torch.cuda.empty_cache()

    def run(overlap, n):
        models.OVERLAP_STREAMS = overlap
        outs = []
        with torch.no_grad():
            for _ in range(n):
                o = net(x1, x2, Hm)
                d = {k: o[k].clone() for k in ("x1_hat", "x2_hat", "y1_hat", "y2_hat")}
                d.update({"lik_" + k: v.clone() for k, v in o["likelihoods"].items()})
                outs.append(d)
                del o
        torch.cuda.synchronize()
        return outs

    keep = models.OVERLAP_STREAMS
    try:
        ref = run(False, 1)[0]
        for _ in range(4):
            perturb()
            for i, got in enumerate(run(True, 4)):
                for k, v in got.items():
                    assert torch.equal(v, ref[k]), (which, i, k)
    finally:
        models.OVERLAP_STREAMS = keep


def test_auto_forward_picks_a_mode_and_returns_the_eager_tensors():
    from hesic_amd import models
    net = build("hsic", torch.bfloat16)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(2, 2, 128, 128))
    auto = models.AutoForward(net, x1, x2, Hm, trial=5)
    assert auto.mode in ("eager", "graph") and set(auto.timings) == {"eager", "graph"}
    with torch.no_grad():
        ref = net(x1, x2, Hm)
    for forced in ("eager", "graph"):
        auto.mode = forced
        out = auto(x1, x2, Hm)
        for k in ("x1_hat", "x2_hat", "y1_hat", "y2_hat"):
            assert torch.equal(out[k], ref[k]), (forced, k)
    a, b, h = (t.to(DEV) for t in synthetic.stereo_batch(3, 1, 128, 128))       # another shape: eager, whatever the mode
    out = auto(a, b, h)
    with torch.no_grad():
        ref1 = net(a, b, h)
    assert torch.equal(out["x2_hat"], ref1["x2_hat"])
    # graph mode is a drop-in: results of successive calls do not alias, and a parameter update after capture is picked up
    auto.mode = "graph"
    x1b, x2b, Hb = (t.to(DEV) for t in synthetic.stereo_batch(7, 2, 128, 128))
    first = auto(x1, x2, Hm)
    second = auto(x1b, x2b, Hb)
    assert first["x2_hat"].data_ptr() != second["x2_hat"].data_ptr()
    assert torch.equal(first["x2_hat"], ref["x2_hat"]) and not torch.equal(second["x2_hat"], ref["x2_hat"])
    with torch.no_grad():
        net.decoder2.after_conv.bias.add_(0.25)
        want = net(x1, x2, Hm)
    got = auto(x1, x2, Hm)
    assert torch.equal(got["x2_hat"], want["x2_hat"]) and not torch.equal(got["x2_hat"], ref["x2_hat"])


def test_graphed_forward_replays_the_eager_result():
    """HIP-graph capture of the whole eval forward (side streams included) + reductions: replay == eager, bit for bit, also
    for new inputs copied into the static buffers."""
    import hesic_amd
    from hesic_amd import models
    prev = hesic_amd.functional.compute_dtype()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        net = models.HSIC()
        synthetic.fill_state_dict_(net.state_dict())
        net = net.cuda().eval()
        a = [t.cuda() for t in synthetic.stereo_batch(0, 2, 128, 128)]
        b = [t.cuda() for t in synthetic.stereo_batch(7, 2, 128, 128)]
        g = models.GraphedForward(net, *a)
        for inp in (a, b, a):
            with torch.no_grad():
                want = net(*inp)
                want_rd = models.rate_distortion(want, inp[0], inp[1])
            got, got_rd = g(*inp)
            for k in ("x1_hat", "x2_hat", "y1_hat", "y2_hat"):
                assert torch.equal(got[k], want[k]), k
            for k in want["likelihoods"]:
                assert torch.equal(got["likelihoods"][k], want["likelihoods"][k]), k
            assert abs(models.metrics_from(got_rd)["bpp"] - models.metrics_from(want_rd)["bpp"]) < 1e-9     # fp64 atomics: order-free to ~1e-16
    finally:
        hesic_amd.set_compute_dtype(prev)


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["f16", "f32"])
def test_segmented_forward_replays_the_eager_result(kind, dtype):
    """The eval forward as a plan of single-stream HIP graphs (``SegmentedForward``: the eager multi-stream schedule, its events
    replayed as events, every stretch of a stream in between as one graph): bit-identical to the eager forward, also for new inputs
    copied into the static buffers, also back to back without a host sync in between."""
    import hesic_amd
    from hesic_amd import models
    hesic_amd.set_compute_dtype(dtype)
    net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.cuda().eval()
    a = [t.cuda() for t in synthetic.stereo_batch(0, 2, 128, 192)]
    b = [t.cuda() for t in synthetic.stereo_batch(7, 2, 128, 192)]
    g = models.SegmentedForward(net, *a)
    assert 4 <= g.n_graphs <= 40, g.n_graphs
    for inp in (a, b, a, b):
        with torch.no_grad():
            want = net(*inp)
        got, _ = g(*inp)
        for k in ("x1_hat", "x2_hat", "y1_hat", "y2_hat"):
            assert torch.equal(got[k], want[k]), k
        for k in want["likelihoods"]:
            assert torch.equal(got["likelihoods"][k], want["likelihoods"][k]), k
    outs = []
    for inp in (a, b, a):                  # no synchronisation between replays: the plan's own events order them
        got, _ = g(*inp)
        outs.append({k: got[k].clone() for k in ("x2_hat", "y2_hat")})
    torch.cuda.synchronize()
    with torch.no_grad():
        wa, wb = net(*a), net(*b)
    for o, w in zip(outs, (wa, wb, wa)):
        assert torch.equal(o["x2_hat"], w["x2_hat"]) and torch.equal(o["y2_hat"], w["y2_hat"])


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_graphed_trainer_follows_the_eager_trace(kind):
    """Whole training step (zero_grad -> forward -> R-D backward -> Adam -> aux backward -> aux Adam) captured into a HIP
    graph: with the same injected noise the replayed steps give the eager Trainer's loss trace and parameters."""
    import hesic_amd
    from hesic_amd import models
    from hesic_amd.train import Trainer, GraphedTrainer
    prev = hesic_amd.functional.compute_dtype()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 2, 128, 128))
        def noise_for(step):
            shp = {"z1": (2, 128, 2, 2), "z2": (2, 128, 2, 2)}
            return {k: synthetic._uniform(f"gt.noise.{step}.{k}", shp.get(k, (2, 192, 8, 8)), -0.5, 0.5).cuda()
                    for k in ("z1", "y1", "y1b", "y1w", "z2", "y2", "y2b")}
        traces, finals = [], []
        for cls, kw in ((Trainer, {}), (GraphedTrainer, {"warmup": 2})):
            net = models.HSIC() if kind == "hsic" else models.HSICJoint()
            synthetic.fill_state_dict_(net.state_dict())
            net = net.cuda()
            tr = cls(net, lr=1e-4, aux_lr=1e-3, lmbda=0.0067, **kw)
            trace = []
            for step in range(5):
                c = tr.step(x1, x2, Hm, noise=noise_for(step))
                trace.append([float(c["loss"]), float(c["bpp_loss"]), float(c["mse_loss"]), float(c["aux_loss"])])
            traces.append(trace)
            finals.append({k: v.detach().float().clone() for k, v in net.named_parameters()})
        assert tr.graph is not None                       # steps 3 and 4 were graph replays
        for a, b in zip(*traces):
            for u, v in zip(a, b):
                # atomics settle in a different order under graph replay; Adam turns a flipped sign of a ~0 gradient into a 2*lr step,
                # and the context model of HESIC+ amplifies that faster than HESIC does (eager reruns agree to 4-5 digits, eager vs
                # graph to 3)
                assert u == pytest.approx(v, rel=2e-3 if kind == "hsic" else 8e-3), traces
        for k in ("encoder1.g_a_conv2.weight", "decoder2.after_conv.bias", "entropy_bottleneck1._biases.0", "entropy_bottleneck1.quantiles"):
            # Adam moves an element by <= lr per step whatever the gradient's size: where a ~0 gradient flips sign between two
            # runs (bf16 + atomics are not run-to-run bit-stable) the two trajectories drift by up to 2 * lr per step
            torch.testing.assert_close(finals[1][k], finals[0][k], rtol=1e-2, atol=2 * 5 * 1e-3 if "entropy_bottleneck" in k else 2 * 5 * 1e-4 + 1e-4)
    finally:
        hesic_amd.set_compute_dtype(prev)


def test_multi_tensor_adam_matches_torch_adam():
    """hesic_adam_step (all tensors of a group in a few launches) against torch.optim.Adam on the same gradients:
    parameters after 6 steps, state layout interchangeable through state_dict."""
    from hesic_amd.train import MultiTensorAdam
    shapes = [(128, 128, 5, 5), (128,), (3, 1, 1), (960, 960, 1, 1), (1,), (37, 5)] * 6        # > 24 tensors: two chunks
    ps_a = [synthetic._uniform(f"ad.p{i}", s, -1, 1).cuda().requires_grad_() for i, s in enumerate(shapes)]
    ps_b = [p.detach().clone().requires_grad_() for p in ps_a]
    a, b = MultiTensorAdam(ps_a, lr=1e-3), torch.optim.Adam(ps_b, lr=1e-3)
    for step in range(6):
        for i, (pa, pb) in enumerate(zip(ps_a, ps_b)):
            g = synthetic._uniform(f"ad.g{step}.{i}", pa.shape, -1, 1).cuda() * (10.0 ** (i % 5 - 3))
            pa.grad, pb.grad = g.clone(), g.clone()
        a.step(); b.step()
    for pa, pb in zip(ps_a, ps_b):
        torch.testing.assert_close(pa.detach(), pb.detach(), rtol=2e-6, atol=2e-7)
    sd = a.state_dict()
    assert float(sd["state"][0]["step"]) == 6.0 and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    c = torch.optim.Adam([p.detach().clone().requires_grad_() for p in ps_a], lr=1e-3, capturable=True)
    c.load_state_dict(sd)                                  # a torch Adam resumes from the HIP optimiser's checkpoint
    torch.testing.assert_close(c.state_dict()["state"][3]["exp_avg_sq"], sd["state"][3]["exp_avg_sq"])


def test_bias_gradient_survives_graph_replay():
    """Regression: buffers that kernels accumulate into are zeroed by kernels, never by hipMemsetAsync -- recorded into a HIP
    graph, the memset node did not reliably precede the accumulating kernel from the second replay on (bias gradients were
    summed on top of the previous replay's values).  Wide conv (bias zero-fill rides on the wgrad launch) and the image-side
    3 -> 128 conv (zero kernel), poisoned output buffers, five replays each against the eager result."""
    import ctypes as C
    from hesic_amd import functional as Fn, _lib as L
    torch.manual_seed(0)
    bf16, cl = torch.bfloat16, torch.channels_last
    x = torch.randn(4, 128, 32, 32, device="cuda").to(bf16).contiguous(memory_format=cl)
    gy = torch.randn(4, 128, 16, 16, device="cuda").to(bf16).contiguous(memory_format=cl)
    d = L.ConvDesc(4, 32, 32, 128, 16, 16, 128, 5, 5, 2, 2, 0, L.dt(x), 0, 0, 128, 0, 128, 0, 0)
    nws = L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(d))

    def wide(poison):
        dwp = torch.empty(25 * 128 * 128, dtype=torch.float32, device="cuda")
        db = torch.empty(128, dtype=torch.float32, device="cuda").fill_(poison)
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device="cuda")
        L.call("hesic_conv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(dwp), L.ptr(db), L.ptr(ws), nws, L.stream())
        return db

    img = torch.rand(4, 3, 64, 64, device="cuda")
    w1 = (torch.randn(128, 3, 5, 5, device="cuda") * 0.1).requires_grad_()
    b1 = torch.zeros(128, device="cuda").requires_grad_()
    g1 = torch.randn(4, 128, 32, 32, device="cuda").to(bf16).contiguous(memory_format=cl)
    prev = Fn.compute_dtype()
    Fn.set_compute_dtype(bf16)
    try:
        def narrow(_poison):
            w1.grad = b1.grad = None
            y = Fn.conv2d(img, w1, b1, kernel_size=5, stride=2, padding=2)
            y.backward(g1)
            return torch.cat((b1.grad.reshape(-1), w1.grad.reshape(-1)[:256]))

        for fn in (wide, narrow):
            ref = fn(0.0).clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn(float("nan"))
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn(float("nan"))
            for _ in range(5):
                g.replay()
                torch.cuda.synchronize()
                assert torch.isfinite(out).all()
                assert float((out - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-5
    finally:
        Fn.set_compute_dtype(prev)


def _fresh_like(net, kind="hsic"):
    from hesic_amd import models
    twin = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    twin.load_state_dict({k: v.detach().clone() for k, v in net.state_dict().items()}, strict=True)
    return twin.cuda().eval()


def test_eval_after_graph_replays_sees_the_updated_parameters():
    """a graph replay runs no Python, and the captured Adam kernel writes parameters through raw pointers -- the
    caches keyed on version counters (bottleneck table, packed GDN parameters, inference weight packs) must not serve the
    pre-update values.  eval -> N replays -> eval must equal a fresh model loaded from the same state_dict, bit for bit."""
    import hesic_amd
    from hesic_amd import models
    from hesic_amd.train import GraphedTrainer
    prev = hesic_amd.functional.compute_dtype()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        net = models.HSIC()
        synthetic.fill_state_dict_(net.state_dict())
        net = net.cuda()
        x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 2, 128, 128))
        tr = GraphedTrainer(net, lr=1e-3, aux_lr=1e-2, lmbda=0.0067, warmup=2)
        net.eval()
        with torch.no_grad():
            before = net(x1, x2, Hm)                      # fills every inference cache
        for _ in range(5):                               # 2 eager warm-ups, capture, 2 replays
            tr.step(x1, x2, Hm)
        assert tr.graph is not None
        net.eval()
        with torch.no_grad():
            after = net(x1, x2, Hm)
            want = _fresh_like(net)(x1, x2, Hm)
        assert not torch.equal(after["likelihoods"]["z1"], before["likelihoods"]["z1"])      # the parameters did move
        for k in ("x1_hat", "x2_hat", "y1_hat", "y2_hat"):
            assert torch.equal(after[k], want[k]), k
        for k in want["likelihoods"]:
            assert torch.equal(after["likelihoods"][k], want["likelihoods"][k]), k
    finally:
        hesic_amd.set_compute_dtype(prev)


def test_multi_tensor_adam_is_self_contained_in_a_user_loop():
    """the reference's loop with the optimiser swapped (loss.backward(); MultiTensorAdam.step(), no Trainer) must
    run the next forward on the updated weights: version counters are bumped and the pack caches invalidated by step()."""
    from hesic_amd import functional as Fn, models
    from hesic_amd.train import MultiTensorAdam
    net = build("hsic")
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(4, 1, 64, 64))
    opt = MultiTensorAdam(list(net.parameters()), lr=1e-3)
    aux = MultiTensorAdam(list(net.aux_parameters()), lr=1e-2)
    noise = {k: synthetic._uniform(f"ul.{k}", (1, 128, 1, 1) if k[0] == "z" else (1, 192, 4, 4), -0.5, 0.5).to(DEV) for k in ("z1", "y1", "y1w", "z2", "y2")}
    for _ in range(2):
        net.train()
        v0 = net.encoder1.g_a_conv2.weight._version
        out = net(x1, x2, Hm, noise=noise)
        loss = Fn.rd_loss(out, x1, x2, 0.0067)["loss"]
        opt.zero_grad(); aux.zero_grad()
        loss.backward()
        opt.step()
        net.aux_loss().backward()
        aux.step()
        assert net.encoder1.g_a_conv2.weight._version > v0
        net.eval()
        with torch.no_grad():
            got = net(x1, x2, Hm)
            want = _fresh_like(net)(x1, x2, Hm)
        for k in ("x1_hat", "x2_hat", "y1_hat"):
            assert torch.equal(got[k], want[k]), k
        assert torch.equal(got["likelihoods"]["z2"], want["likelihoods"]["z2"])


def test_rd_loss_gradient_follows_a_scaled_loss():
    """(loss / accum).backward() must give gradients scaled by 1 / accum (they used to ignore g_loss)."""
    from hesic_amd import functional as Fn
    net = build("hsic")
    net.train()
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(4, 1, 64, 64))
    noise = {k: synthetic._uniform(f"sl.{k}", (1, 128, 1, 1) if k[0] == "z" else (1, 192, 4, 4), -0.5, 0.5).to(DEV) for k in ("z1", "y1", "y1w", "z2", "y2")}
    grads = []
    for scale in (1.0, 0.25):
        net.zero_grad(set_to_none=True)
        for p in net.aux_parameters():        # parameters() skips the bottlenecks (newnet1.py:74-80): zero_grad() does not reach them
            p.grad = None
        c = Fn.rd_loss(net(x1, x2, Hm, noise=noise), x1, x2, 0.0067)
        assert not c["bpp_loss"].requires_grad and not c["mse_loss"].requires_grad
        (c["loss"] * scale).backward()
        grads.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    for n in ("encoder1.g_a_conv1.weight", "decoder2.after_conv.bias", "_h_s2.gmm_means.4.weight", "entropy_bottleneck1._matrices.2"):
        ref = 0.25 * grads[0][n]          # two backward passes: atomics settle in a different order, elements near zero move by ~1e-7 of the scale
        torch.testing.assert_close(grads[1][n], ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))


def _noise_hsic(tag, B, size):
    zs, ys = (B, 128, size // 64, size // 64), (B, 192, size // 16, size // 16)
    return {k: synthetic._uniform(f"{tag}.{k}", zs if k[0] == "z" else ys, -0.5, 0.5).to(DEV) for k in ("z1", "y1", "y1w", "z2", "y2")}


def test_trainer_step_equals_the_average_of_split_batch_gradients():
    """SURVEY 8e on one GPU: the gradient of a batch of 4 equals the mean of the gradients of its two halves (what two DP
    ranks would all-reduce) -- compared on the flat gradient buffers the reducer works on, both optimiser groups -- and an
    Adam step on the averaged buffers lands where ``Trainer.step`` on the whole batch does."""
    from hesic_amd import functional as Fn
    from hesic_amd.train import Trainer
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 4, 64, 64))
    noise = _noise_hsic("sb", 4, 64)

    def grads(tr, sl):
        tr.model.train()
        tr.main_group.zero_grad(); tr.aux_group.zero_grad()
        out = tr.model(x1[sl], x2[sl], Hm[sl], noise={k: v[sl] for k, v in noise.items()})
        Fn.rd_loss(out, x1[sl], x2[sl], 0.0067)["loss"].backward()
        return tr.main_group.flat_g.clone(), tr.aux_group.flat_g.clone()

    half = Trainer(build("hsic"), lr=1e-3, aux_lr=1e-2, lmbda=0.0067)
    gm_full, ga_full = grads(half, slice(0, 4))
    (gm1, ga1), (gm2, ga2) = grads(half, slice(0, 2)), grads(half, slice(2, 4))
    gm_avg, ga_avg = (gm1 + gm2) / 2, (ga1 + ga2) / 2
    for views_full, views_avg, group in ((half.main_group.view_like_params(gm_full), half.main_group.view_like_params(gm_avg), half.main_group),
                                          (half.aux_group.view_like_params(ga_full), half.aux_group.view_like_params(ga_avg), half.aux_group)):
        for a, b in zip(views_full, views_avg):
            scale = float(a.abs().max())
            assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-9, (tuple(a.shape), scale)       # fp32 atomics order differs: 4 vs 2 + 2
    # the update itself: Adam on the averaged buffers vs Trainer.step on the batch (an element moves by <= lr per step; where a ~0
    # gradient changes sign between the two summation orders the two results differ by up to 2 lr)
    half.main_group.flat_g.copy_(gm_avg)
    half.optimizer.step()
    half.aux_group.flat_g.copy_(ga_avg)
    half.model.aux_loss().backward()                # + the (rank-independent) quantile gradient
    half.aux_optimizer.step()
    full = Trainer(build("hsic"), lr=1e-3, aux_lr=1e-2, lmbda=0.0067)
    full.step(x1, x2, Hm, noise=noise)
    a, b = dict(full.model.named_parameters()), dict(half.model.named_parameters())
    for k in a:
        lr = 1e-2 if k.startswith("entropy_bottleneck") else 1e-3
        diff = (a[k].detach() - b[k].detach()).abs()
        assert float(diff.max()) <= 2.1 * lr, k
        assert float((diff > 0.2 * lr).float().mean()) < 0.02, k          # ... and that is rare


def test_flat_gradients_match_autograd_accumulation():
    """The in-place gradient path (finishing kernel adds into the flat slot, None to autograd) against the plain autograd
    path (fresh tensors, AccumulateGrad) on the same model and batch: every parameter, incl. encoder1's twice-used weights."""
    from hesic_amd import functional as Fn
    from hesic_amd.train import FlatGroup
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(1, 2, 64, 64))
    noise = _noise_hsic("fg", 2, 64)
    grads = []
    for flat in (False, True):
        net = build("hsic")
        net.train()
        groups = [FlatGroup(list(net.parameters())), FlatGroup(list(net.aux_parameters()))] if flat else []
        for g in groups:
            g.zero_grad()
        prev = Fn.grad_slots_active(flat)          # the direct-write path is gated: Trainer.step opens it for its own backward passes
        try:
            Fn.rd_loss(net(x1, x2, Hm, noise=noise), x1, x2, 0.0067)["loss"].backward()
        finally:
            Fn.grad_slots_active(prev)
        if flat:
            assert all(s.writes > 0 for g in groups for s, p in zip(g.slots, g.params))          # every gradient took the in-place path
            assert dict(net.named_parameters())["encoder1.g_a_conv2.weight"].grad.data_ptr() == groups[0].grad_views[
                [id(p) for p in groups[0].params].index(id(net.encoder1.g_a_conv2.weight))].data_ptr()
        grads.append({n: p.grad.detach().clone() for n, p in net.named_parameters()})
        del groups
    for n in grads[0]:
        ref = grads[0][n]
        torch.testing.assert_close(grads[1][n], ref, rtol=2e-4, atol=2e-5 * float(ref.abs().max()) + 1e-9, msg=n)


def test_graphed_trainer_with_rccl_collectives_inside_the_graph():
    """A 1-rank nccl (= RCCL) process group on this box: the bucketed in-place all-reduces of the flat gradient are issued
    (force_collectives) and captured into the step's HIP graph; replays follow the plain eager trainer."""
    import torch.distributed as dist
    from hesic_amd.train import GraphedTrainer, Trainer
    import hesic_amd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    prev = hesic_amd.functional.compute_dtype()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 2, 128, 128))
        noise = _noise_hsic("rc", 2, 128)
        traces = []
        for cls, kw in ((Trainer, {}), (GraphedTrainer, {"warmup": 2, "force_collectives": True, "bucket_mb": 16.0})):
            from hesic_amd import models
            net = models.HSIC()
            synthetic.fill_state_dict_(net.state_dict())
            tr = cls(net.cuda(), lr=1e-4, aux_lr=1e-3, lmbda=0.0067, **kw)
            traces.append([[float(v) for v in tr.step(x1, x2, Hm, noise=noise).values()] for _ in range(5)])
        assert tr.graph is not None and tr.main_reducer.active and len(tr.main_reducer.buckets) >= 4
        for a, b in zip(*traces):
            for u, v in zip(a, b):
                assert u == pytest.approx(v, rel=2e-3), traces
    finally:
        hesic_amd.set_compute_dtype(prev)
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("align", [True, False], ids=["ac1", "ac0"])
def test_forward_non_square_both_warp_conventions(kind, align):
    """256 x 320 against the reference golden generated with the matching kornia convention (geometry.DEFAULT_ALIGN_CORNERS:
    True = kornia >= 0.5, False = the 0.4.x sampling torch-1.6-era checkpoints were trained with)."""
    from hesic_amd import geometry, models
    g = load_golden(f"{kind}_256x320{'' if align else '_ac0'}.npz")
    net = build(kind)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 1, 256, 320))
    keep = geometry.DEFAULT_ALIGN_CORNERS
    geometry.DEFAULT_ALIGN_CORNERS = align
    try:
        with torch.no_grad():
            out = net(x1, x2, Hm)
            m = models.metrics_from(models.rate_distortion(out, x1, x2))
    finally:
        geometry.DEFAULT_ALIGN_CORNERS = keep
    assert out["x1_hat"].shape == (1, 3, 256, 320) and out["y1_hat"].shape == (1, 192, 16, 20)
    for k in ("y1_hat", "y2_hat"):
        assert float((out[k].cpu().to(torch.int16) != T(g[k])).float().mean()) < 2e-4, k
    for k in ("y1", "y2", "z1", "z2"):
        assert m["bits"][k] == pytest.approx(float(g["bits_" + k]), rel=1e-3), k
    assert m["mse1"] == pytest.approx(float(g["mse1"]), rel=1e-3) and m["mse2"] == pytest.approx(float(g["mse2"]), rel=1e-3)
    torch.testing.assert_close(torch.nn.functional.avg_pool2d(out["x2_hat"].cpu(), 8), T(g["x2_hat_pool"]), rtol=2e-3, atol=3e-3)


def test_c5_860x1080_padded_matches_the_oracle():
    """BASELINE config C5: an InStereo2K-size 860 x 1080 pair.  The hyper path needs multiples of 64 (the reference raises on
    860 x 1080): pad to 896 x 1088 (zeros, bottom / right; pixel coordinates -- hence the homography -- unchanged),
    run, crop the reconstructions, bpp over the ORIGINAL pixel count.  Checked against the CPU oracle on the same padded input."""
    from hesic_amd import models
    from oracle import hesic_oracle as O
    net = build("hsic")
    P = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    x1, x2, Hm = synthetic.stereo_batch(0, 1, 860, 1080)
    x1p, x2p = models.pad_to_multiple(x1), models.pad_to_multiple(x2)
    assert x1p.shape[-2:] == (896, 1088) and torch.equal(x1p[..., :860, :1080], x1) and float(x1p[..., 860:, :].abs().max()) == 0
    with torch.no_grad():
        out = net(x1p.to(DEV), x2p.to(DEV), Hm.to(DEV))
        m = models.metrics_from(models.rate_distortion(out, x1.to(DEV), x2.to(DEV)))       # crops to 860 x 1080, bpp over 860*1080
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        ref = O.hsic_forward(P, x1p, x2p, Hm)
    for k in ("y1_hat", "y2_hat"):
        assert float((out[k].cpu() != ref[k]).float().mean()) < 2e-4, k
    ref_crop = {"x1_hat": ref["x1_hat"][..., :860, :1080], "x2_hat": ref["x2_hat"][..., :860, :1080], "likelihoods": ref["likelihoods"]}
    mr = O.metrics(ref_crop, x1, x2)
    assert out["y1_hat"].shape == (1, 192, 56, 68)
    assert m["bpp"] == pytest.approx(mr["bpp"], rel=1e-3) and abs(m["psnr"] - mr["psnr"]) < 1e-3
    assert m["bpp"] == pytest.approx(sum(mr["bits"].values()) / (860 * 1080) / 2, rel=1e-3)     # over the original pixels


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_train_trace_256_matches_reference(kind):
    """Row T at 256 x 256 (SURVEY 8a asked for 64 AND 256): two optimiser steps with injected noise against the reference's
    (loss, bpp, mse, aux) trace."""
    from hesic_amd.train import Trainer
    g = load_golden(f"{kind}_train256.npz")
    net = build(kind)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 1, 256, 256))
    tr = Trainer(net, lr=1e-4, aux_lr=1e-3, lmbda=0.0067)
    trace = []
    for step in range(2):
        noise = {}
        for k in (str(s) for s in g["noise_order"]):
            if k[0] == "z":      # reference layout (C, 1, H*W*B), B fastest -> (B, C, H, W)
                nz = synthetic._uniform(f"noise.{kind}.t256.{step}.{k}", (128, 1, 16), -0.5, 0.5).reshape(128, 4, 4, 1).permute(3, 0, 1, 2).contiguous()
            else:
                nz = synthetic._uniform(f"noise.{kind}.t256.{step}.{k}", (1, 192, 16, 16), -0.5, 0.5)
            noise[k] = nz.to(DEV)
        c = tr.step(x1, x2, Hm, noise=noise)
        trace.append([float(c["loss"]), float(c["bpp_loss"]), float(c["mse_loss"]), float(c["aux_loss"])])
    for s in range(2):
        for j, nm in enumerate(("loss", "bpp", "mse", "aux")):
            assert trace[s][j] == pytest.approx(float(g["trace"][s][j]), rel=5e-3), (s, nm, trace, g["trace"])


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16x3"])
def test_forward_edge_cases(kind, dtype):
    """Ragged / empty / strided inputs: sizes that are not multiples of 64 (the reference dies with a size mismatch deep inside,
    SURVEY.md 5), an empty batch and mismatched views raise up front; non-contiguous views of the images and a broadcast
    (1, 3, 3) homography give the result of their contiguous / expanded copies bit for bit; the smallest legal image works."""
    net = build(kind, dtype)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 2, 64, 128))
    with torch.no_grad():
        ref = net(x1, x2, Hm)
        for bad in ((x1[..., :60, :], x2[..., :60, :], Hm), (x1[:0], x2[:0], Hm[:0]), (x1, x2[:, :, :, :64], Hm), (x1, x2, Hm[:, :2])):
            with pytest.raises(RuntimeError):
                net(*bad)
        wide1, wide2 = torch.zeros(2, 3, 64, 256, device=DEV), torch.zeros(2, 3, 64, 256, device=DEV)
        wide1[..., 64:192], wide2[..., 64:192] = x1, x2
        v1, v2 = wide1[..., 64:192], wide2[..., 64:192]                      # row stride 256, not contiguous
        assert not v1.is_contiguous()
        out = net(v1, v2, Hm)
        for k in ("x1_hat", "x2_hat", "y1_hat", "y2_hat"):
            assert torch.equal(out[k], ref[k]), k
        one = net(x1, x2, Hm[:1])                                               # (1, 3, 3) homography for the whole batch
        exp = net(x1, x2, Hm[:1].expand(2, 3, 3).contiguous())
        assert torch.equal(one["x2_hat"], exp["x2_hat"]) and torch.equal(one["y2_hat"], exp["y2_hat"])
        tiny = net(x1[:1, :, :, :64], x2[:1, :, :, :64], Hm[:1])              # 64 x 64: one z value per channel
        assert tiny["likelihoods"]["z1"].shape == (1, 128, 1, 1) and bool(torch.isfinite(tiny["x2_hat"]).all())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 6e-2)], ids=["f32", "bf16"])
def test_independent_en_training_gradients_match_oracle_autograd(dtype, tol):
    """Stage 2 trains the enhancement net with HSIC frozen (newnet1.py:272-311, 1278-1300; newtrain6_real.py): the gradient of an MSE
    loss w.r.t. every Independent_EN parameter on the HIP path -- bf16: forward and data gradients of the eighteen 32 -> 32 convs per
    view on the register-resident-weight kernel -- against torch autograd through the CPU oracle."""
    import hesic_amd
    from hesic_amd import models
    from oracle import hesic_oracle as O
    from test_oracle_golden import _en_params
    hesic_amd.set_compute_dtype(dtype)
    P = _en_params()
    net = models.Independent_EN()
    net.load_state_dict(P, strict=True)
    net = net.to(DEV).train()
    x1, x2, Hm = synthetic.stereo_batch(5, 2, 64, 64)
    t1, t2, _ = synthetic.stereo_batch(6, 2, 64, 64)
    out = net(x1.to(DEV), x2.to(DEV), Hm.to(DEV))
    loss = ((out["x1_hat"].float() - t1.to(DEV)) ** 2).mean() + ((out["x2_hat"].float() - t2.to(DEV)) ** 2).mean()
    loss.backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = O.independent_en(Pr, x1, x2, Hm)
    lref = ((ref["x1_hat"] - t1) ** 2).mean() + ((ref["x2_hat"] - t2) ** 2).mean()
    lref.backward()
    assert float(loss) == pytest.approx(float(lref), rel=tol)
    bad = []
    for name, p in net.named_parameters():
        g, r = p.grad.float().cpu(), Pr[name].grad
        rel = float((g - r).norm() / r.norm().clamp_min(1e-12))
        if rel > tol:
            bad.append((name, rel))
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype,tol_loss,tol_gn", [(torch.float32, 2e-3, 2e-2), (torch.bfloat16, 3e-2, 0.15)], ids=["f32", "bf16"])
def test_stage2_trainer_follows_the_reference_steps(dtype, tol_loss, tol_gn):
    """``train.Stage2Trainer`` (frozen HSIC in eval mode -> Independent_EN trained on lambda * 255^2 * (MSE1 + MSE2), Adam on the
    enhancement net: ywz/mywork/newtrain6_real.py:154-167) against two steps RECORDED FROM THE REFERENCE's own modules
    (tests/golden/stage2_128.npz): loss and mse of both steps, every gradient norm of step 0, every parameter norm after step 1."""
    import hesic_amd
    from hesic_amd import models
    from hesic_amd.train import Stage2Trainer
    from test_oracle_golden import _en_params
    g = load_golden("stage2_128.npz")
    hesic_amd.set_compute_dtype(dtype)
    hs = models.HSIC()
    synthetic.fill_state_dict_(hs.state_dict())
    hs = hs.to(DEV)
    en = models.Independent_EN()
    en.load_state_dict(_en_params(), strict=True)
    en = en.to(DEV)
    tr = Stage2Trainer(hs, en, lr=1e-4, lmbda=0.0067)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(11, 2, 128, 128))
    bad = []
    for step in range(2):
        c = tr.step(x1, x2, Hm)
        assert float(c["loss"]) == pytest.approx(float(g[f"loss{step}"]), rel=tol_loss), step
        assert float(c["mse_loss"]) == pytest.approx(float(g[f"mse{step}"]), rel=tol_loss), step
        if step == 0:
            for name, p in en.named_parameters():
                gn, ref = float(p.grad.double().norm()), float(g["gn_" + name])
                if abs(gn - ref) > tol_gn * ref:
                    bad.append((name, gn, ref))
    assert not bad, bad[:6]
    for name, p in en.named_parameters():
        # Adam's first steps move every element by ~lr whatever the gradient's size: bf16 gradients flip a few update signs
        assert float(p.detach().double().norm()) == pytest.approx(float(g["pn_" + name]), rel=1e-4 if dtype == torch.float32 else 2e-3), name
    assert not hs.training and all(p.grad is None for p in hs.parameters())          # the compression model stayed frozen


def test_stage2_trainer_under_f16_inference_keeps_both_formats_packs():
    """f16 inference library + bf16 enhancer training (the documented combination): ``Stage2Trainer`` switches the 16-bit format twice per
    step.  A pure library switch moves only the format half of the pack-cache epoch, so the frozen HSIC weights are packed once per
    format -- not on every step -- and ``invalidate_weight_cache()`` still drops both."""
    import hesic_amd
    from hesic_amd import functional as Fn, models
    from hesic_amd.train import Stage2Trainer
    hesic_amd.set_compute_dtype(torch.float16)
    hs = models.HSIC()
    synthetic.fill_state_dict_(hs.state_dict())
    hs = hs.to(DEV)
    en = models.Independent_EN().to(DEV)
    tr = Stage2Trainer(hs, en, lr=1e-4, lmbda=0.0067)
    assert tr.train_dtype == torch.bfloat16
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(11, 2, 128, 128))
    c0 = tr.step(x1, x2, Hm)
    assert Fn.compute_dtype() == torch.float16 and math.isfinite(float(c0["loss"]))
    packs = {id(v[1]) for m in hs.modules() for v in getattr(getattr(m, "_packer", None), "_cache", {}).values()}
    assert packs, "no packed conv weights cached after a step"
    epoch = Fn._cache_epoch
    c1 = tr.step(x1, x2, Hm)
    assert Fn._cache_epoch == epoch and Fn.compute_dtype() == torch.float16
    packs1 = {id(v[1]) for m in hs.modules() for v in getattr(getattr(m, "_packer", None), "_cache", {}).values()}
    assert packs <= packs1, "a frozen weight was repacked by the format switch"
    assert float(c1["loss"]) < float(c0["loss"]) * 1.5
    Fn.invalidate_weight_cache()
    assert Fn._cache_epoch != epoch and Fn._cache_epoch[0] == epoch[0]
