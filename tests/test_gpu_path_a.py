"""INTEGRATION.md path A on hardware: the reference's forward, replayed in ITS call order with nothing but module ``__call__``s of
the drop-in package (``hesic_amd/compressai``: ``conv()`` / ``deconv()`` modules, ``GDN``, ``MaskedConv2d``, the entropy models),
plain ``nn.Sequential`` / ``nn.ReLU`` / ``nn.LeakyReLU`` / ``nn.UpsamplingBilinear2d``, ``torch.cat`` / ``torch.abs`` / ``softmax``
and the kornia-shaped ``warp_perspective`` -- what the reference's own ``newnet1{,_joint}.py`` executes after ``import hesic_amd``
(ywz/mywork/newnet1.py:590-601, :615-624, :641-655, :676-692, :433-437, :441-453, :496-512, :562-577, :724-783;
newnet1_joint.py:675-753).  Every module's output goes to the next module as it is, like in the reference; no fused ``run_*`` entry point, no
``_forward_eval`` schedule; at inference the modules hand over among themselves (hesic_amd/handover.py).  The reference .py files do not travel to the GPU box, so the call order is restated here (the
container-only ``test_dropin_reference_model.py`` loads the real files against the same package).

Checked against the reference-recorded goldens (64 x 64 full tensors, 256 x 256 metrics), eval and training mode, fp32 and
bf16 storage."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import T, load_golden
from hesic_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


from hesic_amd.path_a import FWD, hsic_forward, joint_forward  # noqa: E402,F401  (the restated call order lives next to the product)


def build(kind, dtype, train=False):
    import hesic_amd
    from hesic_amd import models
    hesic_amd.set_compute_dtype(dtype)
    net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.to(DEV)
    return net.train() if train else net.eval()


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    import hesic_amd
    hesic_amd.set_compute_dtype(torch.float32)


def _metrics(out, x1, x2):
    bits = {k: float(torch.log2(v.double()).sum() * -1) for k, v in out["likelihoods"].items()}
    mse1 = float(((out["x1_hat"].double() - x1.double()) ** 2).mean())
    mse2 = float(((out["x2_hat"].double() - x2.double()) ** 2).mean())
    return bits, mse1, mse2


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("size,batch", [(64, 2), (256, 1)])
def test_path_a_eval_fp32_matches_reference_golden(kind, size, batch):
    g = load_golden(f"{kind}_{size}.npz")
    net = build(kind, torch.float32)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, batch, size, size))
    with torch.no_grad():
        out = FWD[kind](net, x1, x2, Hm)
    bits, mse1, mse2 = _metrics(out, x1, x2)
    for k in ("y1_hat", "y2_hat"):
        assert float((out[k].cpu().to(torch.int16) != T(g[k])).float().mean()) < 2e-4, k
    for k in ("y1", "y2", "z1", "z2"):
        assert bits[k] == pytest.approx(float(g["bits_" + k]), rel=1e-3), k
    assert mse1 == pytest.approx(float(g["mse1"]), rel=1e-3) and mse2 == pytest.approx(float(g["mse2"]), rel=1e-3)
    if size == 64:
        torch.testing.assert_close(out["x2_hat"].float().cpu(), T(g["x2_hat"]), rtol=2e-3, atol=2e-4)
        torch.testing.assert_close(out["likelihoods"]["y2"].float().cpu().contiguous(), T(g["lik_y2"]), rtol=5e-3, atol=1e-7)


# Round 6: at inference the modules of the package hand over among themselves (hesic_amd/handover.py: deferred conv -> GDN fusion, hi/lo
# pairs between the analysis layers, fp32 latents into round() and the likelihoods), so the reference's call order holds the SAME bars as
# the fused forward of hesic_amd.models -- test_gpu_baseline_workloads.MODE_BARS -- on the exact BASELINE workloads C2 and C4.
@pytest.mark.parametrize("fmt,analysis", [("f16", "x3"), ("bf16", "x3"), ("f16", "x3c2")], ids=["f16-x3", "bf16-x3", "f16-x3c2"])
@pytest.mark.parametrize("kind,batch", [("hsic", 8), ("joint", 4)], ids=["C2-hesic-b8", "C4-hesicplus-b4"])
def test_path_a_16bit_512_batch_holds_the_fused_forwards_bars(kind, batch, fmt, analysis):
    from hesic_amd import functional as Fn
    from test_gpu_baseline_workloads import MODE_BARS, check_against
    g = load_golden(f"{kind}_512_b{batch}.npz")
    net = build(kind, {"f16": torch.float16, "bf16": torch.bfloat16}[fmt])
    prev = Fn.set_analysis_precision(analysis)
    try:
        bars = dict(MODE_BARS[(fmt, analysis)])
        dbpp_bar = bars.pop("dbpp")
        x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, batch, 512, 512))
        with torch.no_grad():
            out = FWD[kind](net, x1, x2, Hm)
            meas = check_against(g, out, x1, x2, **bars)
            assert meas["dbpp_abs_set_mean"] < dbpp_bar and meas["dbpp_rel"] < 1e-3, meas
            if (fmt, analysis, kind) == ("f16", "x3", "hsic"):
                assert meas["dbpp_abs"] < 1e-3, meas                  # every pair of C2 under the absolute bar, as for the fused forward
            # and against the fused forward itself: the transmitted latents of the two routes are the same integers (same kernels, same
            # order on the analysis side); reconstructions / likelihoods within the 16-bit storage noise of the differently grouped hyper-synthesis
            twin = net(x1, x2, Hm)
        for k in ("y1_hat", "y2_hat"):
            assert float((out[k].float() != twin[k].float()).float().mean()) <= 2e-5, k
        for k in ("x1_hat", "x2_hat"):
            assert float((out[k].float() - twin[k].float()).abs().max()) <= 2e-3, k
    finally:
        Fn.set_analysis_precision(prev)


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_path_a_single_operand_mode_and_the_switch_back(kind, monkeypatch):
    """"x1" (single 16-bit operands everywhere) through the module calls stays inside the single-operand bars; with the hand-over switched
    off (``HESIC_NO_HANDOVER``: every module launches when called, 16-bit tensors at every boundary -- round 5's path A) likewise."""
    from hesic_amd import functional as Fn, handover
    g = load_golden(f"{kind}_256.npz")
    net = build(kind, torch.float16)
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 1, 256, 256))
    total = sum(float(g["bits_" + k]) for k in ("y1", "y2", "z1", "z2"))
    prev = Fn.set_analysis_precision("x1")
    try:
        for off in (False, True):
            monkeypatch.setattr(handover, "ENABLED", not off)
            with torch.no_grad():
                out = FWD[kind](net, x1, x2, Hm)
            bits, mse1, mse2 = _metrics(out, x1, x2)
            meas = {"bits_rel": abs(sum(bits.values()) / total - 1), "mse1_rel": abs(mse1 / float(g["mse1"]) - 1), "mse2_rel": abs(mse2 / float(g["mse2"]) - 1)}
            for k in ("y1_hat", "y2_hat"):
                meas["flips_" + k] = float((out[k].float().cpu().to(torch.int16) != T(g[k])).float().mean())
            print("handover", not off, "measured:", {k: float("%.3g" % v) for k, v in meas.items()})
            assert meas["bits_rel"] <= 2e-3 and max(meas["mse1_rel"], meas["mse2_rel"]) <= 5e-4, meas
            assert max(meas["flips_y1_hat"], meas["flips_y2_hat"]) < 6e-3, meas
    finally:
        Fn.set_analysis_precision(prev)


def test_a_carrier_is_an_ordinary_tensor_to_foreign_code():
    """On the device: a deferred conv output used by code that knows nothing about it (arithmetic, indexing, ``.cpu()``) gives the values the
    plain launch stores; a recorded ``relu`` / ``abs`` / ``chunk`` gives the same values as the ATen operator on that tensor."""
    from hesic_amd import handover
    net = build("hsic", torch.float16)
    x1, _, _ = (t.to(DEV) for t in synthetic.stereo_batch(0, 1, 128, 128))
    conv1, conv2 = net.encoder1.g_a_conv1, net.encoder1.g_a_conv2
    with torch.no_grad():
        ref = conv1.run(x1)
        c = conv1(x1)
        assert type(c) is handover.Carrier and c.dtype == torch.float16 and tuple(c.shape) == tuple(ref.shape) and not c._node.resolved()
        assert c.stride() == ref.stride()
        assert torch.equal(c + 0, ref) and torch.equal(c.cpu(), ref.cpu()) and torch.equal(c[:, 5:9].contiguous(), ref[:, 5:9].contiguous())
        assert torch.equal(torch.relu(conv1(x1)) + 0, torch.relu(ref))
        assert torch.equal(torch.abs(conv1(x1)) + 0, torch.abs(ref))
        lo, hi = conv1(x1).chunk(2, 1)
        assert torch.equal(hi + 0, ref[:, 64:])
        # conv -> conv with a recorded activation in between == the two launches with the activation fused
        two = conv2(torch.nn.functional.leaky_relu(conv1(x1)))
        assert torch.equal(two + 0, conv2.run(conv1.run(x1, act=2), act=0))


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_path_a_training_step_matches_the_reference_trace(kind, dtype):
    """Training mode through the plain module calls (autograd through every module's own Function), the entropy models drawing
    their noise themselves -- fed, in the reference's draw order, with the draws its own recorded training step used: loss terms and
    every parameter's gradient norm of the first backward against the reference (row T golden)."""
    from compressai.entropy_models import EntropyModel
    from hesic_amd import functional as Fn
    g = load_golden(f"{kind}_train64.npz")
    x1, x2, Hm = (t.to(DEV) for t in synthetic.stereo_batch(0, 2, 64, 64))
    order = [str(s_) for s_ in g["noise_order"]]
    queue = []
    for k in order:
        if k[0] == "z":      # reference layout (C, 1, B*H*W) with B fastest -> (B, C, 1, 1)
            queue.append(synthetic._uniform(f"noise.{kind}.0.{k}", (128, 1, 2), -0.5, 0.5).reshape(128, 1, 1, 2).permute(3, 0, 1, 2).contiguous())
        else:
            queue.append(synthetic._uniform(f"noise.{kind}.0.{k}", (2, 192, 4, 4), -0.5, 0.5))

    def draw(x):
        n = queue.pop(0)
        assert tuple(n.shape) == tuple(x.shape), (n.shape, x.shape)
        return n.to(x.device, x.dtype)

    net = build(kind, dtype, train=True)
    keep = EntropyModel.__dict__["_noise_like"]
    EntropyModel._noise_like = staticmethod(draw)
    try:
        out = FWD[kind](net, x1, x2, Hm)
    finally:
        EntropyModel._noise_like = keep
    assert not queue
    crit = Fn.rd_loss(out, x1, x2, 0.0067)
    crit["loss"].backward()
    rel, rel_gn = (5e-3, 2e-2) if dtype == torch.float32 else (2e-2, 0.12)      # bf16: the 6 <-> 3 image-side stages see the summed bf16 noise of the whole stack (measured <= 9 %)
    ref = g["trace"][0]
    assert float(crit["loss"]) == pytest.approx(float(ref[0]), rel=rel)
    assert float(crit["bpp_loss"]) == pytest.approx(float(ref[1]), rel=rel)
    assert float(crit["mse_loss"]) == pytest.approx(float(ref[2]), rel=rel)
    bad = []
    for name, p in net.named_parameters():
        live = "gn_live_" + name
        r = float(g[live] if live in g else g["gn_" + name])
        got = float(p.grad.double().norm()) if p.grad is not None else 0.0
        if abs(got - r) > rel_gn * max(r, 1e-6) + 1e-7:
            bad.append((name, got, r))
    assert not bad, bad[:8]
