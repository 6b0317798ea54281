"""The hand-over logic of the drop-in modules (hesic_amd/handover.py) without a GPU: the launches are replaced by the oracle's CPU
functions (tests/cpu_backend.py), the module calls, Carriers, recorded ATen operators and fall-backs are the product's.  The forward in
the REFERENCE's call order (hesic_amd/path_a.py; with /root/reference present also the reference's own newnet1.py / newnet1_joint.py)
must equal the oracle's forward on the same weights."""
import os

import pytest
import torch

import hesic_amd
from hesic_amd import handover, models, path_a, synthetic
from oracle import hesic_oracle as O

from cpu_backend import emulate


def _net(kind):
    hesic_amd.set_compute_dtype(torch.float32)
    net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    synthetic.fill_state_dict_(net.state_dict())
    return net.eval(), {k: v.clone() for k, v in net.state_dict().items()}


def _close(a, b, tol=2e-5):
    a, b = a.float(), b.float()
    return float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_reference_call_order_over_deferred_modules_equals_the_oracle(kind):
    net, P = _net(kind)
    x1, x2, Hm = synthetic.stereo_batch(0, 1, 64, 64)
    with torch.no_grad():
        ref = (O.hsic_forward if kind == "hsic" else O.hsic_joint_forward)(P, x1, x2, Hm)
        with emulate() as calls:
            out = path_a.FWD[kind](net, x1, x2, Hm)
            log = list(calls)
    for k in ("x1_hat", "x2_hat"):
        assert type(out[k]) is torch.Tensor and _close(out[k], ref[k]), k
    for k in ("y1_hat", "y2_hat"):
        assert torch.equal(handover.plain(out[k]).float(), ref[k]), k
    for k, v in out["likelihoods"].items():
        assert _close(v, ref["likelihoods"][k], 1e-4), k
    # the forms the module calls took: every 128-channel conv -> (I)GDN pair of the three analysis passes and the two synthesis stacks
    # went out as ONE call (3 x 3 + 2 x 3), the two 6 <-> 3 stages cat-free with their 3-channel (I)GDN on the same call
    assert log.count("HipConv2d.run_gdn") == 9 and log.count("HipConvTranspose2d.run_gdn") == 6, log
    assert log.count("HipConv2d.run_cat") == 1 and log.count("HipConvTranspose2d.run_cat") == 1, log
    if kind == "hsic":
        assert log.count("upsample4_cat") == 1 and log.count("upsample4") == 0      # nn.UpsamplingBilinear2d + torch.cat: one launch
        assert log.count("pooled_linear") == 2
    else:
        assert log.count("HipConv2d.run_into") == 4                                  # h_s's last conv and the masked conv write the cat buffer in place
        assert log.count("copy_into") == 1                                           # y1_hat_w behind them (view 2)


def test_a_foreign_operator_sees_the_plain_tensor():
    net, P = _net("hsic")
    x1, _, _ = synthetic.stereo_batch(0, 1, 64, 64)
    with torch.no_grad(), emulate():
        c = net.encoder1.g_a_conv1(x1)
        assert type(c) is handover.Carrier and tuple(c.shape) == (1, 128, 32, 32) and not c._node.resolved()
        ref = O.conv(x1, P["encoder1.g_a_conv1.weight"], P["encoder1.g_a_conv1.bias"])
        s = c * 2.0 + 1.0                                  # not an operator the hand-over records
        assert type(s) is torch.Tensor and _close(s, ref * 2 + 1)
        assert _close(c.cpu(), ref) and _close(torch.relu(c[:, 3:9]), torch.relu(ref[:, 3:9]))
        r = torch.relu(net.encoder1.g_a_conv1(x1))         # recorded: the activation rides on the launch
        assert type(r) is handover.Carrier and _close(r + 0, torch.relu(ref))
        a = torch.abs(net.encoder1.g_a_conv1(x1))
        assert type(a) is handover.Carrier and _close(a + 0, ref.abs())
        m = net.encoder1.g_a_conv1(x1)
        m.mul_(3.0)                                        # a foreign in-place operator: the Carrier stands for the written tensor afterwards
        assert _close(m + 0, ref * 3)
        lo, hi = net.encoder1.g_a_conv1(x1).chunk(2, 1)
        assert _close(lo + 0, ref[:, :64]) and _close(hi + 0, ref[:, 64:])


def test_training_mode_and_grad_mode_do_not_defer():
    net, _ = _net("hsic")
    x1, _, _ = synthetic.stereo_batch(0, 1, 64, 64)
    with emulate():
        assert not handover.active(x1)                      # grad mode on
        with torch.no_grad():
            assert handover.active(x1)
    assert not handover.active(x1)                          # CPU tensor outside the emulation


REF = "/root/reference/ywz/mywork"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this machine")
@pytest.mark.parametrize("which", [0, 1], ids=["newnet1.HSIC", "newnet1_joint.HSIC"])
def test_the_reference_model_file_itself_runs_over_the_deferred_modules(which):
    """The REAL newnet1.py / newnet1_joint.py forward (container only), executed line by line over this package's modules with the
    launches emulated: identical to the oracle, i.e. the hand-over is transparent to the reference's own code (its attribute
    assignments, its ``spatial_pool2d`` Python loops, its ``kornia`` calls)."""
    from test_dropin_reference_model import ref_models as _fixture
    gen = _fixture.__wrapped__()
    mods = next(gen)
    try:
        hesic_amd.set_compute_dtype(torch.float32)
        ref_net = mods[which].HSIC().eval()
        ours, P = _net("hsic" if which == 0 else "joint")
        ref_net.load_state_dict(ours.state_dict(), strict=True)
        x1, x2, Hm = synthetic.stereo_batch(0, 1, 64, 64)
        with torch.no_grad():
            want = (O.hsic_forward if which == 0 else O.hsic_joint_forward)(P, x1, x2, Hm)
            with emulate():
                out = ref_net(x1, x2, Hm)
        for k in ("x1_hat", "x2_hat"):
            assert _close(handover.plain(out[k]), want[k]), k
        for k in ("y1_hat", "y2_hat"):
            assert torch.equal(handover.plain(out[k]).float(), want[k]), k
        for k, v in out["likelihoods"].items():
            assert _close(handover.plain(v), want["likelihoods"][k], 1e-4), k
    finally:
        gen.close()
