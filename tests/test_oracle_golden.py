"""Pins the CPU oracle (oracle/hesic_oracle.py) to golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import math

import numpy as np
import pytest
import torch

from conftest import T, load_golden
from hesic_amd import synthetic
import torch.nn.functional as F
from oracle import hesic_oracle as O

TOL = dict(rtol=2e-5, atol=2e-6)


def close(a, b, rtol=2e-5, atol=2e-6):
    a = a.detach() if torch.is_tensor(a) else T(a)
    b = b.detach() if torch.is_tensor(b) else T(b)
    torch.testing.assert_close(a.float().reshape(b.shape), b.float(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("C", [3, 128])
@pytest.mark.parametrize("inv", [False, True])
def test_gdn_fwd_bwd(ops_golden, C, inv):
    g, t = ops_golden, f"gdn_C{C}_{'inv' if inv else 'fwd'}_"
    x = T(g[t + "x"]).requires_grad_()
    beta, gamma = T(g[t + "beta"]).requires_grad_(), T(g[t + "gamma"]).requires_grad_()
    y = O.gdn(x, beta, gamma, inv)
    y.backward(T(g[t + "gy"]))
    close(y, g[t + "y"])
    close(x.grad, g[t + "dx"], 1e-4, 1e-5)
    close(beta.grad, g[t + "dbeta"], 1e-4, 1e-5)
    close(gamma.grad, g[t + "dgamma"], 1e-4, 1e-5)


def test_lower_bound_rule(ops_golden):
    g = ops_golden
    x = T(g["lb_x"]).requires_grad_()
    y = O.lower_bound(x, 0.11)
    y.backward(T(g["lb_g"]))
    close(y, g["lb_y"], 0, 0)
    close(x.grad, g["lb_dx"], 0, 0)


def _eb_params(g, t, grad=False):
    P = {}
    for k in g:
        if k.startswith(t + "p_") and not k.endswith("target"):
            v = T(g[k])
            P["eb." + k[len(t) + 2:]] = v.requires_grad_() if grad else v
    return P


@pytest.mark.parametrize("C", [8, 128])
def test_entropy_bottleneck(ops_golden, C):
    g, t = ops_golden, f"eb_C{C}_"
    P = _eb_params(g, t, grad=True)
    x = T(g[t + "x"]).requires_grad_()
    zh, lik = O.eb_forward(P, "eb.", x, training=False)
    assert torch.equal(zh.detach(), T(g[t + "eval_xhat"]))          # integer + median: exact
    close(lik, g[t + "eval_lik"], 1e-4, 1e-9)
    (lik * T(g[t + "g_lik"])).sum().backward()
    for k, p in P.items():
        close(p.grad if p.grad is not None else torch.zeros_like(p), g[t + "eval_d_" + k[3:]], 2e-3, 1e-6)
    # training mode with the injected noise
    P = _eb_params(g, t, grad=True)
    x = T(g[t + "x"]).requires_grad_()
    B, _, H, W = x.shape
    noise = T(g[t + "noise"]).reshape(C, H, W, B).permute(3, 0, 1, 2)
    zt, lt = O.eb_forward(P, "eb.", x, training=True, noise=noise)
    close(zt, g[t + "train_xhat"], 0, 1e-6)
    close(lt, g[t + "train_lik"], 1e-4, 1e-9)
    ((lt * T(g[t + "g_lik"])).sum() + (zt * T(g[t + "g_xhat"])).sum()).backward()
    close(x.grad, g[t + "train_dx"], 2e-3, 1e-6)
    for k, p in P.items():
        close(p.grad if p.grad is not None else torch.zeros_like(p), g[t + "train_d_" + k[3:]], 2e-3, 1e-6)
    # aux loss
    P = _eb_params(g, t, grad=True)
    loss = O.eb_aux_loss(P, "eb.")
    loss.backward()
    close(loss, g[t + "aux_loss"], 1e-5, 1e-4)
    close(P["eb.quantiles"].grad, g[t + "aux_dquantiles"], 1e-4, 1e-6)


def test_gmm(ops_golden):
    g = ops_golden
    ins = [T(g[k]).requires_grad_() for k in ("gmm_y", "gmm_scales", "gmm_means", "gmm_weights")]
    yh, lik = O.gmm_forward(*ins, K=5)
    assert torch.equal(yh.detach(), T(g["gmm_eval_yhat"]))
    assert torch.equal(O.quantize(ins[0].detach(), "symbols"), T(g["gmm_symbols"]))
    close(lik, g["gmm_eval_lik"], 1e-5, 1e-9)
    assert float(lik.min()) == pytest.approx(1e-9, rel=1e-6)          # the bound is hit by construction
    (lik * T(g["gmm_g_lik"])).sum().backward()
    close(ins[1].grad, g["gmm_eval_dscales"], 1e-3, 1e-6)
    close(ins[2].grad, g["gmm_eval_dmeans"], 1e-3, 1e-6)
    close(ins[3].grad, g["gmm_eval_dweights"], 1e-3, 1e-6)
    ins = [T(g[k]).requires_grad_() for k in ("gmm_y", "gmm_scales", "gmm_means", "gmm_weights")]
    yh, lik = O.gmm_forward(*ins, K=5, training=True, noise=T(g["gmm_noise"]))
    close(yh, g["gmm_train_yhat"], 0, 1e-6)
    close(lik, g["gmm_train_lik"], 1e-5, 1e-9)
    ((lik * T(g["gmm_g_lik"])).sum() + (yh * T(g["gmm_g_yhat"])).sum()).backward()
    for t_, k in zip(ins, ("dy", "dscales", "dmeans", "dweights")):
        close(t_.grad, g["gmm_train_" + k], 1e-3, 1e-6)


def test_gaussian_conditional(ops_golden):
    g = ops_golden
    y, sc, mu = T(g["gmm_y"]), T(g["gmm_scales"])[:, :16].clone(), T(g["gmm_means"])[:, :16].clone()
    ins = [t.requires_grad_() for t in (y, sc, mu)]
    yh, lik = O.gc_forward(*ins)
    close(yh, g["gc_eval_yhat"], 0, 0)
    assert torch.equal(O.quantize(y.detach(), "symbols", mu.detach()), T(g["gc_symbols"]))
    close(lik, g["gc_eval_lik"], 1e-5, 1e-9)
    (lik * T(g["gmm_g_lik"])).sum().backward()
    close(ins[1].grad, g["gc_eval_dscales"], 1e-3, 1e-6)
    close(ins[2].grad, g["gc_eval_dmeans"], 1e-3, 1e-6)
    ins = [T(g["gmm_y"]).requires_grad_(), sc.detach().clone().requires_grad_(), mu.detach().clone().requires_grad_()]
    yh, lik = O.gc_forward(*ins, training=True, noise=T(g["gmm_noise"]))
    ((lik * T(g["gmm_g_lik"])).sum() + (yh * T(g["gmm_g_yhat"])).sum()).backward()
    close(lik, g["gc_train_lik"], 1e-5, 1e-9)
    for t_, k in zip(ins, ("dy", "dscales", "dmeans")):
        close(t_.grad, g["gc_train_" + k], 1e-3, 1e-6)


@pytest.mark.parametrize("mt", ["A", "B"])
def test_masked_conv(ops_golden, mt):
    g = ops_golden
    y = O.masked_conv(T(g[f"mc{mt}_x"]), T(g[f"mc{mt}_w"]), T(g[f"mc{mt}_b"]), mt)
    close(y, g[f"mc{mt}_y"], 1e-5, 1e-6)


@pytest.mark.parametrize("tag,stride,tr", [("c5s2", 2, 0), ("c5s1", 1, 0), ("c3s1", 1, 0), ("d5s2", 2, 1),
                                           ("d5s1", 1, 1), ("c5s2_32", 2, 0), ("d5s2_32", 2, 1)])
def test_conv_factories(ops_golden, tag, stride, tr):
    g = ops_golden
    x, w, b = (T(g[f"{tag}_{k}"]).requires_grad_() for k in ("x", "w", "b"))
    y = (O.deconv if tr else O.conv)(x, w, b, stride)
    close(y, g[tag + "_y"], 1e-5, 1e-6)
    y.backward(T(g[tag + "_gy"]))
    close(x.grad, g[tag + "_dx"], 1e-4, 1e-5)
    close(w.grad, g[tag + "_dw"], 1e-4, 1e-5)
    close(b.grad, g[tag + "_db"], 1e-4, 1e-5)


def test_gmm_hyper_heads(ops_golden):
    g = ops_golden
    P1 = {k[len("hy1_p_"):]: T(v) for k, v in g.items() if k.startswith("hy1_p_")}
    s, m, w = O.gmm_hyper_y1(P1, T(g["hy1_z"]), K=3, M=6)
    close(s, g["hy1_sigma"], 1e-4, 1e-5), close(m, g["hy1_means"], 1e-4, 1e-5), close(w, g["hy1_weights"], 1e-4, 1e-6)
    P2 = {k[len("hy2_p_"):]: T(v) for k, v in g.items() if k.startswith("hy2_p_")}
    close(O.upsample_bilinear_x4(T(g["hy1_z"])), g["hy2_up"], 1e-5, 1e-6)
    s, m, w = O.gmm_hyper_y2(P2, T(g["hy1_z"]), T(g["hy2_y1"]), K=3, M=6)
    close(s, g["hy2_sigma"], 1e-4, 1e-5), close(m, g["hy2_means"], 1e-4, 1e-5), close(w, g["hy2_weights"], 1e-4, 1e-6)


@pytest.mark.parametrize("ac", [True, False])
def test_warp(warp_golden, ac):
    """kornia is third party and absent: the golden comes from the normalised-grid restatement in
    make_golden.py, the oracle uses the pixel-space inverse map -- two routes, same published maths."""
    g = warp_golden
    src = T(g["src"]).requires_grad_()
    out = O.warp_perspective(src, T(g["H"]), (24, 32), align_corners=ac)
    close(out, g[f"out_ac{int(ac)}"], 1e-4, 2e-5)
    out.backward(T(g["g"]))
    close(src.grad, g[f"dsrc_ac{int(ac)}"], 1e-3, 5e-5)


def _model_params(kind):
    """Full-size deterministic HESIC / HESIC+ parameters, keyed like the reference state-dict."""
    import os
    from conftest import GOLDEN
    P = {}
    with open(os.path.join(GOLDEN, f"{kind}_state_keys.txt")) as f:
        for line in f:
            parts = line.split()
            shape = tuple(int(s) for s in parts[1:])
            P[parts[0]] = torch.zeros(shape)
    synthetic.init_reference_defaults_(P)
    synthetic.fill_state_dict_(P)
    return P


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("size,batch", [(64, 2), (256, 1)])
def test_whole_model_forward(kind, size, batch):
    g = load_golden(f"{kind}_{size}.npz")
    P = _model_params(kind)
    x1, x2, Hm = synthetic.stereo_batch(0, batch, size, size)
    with torch.no_grad():
        out = (O.hsic_forward if kind == "hsic" else O.hsic_joint_forward)(P, x1, x2, Hm)
    m = O.metrics(out, x1, x2)
    # integer latents: allow a handful of rounding-boundary flips from summation-order noise
    for k in ("y1_hat", "y2_hat"):
        diff = (out[k].to(torch.int16) != T(g[k])).float().mean()
        assert float(diff) < 2e-4, (k, float(diff))
    for k in ("y1", "y2", "z1", "z2"):
        assert m["bits"][k] == pytest.approx(float(g["bits_" + k]), rel=1e-3)
    assert m["mse1"] == pytest.approx(float(g["mse1"]), rel=1e-3)
    assert m["mse2"] == pytest.approx(float(g["mse2"]), rel=1e-3)
    if size == 64:
        close(out["x1_hat"], g["x1_hat"], 1e-3, 1e-4)
        close(out["x2_hat"], g["x2_hat"], 1e-3, 1e-4)
        close(out["likelihoods"]["z1"], g["lik_z1"], 1e-3, 1e-8)


def test_warp_against_a_third_party_sampler():
    """kornia is absent, so the warp cannot be pinned by a run of it.  What CAN be pinned by code the builder did
    not write: the align_corners=True semantics -- destination pixel p samples the source at H^-1 p, bilinear, zeros outside (what
    kornia >= 0.5 and cv2.warpPerspective document) -- against scipy.ndimage.map_coordinates(order=1, mode="constant")."""
    import numpy as np
    from scipy.ndimage import map_coordinates
    g = torch.Generator().manual_seed(11)
    src = torch.rand(2, 3, 20, 28, generator=g)
    Hs = []
    for b in range(2):
        H = torch.eye(3)
        H[:2, :] += (torch.rand(2, 3, generator=g) - 0.5) * torch.tensor([0.2, 0.2, 6.0])
        H[2, :2] = (torch.rand(2, generator=g) - 0.5) * 4e-3
        Hs.append(H)
    Hm = torch.stack(Hs)
    Ho, Wo = 20, 28
    out = O.warp_perspective(src, Hm, (Ho, Wo), align_corners=True).numpy()
    ys, xs = np.meshgrid(np.arange(Ho, dtype=np.float64), np.arange(Wo, dtype=np.float64), indexing="ij")
    for b in range(2):
        s = np.linalg.inv(Hm[b].double().numpy()) @ np.stack([xs.ravel(), ys.ravel(), np.ones(Ho * Wo)])
        sx, sy = s[0] / s[2], s[1] / s[2]
        for c in range(3):
            ref = map_coordinates(src[b, c].double().numpy(), [sy, sx], order=1, mode="constant", cval=0.0).reshape(Ho, Wo)
            inside = ((sx >= 0) & (sx <= 27) & (sy >= 0) & (sy <= 19)).reshape(Ho, Wo)      # scipy drops the half-covered border taps
            assert inside.mean() > 0.5
            np.testing.assert_allclose(out[b, c][inside], ref[inside], rtol=0, atol=2e-6)


@pytest.mark.parametrize("kind", ["hsic", "joint"])
@pytest.mark.parametrize("align", [True, False], ids=["ac1", "ac0"])
def test_whole_model_forward_non_square_both_warp_conventions(kind, align):
    """Round 2: 256 x 320 (the class of BASELINE config C5's padded 896 x 1088) under align_corners=True (kornia >= 0.5) and
    the legacy align_corners=False sampling (kornia 0.4.x, what the reference's pinned torch 1.6.0 era implies): the oracle's
    warp restatement inside the whole model against the reference run with the matching convention."""
    g = load_golden(f"{kind}_256x320{'' if align else '_ac0'}.npz")
    P = _model_params(kind)
    x1, x2, Hm = synthetic.stereo_batch(0, 1, 256, 320)
    with torch.no_grad():
        out = (O.hsic_forward if kind == "hsic" else O.hsic_joint_forward)(P, x1, x2, Hm, align_corners=align)
    m = O.metrics(out, x1, x2)
    for k in ("y1_hat", "y2_hat"):
        assert float((out[k].to(torch.int16) != T(g[k])).float().mean()) < 2e-4, k
    for k in ("y1", "y2", "z1", "z2"):
        assert m["bits"][k] == pytest.approx(float(g["bits_" + k]), rel=1e-3)
    assert m["mse1"] == pytest.approx(float(g["mse1"]), rel=1e-3) and m["mse2"] == pytest.approx(float(g["mse2"]), rel=1e-3)
    if not align:       # the two conventions really differ on this input (view 2 sees the warped view 1)
        g1 = load_golden(f"{kind}_256x320.npz")
        assert abs(float(g1["bits_y2"]) - float(g["bits_y2"])) > 1e-4 * float(g["bits_y2"])


def _en_params():
    import os
    from conftest import GOLDEN
    P = {}
    with open(os.path.join(GOLDEN, "en_state_keys.txt")) as f:
        for line in f:
            parts = line.split()
            shape = tuple(int(v) for v in parts[1:])
            fan = shape[1] * 9 if len(shape) == 4 else 1
            a = (3.0 / fan) ** 0.5 if len(shape) == 4 else 0.05
            P[parts[0]] = synthetic._uniform("en." + parts[0], shape, -a, a)
    return P


def test_independent_en():
    """the enhancement stage against the reference-generated golden."""
    g = load_golden("en_64.npz")
    x1, x2, Hm = synthetic.stereo_batch(5, 2, 64, 64)
    with torch.no_grad():
        out = O.independent_en(_en_params(), x1, x2, Hm)
    close(out["x1_hat"], g["x1_hat"], 1e-4, 1e-5)
    close(out["x2_hat"], g["x2_hat"], 1e-4, 1e-5)


def test_stage2_training_steps_match_the_reference_run():
    """Two stage-2 steps (ywz/mywork/newtrain6_real.py:154-167: frozen HSIC in eval mode, Independent_EN trained on
    lambda * 255^2 * (MSE1 + MSE2) with Adam) recorded from the reference's own modules -- ``stage2_128.npz``, made by
    ``make_golden.py stage2`` -- replayed through the oracle: loss / mse per step, every gradient norm of step 0, every parameter norm
    after step 1."""
    g = load_golden("stage2_128.npz")
    from hesic_amd import models
    hs = models.HSIC()
    synthetic.fill_state_dict_(hs.state_dict())
    P = {k: v.clone() for k, v in hs.state_dict().items()}
    E = {k: v.clone().requires_grad_(True) for k, v in _en_params().items()}
    names = [k for k in E]
    opt = torch.optim.Adam([E[k] for k in names], lr=1e-4)
    x1, x2, Hm = synthetic.stereo_batch(11, 2, 128, 128)
    for step in range(2):
        opt.zero_grad()
        with torch.no_grad():
            out = O.hsic_forward(P, x1, x2, Hm)
        out2 = O.independent_en(E, out["x1_hat"], out["x2_hat"], Hm)
        mse = F.mse_loss(out2["x1_hat"], x1) + F.mse_loss(out2["x2_hat"], x2)
        loss = 0.0067 * 255 ** 2 * mse
        loss.backward()
        assert float(loss) == pytest.approx(float(g[f"loss{step}"]), rel=2e-4)
        assert float(mse) == pytest.approx(float(g[f"mse{step}"]), rel=2e-4)
        if step == 0:
            for k in names:
                assert float(E[k].grad.double().norm()) == pytest.approx(float(g["gn_" + k]), rel=2e-3), k
        opt.step()
    for k in names:
        assert float(E[k].detach().double().norm()) == pytest.approx(float(g["pn_" + k]), rel=1e-5), k
