#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  Nothing of
the reference is copied: the script imports it through an overlay directory in
/tmp (symlinks + its two pybind11 extensions compiled from the sources where
they lie), feeds it the deterministic weights / inputs of
``hesic_amd.synthetic`` and stores inputs + expected outputs as small .npz files.

Third-party modules the reference imports but the image lacks are stubbed:
cv2, torchvision, range_coder (unused by forward) and kornia.  kornia's
``warp_perspective`` is restated from its published definition
(normalise H with (W-1)/2,(H-1)/2 -> invert -> affine grid in [-1,1] ->
``F.grid_sample``): warp parity is therefore "unpinned" (DESIGN.md).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import os
import subprocess
import sys
import sysconfig
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OVERLAY = "/tmp/hesic_ref_overlay"
sys.path.insert(0, ROOT)

from hesic_amd import synthetic  # noqa: E402


# ------------------------------------------------------------------ reference import
def build_overlay():
    pkg = os.path.join(OVERLAY, "compressai")
    os.makedirs(pkg, exist_ok=True)
    for name in os.listdir(os.path.join(REF, "compressai")):
        dst = os.path.join(pkg, name)
        if not os.path.lexists(dst):
            os.symlink(os.path.join(REF, "compressai", name), dst)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    inc = subprocess.check_output([sys.executable, "-m", "pybind11", "--includes"], text=True).split()
    base = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", *inc]
    jobs = {
        "ans": [f"-I{REF}/third_party/ryg_rans", f"-I{REF}/compressai/cpp_exts/rans",
                f"{REF}/compressai/cpp_exts/rans/rans_interface.cpp"],
        "_CXX": [f"{REF}/compressai/cpp_exts/ops/ops.cpp"],
    }
    for mod, args in jobs.items():
        out = os.path.join(pkg, mod + ext)
        if not os.path.exists(out):
            subprocess.check_call(base + args + ["-o", out])


def _norm_mat(h, w):
    return torch.tensor([[2.0 / (w - 1), 0, -1.0], [0, 2.0 / (h - 1), -1.0], [0, 0, 1.0]], dtype=torch.float64)


KORNIA_ALIGN = {"align_corners": True}


def kornia_warp_perspective(src, M, dsize, **kw):
    B, C, H, W = src.shape
    Ho, Wo = dsize
    dst_norm_src_norm = _norm_mat(Ho, Wo) @ M.double() @ torch.linalg.inv(_norm_mat(H, W))
    src_norm_dst_norm = torch.linalg.inv(dst_norm_src_norm)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, Ho, dtype=torch.float64),
                            torch.linspace(-1, 1, Wo, dtype=torch.float64), indexing="ij")
    pts = torch.stack([xs, ys, torch.ones_like(xs)], -1).reshape(1, -1, 3)
    t = pts @ src_norm_dst_norm.transpose(1, 2)
    grid = (t[..., :2] / t[..., 2:]).reshape(B, Ho, Wo, 2).to(src.dtype)
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros",
                         align_corners=KORNIA_ALIGN["align_corners"])


def install_stubs():
    sys.modules["cv2"] = types.ModuleType("cv2")
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    for n in ("Compose", "ToTensor", "Normalize", "RandomCrop", "CenterCrop"):
        setattr(tvt, n, lambda *a, **k: None)
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    rc = types.ModuleType("range_coder")
    rc.RangeEncoder = rc.RangeDecoder = rc.prob_to_cum_freq = object
    sys.modules["range_coder"] = rc
    kn = types.ModuleType("kornia")
    kn.warp_perspective = kornia_warp_perspective
    sys.modules["kornia"] = kn


def import_reference():
    build_overlay()
    install_stubs()
    sys.path[:0] = [OVERLAY, os.path.join(REF, "ywz", "mywork")]
    import compressai  # noqa: F401
    import newnet1
    import newnet1_joint
    return newnet1, newnet1_joint


class NoiseQueue:
    """Replaces EntropyModel._get_noise_cached so the U(-1/2,1/2) draws are known tensors."""

    def __init__(self):
        self.items, self.used = [], []

    def __call__(self, module, x):
        n = self.items.pop(0)
        assert n.shape == x.shape, (n.shape, x.shape)
        self.used.append(n)
        return n


def det_noise(name, shape):
    return synthetic._uniform("noise." + name, shape, -0.5, 0.5)


def det_tensor(name, shape, lo=-1.0, hi=1.0):
    return synthetic._uniform("input." + name, shape, lo, hi)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


# ------------------------------------------------------------------------ fixtures
def fx_ops(newnet1):
    from compressai.entropy_models import EntropyBottleneck, GaussianConditional, GaussianMixtureConditional, EntropyModel
    from compressai.layers import GDN, MaskedConv2d
    from compressai.models.utils import conv, deconv
    from compressai.ops import LowerBound
    A = {}

    # G1 GDN / IGDN fwd+bwd
    for C in (3, 128):
        for inv in (False, True):
            m = GDN(C, inverse=inv)
            synthetic.fill_state_dict_(m.state_dict(), salt=C + inv)
            # push a few gamma/beta entries under their bounds to exercise LowerBound's grad rule
            with torch.no_grad():
                m.gamma.view(-1)[::7] = 1e-7
                m.beta[0] = 1e-4
            x = det_tensor(f"gdn{C}{inv}", (2, C, 8, 8), -3, 3).requires_grad_()
            gy = det_tensor(f"gdn_g{C}{inv}", (2, C, 8, 8))
            y = m(x)
            y.backward(gy)
            t = f"gdn_C{C}_{'inv' if inv else 'fwd'}_"
            A.update({t + "beta": m.beta, t + "gamma": m.gamma, t + "x": x, t + "gy": gy, t + "y": y,
                      t + "dx": x.grad, t + "dbeta": m.beta.grad, t + "dgamma": m.gamma.grad})

    # LowerBound value + grad rule
    lb = LowerBound(0.11)
    x = det_tensor("lb", (64,), -0.5, 0.8).requires_grad_()
    g = det_tensor("lbg", (64,))
    yb = lb(x)
    yb.backward(g)
    A.update(lb_x=x, lb_g=g, lb_y=yb, lb_dx=x.grad)

    # G2 EntropyBottleneck
    nq = NoiseQueue()
    EntropyModel._get_noise_cached = lambda self, x: nq(self, x)
    for C in (8, 128):
        eb = EntropyBottleneck(C)
        synthetic.fill_state_dict_(eb.state_dict(), salt=C)
        t = f"eb_C{C}_"
        for k, v in eb.state_dict().items():
            if v.numel() and v.dtype.is_floating_point and "bound" not in k:
                A[t + "p_" + k] = v.clone()
        x = det_tensor(f"eb{C}", (2, C, 4, 4), -6, 6)
        x.view(-1)[:5] = torch.tensor([0.5, 1.5, -0.5, -2.5, 40.0])   # ties + a tail value
        eb.eval()
        xe = x.clone().requires_grad_()
        zh, lik = eb(xe)
        gl = det_tensor(f"ebg{C}", lik.shape)
        (lik * gl).sum().backward()
        A.update({t + "x": x, t + "eval_xhat": zh, t + "eval_lik": lik, t + "g_lik": gl, t + "eval_dx": xe.grad})
        for n_, p_ in eb.named_parameters():
            A[t + "eval_d_" + n_] = p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_)
        eb.zero_grad()
        eb.train()
        noise = det_noise(f"eb{C}", (C, 1, 2 * 4 * 4))
        nq.items.append(noise)
        xt = x.clone().requires_grad_()
        zt, lt = eb(xt)
        gx = det_tensor(f"ebgx{C}", zt.shape)
        ((lt * gl).sum() + (zt * gx).sum()).backward()
        A.update({t + "noise": noise, t + "train_xhat": zt, t + "train_lik": lt, t + "g_xhat": gx,
                  t + "train_dx": xt.grad})
        for n_, p_ in eb.named_parameters():
            A[t + "train_d_" + n_] = p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_)
        eb.zero_grad()
        loss = eb.loss()
        loss.backward()
        A[t + "aux_loss"] = loss
        A[t + "aux_dquantiles"] = eb.quantiles.grad
        eb.update(force=True)
        A.update({t + "offset": eb._offset, t + "quantized_cdf": eb._quantized_cdf, t + "cdf_length": eb._cdf_length})
        eb.eval()
        # batch 1 only: the reference's decompress rejects medians (1,C,1,1) against N>1 indexes
        strings = eb.compress(x[:1])
        A[t + "string0"] = np.frombuffer(strings[0], dtype=np.uint8)
        A[t + "decompressed0"] = eb.decompress(strings, x.shape[-2:])

    # G3 GMM fwd+bwd with symbols
    K, M = 5, 16
    gm = GaussianMixtureConditional(K=K)
    y = det_tensor("gmm_y", (2, M, 4, 4), -6, 6)
    y.view(-1)[:4] = torch.tensor([0.5, 1.5, -2.5, 60.0])       # ties; 60 drives lik under 1e-9
    sc = det_tensor("gmm_s", (2, M * K, 4, 4), 0.02, 3.0)      # includes sigma < 0.11
    mu = det_tensor("gmm_m", (2, M * K, 4, 4), -3, 3)
    w = torch.softmax(det_tensor("gmm_w", (2, K, M, 1, 1), -2, 2), 1).reshape(2, K * M, 1, 1)
    ins = [t_.clone().requires_grad_() for t_ in (y, sc, mu, w)]
    gm.eval()
    yh, lik = gm(*ins)
    gl = det_tensor("gmm_gl", lik.shape)
    (lik * gl).sum().backward()
    A.update(gmm_y=y, gmm_scales=sc, gmm_means=mu, gmm_weights=w, gmm_g_lik=gl, gmm_eval_yhat=yh, gmm_eval_lik=lik,
             gmm_symbols=gm._quantize(y, "symbols"), gmm_eval_dscales=ins[1].grad, gmm_eval_dmeans=ins[2].grad,
             gmm_eval_dweights=ins[3].grad)
    ins = [t_.clone().requires_grad_() for t_ in (y, sc, mu, w)]
    gm.train()
    noise = det_noise("gmm", y.shape)
    nq.items.append(noise)
    yh, lik = gm(*ins)
    gyh = det_tensor("gmm_gyh", yh.shape)
    ((lik * gl).sum() + (yh * gyh).sum()).backward()
    A.update(gmm_noise=noise, gmm_g_yhat=gyh, gmm_train_yhat=yh, gmm_train_lik=lik, gmm_train_dy=ins[0].grad,
             gmm_train_dscales=ins[1].grad, gmm_train_dmeans=ins[2].grad, gmm_train_dweights=ins[3].grad)

    # G4 GaussianConditional with means
    gc = GaussianConditional(None)
    sc1, mu1 = sc[:, :M].clone(), mu[:, :M].clone()
    ins = [t_.clone().requires_grad_() for t_ in (y, sc1, mu1)]
    gc.eval()
    yh, lik = gc(ins[0], ins[1], means=ins[2])
    (lik * gl).sum().backward()
    A.update(gc_eval_yhat=yh, gc_eval_lik=lik, gc_symbols=gc._quantize(y, "symbols", mu1),
             gc_eval_dscales=ins[1].grad, gc_eval_dmeans=ins[2].grad)
    ins = [t_.clone().requires_grad_() for t_ in (y, sc1, mu1)]
    gc.train()
    nq.items.append(noise)
    yh, lik = gc(ins[0], ins[1], means=ins[2])
    ((lik * gl).sum() + (yh * gyh).sum()).backward()
    A.update(gc_train_yhat=yh, gc_train_lik=lik, gc_train_dy=ins[0].grad, gc_train_dscales=ins[1].grad,
             gc_train_dmeans=ins[2].grad)

    # G5 MaskedConv2d
    for mt in ("A", "B"):
        mc = MaskedConv2d(8, 16, kernel_size=5, padding=2, stride=1, mask_type=mt)
        synthetic.fill_state_dict_(mc.state_dict(), salt=ord(mt))
        w0 = mc.weight.detach().clone()
        x = det_tensor("mc" + mt, (2, 8, 6, 7))
        A.update({f"mc{mt}_w": w0, f"mc{mt}_b": mc.bias, f"mc{mt}_x": x, f"mc{mt}_mask": mc.mask, f"mc{mt}_y": mc(x)})

    # G6 conv / deconv factories fwd + bwd (odd sizes included)
    cases = [("c5s2", conv, 8, 16, 5, 2, (2, 8, 9, 12)), ("c5s1", conv, 6, 3, 5, 1, (1, 6, 7, 10)),
             ("c3s1", conv, 8, 8, 3, 1, (2, 8, 5, 6)), ("d5s2", deconv, 16, 8, 5, 2, (2, 16, 5, 7)),
             ("d5s1", deconv, 6, 3, 5, 1, (1, 6, 7, 10)), ("c5s2_32", conv, 32, 64, 5, 2, (1, 32, 16, 16)),
             ("d5s2_32", deconv, 64, 32, 5, 2, (1, 64, 8, 8))]
    for tag, fac, ci, co, k, s, shp in cases:
        m = fac(ci, co, kernel_size=k, stride=s)
        sd = {f"{'g_s_conv' if fac is deconv else 'g_a_conv'}.{n_}": v for n_, v in m.state_dict().items()}
        synthetic.fill_state_dict_(sd, salt=len(tag))
        x = det_tensor(tag, shp).requires_grad_()
        yv = m(x)
        gy = det_tensor(tag + "g", yv.shape)
        yv.backward(gy)
        A.update({f"{tag}_w": m.weight, f"{tag}_b": m.bias, f"{tag}_x": x, f"{tag}_y": yv, f"{tag}_gy": gy,
                  f"{tag}_dx": x.grad, f"{tag}_dw": m.weight.grad, f"{tag}_db": m.bias.grad})

    # G7 gmm_hyper_y1 / y2 (pool + softmax layout), small N,M,K
    N_, M_, K_ = 8, 6, 3
    h1 = newnet1.gmm_hyper_y1(N_, M_, K_)
    sd = {"_h_s1." + k: v for k, v in h1.state_dict().items()}
    synthetic.fill_state_dict_(sd, salt=71)
    z = det_tensor("hy1", (2, N_, 2, 3), -3, 3)
    s_, m_, w_ = h1(z)
    for k, v in sd.items():
        A["hy1_p_" + k] = v
    A.update(hy1_z=z, hy1_sigma=s_, hy1_means=m_, hy1_weights=w_)
    h2 = newnet1.gmm_hyper_y2(N_, M_, K_)
    sd = {"_h_s2." + k: v for k, v in h2.state_dict().items()}
    synthetic.fill_state_dict_(sd, salt=72)
    y1 = det_tensor("hy2y", (2, M_, 8, 12), -3, 3)
    s_, m_, w_ = h2(z, y1)
    for k, v in sd.items():
        A["hy2_p_" + k] = v
    A.update(hy2_y1=y1, hy2_sigma=s_, hy2_means=m_, hy2_weights=w_, hy2_up=h2.up_z2)
    npz("ops.npz", **A)


def fx_warp():
    A = {}
    src = det_tensor("warp_src", (3, 3, 24, 32), 0, 1)
    Hs = torch.tensor([
        [[1, 0, 0], [0, 1, 0], [0, 0, 1]],
        [[1, 0, 3.25], [0, 1, -2.5], [0, 0, 1]],
        [[1.02, 0.01, 2.0], [-0.015, 0.97, 1.5], [2e-4, -1e-4, 1]],
    ], dtype=torch.float32)
    g = det_tensor("warp_g", src.shape)
    for ac in (True, False):
        KORNIA_ALIGN["align_corners"] = ac
        s = src.clone().requires_grad_()
        out = kornia_warp_perspective(s, Hs, (24, 32))
        out.backward(g)
        A[f"out_ac{int(ac)}"], A[f"dsrc_ac{int(ac)}"] = out, s.grad
    KORNIA_ALIGN["align_corners"] = True
    npz("warp.npz", src=src, H=Hs, g=g, **A)


def fx_msssim():
    """MS-SSIM golden.  ``pytorch_msssim`` (what ywz/mywork/test3real.py:107-109 calls) is third party, absent from the reference tree
    and the image, and unpinned -- like kornia.  The expected values below come from a route INDEPENDENT of oracle/hesic_oracle.py::ms_ssim:
    fp64 numpy, scipy.ndimage.correlate1d for the Gaussian (cropped to the valid region), reshape-mean for the 2 x 2 pool.  Two routes
    to the same published algorithm, not a run of the package itself."""
    import numpy as np
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(20260930)
    N, C, H, W = 2, 3, 177, 208                      # odd height: exercises the padded pool
    base = rng.random((N, C, H // 8 + 2, W // 8 + 2))
    x = np.clip(np.kron(base, np.ones((8, 8)))[:, :, :H, :W] * 0.8 + 0.1 * rng.random((N, C, H, W)), 0, 1).astype(np.float32)
    y = np.clip(x + rng.normal(0, [[[[0.02]]], [[[0.08]]]], (N, C, H, W)), 0, 1).astype(np.float32)
    xq, yq = np.round(x * 255).astype(np.uint8), np.round(y * 255).astype(np.uint8)         # 8-bit images: a small fixture
    x, y = xq.astype(np.float32) / np.float32(255), yq.astype(np.float32) / np.float32(255)
    co = np.arange(11, dtype=np.float32) - 5
    g = np.exp(-(co ** 2) / np.float32(2 * 1.5 ** 2)).astype(np.float32)
    g = (g / g.sum(dtype=np.float32)).astype(np.float64)

    def blur(t):
        t = correlate1d(correlate1d(t, g, axis=-1, mode="constant"), g, axis=-2, mode="constant")
        return t[..., 5:-5, 5:-5]

    def pool(t):
        h, w = t.shape[-2:]
        t = np.pad(t, ((0, 0), (0, 0), (h % 2, h % 2), (w % 2, w % 2)))       # F.avg_pool2d(padding = s % 2): zeros on both sides
        h2, w2 = (t.shape[-2] // 2) * 2, (t.shape[-1] // 2) * 2
        t = t[..., :h2, :w2]
        return t.reshape(*t.shape[:2], h2 // 2, 2, w2 // 2, 2).mean((3, 5))

    wts = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    a, b = x.astype(np.float64), y.astype(np.float64)
    out, per_scale = np.ones((N, C)), []
    for i, wt in enumerate(wts):
        mx, my = blur(a), blur(b)
        sxx, syy, sxy = blur(a * a) - mx * mx, blur(b * b) - my * my, blur(a * b) - mx * my
        cs = (2 * sxy + C2) / (sxx + syy + C2)
        ss = (2 * mx * my + C1) / (mx * mx + my * my + C1) * cs
        v = np.maximum((cs if i < 4 else ss).mean((2, 3)), 0)
        per_scale.append(np.stack([ss.mean((2, 3)), cs.mean((2, 3))], -1))
        out *= v ** wt
        if i < 4:
            a, b = pool(a), pool(b)
    npz("msssim.npz", x_u8=xq, y_u8=yq, ms_ssim=out.mean(1), per_scale=np.stack(per_scale))      # x = x_u8 / 255 in float32


def _run_model(net, size, batch, training, noise_names, nq, tag):
    x1, x2, Hm = synthetic.stereo_batch(0, batch, size, size)
    noises = {}
    if training:
        net.train()
    else:
        net.eval()
    return x1, x2, Hm, noises


def fx_models(newnet1, newnet1_joint):
    from compressai.entropy_models import EntropyModel
    nq = NoiseQueue()
    EntropyModel._get_noise_cached = lambda self, x: nq(self, x)
    for tag, mod in (("hsic", newnet1), ("joint", newnet1_joint)):
        torch.manual_seed(0)
        net = mod.HSIC()
        synthetic.fill_state_dict_(net.state_dict())
        keys = list(net.state_dict().keys())
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.eval()
        for size, batch in ((64, 2), (256, 1)):
            x1, x2, Hm = synthetic.stereo_batch(0, batch, size, size)
            with torch.no_grad():
                out = net(x1, x2, Hm)
            n, _, h, w = x1.shape
            import math
            bits = {k: float(torch.log(v.double()).sum() / -math.log(2)) for k, v in out["likelihoods"].items()}
            mse1 = float(((out["x1_hat"].double() - x1.double()) ** 2).mean())
            mse2 = float(((out["x2_hat"].double() - x2.double()) ** 2).mean())
            A = dict(H=Hm, bits_y1=bits["y1"], bits_y2=bits["y2"], bits_z1=bits["z1"], bits_z2=bits["z2"],
                     mse1=mse1, mse2=mse2, y1_hat=out["y1_hat"].to(torch.int16), y2_hat=out["y2_hat"].to(torch.int16))
            if size == 64:
                A.update(x1=x1, x2=x2, x1_hat=out["x1_hat"], x2_hat=out["x2_hat"],
                         lik_y1=out["likelihoods"]["y1"], lik_y2=out["likelihoods"]["y2"],
                         lik_z1=out["likelihoods"]["z1"], lik_z2=out["likelihoods"]["z2"])
                if tag == "hsic":
                    A.update(enc1_g1=net.encoder1.g_a_g1[:, :8], sigma1=net._h_s1.sigma[:, ::64],
                             weights1=net._h_s1.weights, weights2=net._h_s2.weights)
            else:
                # pooled reconstructions: enough to catch gross errors without shipping megabytes
                A.update(x1_hat_pool=F.avg_pool2d(out["x1_hat"], 8), x2_hat_pool=F.avg_pool2d(out["x2_hat"], 8))
            npz(f"{tag}_{size}.npz", **A)
            print(tag, size, "bpp_loss", sum(bits.values()) / (n * h * w), "mse", mse1, mse2,
                  "|y1|max", int(out["y1_hat"].abs().max()), "nonzero y1 %.3f" % float((out["y1_hat"] != 0).float().mean()))

        # training-mode forward + R-D loss + aux at 64x64 with injected noise, and the 2-step Adam trace (row T)
        import math
        net.train()
        x1, x2, Hm = synthetic.stereo_batch(0, 2, 64, 64)
        order = ["z1", "y1", "y1w", "z2", "y2"] if tag == "hsic" else ["z1", "y1", "y1b", "z2", "y1w", "y2", "y2b"]
        shapes_n = {"z1": (128, 1, 2), "z2": (128, 1, 2)}
        for k in order:
            if k not in shapes_n:
                shapes_n[k] = (2, 192, 4, 4)
        lam = 0.0067
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        aux_opt = torch.optim.Adam(net.aux_parameters(), lr=1e-3)
        trace = []
        T = {}
        for step in range(2):
            nq.items = [det_noise(f"{tag}.{step}.{k}", shapes_n[k]) for k in order]
            opt.zero_grad()
            aux_opt.zero_grad()
            out = net(x1, x2, Hm)
            npix = 2 * 64 * 64
            bpp = sum(torch.log(l).sum() / (-math.log(2) * npix) for l in out["likelihoods"].values())
            mse = F.mse_loss(out["x1_hat"], x1) + F.mse_loss(out["x2_hat"], x2)
            loss = lam * 255 ** 2 * mse + bpp
            loss.backward()
            if step == 0:
                for n_, p_ in net.named_parameters():
                    T["gn_" + n_] = float(p_.grad.double().norm()) if p_.grad is not None else 0.0
                for n_, m_ in net.named_modules():
                    if hasattr(m_, "mask") and getattr(m_, "weight", None) is not None:
                        T["gn_live_" + n_ + ".weight"] = float((m_.weight.grad * m_.mask).double().norm())
                T["g_encoder1.g_a_conv4.bias"] = net.encoder1.g_a_conv4.bias.grad.clone()
                T["g_decoder2.after_conv.bias"] = net.decoder2.after_conv.bias.grad.clone()
                T["g_entropy_bottleneck1._biases.0"] = net.entropy_bottleneck1._biases[0].grad.clone()
            opt.step()
            aux = net.aux_loss()
            aux.backward()
            aux_opt.step()
            trace.append([float(loss), float(bpp), float(mse), float(aux)])
        T["trace"] = np.array(trace)
        T["noise_order"] = np.array(order)
        npz(f"{tag}_train64.npz", **T)
        print(tag, "train trace", trace)
        if tag == "hsic":
            with open(os.path.join(HERE, "hsic_state_keys.txt"), "w") as f:
                for k in keys:
                    f.write(f"{k} {' '.join(map(str, shapes[k]))}\n")
        else:
            with open(os.path.join(HERE, "joint_state_keys.txt"), "w") as f:
                for k in keys:
                    f.write(f"{k} {' '.join(map(str, shapes[k]))}\n")


def fx_models_r2(newnet1, newnet1_joint):
    """Round 2: (a) a NON-SQUARE eval forward (256 x 320, the class of BASELINE config C5's 896 x 1088) under BOTH warp
    conventions -- align_corners=True (kornia >= 0.5) and the legacy align_corners=False sampling of the kornia 0.4.x era
    the reference's pinned torch 1.6.0 implies (Readme.md:11,16); (b) the 2-step Adam trace of row T at 256 x 256."""
    import math
    from compressai.entropy_models import EntropyModel
    nq = NoiseQueue()
    EntropyModel._get_noise_cached = lambda self, x: nq(self, x)
    for tag, mod in (("hsic", newnet1), ("joint", newnet1_joint)):
        torch.manual_seed(0)
        net = mod.HSIC()
        synthetic.fill_state_dict_(net.state_dict())
        net.eval()
        x1, x2, Hm = synthetic.stereo_batch(0, 1, 256, 320)
        for ac in (True, False):
            KORNIA_ALIGN["align_corners"] = ac
            with torch.no_grad():
                out = net(x1, x2, Hm)
            bits = {k: float(torch.log(v.double()).sum() / -math.log(2)) for k, v in out["likelihoods"].items()}
            mse1 = float(((out["x1_hat"].double() - x1.double()) ** 2).mean())
            mse2 = float(((out["x2_hat"].double() - x2.double()) ** 2).mean())
            npz(f"{tag}_256x320{'' if ac else '_ac0'}.npz", H=Hm, bits_y1=bits["y1"], bits_y2=bits["y2"], bits_z1=bits["z1"], bits_z2=bits["z2"],
                mse1=mse1, mse2=mse2, y1_hat=out["y1_hat"].to(torch.int16), y2_hat=out["y2_hat"].to(torch.int16),
                x2_hat_pool=F.avg_pool2d(out["x2_hat"], 8))
            print(tag, "256x320 align_corners", ac, "bpp_loss", sum(bits.values()) / (256 * 320), "mse", mse1, mse2)
        KORNIA_ALIGN["align_corners"] = True

        net.train()
        x1, x2, Hm = synthetic.stereo_batch(0, 1, 256, 256)
        order = ["z1", "y1", "y1w", "z2", "y2"] if tag == "hsic" else ["z1", "y1", "y1b", "z2", "y1w", "y2", "y2b"]
        shapes_n = {k: ((128, 1, 16) if k[0] == "z" else (1, 192, 16, 16)) for k in order}
        lam = 0.0067
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        aux_opt = torch.optim.Adam(net.aux_parameters(), lr=1e-3)
        trace = []
        for step in range(2):
            nq.items = [det_noise(f"{tag}.t256.{step}.{k}", shapes_n[k]) for k in order]
            opt.zero_grad()
            aux_opt.zero_grad()
            out = net(x1, x2, Hm)
            npix = 256 * 256
            bpp = sum(torch.log(l).sum() / (-math.log(2) * npix) for l in out["likelihoods"].values())
            mse = F.mse_loss(out["x1_hat"], x1) + F.mse_loss(out["x2_hat"], x2)
            loss = lam * 255 ** 2 * mse + bpp
            loss.backward()
            opt.step()
            aux = net.aux_loss()
            aux.backward()
            aux_opt.step()
            trace.append([float(loss), float(bpp), float(mse), float(aux)])
        npz(f"{tag}_train256.npz", trace=np.array(trace), noise_order=np.array(order))
        print(tag, "train256 trace", trace)


def fx_models_r3(newnet1, newnet1_joint, which=("c2", "c4", "c5", "c3")):
    """Round 3: the reference run on the EXACT BASELINE workloads (BASELINE.json configs): C2 = HESIC, 8 pairs of 512 x 512; C4 = HESIC+,
    4 pairs of 512 x 512; C5 = 860 x 1080 pairs zero-padded to 896 x 1088, the four lambda-models of the sweep (weight salts 0..3, as
    bench.py --sweep fills them), metrics over the original pixels; C3 = one training step (R-D loss, both optimisers) on 8 pairs of
    512 x 512.  Stored per pair: bits per latent tensor, squared error per view, the rounded latents as int8."""
    import math
    from compressai.entropy_models import EntropyModel
    nq = NoiseQueue()
    EntropyModel._get_noise_cached = lambda self, x: nq(self, x)

    def per_pair(out, x1, x2, crop=None):
        B = x1.shape[0]
        A = {}
        for k, v in out["likelihoods"].items():
            A["bits_" + k] = (torch.log(v.double()).reshape(B, -1).sum(1) / -math.log(2)).numpy()
        xh1, xh2 = out["x1_hat"], out["x2_hat"]
        if crop is not None:
            xh1, xh2 = xh1[..., :crop[0], :crop[1]], xh2[..., :crop[0], :crop[1]]
        A["sse1"] = ((xh1.double() - x1.double()) ** 2).reshape(B, -1).sum(1).numpy()
        A["sse2"] = ((xh2.double() - x2.double()) ** 2).reshape(B, -1).sum(1).numpy()
        for k in ("y1_hat", "y2_hat"):
            assert float(out[k].abs().max()) < 128
            A[k] = out[k].to(torch.int8)
        return A

    for tag, mod, batch, cfg in (("hsic", newnet1, 8, "c2"), ("joint", newnet1_joint, 4, "c4")):
        if cfg not in which:
            continue
        net = mod.HSIC()
        synthetic.fill_state_dict_(net.state_dict())
        net.eval()
        x1, x2, Hm = synthetic.stereo_batch(0, batch, 512, 512)
        with torch.no_grad():
            out = net(x1, x2, Hm)
        A = per_pair(out, x1, x2)
        npz(f"{tag}_512_b{batch}.npz", H=Hm, **A)
        print(tag, "512 x", batch, "bpp", sum(float(A["bits_" + k].sum()) for k in ("y1", "y2", "z1", "z2")) / (batch * 512 * 512 * 2))

    if "c5" in which:
        Himg, Wimg = 860, 1080
        for tag, mod in (("hsic", newnet1), ("joint", newnet1_joint)):
            recs = {}
            for salt in range(4):
                net = mod.HSIC()
                synthetic.fill_state_dict_(net.state_dict(), salt=salt)
                net.eval()
                x1, x2, Hm = synthetic.stereo_batch(0, 1, Himg, Wimg)
                ph, pw = (-Himg) % 64, (-Wimg) % 64
                x1p, x2p = F.pad(x1, (0, pw, 0, ph)), F.pad(x2, (0, pw, 0, ph))
                with torch.no_grad():
                    out = net(x1p, x2p, Hm)
                A = per_pair(out, x1, x2, crop=(Himg, Wimg))
                for k, v in A.items():
                    if k.endswith("_hat"):
                        if salt == 0:
                            recs[k] = v
                        recs[f"{k}_abs_sum_{salt}"] = float(v.double().abs().sum())
                    else:
                        recs[f"{k}_{salt}"] = v
                print(tag, "c5 salt", salt, "bpp", sum(float(A["bits_" + k].sum()) for k in ("y1", "y2", "z1", "z2")) / (Himg * Wimg * 2))
            npz(f"{tag}_c5.npz", H=Hm, **recs)

    if "c3" in which:
        for tag, mod in (("hsic", newnet1), ("joint", newnet1_joint)):
            net = mod.HSIC()
            synthetic.fill_state_dict_(net.state_dict())
            net.train()
            B = 8
            x1, x2, Hm = synthetic.stereo_batch(0, B, 512, 512)
            order = ["z1", "y1", "y1w", "z2", "y2"] if tag == "hsic" else ["z1", "y1", "y1b", "z2", "y1w", "y2", "y2b"]
            shapes_n = {k: ((128, 1, B * 64) if k[0] == "z" else (B, 192, 32, 32)) for k in order}
            lam = 0.0067
            opt = torch.optim.Adam(net.parameters(), lr=1e-4)
            aux_opt = torch.optim.Adam(net.aux_parameters(), lr=1e-3)
            nq.items = [det_noise(f"{tag}.t512.0.{k}", shapes_n[k]) for k in order]
            opt.zero_grad()
            aux_opt.zero_grad()
            out = net(x1, x2, Hm)
            npix = B * 512 * 512
            bpp = sum(torch.log(l).sum() / (-math.log(2) * npix) for l in out["likelihoods"].values())
            mse = F.mse_loss(out["x1_hat"], x1) + F.mse_loss(out["x2_hat"], x2)
            loss = lam * 255 ** 2 * mse + bpp
            loss.backward()
            T = {"loss": float(loss), "bpp": float(bpp), "mse": float(mse), "noise_order": np.array(order)}
            for n_, p_ in net.named_parameters():
                T["gn_" + n_] = float(p_.grad.double().norm()) if p_.grad is not None else 0.0
            for n_, m_ in net.named_modules():
                if hasattr(m_, "mask") and getattr(m_, "weight", None) is not None:
                    T["gn_live_" + n_ + ".weight"] = float((m_.weight.grad * m_.mask).double().norm())
            opt.step()
            aux = net.aux_loss()
            aux.backward()
            aux_opt.step()
            T["aux"] = float(aux)
            # the step taken: a digest of the updated parameters (sum and sum of squares per tensor would be 2 x 222 numbers; one
            # norm of the whole update is what a test needs)
            npz(f"{tag}_train512.npz", **T)
            print(tag, "train512", T["loss"], T["bpp"], T["mse"], T["aux"])
            del net, out, loss


def fx_enhance(newnet1):
    """Independent_EN (cross-view enhancement, newnet1.py:272-311,1278-1300)."""
    net = newnet1.Independent_EN().eval()
    sd = net.state_dict()
    for name, t in sd.items():          # name-keyed deterministic fill (fan-in scaled), small so the residual stays tame
        fan = t.shape[1] * 9 if t.dim() == 4 else 1
        a = (3.0 / fan) ** 0.5 if t.dim() == 4 else 0.05
        t.copy_(synthetic._uniform("en." + name, t.shape, -a, a))
    x1, x2, Hm = synthetic.stereo_batch(5, 2, 64, 64)
    with torch.no_grad():
        out = net(x1, x2, Hm)
    with open(os.path.join(HERE, "en_state_keys.txt"), "w") as f:
        for k, v in sd.items():
            f.write(f"{k} {' '.join(map(str, v.shape))}\n")
    npz("en_64.npz", x1_hat=out["x1_hat"], x2_hat=out["x2_hat"])


def fx_stage2(newnet1):
    """Stage-2 training steps as ywz/mywork/newtrain6_real.py:154-167 runs them, recorded from the reference's own modules: HSIC in eval
    mode (frozen), Independent_EN in train mode, loss = lambda * 255^2 * (MSE(x1_hat', x1) + MSE(x2_hat', x2)) (:83-91, kind=0), Adam on
    the enhancement net only.  Two steps on two 128 x 128 pairs: loss / mse per step, every parameter's gradient norm of step 0 and the
    parameter norms after step 1."""
    hs = newnet1.HSIC()
    synthetic.fill_state_dict_(hs.state_dict())
    hs.eval()
    en = newnet1.Independent_EN()
    sd = en.state_dict()
    for name, t in sd.items():
        fan = t.shape[1] * 9 if t.dim() == 4 else 1
        a = (3.0 / fan) ** 0.5 if t.dim() == 4 else 0.05
        t.copy_(synthetic._uniform("en." + name, t.shape, -a, a))
    en.train()
    lam = 0.0067
    opt = torch.optim.Adam(en.parameters(), lr=1e-4)
    x1, x2, Hm = synthetic.stereo_batch(11, 2, 128, 128)
    T = {}
    for step in range(2):
        opt.zero_grad()
        out = hs(x1, x2, Hm)
        out2 = en(out["x1_hat"], out["x2_hat"], Hm)
        mse = F.mse_loss(out2["x1_hat"], x1) + F.mse_loss(out2["x2_hat"], x2)
        loss = lam * 255 ** 2 * mse
        loss.backward()
        T[f"loss{step}"], T[f"mse{step}"] = float(loss), float(mse)
        if step == 0:
            for n_, p_ in en.named_parameters():
                T["gn_" + n_] = float(p_.grad.double().norm())
        opt.step()
    for n_, p_ in en.named_parameters():
        T["pn_" + n_] = float(p_.detach().double().norm())
    npz("stage2_128.npz", **T)


def fx_homo():
    """HomographyNet forward (ywz/mywork/model.py:73-101) from the reference's own module.  Weights
    and patches are the name-keyed synthetic ones (regenerated by the tests), so only outputs are stored."""
    import model as ref_model                                   # ywz/mywork/model.py
    net = ref_model.Net().eval()
    sd = net.state_dict()
    synthetic.fill_homography_state_dict_(sd)
    a, b, corners = synthetic.homography_batch(0, 2)
    feats = {}
    net.cnn.register_forward_hook(lambda m, i, o: feats.__setitem__("cnn", o))
    with torch.no_grad():
        delta = net(a, b)
    with open(os.path.join(HERE, "homo_state_keys.txt"), "w") as f:
        for k, v in sd.items():
            f.write(f"{k} {' '.join(map(str, v.shape))}\n")
    npz("homo.npz", delta=delta, cnn_sub=feats["cnn"][:, ::8, ::2, ::2], cnn_absmean=feats["cnn"].abs().mean())


def fx_codec():
    """G11: pmf_to_quantized_cdf cases and rANS byte strings from the reference's C++ extensions."""
    from compressai._CXX import pmf_to_quantized_cdf
    from compressai import ans
    A = {}
    r = np.random.Generator(np.random.PCG64(11))
    pmfs = [np.array([0.25, 0.25, 0.25, 0.25]), np.array([0.9, 0.05, 0.05, 1e-12, 1e-12, 0.0]),
            r.dirichlet(np.ones(33)), np.concatenate([r.dirichlet(np.ones(5)) * 0.999999, np.full(200, 1e-9 / 200)])]
    for i, p in enumerate(pmfs):
        A[f"pmf{i}"] = p.astype(np.float32)
        A[f"cdf{i}"] = np.array(pmf_to_quantized_cdf(p.astype(np.float32).tolist(), 16), dtype=np.uint32)
    # rANS: 3 cdf tables, symbols incl. out-of-range (bypass) ones
    cdfs = [pmf_to_quantized_cdf(r.dirichlet(np.ones(n)).astype(np.float32).tolist(), 16) for n in (4, 9, 17)]
    sizes = [len(c) for c in cdfs]
    L = max(sizes)
    table = [list(c) + [0] * (L - len(c)) for c in cdfs]
    offsets = [-2, -4, -8]
    n = 500
    idx = r.integers(0, 3, n).tolist()
    sym = [int(r.integers(offsets[i] - 3, offsets[i] + sizes[i] + 2)) for i in idx]
    enc = ans.RansEncoder()
    s = enc.encode_with_indexes(sym, idx, table, sizes, offsets)
    dec = ans.RansDecoder()
    back = dec.decode_with_indexes(s, idx, table, sizes, offsets)
    assert back == sym
    A.update(rans_cdfs=np.array(table, dtype=np.int32), rans_sizes=np.array(sizes), rans_offsets=np.array(offsets),
             rans_indexes=np.array(idx), rans_symbols=np.array(sym), rans_bytes=np.frombuffer(s, dtype=np.uint8))
    npz("codec.npz", **A)


def fx_codec_model(newnet1):
    """Round 2: the reference's OWN ``HSIC.compress`` (newnet1.py:823-1066) run here at 64 x 64 with a
    recording stand-in for the third-party range coder object it drives (``encoder.encode([symbol], cdf)`` calls are logged, no
    bytes are produced: the package is absent and its stream format is not what is pinned) and ``.to('cuda:0')`` mapped to
    the host.  What it pins: the side-information file byte for byte, the coding order and symbols, and every per-latent
    cumulative-frequency table the reference computes (all tables of view 1, every third of view 2)."""
    import tempfile
    np.int = int                                   # newnet1.py:880,980 still use the alias NumPy 2 removed
    log = []

    class Recorder:
        def __init__(self, path):
            self.path = path
            open(path, "wb").close()           # the reference stats the file afterwards (newnet1.py:1048)

        def encode(self, symbols, cdf):
            log.append((int(symbols[0]), np.asarray(cdf, dtype=np.int64)))

        def close(self):
            pass

    newnet1.RangeEncoder = Recorder
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else orig_to(self, *a, **k)
    try:
        torch.manual_seed(0)
        net = newnet1.HSIC()
        synthetic.fill_state_dict_(net.state_dict())
        net.eval()
        net.entropy_bottleneck1.update(force=True)
        net.entropy_bottleneck2.update(force=True)
        x1, x2, Hm = synthetic.stereo_batch(0, 1, 64, 64)
        with tempfile.TemporaryDirectory() as td, torch.no_grad():
            net.compress(x1, x2, Hm, "pair0", td)
            head = np.frombuffer(open(os.path.join(td, "pair0.npz"), "rb").read(), dtype=np.uint8).copy()
    finally:
        torch.Tensor.to = orig_to
    # header: uint16 H, W | per view: uint16 len(z), uint16 minmax, M/8 flag bytes, z bytes
    pos, views = 4, []
    for _ in range(2):
        ln, mm = (int(v) for v in np.frombuffer(head[pos:pos + 4].tobytes(), dtype=np.uint16))
        flags = np.unpackbits(head[pos + 4:pos + 4 + 24])
        views.append((ln, mm, int(flags.sum())))
        pos += 4 + 24 + ln
    n1 = views[0][2] * 16                          # latents of view 1: non-zero channels x 4 x 4
    assert len(log) == n1 + views[1][2] * 16, (len(log), views)
    sym = np.array([s_ for s_, _ in log], dtype=np.int32)
    t1 = np.stack([c for _, c in log[:n1]]).astype(np.uint32)
    t2 = np.stack([c for _, c in log[n1::3]]).astype(np.uint32)
    npz("codec_model_64.npz", header=head, symbols=sym, n_view1=np.int64(n1), tables1=t1, tables2_every3=t2,
        minmax=np.array([views[0][1], views[1][1]]), zlen=np.array([views[0][0], views[1][0]]))
    print("codec model: views (len z, minmax, non-zero channels)", views, "calls", len(log), "table widths", t1.shape, t2.shape)


def fx_codec_model_joint(newnet1_joint):
    """Round 3: the reference's own HESIC+ ``HSIC.compress`` (newnet1_joint.py:793-1079) at 64 x 64 with the same recording stand-in
    for the absent range coder as ``fx_codec_model``: the side-information file byte for byte, the coding order (raster over the
    pixels, all non-zero channels of a pixel together; view 1 then view 2), every symbol and every cumulative-frequency table the
    per-pixel crop -> masked conv -> entropy-parameter net of the reference produces."""
    import tempfile
    np.int = int                                   # newnet1_joint.py:858,967 still use the alias NumPy 2 removed
    log = []

    class Recorder:
        def __init__(self, path):
            open(path, "wb").close()           # the reference stats the file afterwards

        def encode(self, symbols, cdf):
            log.append((int(symbols[0]), np.asarray(cdf, dtype=np.int64)))

        def close(self):
            pass

    newnet1_joint.RangeEncoder = Recorder
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else orig_to(self, *a, **k)
    try:
        torch.manual_seed(0)
        net = newnet1_joint.HSIC()
        synthetic.fill_state_dict_(net.state_dict())
        net.eval()
        net.entropy_bottleneck1.update(force=True)
        net.entropy_bottleneck2.update(force=True)
        x1, x2, Hm = synthetic.stereo_batch(0, 1, 64, 64)
        with tempfile.TemporaryDirectory() as td, torch.no_grad():
            net.compress(x1, x2, Hm, "pair0", td)
            head = np.frombuffer(open(os.path.join(td, "pair0.npz"), "rb").read(), dtype=np.uint8).copy()
    finally:
        torch.Tensor.to = orig_to
    pos, views = 4, []
    for _ in range(2):
        ln, mm = (int(v) for v in np.frombuffer(head[pos:pos + 4].tobytes(), dtype=np.uint16))
        flags = np.unpackbits(head[pos + 4:pos + 4 + 24])
        views.append((ln, mm, int(flags.sum())))
        pos += 4 + 24 + ln
    n1 = views[0][2] * 16                          # latents of view 1: 4 x 4 pixels x non-zero channels
    assert len(log) == n1 + views[1][2] * 16, (len(log), views)
    sym = np.array([s_ for s_, _ in log], dtype=np.int32)
    t1 = np.stack([c for _, c in log[:n1]]).astype(np.uint32)
    t2 = np.stack([c for _, c in log[n1:]]).astype(np.uint32)
    npz("codec_model_joint_64.npz", header=head, symbols=sym, n_view1=np.int64(n1), tables1=t1, tables2=t2,
        minmax=np.array([views[0][1], views[1][1]]), zlen=np.array([views[0][0], views[1][0]]))
    print("codec model joint: views (len z, minmax, non-zero channels)", views, "calls", len(log), "table widths", t1.shape, t2.shape)


def fx_dataset():
    """Round 3: the reference's own stereo ``ImageFolder`` (compressai/datasets/utils.py:68-214) run on a small
    synthetic left/right folder with Python's ``random`` seeded per item.  Absent third parties are stood in for by their
    published semantics -- ``cv2.imread`` / ``cvtColor`` (PIL + channel flip), ``cv2.resize`` (INTER_LINEAR: half-pixel centres, no
    anti-aliasing; float form rounded to uint8), torchvision ``ToTensor`` / ``Normalize`` / ``Compose`` -- and ``get_H`` (SURF + RANSAC,
    opencv-contrib non-free) by a fixed matrix.  What the fixture pins is everything the reference's code itself decides: the
    pairing, the crop rule and its random draws, the shared offset of the two views, the 256 -> 128 grey windows with their
    normalisation and corner order, the item layouts."""
    import random
    import tempfile
    from PIL import Image
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB = 4

    def imread(path):
        with Image.open(path) as im:
            return np.array(im.convert("RGB"))[:, :, ::-1].copy()
    cv2.imread = imread
    cv2.cvtColor = lambda img, code: img[:, :, ::-1].copy()

    def resize(img, size):
        t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).unsqueeze(0).float()
        t = F.interpolate(t, size=(size[1], size[0]), mode="bilinear", align_corners=False)
        return t.round().clamp(0, 255).to(torch.uint8)[0].permute(1, 2, 0).numpy()
    cv2.resize = resize
    sys.modules["cv2"] = cv2
    tvt = sys.modules["torchvision.transforms"]

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255.0)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            return (t - self.mean.view(-1, 1, 1)) / self.std.view(-1, 1, 1)
    tvt.Compose, tvt.ToTensor, tvt.Normalize = Compose, ToTensor, Normalize
    for m in [k for k in sys.modules if k.startswith("compressai.datasets")]:
        del sys.modules[m]
    import compressai.datasets.utils as U
    Hfix = np.array([[1.01, 0.02, -3.0], [-0.01, 0.99, 1.5], [1e-5, -2e-5, 1.0]])
    U.get_H = lambda a, b: [torch.from_numpy(Hfix.astype(np.float32))]
    Himg, Wimg, n = 96, 128, 3
    x1, x2, _ = synthetic.stereo_batch(0, n, Himg, Wimg)
    A = {"Hfix": Hfix.astype(np.float32)}
    with tempfile.TemporaryDirectory() as td:
        for side, x in (("left", x1), ("right", x2)):
            os.makedirs(os.path.join(td, "train", side))
            for i in range(n):
                Image.fromarray((x[i].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)).save(os.path.join(td, "train", side, f"{i:04d}.png"))
        ds = U.ImageFolder(td, transform=ToTensor(), patch_size=(64, 80), split="train")
        for i in range(n):
            random.seed(1000 + i)
            a, b, h, g1, g2, corners = ds[i]
            A[f"x1_{i}"], A[f"x2_{i}"] = (a * 255).round().to(torch.uint8), (b * 255).round().to(torch.uint8)
            A[f"H_{i}"], A[f"corners_{i}"] = h, corners
            A[f"homo1_sub_{i}"], A[f"homo2_sub_{i}"] = g1[:, ::4, ::4], g2[:, ::4, ::4]
            A[f"homo_sums_{i}"] = np.array([float(g1.double().sum()), float((g1.double() ** 2).sum()), float(g2.double().sum()), float((g2.double() ** 2).sum())])
        full = U.ImageFolder(td, transform=ToTensor(), patch_size=(96, 128), split="train", need_file_name=True)
        random.seed(7)
        item = full[1]
        A["full_len"], A["full_name"] = np.int64(len(item)), np.array(item[3])
        A["full_x1_sum"], A["full_corners"] = float(item[0].double().sum()), item[6]
        U.get_H = lambda a, b: [None]
        random.seed(8)
        A["none_len"] = np.int64(len(ds[0]))
    npz("dataset.npz", **A)


def main():
    torch.set_num_threads(8)
    if sys.argv[1:] == ["msssim"]:           # needs nothing of the reference
        return fx_msssim()
    newnet1, newnet1_joint = import_reference()
    which = sys.argv[1:] or ["ops", "warp", "models", "models2", "codec", "codec_model", "codec_model_joint", "dataset", "enhance", "homo", "models3"]
    if "ops" in which:
        fx_ops(newnet1)
    if "warp" in which:
        fx_warp()
    if "codec" in which:
        fx_codec()
    if "enhance" in which:
        fx_enhance(newnet1)
    if "stage2" in which:
        fx_stage2(newnet1)
    if "homo" in which:
        fx_homo()
    if "models" in which:
        fx_models(newnet1, newnet1_joint)
    if any(w in which for w in ("c2", "c3", "c4", "c5", "models3")):
        sel = ("c2", "c4", "c5", "c3") if "models3" in which else tuple(w for w in which if w in ("c2", "c3", "c4", "c5"))
        fx_models_r3(newnet1, newnet1_joint, which=sel)
    if "models2" in which:
        fx_models_r2(newnet1, newnet1_joint)
    if "codec_model" in which:
        fx_codec_model(newnet1)
    if "dataset" in which:
        fx_dataset()
    if "codec_model_joint" in which:
        fx_codec_model_joint(newnet1_joint)


if __name__ == "__main__":
    main()
