"""CPU oracle for the HESIC / HESIC+ hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file.  The product path (``hesic_amd``) never does:
it fails loudly when the HIP library is missing.

What it is: a functional, module-free restatement of the reference's algorithm
for the path ``HSIC.forward`` (``ywz/mywork/newnet1.py:724-783``) and the HESIC+
variant (``ywz/mywork/newnet1_joint.py:675-753``), written against plain CPU
tensors in NCHW.  Parameters arrive as a flat ``{state_dict key: tensor}`` dict
with the reference's key names.  The contractions themselves
(``F.conv2d`` / ``F.conv_transpose2d`` / ``F.grid_sample``) are the same L1
PyTorch primitives the reference calls; everything above them is restated here.

Pinning: every function below is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by importing the reference itself in the build
container (``tests/golden/make_golden.py``).  One exception, stated in
DESIGN.md: ``warp_perspective`` comes from the third-party ``kornia`` package,
which is neither vendored in the reference nor installed; its two historical
semantics are restated from its published definition -> *parity unpinned* for
that op only.  Likewise ``ms_ssim`` (round 4): ``pytorch_msssim`` is third party,
absent and unpinned by the reference; its published algorithm is restated here and
cross-checked against an independent numpy / scipy route (``make_golden.py msssim``).

Reference files are cited as path:line relative to the reference root.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LIKELIHOOD_BOUND = 1e-9          # compressai/entropy_models/entropy_models.py:66
SCALE_BOUND = 0.11               # compressai/entropy_models/entropy_models.py:446,587
REPARAM_OFFSET = 2.0 ** -18      # compressai/ops/parametrizers.py:27
PEDESTAL = REPARAM_OFFSET ** 2


# --------------------------------------------------------------------------- ops
class _LowerBoundFn(torch.autograd.Function):
    """max(x, b) whose gradient passes where x >= b or the step moves x up
    (compressai/ops/bound_ops.py:19-31)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x)
        ctx.bound = bound
        return x.clamp(min=bound)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        keep = (x >= ctx.bound) | (g < 0)
        return g * keep.to(g.dtype), None


def lower_bound(x, bound: float):
    b = float(torch.tensor(bound, dtype=torch.float32))   # the reference stores the bound as fp32
    return _LowerBoundFn.apply(x, b)


def nonneg(theta, minimum: float = 0.0):
    """NonNegativeParametrizer.forward (compressai/ops/parametrizers.py:41-44)."""
    bound = (minimum + PEDESTAL) ** 0.5
    ped = float(torch.tensor(PEDESTAL, dtype=torch.float32))
    return lower_bound(theta, bound) ** 2 - ped


def gdn(x, beta, gamma, inverse=False, beta_min=1e-6):
    """GDN / IGDN (compressai/layers/gdn.py:55-70)."""
    C = x.shape[1]
    b = nonneg(beta, beta_min)
    g = nonneg(gamma).reshape(C, C, 1, 1)
    norm = F.conv2d(x * x, g, b)
    return x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm))


def conv(x, w, b, stride=2):
    """conv() factory: Conv2d(k, stride, padding=k//2) (compressai/models/utils.py:104-109)."""
    return F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)


def deconv(x, w, b, stride=2):
    """deconv() factory: ConvTranspose2d(k, stride, output_padding=stride-1, padding=k//2)
    (compressai/models/utils.py:112-118)."""
    return F.conv_transpose2d(x, w, b, stride=stride, padding=w.shape[-1] // 2,
                              output_padding=stride - 1)


def masked_conv(x, w, b, mask_type="A"):
    """MaskedConv2d (compressai/layers/layers.py:21-45): taps at/after the centre are zeroed."""
    kh, kw = w.shape[-2:]
    m = torch.ones_like(w)
    m[:, :, kh // 2, kw // 2 + (mask_type == "B"):] = 0
    m[:, :, kh // 2 + 1:] = 0
    return F.conv2d(x, w * m, b, stride=1, padding=kh // 2)


def quantize(x, mode, means=None, noise=None):
    """EntropyModel._quantize (compressai/entropy_models/entropy_models.py:98-125).
    ``noise`` replaces the reference's U(-1/2,1/2) draw so runs are reproducible."""
    if mode not in ("noise", "dequantize", "symbols"):
        raise ValueError(f'Invalid quantization mode: "{mode}"')
    if mode == "noise":
        return x + noise
    v = x if means is None else x - means
    v = torch.round(v)
    if mode == "dequantize":
        return v if means is None else v + means
    return v.int()


def std_cumulative(x):
    """Phi(x) = erfc(-x/sqrt2)/2 (entropy_models.py:485-490)."""
    return 0.5 * torch.erfc(-(2 ** -0.5) * x)


# ------------------------------------------------------------------ entropy models
def eb_logits_cumulative(P, prefix, v, stop_gradient=False):
    """EntropyBottleneck._logits_cumulative (entropy_models.py:350-369). v: (C,1,L)."""
    n = sum(1 for k in P if k.startswith(prefix + "_matrices."))
    h = v
    for i in range(n):
        m = P[f"{prefix}_matrices.{i}"]
        b = P[f"{prefix}_biases.{i}"]
        if stop_gradient:
            m, b = m.detach(), b.detach()
        h = torch.matmul(F.softplus(m), h) + b
        if i < n - 1:
            f = P[f"{prefix}_factors.{i}"]
            if stop_gradient:
                f = f.detach()
            h = h + torch.tanh(f) * torch.tanh(h)
    return h


def eb_forward(P, prefix, x, training=False, noise=None):
    """EntropyBottleneck.forward (entropy_models.py:384-411). Returns (x_hat, likelihood)."""
    B, C, H, W = x.shape
    v = x.permute(1, 2, 3, 0).reshape(C, 1, -1)
    med = P[prefix + "quantiles"][:, :, 1:2]
    if training:
        nz = noise.permute(1, 2, 3, 0).reshape(C, 1, -1)
        out = quantize(v, "noise", med, nz)
    else:
        out = quantize(v, "dequantize", med)
    lo = eb_logits_cumulative(P, prefix, out - 0.5)
    up = eb_logits_cumulative(P, prefix, out + 0.5)
    sign = -torch.sign(lo + up).detach()
    lik = torch.abs(torch.sigmoid(sign * up) - torch.sigmoid(sign * lo))
    lik = lower_bound(lik, LIKELIHOOD_BOUND)
    back = lambda t: t.reshape(C, H, W, B).permute(3, 0, 1, 2).contiguous()
    return back(out), back(lik)


def eb_aux_loss(P, prefix):
    """EntropyBottleneck.loss (entropy_models.py:345-348)."""
    t = math.log(2 / 1e-9 - 1)
    target = torch.tensor([-t, 0.0, t], dtype=P[prefix + "quantiles"].dtype)
    logits = eb_logits_cumulative(P, prefix, P[prefix + "quantiles"], stop_gradient=True)
    return torch.abs(logits - target).sum()


def gaussian_likelihood(y_hat, scales, means=None):
    """GaussianConditional._likelihood (entropy_models.py:528-544), without the final bound."""
    v = y_hat if means is None else y_hat - means
    s = lower_bound(scales, SCALE_BOUND)
    v = torch.abs(v)
    return std_cumulative((0.5 - v) / s) - std_cumulative((-0.5 - v) / s)


def gc_forward(y, scales, means=None, training=False, noise=None):
    """GaussianConditional.forward (entropy_models.py:546-554)."""
    out = quantize(y, "noise", means, noise) if training else quantize(y, "dequantize", means)
    return out, lower_bound(gaussian_likelihood(out, scales, means), LIKELIHOOD_BOUND)


def gmm_forward(y, scales, means, weights, K, training=False, noise=None):
    """GaussianMixtureConditional.forward (entropy_models.py:661-702): quantise WITHOUT
    means, likelihood = sum_k w[:,kM:(k+1)M] * (Phi(u_k) - Phi(l_k))."""
    M = y.shape[1]
    out = quantize(y, "noise", None, noise) if training else quantize(y, "dequantize", None)
    lik = None
    for k in range(K):
        sl = slice(M * k, M * (k + 1))
        term = gaussian_likelihood(out, scales[:, sl], means[:, sl]) * weights[:, sl]
        lik = term if lik is None else lik + term
    return out, lower_bound(lik, LIKELIHOOD_BOUND)


# ----------------------------------------------------------------------- geometry
def warp_perspective(src, Mat, dsize, align_corners=True):
    """kornia.warp_perspective(src, M, dsize) restated (third party).

    ``Mat`` (B,3,3) maps source pixel coords to destination pixel coords.
    align_corners=True  : kornia >= 0.5 default == exact inverse-map bilinear with zero
                          padding (also cv2.warpPerspective).
    align_corners=False : kornia <= 0.4 default: the same normalised grid handed to
                          grid_sample(align_corners=False), i.e. sampling at
                          xs*W/(W-1) - 1/2.
    """
    B, C, H, W = src.shape
    Ho, Wo = dsize
    Minv = torch.linalg.inv(Mat.double())
    ys, xs = torch.meshgrid(torch.arange(Ho, dtype=torch.float64),
                            torch.arange(Wo, dtype=torch.float64), indexing="ij")
    ones = torch.ones_like(xs)
    pts = torch.stack([xs, ys, ones], 0).reshape(1, 3, -1)
    s = Minv @ pts
    sx = (s[:, 0] / s[:, 2]).reshape(B, Ho, Wo)
    sy = (s[:, 1] / s[:, 2]).reshape(B, Ho, Wo)
    gx = 2.0 * sx / (W - 1) - 1.0
    gy = 2.0 * sy / (H - 1) - 1.0
    grid = torch.stack([gx, gy], -1).to(src.dtype)
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros",
                         align_corners=align_corners)


def upsample_bilinear_x4(x):
    """nn.UpsamplingBilinear2d(scale_factor=4) == align_corners=True (newnet1.py:524)."""
    return F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=True)


# --------------------------------------------------------------------- sub-networks
def _cv(P, key, x, stride=2):
    return conv(x, P[key + ".weight"], P[key + ".bias"], stride)


def _dc(P, key, x, stride=2):
    return deconv(x, P[key + ".weight"], P[key + ".bias"], stride)


def _gdn(P, key, x, inverse=False):
    return gdn(x, P[key + ".beta"], P[key + ".gamma"], inverse)


def g_a(P, pre, x):
    """Encoder1 stack (newnet1.py:580-601); also the tail of Encoder2."""
    x = _gdn(P, pre + "g_a_gdn1", _cv(P, pre + "g_a_conv1", x))
    x = _gdn(P, pre + "g_a_gdn2", _cv(P, pre + "g_a_conv2", x))
    x = _gdn(P, pre + "g_a_gdn3", _cv(P, pre + "g_a_conv3", x))
    return _cv(P, pre + "g_a_conv4", x)


def g_s(P, pre, y):
    """Decoder1 stack (newnet1.py:603-624); also the head of Decoder2."""
    y = _gdn(P, pre + "g_s_gdn1", _dc(P, pre + "g_s_conv1", y), True)
    y = _gdn(P, pre + "g_s_gdn2", _dc(P, pre + "g_s_conv2", y), True)
    y = _gdn(P, pre + "g_s_gdn3", _dc(P, pre + "g_s_conv3", y), True)
    return _dc(P, pre + "g_s_conv4", y)


def encoder2(P, x1_warp, x2):
    """Encoder2.forward (newnet1.py:626-655)."""
    t = _cv(P, "encoder2.pre_conv", torch.cat((x1_warp, x2), 1), stride=1)
    t = _gdn(P, "encoder2.pre_gdn", t)
    return g_a(P, "encoder2.", t)


def decoder2(P, y_hat, x1_hat_warp):
    """Decoder2.forward (newnet1.py:657-692)."""
    t = _gdn(P, "decoder2.after_gdn", g_s(P, "decoder2.", y_hat), True)
    return _dc(P, "decoder2.after_conv", torch.cat((t, x1_hat_warp), 1), stride=1)


def encode_hyper(P, pre, y):
    """encode_hyper.forward (newnet1.py:420-437)."""
    t = F.relu(_cv(P, pre + "encode_hyper.0", torch.abs(y), 1))
    t = F.relu(_cv(P, pre + "encode_hyper.2", t))
    return _cv(P, pre + "encode_hyper.4", t)


def _mix_weights(t, K, M):
    """(B,K*M,1,1) logits -> softmax over K with channel = k*M+m (newnet1.py:510-512)."""
    B = t.shape[0]
    return F.softmax(t.reshape(B, K, M, 1, 1), dim=1).reshape(B, K * M, 1, 1)


def gmm_hyper_y1(P, z_hat, K, M):
    """gmm_hyper_y1.forward (newnet1.py:456-514)."""
    p = "_h_s1."
    s = F.relu(_dc(P, p + "gmm_sigma.0", z_hat))
    s = F.relu(_dc(P, p + "gmm_sigma.2", s))
    s = F.relu(_cv(P, p + "gmm_sigma.4", s, 1))
    m = F.leaky_relu(_dc(P, p + "gmm_means.0", z_hat), 0.01)
    m = F.leaky_relu(_dc(P, p + "gmm_means.2", m), 0.01)
    m = _cv(P, p + "gmm_means.4", m, 1)
    w = F.leaky_relu(_dc(P, p + "gmm_weights.0", z_hat), 0.01)
    w = _dc(P, p + "gmm_weights.2", w)
    w = F.leaky_relu(torch.amax(w, dim=(2, 3), keepdim=True), 0.01)   # spatial_pool2d :441-453
    w = _cv(P, p + "gmm_weights.5", w, 1)
    return s, m, _mix_weights(w, K, M)


def gmm_hyper_y2(P, z_hat, y1, K, M):
    """gmm_hyper_y2.forward (newnet1.py:517-577)."""
    p = "_h_s2."
    c = torch.cat((upsample_bilinear_x4(z_hat), y1), 1)
    s = F.relu(_cv(P, p + "gmm_sigma.0", c, 1))
    s = F.relu(_cv(P, p + "gmm_sigma.2", s, 1))
    s = F.relu(_cv(P, p + "gmm_sigma.4", s, 1))
    m = F.leaky_relu(_cv(P, p + "gmm_means.0", c, 1), 0.01)
    m = F.leaky_relu(_cv(P, p + "gmm_means.2", m, 1), 0.01)
    m = _cv(P, p + "gmm_means.4", m, 1)
    w = F.leaky_relu(_cv(P, p + "gmm_weights.0", c, 1), 0.01)
    w = _cv(P, p + "gmm_weights.2", w, 1)
    w = F.leaky_relu(torch.amax(w, dim=(2, 3), keepdim=True), 0.01)
    w = _cv(P, p + "gmm_weights.5", w, 1)
    return s, m, _mix_weights(w, K, M)


# -------------------------------------------------------------------- whole models
def hsic_forward(P, x1, x2, Hm, K=5, M=192, training=False, noise=None, align_corners=True, return_gmm=False):
    """HSIC.forward (ywz/mywork/newnet1.py:724-783).  ``return_gmm``: also hand back the (sigma, means, weights) triples and the
    un-quantised z of both views (what ``HSIC.compress``, :823-906, codes).

    ``noise`` (training only): dict with keys z1,y1,y1w,z2,y2 -- the five U(-1/2,1/2)
    draws in the order the reference makes them (SURVEY.md §7 hard parts).
    """
    nz = noise or {}
    size = x1.shape[-2:]
    y1 = g_a(P, "encoder1.", x1)
    z1 = encode_hyper(P, "_h_a1.", y1)
    z1_hat, z1_lik = eb_forward(P, "entropy_bottleneck1.", z1, training, nz.get("z1"))
    s1, m1, w1 = gmm_hyper_y1(P, z1_hat, K, M)
    y1_hat, y1_lik = gmm_forward(y1, s1, m1, w1, K, training, nz.get("y1"))
    x1_hat = g_s(P, "decoder1.", y1_hat)

    x1_warp = warp_perspective(x1, Hm, size, align_corners)
    y2 = encoder2(P, x1_warp, x2)
    x1_hat_warp = warp_perspective(x1_hat, Hm, size, align_corners)     # :753 == :767
    y1_w = g_a(P, "encoder1.", x1_hat_warp)
    y1_hat_w = quantize(y1_w, "noise", None, nz.get("y1w")) if training else quantize(y1_w, "dequantize")

    z2 = encode_hyper(P, "_h_a2.", y2)
    z2_hat, z2_lik = eb_forward(P, "entropy_bottleneck2.", z2, training, nz.get("z2"))
    s2, m2, w2 = gmm_hyper_y2(P, z2_hat, y1_hat_w, K, M)
    y2_hat, y2_lik = gmm_forward(y2, s2, m2, w2, K, training, nz.get("y2"))
    x2_hat = decoder2(P, y2_hat, x1_hat_warp)
    out = {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat,
           "z1_hat": z1_hat, "z2_hat": z2_hat,
           "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}
    if return_gmm:
        out.update(gmm1=(s1, m1, w1), gmm2=(s2, m2, w2), z1=z1, z2=z2)
    return out


def _seq(P, pre, x, spec):
    """Run a reference nn.Sequential given as [(index, kind, stride)], LeakyReLU(0.01) between."""
    for j, (idx, kind, stride) in enumerate(spec):
        key = f"{pre}.{idx}"
        if kind == "conv":
            x = _cv(P, key, x, stride)
        elif kind == "deconv":
            x = _dc(P, key, x, stride)
        else:  # plain nn.Conv2d 1x1
            x = F.conv2d(x, P[key + ".weight"], P[key + ".bias"])
        if j < len(spec) - 1:
            x = F.leaky_relu(x, 0.01)
    return x


_H_A = [(0, "conv", 1), (2, "conv", 2), (4, "conv", 2)]
_H_S = [(0, "deconv", 2), (2, "deconv", 2), (4, "conv", 1)]
_EP = [(0, "1x1", 1), (2, "1x1", 1), (4, "1x1", 1)]


def hsic_joint_forward(P, x1, x2, Hm, training=False, noise=None, align_corners=True, return_params=False):
    """HESIC+ forward (ywz/mywork/newnet1_joint.py:675-753).  ``return_params``: also out["gauss1"/"gauss2"] = (scales, means),
    what compress() builds its per-pixel tables from (:903-911 -- the per-pixel crop evaluates the same masked conv)."""
    nz = noise or {}
    size = x1.shape[-2:]
    q = (lambda t, k: quantize(t, "noise", None, nz.get(k))) if training else (lambda t, k: quantize(t, "dequantize"))
    y1 = g_a(P, "encoder1.", x1)
    z1 = _seq(P, "h_a1", y1, _H_A)
    z1_hat, z1_lik = eb_forward(P, "entropy_bottleneck1.", z1, training, nz.get("z1"))
    params1 = _seq(P, "h_s1", z1_hat, _H_S)
    y1_hat = q(y1, "y1")
    ctx1 = masked_conv(y1_hat, P["context_prediction1.weight"], P["context_prediction1.bias"])
    gp1 = _seq(P, "entropy_parameters1", torch.cat((params1, ctx1), 1), _EP)
    sc1, mu1 = gp1.chunk(2, 1)
    _, y1_lik = gc_forward(y1, sc1, mu1, training, nz.get("y1b"))
    x1_hat = g_s(P, "decoder1.", y1_hat)

    x1_warp = warp_perspective(x1, Hm, size, align_corners)
    y2 = encoder2(P, x1_warp, x2)
    z2 = _seq(P, "h_a2", y2, _H_A)
    z2_hat, z2_lik = eb_forward(P, "entropy_bottleneck2.", z2, training, nz.get("z2"))
    x1_hat_warp = warp_perspective(x1_hat, Hm, size, align_corners)
    y1_hat_w = q(g_a(P, "encoder1.", x1_hat_warp), "y1w")
    params2 = _seq(P, "h_s2", z2_hat, _H_S)
    y2_hat = q(y2, "y2")
    ctx2 = masked_conv(y2_hat, P["context_prediction2.weight"], P["context_prediction2.bias"])
    gp2 = _seq(P, "entropy_parameters2", torch.cat((params2, ctx2, y1_hat_w), 1), _EP)
    sc2, mu2 = gp2.chunk(2, 1)
    _, y2_lik = gc_forward(y2, sc2, mu2, training, nz.get("y2b"))
    x2_hat = decoder2(P, y2_hat, x1_hat_warp)
    out = {"x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "y2_hat": y2_hat,
           "z1_hat": z1_hat, "z2_hat": z2_hat,
           "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik}}
    if return_params:
        out["gauss1"], out["gauss2"], out["z1"], out["z2"] = (sc1, mu1), (sc2, mu2), z1, z2
    return out


# ------------------------------------------------------ enhancement
def _residual_block(P, pre, x):
    """ResidualBlock (compressai/layers/layers.py:125-147): conv3x3 -> LeakyReLU -> conv3x3 -> LeakyReLU, + x."""
    t = F.leaky_relu(F.conv2d(x, P[pre + "conv1.weight"], P[pre + "conv1.bias"], padding=1), 0.01)
    t = F.leaky_relu(F.conv2d(t, P[pre + "conv2.weight"], P[pre + "conv2.bias"], padding=1), 0.01)
    return t + x


def enhancement(P, pre, x, x_other_warp):
    """Enhancement.forward (ywz/mywork/newnet1.py:288-311): conv3x3(6->32), 3 blocks of 3 residual blocks (each block
    with its own skip, :272-286), conv3x3(32->3), + x."""
    t = F.conv2d(torch.cat((x, x_other_warp), 1), P[pre + "conv1.weight"], P[pre + "conv1.bias"], padding=1)
    for eb in ("EB1.", "EB2.", "EB3."):
        u = t
        for rb in ("RB1.", "RB2.", "RB3."):
            u = _residual_block(P, pre + eb + rb, u)
        t = u + t
    return F.conv2d(t, P[pre + "conv2.weight"], P[pre + "conv2.bias"], padding=1) + x


def independent_en(P, x1_hat, x2_hat, Hm, align_corners=True):
    """Independent_EN.forward (newnet1.py:1278-1300): each view is enhanced with the other view warped onto it."""
    size = x1_hat.shape[-2:]
    x1w = warp_perspective(x1_hat, Hm, size, align_corners)
    x2w = warp_perspective(x2_hat, torch.inverse(Hm), size, align_corners)      # fp32 inverse, as the reference (:1290)
    return {"x1_hat": enhancement(P, "EH1.", x1_hat, x2w), "x2_hat": enhancement(P, "EH2.", x2_hat, x1w)}


# --------------------------------------------------------------- loss and metrics
# --------------------------------------------------------------------------- real bit-stream (HSIC.compress / decompress)
def compress_cdf_tables(scales, means, weights, channels, minmax, K, M):
    """The per-pixel cumulative-frequency tables ``HSIC.compress`` feeds its range coder (newnet1.py:925-978; the
    decoder repeats it, :1137-1175), for image 0: (len(channels), H, W, 2*minmax+2) uint32.  Same operations in the same
    order as the reference: torch fp32 pmf of the K-mixture over the shifted alphabet, then numpy float32 clip / sum /
    round / add.accumulate."""
    import numpy as np
    H, W = scales.shape[-2:]
    samples = torch.arange(0, minmax * 2 + 1, dtype=torch.float32).reshape(-1, 1, 1).expand(-1, H, W)
    out = np.zeros((len(channels), H, W, 2 * minmax + 2), dtype=np.uint32)
    for j, ch in enumerate(channels):
        idx = [int(ch) + k * M for k in range(K)]
        sig, mu, w = scales[0, idx], means[0, idx] + minmax, weights.reshape(weights.shape[0], -1)[0, idx]
        pmf = None
        for k in range(K):
            v = (samples - mu[k]).abs()
            s_ = lower_bound(sig[k], SCALE_BOUND)
            t = (std_cumulative((0.5 - v) / s_) - std_cumulative((-0.5 - v) / s_)) * w[k]
            pmf = t if pmf is None else pmf + t
        pmf = pmf.numpy()
        for h in range(H):
            for x in range(W):
                pc = np.clip(pmf[:, h, x], 1.0 / 65536, 1.0)
                pc = np.round(pc / np.sum(pc) * 65536)
                out[j, h, x, 1:] = np.add.accumulate(pc).astype(np.uint32)
    return out


# --------------------------------------------------------------------------- in front of the path: HomographyNet -> h_matrix
def homography_net(P, a, b, return_features=False):
    """``Net.forward`` in eval mode (ywz/mywork/model.py:73-101): cat -> 4 Blocks (conv3x3+ReLU twice, MaxPool2d(2,2)
    after the first three, :50-71) -> flatten (NCHW order) -> Linear+ReLU -> Linear -> (B,4,2) corner deltas.
    Dropout (:80,84) is the identity in eval mode.  ``P`` holds the reference's state-dict keys."""
    x = torch.cat((a, b), 1)
    for blk in range(4):
        pre = f"cnn.{blk}.layers."
        x = F.relu(F.conv2d(x, P[pre + "0.weight"], P[pre + "0.bias"], padding=1))
        x = F.relu(F.conv2d(x, P[pre + "2.weight"], P[pre + "2.bias"], padding=1))
        if blk < 3:
            x = F.max_pool2d(x, 2, 2)
    feat = x
    x = F.relu(F.linear(x.flatten(1), P["fc.2.weight"], P["fc.2.bias"]))
    x = F.linear(x, P["fc.5.weight"], P["fc.5.bias"])
    delta = x.view(-1, 4, 2)
    return (delta, feat) if return_features else delta


def get_perspective_transform(src, dst):
    """kornia.get_perspective_transform(src, dst) (call sites model.py:26,108; newtrain1_real.py:116): the (B,3,3) H
    with dst ~ H src from 4 point pairs -- third-party and unpinned like warp_perspective, restated from its published
    definition: the 8x8 DLT system with h33 = 1 (rows [x y 1 0 0 0 -xu -yu | u], [0 0 0 x y 1 -xv -yv | v])."""
    src, dst = src.double(), dst.double()
    B = src.shape[0]
    A = torch.zeros(B, 8, 8, dtype=torch.float64)
    rhs = torch.zeros(B, 8, dtype=torch.float64)
    for i in range(4):
        x, y, u, v = src[:, i, 0], src[:, i, 1], dst[:, i, 0], dst[:, i, 1]
        A[:, 2 * i, 0], A[:, 2 * i, 1], A[:, 2 * i, 2] = x, y, 1.0
        A[:, 2 * i, 6], A[:, 2 * i, 7] = -x * u, -y * u
        A[:, 2 * i + 1, 3], A[:, 2 * i + 1, 4], A[:, 2 * i + 1, 5] = x, y, 1.0
        A[:, 2 * i + 1, 6], A[:, 2 * i + 1, 7] = -x * v, -y * v
        rhs[:, 2 * i], rhs[:, 2 * i + 1] = u, v
    h = torch.linalg.solve(A, rhs)
    return torch.cat((h, torch.ones(B, 1, dtype=torch.float64)), 1).view(B, 3, 3).float()


def h_adjust(orishapea, orishapeb, resizeshapea, resizeshapeb, h):
    """newtrain1_real.py:47-57 (in-place there): rescale an H estimated in the pic_size frame to the image frame --
    including the reference's quirk of scaling the x row by the HEIGHT ratio."""
    a, b = orishapea / resizeshapea, orishapeb / resizeshapeb
    h = h.clone()
    h[:, 0, :] = a * h[:, 0, :]
    h[:, :, 0] = (1. / a) * h[:, :, 0]
    h[:, 1, :] = b * h[:, 1, :]
    h[:, :, 1] = (1. / b) * h[:, :, 1]
    return h


def h_matrix_from_delta(corners, delta, img_h, img_w, pic_size):
    """newtrain1_real.py:113-123: the h_matrix handed to HSIC.forward, from HomographyNet's corner deltas."""
    c0 = corners - corners[:, 0].view(-1, 1, 2)
    h = get_perspective_transform(c0, c0 + delta)
    return h_adjust(img_h, img_w, pic_size, pic_size, torch.inverse(h))


def rd_loss(out, x1, x2, lmbda):
    """RateDistortionLoss (ywz/mywork/newtrain1.py:37-56)."""
    n, _, h, w = x1.shape
    bpp = sum(torch.log(l).sum() / (-math.log(2) * n * h * w) for l in out["likelihoods"].values())
    mse = F.mse_loss(out["x1_hat"], x1) + F.mse_loss(out["x2_hat"], x2)
    return {"bpp_loss": bpp, "mse_loss": mse, "loss": lmbda * 255 ** 2 * mse + bpp}


def metrics(out, x1, x2):
    """PSNR / bpp conventions (ywz/mywork/test3real.py:69-72,110-122; newtrain1.py:141-142)."""
    n, _, h, w = x1.shape
    bits = {k: float(torch.log(v.double()).sum() / -math.log(2)) for k, v in out["likelihoods"].items()}
    mse1 = float(F.mse_loss(out["x1_hat"].double(), x1.double()))
    mse2 = float(F.mse_loss(out["x2_hat"].double(), x2.double()))
    psnr1, psnr2 = 10 * math.log10(1 / mse1), 10 * math.log10(1 / mse2)
    bpp_loss = sum(bits.values()) / (n * h * w)
    return {"bits": bits, "bpp_loss": bpp_loss, "bpp": bpp_loss / 2, "mse1": mse1, "mse2": mse2,
            "psnr1": psnr1, "psnr2": psnr2, "psnr": (psnr1 + psnr2) / 2}


MS_SSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def ms_ssim(x, y, data_range=1.0, win_size=11, win_sigma=1.5, weights=MS_SSIM_WEIGHTS, K=(0.01, 0.03)):
    """``pytorch_msssim.ms_ssim(X, Y, data_range, size_average=False)`` as the reference's evaluation calls it
    (ywz/mywork/test3real.py:107-109): per-image multi-scale SSIM (Wang, Simoncelli, Bovik 2003), (N,) tensor.
    THIRD PARTY, absent and unpinned -> restated from the published algorithm (parity unpinned for this metric):
    a normalised 11-tap Gaussian (sigma 1.5) applied separably per channel WITHOUT padding; per scale
    cs = (2 s_xy + C2) / (s_x^2 + s_y^2 + C2), ssim = (2 mu_x mu_y + C1) / (mu_x^2 + mu_y^2 + C1) * cs, averaged over the
    valid positions per channel; between scales a 2 x 2 average pool (zero padding of the odd sides, divisor always 4);
    the first four scales contribute relu(cs)^w, the last relu(ssim)^w; product over scales, mean over channels."""
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("ms_ssim: two (N, C, H, W) tensors of the same shape")
    if min(x.shape[-2:]) <= (win_size - 1) * 2 ** 4:
        raise ValueError("ms_ssim: the smaller image side must exceed (win_size - 1) * 2^4 = %d" % ((win_size - 1) * 16))
    x, y = x.double(), y.double()
    co = torch.arange(win_size, dtype=torch.float32) - win_size // 2
    g = torch.exp(-(co ** 2) / (2 * win_sigma ** 2))
    g = (g / g.sum()).double()
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    ch = x.shape[1]
    wh, wv = g.reshape(1, 1, 1, -1).repeat(ch, 1, 1, 1), g.reshape(1, 1, -1, 1).repeat(ch, 1, 1, 1)

    def blur(t):
        return F.conv2d(F.conv2d(t, wh, groups=ch), wv, groups=ch)

    vals = []
    for i in range(len(weights)):
        mx, my = blur(x), blur(y)
        sxx, syy, sxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
        cs_map = (2 * sxy + C2) / (sxx + syy + C2)
        ssim_map = (2 * mx * my + C1) / (mx * mx + my * my + C1) * cs_map
        if i < len(weights) - 1:
            vals.append(torch.relu(cs_map.flatten(2).mean(-1)))
            pad = [s % 2 for s in x.shape[2:]]
            x, y = F.avg_pool2d(x, 2, padding=pad), F.avg_pool2d(y, 2, padding=pad)
        else:
            vals.append(torch.relu(ssim_map.flatten(2).mean(-1)))
    v = torch.stack(vals, 0)                                        # (levels, N, C)
    w = torch.tensor(weights, dtype=torch.float64).reshape(-1, 1, 1)
    return torch.prod(v ** w, 0).mean(1)


def aux_loss(P):
    """CompressionModel.aux_loss (newnet1.py:56-62)."""
    return eb_aux_loss(P, "entropy_bottleneck1.") + eb_aux_loss(P, "entropy_bottleneck2.")
