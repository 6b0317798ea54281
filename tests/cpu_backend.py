"""Test infrastructure: CPU stand-ins for the launches ``hesic_amd.handover`` and the drop-in modules issue, built from the oracle's
functions (fp32, torch CPU ops), so that the hand-over LOGIC -- which module call is deferred, what it fuses with, what a foreign operator
gets to see -- can be checked without a GPU.  The product never imports this file; on a GPU box the same calls reach the HIP kernels
(tests/test_gpu_path_a.py)."""
import contextlib

import torch
import torch.nn.functional as F

from oracle import hesic_oracle as O


def _act(y, act):
    from hesic_amd import _lib as L
    if act == L.ACT_RELU:
        return F.relu(y)
    if act == L.ACT_LEAKY:
        return F.leaky_relu(y, 0.01)
    return y


def _conv2d(x, weight, bias, *, kernel_size, stride, padding, transposed=False, act=0, in_abs=False, tap_mask=0, packer=None, mask=None):
    from hesic_amd import handover
    x = handover.plain(x).float()
    if in_abs:
        x = x.abs()
    w = weight.detach() if mask is None else weight.detach() * mask
    b = None if bias is None else bias.detach()
    if transposed:
        y = F.conv_transpose2d(x, w, b, stride=stride, padding=padding, output_padding=stride - 1)
    else:
        y = F.conv2d(x, w, b, stride=stride, padding=padding)
    return _act(y, act)


def _gdn(x, beta, gamma, inverse=False, beta_min=1e-6):
    return O.gdn(x.float(), beta.detach(), gamma.detach(), inverse, beta_min)


def _conv2d_into(x, weight, bias, out, out_c_off, *, kernel_size, stride, padding, transposed=False, act=0, packer=None, mask=None, tap_mask=0):
    y = _conv2d(x, weight, bias, kernel_size=kernel_size, stride=stride, padding=padding, transposed=transposed, act=act, mask=mask)
    out[:, out_c_off:out_c_off + y.shape[1]] = y
    return out


def _copy_into(x, out, off):
    out[:, off:off + x.shape[1]] = x
    return out


def _eb(z, matrices, biases, factors, quantiles, noise=None, packer=None, out_dtype=None, lik_bound=1e-9):
    P = {"eb.quantiles": quantiles.detach()}
    for i, m in enumerate(matrices):
        P[f"eb._matrices.{i}"] = m.detach()
    for i, b in enumerate(biases):
        P[f"eb._biases.{i}"] = b.detach()
    for i, f in enumerate(factors):
        P[f"eb._factors.{i}"] = f.detach()
    return O.eb_forward(P, "eb.", z.float())


def _gmm(y, scales, means, weights, K, noise=None, scale_bound=0.11, lik_bound=1e-9, out_dtype=None):
    return O.gmm_forward(y.float(), scales.float(), means.float(), weights.float(), K)


def _gc(y, scales, means=None, noise=None, scale_bound=0.11, lik_bound=1e-9, out_dtype=None):
    return O.gc_forward(y.float(), scales.float(), None if means is None else means.float())


def _warp(src, M, dsize, align_corners=True, inverse_map=False):
    assert not inverse_map
    return O.warp_perspective(src.float(), M, dsize, align_corners)


def _pooled_linear(pooled, weight, bias):
    return F.conv2d(pooled.float(), weight.detach(), None if bias is None else bias.detach())


CALLS = []


def _logged(name, fn):
    def wrapped(*a, **k):
        CALLS.append(name)
        return fn(*a, **k)
    return wrapped


def _logged_method(name, fn):
    def wrapped(self, *a, **k):
        CALLS.append(name)
        return fn(self, *a, **k)
    return wrapped


@contextlib.contextmanager
def emulate():
    """Route the launches through the oracle's CPU functions and let the hand-over run on CPU tensors; ``CALLS`` lists the launches made."""
    from hesic_amd import functional as Fn, handover
    stubs = {
        "conv2d": _conv2d, "gdn": _gdn, "_gdn_op": _gdn, "conv2d_into": _conv2d_into, "copy_into": _copy_into,
        "entropy_bottleneck": _eb, "gaussian_mixture": _gmm, "gaussian_conditional": _gc, "warp_perspective": _warp,
        "pooled_linear": _pooled_linear, "upsample4": lambda z: O.upsample_bilinear_x4(z.float()),
        "upsample4_cat": lambda z, y1: torch.cat((O.upsample_bilinear_x4(z.float()), y1.float()), 1),
        "spatial_max": lambda x, leaky=False: (lambda m: F.leaky_relu(m, 0.01) if leaky else m)(torch.amax(x.float(), dim=(2, 3), keepdim=True)),
        "round_to": lambda x, dtype: torch.round(x),
    }
    keep = {k: getattr(Fn, k) for k in stubs}
    keep_force = handover._FORCE_CPU
    CALLS.clear()
    for k, v in stubs.items():
        setattr(Fn, k, _logged(k, v))
    handover._FORCE_CPU = True
    # the modules' own entry points, logged by name: which FORM a module call took (fused conv + GDN, cat-free 6 -> 3, in-place slice write)
    from compressai.models.utils import HipConv2d, HipConvTranspose2d
    keep_m = []
    for cls in (HipConv2d, HipConvTranspose2d):
        for name in ("run", "run_gdn", "run_cat", "run_into", "run_latent"):
            if name in cls.__dict__:
                keep_m.append((cls, name, cls.__dict__[name]))
                setattr(cls, name, _logged_method(f"{cls.__name__}.{name}", cls.__dict__[name]))
    try:
        yield CALLS
    finally:
        for cls, name, fn in keep_m:
            setattr(cls, name, fn)
        handover._FORCE_CPU = keep_force
        for k, v in keep.items():
            setattr(Fn, k, v)
