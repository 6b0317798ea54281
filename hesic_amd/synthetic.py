"""Deterministic synthetic weights and stereo pairs.

Nothing here depends on torch's RNG: every tensor is drawn from a NumPy PCG64
stream keyed by ``crc32(name)`` (weights) or by the pair index (images), so the
golden fixtures made in the reference container, the oracle, the HIP path and
``bench.py`` all see the very same numbers on any machine.

Weights are *not* the reference's default init (that leaves |y| < 0.5 so every
latent rounds to 0 and parity checks would be vacuous); gains are chosen so that
y spreads over a few integer bins and z over a couple, with non-zero EB medians
and factors so every term of the entropy models is exercised.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(name: str, salt: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) + 7919 * salt) & 0xFFFFFFFF))


def _uniform(name, shape, lo, hi, salt=0):
    r = _rng(name, salt)
    return torch.from_numpy(r.uniform(lo, hi, size=tuple(shape)).astype(np.float32))


# gain applied on top of the fan-in scaled uniform, by state-dict key suffix
_GAINS = (
    ("g_a_conv4.weight", 2.2),    # no GDN after it: sets the spread of y
    ("g_a_conv1.weight", 2.5),
    ("g_a_conv", 1.6),
    ("g_s_conv1.weight", 0.45),   # IGDN grows ~cubically: keep the synthesis side tame
    ("g_s_conv4.weight", 0.35),
    ("g_s_conv", 0.6),
    ("pre_conv.weight", 1.5),
    ("after_conv.weight", 0.8),
    ("encode_hyper.4.weight", 1.6),  # last h_a conv of HESIC: spread of z
    ("encode_hyper", 1.2),
    ("h_a1.4.weight", 1.6), ("h_a2.4.weight", 1.6), ("h_a", 1.2),
    ("gmm_sigma.4", 0.9), ("gmm_means.4", 0.8), ("gmm_weights.5", 0.25),
    ("gmm_sigma", 1.2), ("gmm_means", 1.2), ("gmm_weights", 1.2),
    ("h_s", 1.2), ("entropy_parameters", 1.3), ("context_prediction", 1.0),
)


def _gain(name: str) -> float:
    for key, g in _GAINS:
        if key in name:
            return g
    return 1.0


def fill_state_dict_(sd: dict, salt: int = 0) -> dict:
    """Overwrite every learnable/buffer tensor of a HESIC / HESIC+ state-dict in place.

    Works on the reference's own ``HSIC().state_dict()`` and on ours (same keys,
    SURVEY.md §8b), which is what makes the fixtures transferable.
    """
    for name, t in sd.items():
        if t.numel() == 0 or not t.dtype.is_floating_point:
            continue
        leaf = name.rsplit(".", 1)[-1]
        if ".beta_reparam." in name or ".gamma_reparam." in name or leaf in ("bound", "pedestal"):
            continue  # constants of the reparametrisation / LowerBound
        if leaf in ("target", "scale_table", "scale_bound", "mask"):
            continue
        if leaf == "beta":      # GDN beta (reparam domain): beta' in [0.6, 1.6]
            v = _uniform(name, t.shape, 0.6, 1.6, salt)
            t.copy_(torch.sqrt(v + 2.0 ** -36))
        elif leaf == "gamma":   # GDN gamma (reparam domain): 0.1*I + small dense positive part
            c = t.shape[0]
            v = _uniform(name, t.shape, 0.0, 0.02 * min(1.0, 16.0 / c), salt) + 0.1 * torch.eye(c)
            t.copy_(torch.sqrt(v + 2.0 ** -36))
        elif leaf == "quantiles":
            med = _uniform(name, (t.shape[0],), -0.4, 0.4, salt)
            t[:, 0, 0] = med - 10.0
            t[:, 0, 1] = med
            t[:, 0, 2] = med + 10.0
        elif "_matrices" in name:
            t.add_(_uniform(name, t.shape, -0.3, 0.3, salt))
        elif "_biases" in name:
            t.copy_(_uniform(name, t.shape, -0.5, 0.5, salt))
        elif "_factors" in name:
            t.copy_(_uniform(name, t.shape, -0.4, 0.4, salt))
        elif leaf == "weight" and t.dim() == 4:
            # Conv2d (Cout,Cin,kh,kw) / ConvTranspose2d (Cin,Cout,kh,kw): fan-in of the op
            transposed = ("g_s_conv" in name or "after_conv" in name or
                          _is_deconv_key(name))
            fan_in = (t.shape[0] if transposed else t.shape[1]) * t.shape[2] * t.shape[3]
            if transposed and t.shape[2] > 1:
                fan_in = fan_in / 4.0 if "after_conv" not in name else fan_in  # stride-2: ~1/4 of taps hit
            a = _gain(name) * (3.0 / fan_in) ** 0.5
            t.copy_(_uniform(name, t.shape, -a, a, salt))
        elif leaf == "bias":
            t.copy_(_uniform(name, t.shape, -0.1, 0.1, salt))
        else:
            t.copy_(_uniform(name, t.shape, -0.1, 0.1, salt))
    return sd


def init_reference_defaults_(sd: dict) -> dict:
    """Give a zero-filled state-dict the reference's constructor constants (the entries
    ``fill_state_dict_`` deliberately leaves alone or perturbs additively):
    EB matrices log(expm1(1/scale/f)) (entropy_models.py:275-279), LowerBound bounds,
    pedestals, EB targets."""
    import math
    scale = 10.0 ** (1 / 5)
    filt = (1, 3, 3, 3, 3, 1)
    for name, t in sd.items():
        leaf = name.rsplit(".", 1)[-1]
        if t.numel() == 0:
            continue
        if "_matrices." in name:
            i = int(leaf)
            t.fill_(math.log(math.expm1(1 / scale / filt[i + 1])))
        elif leaf == "pedestal":
            t.fill_(2.0 ** -36)
        elif leaf == "bound":
            if "beta_reparam" in name:
                t.fill_((1e-6 + 2.0 ** -36) ** 0.5)
            elif "gamma_reparam" in name:
                t.fill_(2.0 ** -18)
            elif "lower_bound_scale" in name:
                t.fill_(0.11)
            else:
                t.fill_(1e-9)
        elif leaf == "target":
            v = math.log(2 / 1e-9 - 1)
            t.copy_(torch.tensor([-v, 0.0, v]))
        elif leaf == "scale_bound":
            t.fill_(0.11)
    return sd


# ConvTranspose2d layers inside the hyper-synthesis Sequentials (index within the Sequential)
_DECONV_KEYS = (
    "_h_s1.gmm_sigma.0.", "_h_s1.gmm_sigma.2.", "_h_s1.gmm_means.0.", "_h_s1.gmm_means.2.",
    "_h_s1.gmm_weights.0.", "_h_s1.gmm_weights.2.",
    "h_s1.0.", "h_s1.2.", "h_s2.0.", "h_s2.2.",
)


def _is_deconv_key(name: str) -> bool:
    return any(name.startswith(k) for k in _DECONV_KEYS)


def homography(seed: int) -> np.ndarray:
    """Mild projective src->dst pixel map of SURVEY.md §8d, 3x3 float64."""
    r = np.random.Generator(np.random.PCG64(1000003 + seed))
    a, d = r.uniform(-0.02, 0.02, 2)
    b, c = r.uniform(-0.01, 0.01, 2)
    tx = r.uniform(-16, 16)
    ty = r.uniform(-4, 4)
    e, f = r.uniform(-2e-5, 2e-5, 2)
    return np.array([[1 + a, b, tx], [c, 1 + d, ty], [e, f, 1.0]], dtype=np.float64)


def _box3(img: np.ndarray) -> np.ndarray:
    p = np.pad(img, ((0, 0), (1, 1), (1, 1)), mode="edge")
    out = np.zeros_like(img)
    for dy in range(3):
        for dx in range(3):
            out += p[:, dy:dy + img.shape[1], dx:dx + img.shape[2]]
    return out / 9.0


def _warp_np(src: np.ndarray, M: np.ndarray) -> np.ndarray:
    """Exact inverse-map bilinear warp, zeros outside (float64, CHW)."""
    C, H, W = src.shape
    Minv = np.linalg.inv(M)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    X = Minv[0, 0] * xs + Minv[0, 1] * ys + Minv[0, 2]
    Y = Minv[1, 0] * xs + Minv[1, 1] * ys + Minv[1, 2]
    Z = Minv[2, 0] * xs + Minv[2, 1] * ys + Minv[2, 2]
    sx, sy = X / Z, Y / Z
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = sx - x0, sy - y0
    out = np.zeros_like(src)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            v = src[:, np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
            out += v * (wx * wy * ok)[None]
    return out


def stereo_pair(seed: int, height: int, width: int):
    """(x1, x2, H) for pair ``seed``: x1 low-passed U[0,1), x2 = warp(x1,H)+N(0,0.01)."""
    r = np.random.Generator(np.random.PCG64(seed))
    x1 = _box3(r.uniform(0.0, 1.0, size=(3, height, width)))
    H = homography(seed)
    x2 = np.clip(_warp_np(x1, H) + r.normal(0.0, 0.01, size=x1.shape), 0.0, 1.0)
    return x1.astype(np.float32), x2.astype(np.float32), H.astype(np.float32)


def _upsample_linear(grid: np.ndarray, height: int, width: int) -> np.ndarray:
    """(C, h, w) coarse grid -> (C, height, width), separable linear interpolation with the grid's corners on the image corners."""
    C, h, w = grid.shape
    ys, xs = np.linspace(0.0, h - 1.0, height), np.linspace(0.0, w - 1.0, width)
    y0, x0 = np.minimum(ys.astype(np.int64), h - 2), np.minimum(xs.astype(np.int64), w - 2)
    fy, fx = (ys - y0)[None, :, None], (xs - x0)[None, None, :]
    rows = grid[:, y0, :] * (1 - fy) + grid[:, y0 + 1, :] * fy
    return rows[:, :, x0] * (1 - fx) + rows[:, :, x0 + 1] * fx


def smooth_stereo_pair(seed: int, height: int, width: int):
    """(x1, x2, H) for a PIECEWISE-SMOOTH pair (round 5): the content a trained codec sees on photographs, where ``stereo_pair``'s
    low-passed noise is close to incompressible (a model trained on it stops near 20 dB).  x1 = a luminance field interpolated from a
    coarse (one knot per 64 pixels) random grid with a weaker per-channel chroma field, plus four soft-edged regions (discs / half
    planes) that shift the level by up to 0.2, plus 3x3-boxed texture of amplitude 0.02; x2 = warp(x1, H) + N(0, 0.004).  PCG64 seeded
    like ``stereo_pair`` (pair i = seed i), same homographies."""
    r = np.random.Generator(np.random.PCG64(7_000_003 + seed))
    gh, gw = max(2, height // 64 + 1), max(2, width // 64 + 1)
    lum = _upsample_linear(r.uniform(0.2, 0.8, size=(1, gh, gw)), height, width)
    chroma = _upsample_linear(r.uniform(-0.12, 0.12, size=(3, gh, gw)), height, width)
    img = lum + chroma
    ys, xs = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    for _ in range(4):
        kind, cy, cx = r.integers(0, 2), r.uniform(0, height), r.uniform(0, width)
        level = r.uniform(-0.2, 0.2, size=(3, 1, 1)) * np.array([1.0, 0.9, 0.8]).reshape(3, 1, 1)
        if kind == 0:
            rad = r.uniform(0.08, 0.3) * min(height, width)
            d = rad - np.sqrt((ys - cy) ** 2 + (xs - cx) ** 2)
        else:
            th = r.uniform(0, 2 * np.pi)
            d = (ys - cy) * np.sin(th) + (xs - cx) * np.cos(th)
        img = img + level * np.clip(d / 3.0 + 0.5, 0.0, 1.0)[None]          # a 3-pixel ramp: an edge, not a step the warp would alias
    img = img + 0.02 * _box3(r.uniform(-1.0, 1.0, size=(3, height, width)))
    x1 = np.clip(img, 0.0, 1.0)
    H = homography(seed)
    x2 = np.clip(_warp_np(x1, H) + r.normal(0.0, 0.004, size=x1.shape), 0.0, 1.0)
    return x1.astype(np.float32), x2.astype(np.float32), H.astype(np.float32)


def smooth_stereo_batch(first_seed: int, batch: int, height: int, width: int):
    xs1, xs2, hs = zip(*(smooth_stereo_pair(first_seed + i, height, width) for i in range(batch)))
    return (torch.from_numpy(np.stack(xs1)), torch.from_numpy(np.stack(xs2)),
            torch.from_numpy(np.stack(hs)))


def stereo_batch(first_seed: int, batch: int, height: int, width: int):
    xs1, xs2, hs = zip(*(stereo_pair(first_seed + i, height, width) for i in range(batch)))
    return (torch.from_numpy(np.stack(xs1)), torch.from_numpy(np.stack(xs2)),
            torch.from_numpy(np.stack(hs)))


# ------------------------------------------------------------------ HomographyNet
def fill_homography_state_dict_(sd: dict, salt: int = 0) -> dict:
    """Name-keyed fill of a HomographyNet state-dict (reference ``model.Net`` keys: cnn.N.layers.{0,2}.*, fc.{2,5}.*):
    He-style fan-in scaled uniform weights so activations survive the 8 ReLU layers, small biases, and a last layer
    scaled so the corner deltas come out at a few pixels."""
    for name, t in sd.items():
        if not t.dtype.is_floating_point or t.numel() == 0:
            continue
        if name.endswith(".weight"):
            fan = t[0].numel()
            a = (6.0 / fan) ** 0.5 * (4.0 if name.startswith("fc.5") else 1.0)
            t.copy_(_uniform("homo." + name, t.shape, -a, a, salt))
        else:
            t.copy_(_uniform("homo." + name, t.shape, -0.05, 0.05, salt))
    return sd


def homography_batch(seed: int, batch: int, patch: int = 128, rho: float = 32.0):
    """Grey patch pairs (B,1,patch,patch) in [0,1) and the 4 patch corners (B,4,2) in the pic-frame, the inputs
    ``Net.forward`` / the h_matrix derivation take (compressai/datasets/utils.py:161-186 produces them from images)."""
    a = _uniform(f"homo.a{seed}", (batch, 1, patch, patch), 0.0, 1.0)
    b = (0.7 * a.roll(3, -1) + 0.3 * _uniform(f"homo.b{seed}", (batch, 1, patch, patch), 0.0, 1.0))
    tl = _uniform(f"homo.tl{seed}", (batch, 1, 2), rho, 2 * rho).round()
    box = torch.tensor([[0.0, 0.0], [patch, 0.0], [patch, patch], [0.0, patch]])
    return a, b, tl + box
