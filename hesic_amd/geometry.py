"""The two ``kornia`` calls of the HESIC path (third party, neither vendored by the reference nor
installed here): ``warp_perspective`` (hot path, ywz/mywork/newnet1.py:746,753,767) and
``get_perspective_transform`` (the step before it in the ``_real`` scripts,
ywz/mywork/newtrain1_real.py:113-131).

Semantics: kornia's default changed across releases and the reference does not pin a
version.  ``DEFAULT_ALIGN_CORNERS = True`` selects the exact inverse-map bilinear resample
(kornia >= 0.5, == cv2.warpPerspective); pass ``align_corners=False`` for the <= 0.4 behaviour.
"""
import torch

from . import functional as Fn

# Checkpoints trained with the reference's pinned environment (torch 1.6.0 / unpinned kornia of that era, Readme.md:11,16:
# kornia 0.4.x, whose warp_perspective defaulted to align_corners=False) expect the legacy sampling: set this to False (or
# HESIC_WARP_ALIGN_CORNERS=0) for them.  True = kornia >= 0.5 = cv2.warpPerspective.
import os as _os
DEFAULT_ALIGN_CORNERS = _os.environ.get("HESIC_WARP_ALIGN_CORNERS", "1") not in ("0", "false", "False")
_CONVENTION_CHOSEN = "HESIC_WARP_ALIGN_CORNERS" in _os.environ      # True once the user has picked a convention (env or the call below)


def use_reference_era_warp(enable=True):
    """Pick the warp convention explicitly.  ``enable=True``: kornia <= 0.4.x sampling (``align_corners=False``) -- what checkpoints
    trained in the reference's pinned environment (torch 1.6.0, ``pip install kornia`` of that time; Readme.md:11,16) were trained
    with.  ``enable=False``: kornia >= 0.5 / cv2.warpPerspective (``align_corners=True``, this package's default).  Returns the
    previous ``DEFAULT_ALIGN_CORNERS``."""
    global DEFAULT_ALIGN_CORNERS, _CONVENTION_CHOSEN
    prev, DEFAULT_ALIGN_CORNERS, _CONVENTION_CHOSEN = DEFAULT_ALIGN_CORNERS, not bool(enable), True
    return prev


def warp_convention():
    return "kornia>=0.5 (align_corners=True)" if DEFAULT_ALIGN_CORNERS else "kornia<=0.4 (align_corners=False)"


def warp_perspective(src, M, dsize, flags="bilinear", border_mode=None, align_corners=None, inverse_map=False):
    """dst(x', y') = bilinear(src, M^-1 (x', y', 1)), zeros outside.  src (B,C,H,W), M (B,3,3) maps source
    pixels to destination pixels, dsize = (H_out, W_out).  HIP kernel: csrc/warp.hip."""
    if flags != "bilinear":
        raise NotImplementedError("hesic_amd.warp_perspective: bilinear only")
    if border_mode not in (None, "zeros"):
        raise NotImplementedError("hesic_amd.warp_perspective: zero padding only")
    if src.dim() != 4 or M.shape[-2:] != (3, 3):
        raise ValueError(f"warp_perspective: expected src (B,C,H,W) and M (B,3,3), got {tuple(src.shape)}, {tuple(M.shape)}")
    ac = DEFAULT_ALIGN_CORNERS if align_corners is None else bool(align_corners)
    if M.shape[0] != src.shape[0]:
        M = M.expand(src.shape[0], 3, 3)
    from . import handover as _ho
    src = _ho.plain(src)
    # a warped reconstruction is still a reconstruction: the analysis stack that reads it runs on single operands (handover.py)
    return _ho.inherit_tags(Fn.warp_perspective(src, M, dsize, ac, inverse_map), src)


def get_perspective_transform(src, dst):
    """4-point DLT: (B,4,2) x (B,4,2) -> (B,3,3) with M @ [x,y,1] ~ [x',y',1].  An 8x8 solve per pair --
    host-side plumbing upstream of the path."""
    if src.shape[-2:] != (4, 2) or dst.shape != src.shape:
        raise ValueError("get_perspective_transform: expected (B,4,2) point sets")
    B = src.shape[0]
    x, y, u, v = src[..., 0], src[..., 1], dst[..., 0], dst[..., 1]
    zeros, ones = torch.zeros_like(x), torch.ones_like(x)
    ax = torch.stack([x, y, ones, zeros, zeros, zeros, -x * u, -y * u], -1)
    ay = torch.stack([zeros, zeros, zeros, x, y, ones, -x * v, -y * v], -1)
    A = torch.cat([ax, ay], 1).double()
    b = torch.cat([u, v], 1).double().unsqueeze(-1)
    h = torch.linalg.solve(A, b).squeeze(-1)
    return torch.cat([h, torch.ones(B, 1, dtype=h.dtype, device=h.device)], 1).reshape(B, 3, 3).to(src.dtype)
