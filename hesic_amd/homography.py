"""HomographyNet and the h_matrix derivation -- the step immediately in front of ``HSIC.forward`` in the reference's
``_real`` scripts: ``Net`` of ``ywz/mywork/model.py:73-111`` and ``newtrain1_real.py:113-123``.

Same module tree and state-dict keys as the reference (``cnn.N.layers.{0,2}.{weight,bias}``, ``fc.{2,5}.*``), so a
``homo_best.pth.tar`` checkpoint loads strictly.  Inference only, like its use on the path (the reference keeps it
frozen, ``newtrain1_real.py:78``): every layer runs on the HIP kernels of ``libhesic_hip.so`` --

* conv3x3 + ReLU: the narrow (Cin = 2) / implicit-GEMM conv kernels with the activation fused;
* MaxPool2d(2,2): ``hesic_maxpool2_forward`` on the NHWC map;
* Linear: a 1x1 implicit-GEMM conv over the NHWC-flattened map (the first Linear's columns are permuted once from the
  reference's NCHW flatten order; the low-resolution split-K launch spreads its 32768-deep contraction over the GPU);
* corner deltas -> h_matrix: ``hesic_h_from_delta`` (4-point DLT, 3x3 inverse and the reference's ``h_adjust``).

Feature maps are fp32 by default (the deltas are pixel offsets that steer a full-resolution warp); pass
``dtype=torch.bfloat16`` for bf16 storage.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as Fn
from .compressai.models.utils import HipConv2d

__all__ = ["Net", "Block", "Flatten", "max_pool2", "get_perspective_transform", "h_matrix_from_delta", "h_matrix"]


def max_pool2(x):
    """nn.MaxPool2d(2, 2) on a channels_last map (model.py:62-63)."""
    L.require_cuda(x)
    x = x.contiguous(memory_format=torch.channels_last)
    B, C, H, W = x.shape
    y = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    L.call("hesic_maxpool2_forward", L.ptr(x), L.ptr(y), B, H, W, C, L.dt(x), L.stream())
    return y


class _HipMaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        if self.kernel_size not in (2, (2, 2)) or self.stride not in (2, (2, 2)) or self.padding not in (0, (0, 0)):
            raise NotImplementedError("hesic_amd max pool: 2x2 stride 2")
        return max_pool2(x)


class Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


class Block(nn.Module):
    """conv3x3 + ReLU, conv3x3 + ReLU, [MaxPool2d(2,2)] (model.py:50-71; batch_norm=False is what the scripts use)."""

    def __init__(self, inchannels, outchannels, batch_norm=False, pool=True):
        super().__init__()
        if batch_norm:
            raise NotImplementedError("HomographyNet on the HIP path: batch_norm=False (the configuration the reference trains)")
        layers = [HipConv2d(inchannels, outchannels, kernel_size=3, padding=1), nn.ReLU(),
                  HipConv2d(outchannels, outchannels, kernel_size=3, padding=1), nn.ReLU()]
        if pool:
            layers.append(_HipMaxPool2d(2, 2))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        x = self.layers[0].run(x, act=L.ACT_RELU)
        x = self.layers[2].run(x, act=L.ACT_RELU)
        return self.layers[4](x) if len(self.layers) > 4 else x


class Net(nn.Module):
    """HomographyNet: two grey patches -> (B,4,2) corner deltas (model.py:73-101)."""

    def __init__(self, batch_norm=False, patch_size=128, dtype=torch.float32):
        super().__init__()
        self.cnn = nn.Sequential(Block(2, 64, batch_norm), Block(64, 64, batch_norm), Block(64, 128, batch_norm),
                                 Block(128, 128, batch_norm, pool=False))
        self.side = patch_size // 8
        self.fc = nn.Sequential(Flatten(), nn.Dropout(p=0.5), nn.Linear(128 * self.side * self.side, 1024), nn.ReLU(),
                                nn.Dropout(p=0.5), nn.Linear(1024, 4 * 2))
        self.dtype = dtype
        self._fc_pack = [Fn.PackedWeight(), Fn.PackedWeight()]
        self._fc1_nhwc = None

    def _fc1_weight(self):
        """fc.2.weight with its columns moved from NCHW-flatten (c, y, x) to NHWC-flatten (y, x, c) order; cached."""
        w = self.fc[2].weight
        tag = (w.data_ptr(), w._version)
        if self._fc1_nhwc is None or self._fc1_nhwc[0] != tag:
            s = self.side
            wn = w.detach().view(-1, 128, s, s).permute(0, 2, 3, 1).reshape(w.shape[0], -1, 1, 1).contiguous()
            self._fc1_nhwc = (tag, wn)
        return self._fc1_nhwc[1]

    def forward(self, a, b):
        if self.training or torch.is_grad_enabled():
            raise RuntimeError("hesic_amd HomographyNet is inference-only (call under torch.no_grad() in eval mode)")
        L.require_cuda(a, b)
        x = torch.cat((a, b), dim=1).to(self.dtype)
        prev = Fn.compute_dtype()
        Fn.set_compute_dtype(self.dtype)            # storage type of the maps the narrow first conv produces
        try:
            x = self.cnn(x)                          # (B,128,side,side), NHWC in memory
            B = x.shape[0]
            x = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(B, -1, 1, 1)
            x = x.contiguous(memory_format=torch.channels_last)
            x = Fn.conv2d(x, self._fc1_weight(), self.fc[2].bias, kernel_size=1, stride=1, padding=0, act=L.ACT_RELU,
                          packer=self._fc_pack[0])
            w2 = self.fc[5].weight                   # 8 outputs: still the implicit-GEMM kernel (one padded cout tile)
            wp = self._fc_pack[1].get(w2.view(w2.shape[0], w2.shape[1], 1, 1), None, w2.shape[0], w2.shape[1], 1, 1, False,
                                      False, x.dtype)
            x = Fn._wide_conv(x, wp, self.fc[5].bias, B, 1, 1, w2.shape[1], 1, 1, w2.shape[0], 1, 1, 0, False)
        finally:
            Fn.set_compute_dtype(prev)
        return x.reshape(-1, 4, 2).float()

    def get_h(self, a, b, corners):
        """inverse(get_perspective_transform(corners, corners + delta)) (model.py:99-111)."""
        delta = self.forward(a, b)
        return h_matrix_from_delta(corners, delta, 1.0, 1.0, 1.0, subtract_origin=False)


def get_perspective_transform(src, dst):
    """kornia.get_perspective_transform (B,4,2),(B,4,2) -> (B,3,3) with dst ~ H src."""
    L.require_cuda(src, dst)
    src, dst = src.contiguous().float(), dst.contiguous().float()
    H = torch.empty((src.shape[0], 3, 3), dtype=torch.float32, device=src.device)
    L.call("hesic_perspective_transform", L.ptr(src), L.ptr(dst), L.ptr(H), src.shape[0], L.stream())
    return H


def h_matrix_from_delta(corners, delta, img_h, img_w, pic_size, subtract_origin=True):
    """newtrain1_real.py:113-123: corners0 = corners - corners[:,0]; h = gpt(corners0, corners0 + delta);
    h_matrix = h_adjust(img_h, img_w, pic_size, pic_size, inverse(h)) -- one kernel, no host round trip."""
    L.require_cuda(corners, delta)
    corners, delta = corners.contiguous().float(), delta.contiguous().float()
    H = torch.empty((corners.shape[0], 3, 3), dtype=torch.float32, device=corners.device)
    L.call("hesic_h_from_delta", L.ptr(corners), L.ptr(delta), float(img_h) / float(pic_size), float(img_w) / float(pic_size),
           int(subtract_origin), L.ptr(H), corners.shape[0], L.stream())
    return H


def h_matrix(net, homo_img1, homo_img2, homo_corners, img_h, img_w, pic_size=256):
    """The h_matrix of a stereo pair as the `_real` scripts derive it (newtrain1_real.py:108-123)."""
    with torch.no_grad():
        delta = net(homo_img1, homo_img2)
        return h_matrix_from_delta(homo_corners, delta, img_h, img_w, pic_size)
