"""MaskedConv2d and the small residual building blocks (reference: compressai/layers/layers.py)."""
import torch
import torch.nn as nn

from compressai.models.utils import HipConv2d
from hesic_amd import _lib as L
from hesic_amd import functional as Fn
from hesic_amd import handover as _ho

from .gdn import GDN


class MaskedConv2d(HipConv2d):
    r"""PixelCNN-style masked convolution (reference layers.py:21-45): mask 'A' zeroes the centre tap and
    everything after it in raster order, 'B' keeps the centre.  As in the reference the mask is folded
    into ``weight.data`` on every forward; the kernel additionally skips the dead taps (12 of 25 live
    for a 5x5 'A' mask), so no multiply-by-zero work is issued."""

    def __init__(self, *args, mask_type="A", **kwargs):
        super().__init__(*args, **kwargs)
        if mask_type not in ("A", "B"):
            raise ValueError(f'Invalid "mask_type" value "{mask_type}"')
        self.register_buffer("mask", torch.ones_like(self.weight.data))
        _, _, h, w = self.mask.size()
        self.mask[:, :, h // 2, w // 2 + (mask_type == "B"):] = 0
        self.mask[:, :, h // 2 + 1:] = 0
        live = (h // 2) * w + w // 2 + (mask_type == "B")
        self._tap_mask = (1 << live) - 1

    def _fold_mask(self):
        """``weight.data *= mask`` (reference layers.py:43).  Idempotent, so at inference it is skipped while the parameter's
        version counter stands where the last fold left it (a load_state_dict / optimiser step moves it)."""
        tag = (self.weight.data_ptr(), self.weight._version, Fn._cache_epoch)      # the epoch moves on raw-pointer updates (graph replays)
        if torch.is_grad_enabled() or getattr(self, "_folded", None) != tag:
            self.weight.data *= self.mask
            self._folded = (self.weight.data_ptr(), self.weight._version, Fn._cache_epoch)

    def forward(self, x):
        self._fold_mask()
        if _ho.active(x):
            return _ho.conv(self, x)
        return self.run(x, mask=self.mask, tap_mask=self._tap_mask)

    def forward_into(self, x, out, c_off):
        """``out[:, c_off:c_off + out_channels] = self(x)`` at inference (see ``HipConv2d.run_into``)."""
        self._fold_mask()
        return self.run_into(x, out, c_off, mask=self.mask, tap_mask=self._tap_mask)


def conv3x3(in_ch, out_ch, stride=1):
    """3x3 convolution with padding."""
    return HipConv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)


def conv1x1(in_ch, out_ch, stride=1):
    return HipConv2d(in_ch, out_ch, kernel_size=1, stride=stride)


def subpel_conv3x3(in_ch, out_ch, r=1):
    """3x3 sub-pixel convolution for up-sampling (Cheng2020 only; kept importable)."""
    return nn.Sequential(HipConv2d(in_ch, out_ch * r ** 2, kernel_size=3, padding=1), nn.PixelShuffle(r))


class ResidualBlock(nn.Module):
    """Two 3x3 convs + LeakyReLU with an identity / 1x1 skip (reference layers.py:125-147; used by the
    stage-2 enhancement net ``Independent_EN``, ywz/mywork/newnet1.py:272-311)."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None

    def forward(self, x, outer_skip=None):
        """``outer_skip``: an extra tensor added to the result (the Enhancement_Block's ``+ x``, newnet1.py:286), fused
        into the last conv's epilogue on the 32-channel inference path."""
        x = _ho.plain(x)
        if self.skip is None and Fn.conv3x3_c32_ok(x, self.conv1.weight) and Fn.conv3x3_c32_ok(x, self.conv2.weight):
            if Fn.RESBLOCK_FUSED:       # both convs, the identity and the outer skip in one launch: the intermediate map stays on the CU
                return Fn.resblock_c32(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias, act=L.ACT_LEAKY, res2=outer_skip)
            out = Fn.conv3x3_c32(x, self.conv1.weight, self.conv1.bias, act=L.ACT_LEAKY)
            return Fn.conv3x3_c32(out, self.conv2.weight, self.conv2.bias, act=L.ACT_LEAKY, res1=x, res2=outer_skip)
        if self.skip is None and Fn.conv3x3_c32_train_ok(x, self.conv1.weight) and Fn.conv3x3_c32_train_ok(x, self.conv2.weight):
            # training (stage 2): forward and data gradient of both convs on the 32-channel kernel
            out = Fn.conv3x3_c32_train(x, self.conv1.weight, self.conv1.bias, act=L.ACT_LEAKY)
            out = Fn.conv3x3_c32_train(out, self.conv2.weight, self.conv2.bias, act=L.ACT_LEAKY)
        else:
            out = self.conv1.run(x, act=L.ACT_LEAKY)
            out = self.conv2.run(out, act=L.ACT_LEAKY)
        identity = x if self.skip is None else self.skip(x)
        out = out + identity.to(out.dtype)
        return out if outer_skip is None else out + outer_skip.to(out.dtype)


class ResidualBlockWithStride(nn.Module):
    """Residual block with a stride on the first convolution (Cheng2020 only)."""

    def __init__(self, in_ch, out_ch, stride=2):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch, stride=stride)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.gdn = GDN(out_ch)
        self.skip = conv1x1(in_ch, out_ch, stride=stride) if stride != 1 or in_ch != out_ch else None

    def forward(self, x):
        out = self.conv1.run(x, act=L.ACT_LEAKY)
        out = self.gdn(self.conv2(out))
        identity = x if self.skip is None else self.skip(x)
        return out + identity.to(out.dtype)


class ResidualBlockUpsample(nn.Module):
    """Residual block with sub-pixel upsampling on the last convolution (Cheng2020 only)."""

    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(out_ch, out_ch)
        self.igdn = GDN(out_ch, inverse=True)
        self.upsample = subpel_conv3x3(in_ch, out_ch, upsample)

    def forward(self, x):
        out = self.leaky_relu(self.subpel_conv(x))
        out = self.igdn(self.conv(out))
        return out + self.upsample(x).to(out.dtype)


class AttentionBlock(nn.Module):
    """Self-attention block of Cheng2020 (simplified, no non-local block).  Not used by HESIC."""

    def __init__(self, N):
        super().__init__()

        class ResidualUnit(nn.Module):
            def __init__(self):
                super().__init__()
                self.conv = nn.Sequential(conv1x1(N, N // 2), nn.ReLU(inplace=True), conv3x3(N // 2, N // 2),
                                          nn.ReLU(inplace=True), conv1x1(N // 2, N))
                self.relu = nn.ReLU(inplace=True)

            def forward(self, x):
                return self.relu(self.conv(x) + x)

        self.conv_a = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit())
        self.conv_b = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit(), conv1x1(N, N))

    def forward(self, x):
        return self.conv_a(x) * torch.sigmoid(self.conv_b(x)) + x
