"""wgrad_row_kernel (csrc/wgrad.hip, round 5): the weight gradient of the 5x5 stride-2 wide convs with one kernel ROW of taps per block
(autograd's wgrad of conv() / deconv(), compressai/models/utils.py:104-118, as run by newtrain1.py:85-96).  Checked against torch's
fp32 weight / bias gradient of the same bf16-rounded operands, and against wgrad_tr_kernel (one tap per block) on the same call: both sum
fp32 products of the same 16-bit values, only the order of the K slices differs."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from hesic_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"

# Cin, Cout, transposed, (B, H, W) of the conv input (5x5, stride 2, pad 2; 64-pixel stages need QW % 64 == 0)
CASES = [
    (128, 128, 0, (2, 128, 128)),
    (128, 128, 1, (2, 64, 64)),
    (192, 128, 0, (1, 256, 128)),          # two ci tiles, the second half empty
    (128, 192, 1, (1, 32, 128)),           # two co tiles on the shifted side
    (64, 128, 0, (3, 128, 256)),           # QW = 128: two stages per row; ragged ci tile
    (128, 64, 1, (1, 64, 192)),
]


def _run(L, d, x, gy, wshape, Cout, row):
    os.environ["HESIC_WGRAD_ROW"] = "1" if row else "0"
    os.environ["HESIC_WGRAD_ROW_MINQ"] = "0"
    try:
        nws = int(L.lib().hesic_conv2d_wgrad_ws_bytes(C.byref(d)))
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV)
        dw = torch.full(wshape, 0.25, dtype=torch.float32, device=DEV)
        db = torch.full((Cout,), -0.5, dtype=torch.float32, device=DEV)
        L.call("hesic_conv2d_wgrad_direct", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(db), 1, L.ptr(ws), nws, L.stream())
        torch.cuda.synchronize()
    finally:
        os.environ.pop("HESIC_WGRAD_ROW", None)
        os.environ.pop("HESIC_WGRAD_ROW_MINQ", None)
    return dw - 0.25, db + 0.5, nws


@pytest.mark.parametrize("case", range(len(CASES)))
def test_row_kernel_matches_torch_and_the_tap_kernel(case):
    import hesic_amd
    from hesic_amd import _lib as L
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        Cin, Cout, tr, (B, H, W) = CASES[case]
        Ho, Wo = (H * 2, W * 2) if tr else (H // 2, W // 2)
        x = synthetic._uniform(f"wr.x{case}", (B, Cin, H, W), -2, 2).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gy = synthetic._uniform(f"wr.g{case}", (B, Cout, Ho, Wo), -1, 1).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
        d = L.ConvDesc(B, H, W, Cin, Ho, Wo, Cout, 5, 5, 2, 2, tr, L.BF16, 0, 0, Cin, 0, Cout, 0, 0)
        wshape = (Cin, Cout, 5, 5) if tr else (Cout, Cin, 5, 5)
        dw1, db1, nws1 = _run(L, d, x, gy, wshape, Cout, True)
        dw0, db0, nws0 = _run(L, d, x, gy, wshape, Cout, False)
        assert nws1 != nws0, "the row kernel did not take this layer (same workspace as the tap kernel)"
        w = torch.zeros(wshape, device=DEV, requires_grad=True)
        b = torch.zeros(Cout, device=DEV, requires_grad=True)
        y = (F.conv_transpose2d(x.float(), w, b, stride=2, padding=2, output_padding=1) if tr else F.conv2d(x.float(), w, b, stride=2, padding=2))
        y.backward(gy.float())
        scale = float(w.grad.abs().max())
        assert scale > 1.0
        assert float((dw1 - w.grad).abs().max()) <= 2e-5 * scale, "row kernel vs torch fp32"
        assert float((dw1 - dw0).abs().max()) <= 2e-5 * scale, "row kernel vs tap kernel"
        bs = float(b.grad.abs().max()) + 1.0
        assert float((db1 - b.grad).abs().max()) <= 2e-5 * bs, "bias gradient vs torch"
        assert float((db1 - db0).abs().max()) <= 2e-5 * bs
    finally:
        hesic_amd.set_compute_dtype(torch.float32)


def test_row_kernel_is_deterministic():
    import hesic_amd
    from hesic_amd import _lib as L
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        Cin, Cout, tr, (B, H, W) = CASES[0]
        x = synthetic._uniform("wr.dx", (B, Cin, H, W), -2, 2).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gy = synthetic._uniform("wr.dg", (B, Cout, H // 2, W // 2), -1, 1).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
        d = L.ConvDesc(B, H, W, Cin, H // 2, W // 2, Cout, 5, 5, 2, 2, 0, L.BF16, 0, 0, Cin, 0, Cout, 0, 0)
        a = _run(L, d, x, gy, (Cout, Cin, 5, 5), Cout, True)
        b = _run(L, d, x, gy, (Cout, Cin, 5, 5), Cout, True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    finally:
        hesic_amd.set_compute_dtype(torch.float32)


# ---- wgrad_nw_fused_kernel: g_a_conv1 (3 -> 128) / g_s_conv4 (128 -> 3 transposed) weight gradients in one launch
NW_CASES = [
    # transposed, (B, H, W) of the NARROW (image-side) grid
    (0, (2, 128, 128)),
    (0, (1, 96, 256)),
    (1, (2, 128, 128)),
    (1, (3, 64, 256)),
]


def _run_nw(L, d, x, gy, wshape, Cout, fused):
    os.environ["HESIC_NW_FUSED"] = "1" if fused else "0"
    try:
        nws = int(L.lib().hesic_sconv2d_wgrad_ws_bytes(C.byref(d)))
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV)
        dw = torch.full(wshape, 7.0, dtype=torch.float32, device=DEV)          # the call overwrites
        db = torch.full((Cout,), 7.0, dtype=torch.float32, device=DEV)
        L.call("hesic_sconv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(gy), L.ptr(dw), L.ptr(db), L.ptr(ws), nws, L.stream())
        torch.cuda.synchronize()
    finally:
        os.environ.pop("HESIC_NW_FUSED", None)
    return dw, db


@pytest.mark.parametrize("case", range(len(NW_CASES)))
def test_fused_narrow_wide_wgrad_matches_torch_and_the_im2col_route(case):
    import hesic_amd
    from hesic_amd import _lib as L
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        tr, (B, H, W) = NW_CASES[case]
        if not tr:        # conv1: x = image (fp32 planar), gy = wide map on the half grid
            x = synthetic._uniform(f"nw.x{case}", (B, 3, H, W), 0, 1).to(DEV)
            gy = synthetic._uniform(f"nw.g{case}", (B, 128, H // 2, W // 2), -1, 1).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
            Cin, Cout, wshape = 3, 128, (128, 3, 5, 5)
            xs, ys = x.stride(), gy.stride()
            d = L.SConvDesc(B, H, W, Cin, H // 2, W // 2, Cout, 5, 5, 2, 2, 0, L.F32, L.BF16, 0, 0, xs[0], xs[1], xs[2], xs[3], ys[0], ys[1], ys[2], ys[3])
        else:             # deconv4: x = wide map on the half grid, gy = image-side gradient (fp32 planar)
            x = synthetic._uniform(f"nw.x{case}", (B, 128, H // 2, W // 2), -2, 2).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
            gy = synthetic._uniform(f"nw.g{case}", (B, 3, H, W), -1, 1).to(DEV)
            Cin, Cout, wshape = 128, 3, (128, 3, 5, 5)
            xs, ys = x.stride(), gy.stride()
            d = L.SConvDesc(B, H // 2, W // 2, Cin, H, W, Cout, 5, 5, 2, 2, 1, L.BF16, L.F32, 0, 0, xs[0], xs[1], xs[2], xs[3], ys[0], ys[1], ys[2], ys[3])
        dw1, db1 = _run_nw(L, d, x, gy, wshape, Cout, True)
        dw0, db0 = _run_nw(L, d, x, gy, wshape, Cout, False)
        w = torch.zeros(wshape, device=DEV, requires_grad=True)
        b = torch.zeros(Cout, device=DEV, requires_grad=True)
        y = (F.conv_transpose2d(x.float(), w, b, stride=2, padding=2, output_padding=1) if tr else F.conv2d(x.float(), w, b, stride=2, padding=2))
        y.backward(gy.float())
        scale = float(w.grad.abs().max())
        # the narrow operand enters the matrix cores as bf16 in both routes (the im2col matrix / the patch tile): 2^-9 per element, random sign
        assert float((dw1 - w.grad).abs().max()) <= 3e-3 * scale, "fused vs torch fp32"
        assert float((dw1 - dw0).abs().max()) <= 2e-5 * scale, "fused vs im2col route (same bf16 operands, other summation order)"
        bs = float(b.grad.abs().max()) + 1.0
        assert float((db1 - b.grad).abs().max()) <= 2e-5 * bs
        assert float((db1 - db0).abs().max()) <= 2e-5 * bs
    finally:
        hesic_amd.set_compute_dtype(torch.float32)
