"""Operator parity on a real MI355X: every op goes through the C ABI (libhesic_hip.so) and is compared
with the CPU oracle on the same seeded inputs and with the reference-generated golden vectors.

Bars: integer symbols / rounded latents bit-exact; fp32 storage (exact-fp32 MFMA) within 1e-4 of the
fp32 CPU result (summation order differs); bf16 storage within 2e-2 of the output scale against the
oracle run on the bf16-rounded inputs."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden
import hesic_amd
from hesic_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _imp():
    from hesic_amd import functional as Fn
    from oracle import hesic_oracle as O
    return Fn, O


def rnd(name, shape, lo=-1.0, hi=1.0):
    return synthetic._uniform("t." + name, shape, lo, hi)


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def bf(x):
    return x.to(torch.bfloat16).float()


CONV_CASES = [
    # tag, Cin, Cout, k, stride, transposed, (B,H,W)
    ("c5s2_128", 128, 128, 5, 2, 0, (2, 32, 32)),
    ("c5s2_128_192", 128, 192, 5, 2, 0, (1, 16, 24)),
    ("c5s1_192_128", 192, 128, 5, 1, 0, (2, 8, 8)),
    ("c5s1_320_128", 320, 128, 5, 1, 0, (1, 12, 12)),
    ("c5s1_128_960", 128, 960, 5, 1, 0, (1, 8, 8)),
    ("c3s1_288_384", 288, 384, 3, 1, 0, (1, 8, 8)),
    ("c1_768_640", 768, 640, 1, 1, 0, (2, 4, 4)),
    ("c5s2_tiny", 128, 128, 5, 2, 0, (2, 2, 2)),
    ("c5s2_parity_walk", 128, 128, 5, 2, 0, (4, 64, 48)),      # big enough for the plain launch: taps walked by parity class
    ("c3s2_parity_walk", 64, 128, 3, 2, 0, (4, 64, 64)),
    ("c5s2_64_72", 64, 72, 5, 2, 0, (1, 10, 6)),
    ("d5s2_128", 128, 128, 5, 2, 1, (2, 16, 16)),
    ("d5s2_192_128", 192, 128, 5, 2, 1, (1, 4, 4)),
    ("d5s2_128_288", 128, 288, 5, 2, 1, (1, 8, 4)),
    ("d5s2_1x1", 128, 128, 5, 2, 1, (2, 1, 1)),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_wide_conv_forward_backward(case, dtype):
    Fn, O = _imp()
    tag, Cin, Cout, k, s, tr, (B, H, W) = case
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    fan = Cin * k * k / (4 if tr and s == 2 else 1)
    w = rnd(tag + "w", wshape) * (3.0 / fan) ** 0.5
    b = rnd(tag + "b", (Cout,), -0.1, 0.1)
    x = rnd(tag + "x", (B, Cin, H, W))
    if dtype == torch.bfloat16:
        x, w = bf(x), bf(w)
    xo, wo, bo = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yo = (O.deconv if tr else O.conv)(xo, wo, bo, s)
    gy = rnd(tag + "g", yo.shape)
    if dtype == torch.bfloat16:
        gy = bf(gy)
    yo.backward(gy)

    xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    wd, bd = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = Fn.conv2d(xd, wd, bd, kernel_size=k, stride=s, padding=k // 2, transposed=bool(tr))
    assert y.shape == yo.shape and y.dtype == dtype
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(y, yo) < tol
    y.backward(gy.to(DEV, dtype))
    assert rel_err(xd.grad, xo.grad) < tol
    assert rel_err(wd.grad, wo.grad) < (2e-4 if dtype == torch.float32 else 3e-2)
    assert rel_err(bd.grad, bo.grad) < (2e-4 if dtype == torch.float32 else 3e-2)


def test_wide_conv_fused_abs_act_and_concat_write():
    Fn, O = _imp()
    from hesic_amd import _lib as L
    import ctypes as C
    x = rnd("fa_x", (2, 192, 8, 8), -2, 2)
    w = rnd("fa_w", (128, 192, 5, 5)) * 0.02
    b = rnd("fa_b", (128,), -0.1, 0.1)
    ref = torch.relu(O.conv(x.abs(), w, b, 1))
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    y = Fn.conv2d(xd, w.to(DEV), b.to(DEV), kernel_size=5, stride=1, padding=2, act=L.ACT_RELU, in_abs=True)
    assert rel_err(y, ref) < 1e-4
    # same conv written into channels [64,192) of a 256-channel buffer
    buf = torch.zeros((2, 256, 8, 8), device=DEV).contiguous(memory_format=torch.channels_last)
    wp = Fn.PackedWeight().get(w.to(DEV), None, 128, 192, 5, 5, False, False, torch.float32)
    Fn._wide_conv(xd, wp, b.to(DEV), 2, 8, 8, 192, 8, 8, 128, 5, 1, 2, False, act=L.ACT_RELU, in_abs=1, out=buf, out_c_off=64)
    assert rel_err(buf[:, 64:192], ref) < 1e-4
    assert float(buf[:, :64].abs().max()) == 0 and float(buf[:, 192:].abs().max()) == 0


def test_wide_conv_split_k_matches_plain_launch():
    """Low-resolution layers take the split-K launch (hesic_conv2d_forward_ws); it must agree with the one-block-per-tile
    launch of the same op (fp32 partial sums in a different order, same bf16 rounding) and honour act / channel offsets."""
    Fn, O = _imp()
    from hesic_amd import _lib as L
    import ctypes as C
    B, Cin, Cout, H = 2, 128, 128, 32
    x = bf(rnd("sk_x", (B, Cin, H, H), -2, 2))
    w = bf(rnd("sk_w", (Cout, Cin, 5, 5)) * 0.03)
    b = rnd("sk_b", (Cout,), -0.1, 0.1)
    ref = torch.nn.functional.leaky_relu(O.conv(x, w, b, 2), 0.01)
    xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wp = Fn.PackedWeight().get(w.to(DEV), None, Cout, Cin, 5, 5, False, False, torch.bfloat16)
    d = L.ConvDesc(B, H, H, Cin, 16, 16, Cout, 5, 5, 2, 2, 0, L.dt(torch.bfloat16), L.ACT_LEAKY, 0, Cin, 0, 192, 32, 0)
    need = int(L.lib().hesic_conv2d_ws_bytes(C.byref(d)))
    assert need >= 2 * B * 16 * 16 * Cout * 4                   # at least two K slices of fp32 partial tiles
    big = L.ConvDesc(8, 256, 256, Cin, 128, 128, Cout, 5, 5, 2, 2, 0, L.dt(torch.bfloat16), 0, 0, Cin, 0, Cout, 0, 0)
    assert int(L.lib().hesic_conv2d_ws_bytes(C.byref(big))) == 0   # full-size layers keep the plain launch
    outs = []
    for split in (True, False):
        buf = torch.zeros((B, 192, 16, 16), device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if split:
            ws = torch.empty(need, dtype=torch.uint8, device=DEV)
            L.call("hesic_conv2d_forward_ws", C.byref(d), L.ptr(xd), L.ptr(wp), L.ptr(b.to(DEV)), L.ptr(buf), L.ptr(ws), need, L.stream())
        else:
            L.call("hesic_conv2d_forward", C.byref(d), L.ptr(xd), L.ptr(wp), L.ptr(b.to(DEV)), L.ptr(buf), L.stream())
        assert float(buf[:, :32].float().abs().max()) == 0 and float(buf[:, 160:].float().abs().max()) == 0
        outs.append(buf[:, 32:160].float())
        assert rel_err(outs[-1], ref) < 2e-2
    assert rel_err(outs[0], outs[1]) < 4e-3                      # one bf16 ulp of the output scale
    with pytest.raises(RuntimeError):                            # too small a workspace is an error, not a silent fallback
        L.call("hesic_conv2d_forward_ws", C.byref(d), L.ptr(xd), L.ptr(wp), L.ptr(b.to(DEV)), L.ptr(buf), L.ptr(ws), 16, L.stream())


def test_stride1_transposed_conv_takes_split_k():
    """The data gradient of a stride-1 conv is a stride-1 TRANSPOSED conv: one output phase with the full K loop.  With Cin = 960
    (the 128 -> 960 layers of gmm_hyper_y1/y2 backwards) on a 32 x 32 map it is a long-K, few-tile launch and takes the split-K form;
    same result as the one-block-per-tile launch and as the oracle's conv_transpose2d."""
    Fn, O = _imp()
    B, Cin, Cout, H = 2, 960, 128, 32
    x = bf(rnd("st_x", (B, Cin, H, H), -1, 1))
    w = bf(rnd("st_w", (Cin, Cout, 5, 5)) * 0.02)
    b = rnd("st_b", (Cout,), -0.1, 0.1)
    ref = torch.nn.functional.conv_transpose2d(x, w, b, stride=1, padding=2)
    xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    from hesic_amd import _lib as L
    import ctypes as C
    d = L.ConvDesc(B, H, H, Cin, H, H, Cout, 5, 5, 1, 2, 1, L.dt(torch.bfloat16), 0, 0, Cin, 0, Cout, 0, 0)
    assert int(L.lib().hesic_conv2d_ws_bytes(C.byref(d))) >= 2 * B * H * H * Cout * 4
    keep = hesic_amd.functional.compute_dtype()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            y_split = Fn.conv2d(xd, w.to(DEV), b.to(DEV), kernel_size=5, stride=1, padding=2, transposed=True).float()
            with Fn.no_split_k():
                y_plain = Fn.conv2d(xd, w.to(DEV), b.to(DEV), kernel_size=5, stride=1, padding=2, transposed=True).float()
    finally:
        hesic_amd.set_compute_dtype(keep)
    assert rel_err(y_split, ref) < 2e-2 and rel_err(y_plain, ref) < 2e-2
    assert rel_err(y_split, y_plain) < 4e-3


@pytest.mark.parametrize("tr", [0, 1], ids=["pre_conv", "after_conv"])
def test_cat_free_6to3_conv(tr):
    """pre_conv(cat(x1_warp, x2)) / after_conv(cat(t, x1_hat_warp)) read their halves from two tensors of different
    layout / dtype (hesic_sconv2d_forward_cat); sizes that end in partial tiles."""
    Fn, O = _imp()
    B, H, W = 2, 37, 200
    xa = rnd("cat_a", (B, 3, H, W))
    xb = bf(rnd("cat_b", (B, 3, H, W)))
    w = rnd("cat_w", (6, 3, 5, 5) if tr else (3, 6, 5, 5)) * 0.1
    b = rnd("cat_bias", (3,), -0.1, 0.1)
    ref = (O.deconv if tr else O.conv)(torch.cat((xa, xb), 1), w, b, 1)
    xad = xa.to(DEV).contiguous(memory_format=torch.channels_last)            # NHWC fp32 (what GDN(3) hands over)
    xbd = xb.to(DEV, torch.bfloat16)                                          # planar bf16
    with torch.no_grad():
        y = Fn.conv2d_cat(xad, xbd, w.to(DEV), b.to(DEV), kernel_size=5, stride=1, padding=2, transposed=bool(tr))
        y2 = Fn.conv2d_cat(xa.to(DEV), xb.to(DEV), w.to(DEV), b.to(DEV), kernel_size=5, stride=1, padding=2, transposed=bool(tr))
        y3 = Fn.conv2d(torch.cat((xa, xb), 1).to(DEV), w.to(DEV), b.to(DEV), kernel_size=5, stride=1, padding=2, transposed=bool(tr))
    assert y.shape == ref.shape and y.dtype == torch.float32
    assert rel_err(y, ref) < 1e-5 and rel_err(y2, ref) < 1e-5 and rel_err(y3, ref) < 1e-5
    # with autograd on, the same call is the ordinary cat + conv
    xg = xa.to(DEV).requires_grad_()
    yg = Fn.conv2d_cat(xg, xb.to(DEV), w.to(DEV), b.to(DEV), kernel_size=5, stride=1, padding=2, transposed=bool(tr))
    yg.sum().backward()
    assert rel_err(yg, ref) < 1e-5 and xg.grad is not None


@pytest.mark.parametrize("on_input", [0, 1])
def test_cat_conv_with_fused_three_channel_gdn(on_input):
    """pre_conv -> GDN(3) (newnet1.py:643-644) and IGDN(3) -> cat -> after_conv (:684-686) as ONE launch
    (hesic_sconv2d_forward_cat_gdn) against the oracle and against the two-launch path; partial tiles, lower-bounded beta."""
    Fn, O = _imp()
    from compressai.layers import GDN
    B, H, W = 2, 37, 200
    xa, xb = rnd("cg_a", (B, 3, H, W)), rnd("cg_b", (B, 3, H, W))
    w = rnd("cg_w", (6, 3, 5, 5) if on_input else (3, 6, 5, 5)) * 0.1
    b = rnd("cg_bias", (3,), -0.1, 0.1)
    g = GDN(3, inverse=bool(on_input))
    with torch.no_grad():
        g.beta.copy_(rnd("cg_beta", (3,), 0.0, 1.5))                  # some below the lower bound
        g.gamma.copy_(rnd("cg_gamma", (3, 3), -0.1, 0.6))
    if on_input:
        ref = O.deconv(torch.cat((O.gdn(xa, g.beta.detach(), g.gamma.detach(), True), xb), 1), w, b, 1)
    else:
        ref = O.gdn(O.conv(torch.cat((xa, xb), 1), w, b, 1), g.beta.detach(), g.gamma.detach(), False)
    g = g.to(DEV)
    args = dict(kernel_size=5, stride=1, padding=2, transposed=bool(on_input), gdn=g, gdn_on_input=bool(on_input))
    keep = Fn.FUSE_GDN3
    try:
        with torch.no_grad():
            Fn.FUSE_GDN3 = True
            y = Fn.conv2d_cat(xa.to(DEV), xb.to(DEV), w.to(DEV), b.to(DEV), **args)
            Fn.FUSE_GDN3 = False
            y2 = Fn.conv2d_cat(xa.to(DEV), xb.to(DEV), w.to(DEV), b.to(DEV), **args)
    finally:
        Fn.FUSE_GDN3 = keep
    assert rel_err(y, ref) < 2e-5 and rel_err(y2, ref) < 2e-5
    assert rel_err(y, y2) < 2e-6
    if on_input:         # the input-side form covers exactly the first tensor's three channels: anything else is an error, not a guess
        from hesic_amd import _lib as L
        import ctypes as C
        out = torch.empty((B, 3, H, W), device=DEV)
        d = Fn._sdesc(xa.to(DEV), out, 6, 3, 5, 1, 2, True)
        xb4 = torch.cat((xb, xb[:, :1]), 1).to(DEV)
        with pytest.raises(RuntimeError, match="first tensor's three channels"):
            L.call("hesic_sconv2d_forward_cat_gdn", C.byref(d), L.ptr(xa[:, :2].contiguous().to(DEV)), L.ptr(xb4), (C.c_int64 * 4)(*xb4.stride()),
                   L.dt(xb4), 2, L.ptr(w.to(DEV)), L.ptr(b.to(DEV)), L.ptr(g.beta), L.ptr(g.gamma), float(g.beta_min), 1, 1, L.ptr(out), L.stream())


def test_masked_conv_matches_golden(ops_golden):
    Fn, O = _imp()
    g = ops_golden
    # golden case is 8->16 (narrow path); the wide path is checked against the oracle
    x = rnd("mc_x", (1, 192, 8, 8), -3, 3).round()
    w = rnd("mc_w", (384, 192, 5, 5)) * 0.02
    b = rnd("mc_b", (384,), -0.1, 0.1)
    ref = O.masked_conv(x, w, b, "A")
    mask = torch.ones_like(w)
    mask[:, :, 2, 2:] = 0
    mask[:, :, 3:] = 0
    y = Fn.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), kernel_size=5, stride=1, padding=2, tap_mask=(1 << 12) - 1,
                  mask=mask.to(DEV))
    assert rel_err(y, ref) < 1e-4
    for mt in ("A", "B"):
        mk = T(g[f"mc{mt}_mask"]).to(DEV)
        y = Fn.conv2d(T(g[f"mc{mt}_x"]).to(DEV), T(g[f"mc{mt}_w"]).to(DEV), T(g[f"mc{mt}_b"]).to(DEV), kernel_size=5,
                      stride=1, padding=2, mask=mk)
        assert rel_err(y, T(g[f"mc{mt}_y"])) < 1e-5


@pytest.mark.parametrize("tag,stride,tr", [("c5s2", 2, 0), ("c5s1", 1, 0), ("c3s1", 1, 0), ("d5s2", 2, 1),
                                           ("d5s1", 1, 1), ("c5s2_32", 2, 0), ("d5s2_32", 2, 1)])
def test_conv_golden_fwd_bwd(ops_golden, tag, stride, tr):
    """The reference-generated conv()/deconv() vectors (odd sizes, narrow channels) through the HIP path."""
    Fn, _ = _imp()
    g = ops_golden
    x, w, b = (T(g[f"{tag}_{k}"]).to(DEV).requires_grad_() for k in ("x", "w", "b"))
    k = w.shape[-1]
    y = Fn.conv2d(x, w, b, kernel_size=k, stride=stride, padding=k // 2, transposed=bool(tr))
    assert rel_err(y, T(g[tag + "_y"])) < 1e-4
    y.backward(T(g[tag + "_gy"]).to(DEV).to(y.dtype))
    assert rel_err(x.grad, T(g[tag + "_dx"])) < 2e-4
    assert rel_err(w.grad, T(g[tag + "_dw"])) < 2e-4
    assert rel_err(b.grad, T(g[tag + "_db"])) < 2e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_image_side_convs(dtype):
    """g_a_conv1 (3->128 s2), g_s_conv4 (128->3 transposed), pre_conv / after_conv (6->3 s1)."""
    Fn, O = _imp()
    Fn.set_compute_dtype(dtype)
    try:
        tol = 1e-4 if dtype == torch.float32 else 2e-2
        x = rnd("ic_x", (2, 3, 32, 48), 0, 1)
        w = rnd("ic_w", (128, 3, 5, 5)) * 0.2
        b = rnd("ic_b", (128,), -0.1, 0.1)
        y = Fn.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), kernel_size=5, stride=2, padding=2)
        assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
        # bf16 storage: operands are rounded to bf16 before the MFMA -> compare with the oracle on rounded operands
        r = bf if dtype == torch.bfloat16 else (lambda t: t)
        assert rel_err(y, O.conv(r(x), r(w), b, 2)) < (1e-4 if dtype == torch.float32 else 6e-3)
        f = rnd("ic_f", (2, 128, 16, 24))
        wt = rnd("ic_wt", (128, 3, 5, 5)) * 0.05
        bt = rnd("ic_bt", (3,), -0.1, 0.1)
        fi = bf(f) if dtype == torch.bfloat16 else f
        keep_sh, Fn.SHAPED_WEIGHTS = Fn.SHAPED_WEIGHTS, False      # the kernel's arithmetic against round-to-nearest weights (the error-feedback panel: test below)
        try:
            yt = Fn.conv2d(fi.to(DEV, dtype), wt.to(DEV), bt.to(DEV), kernel_size=5, stride=2, padding=2, transposed=True)
        finally:
            Fn.SHAPED_WEIGHTS = keep_sh
        assert yt.dtype == torch.float32 and yt.shape == (2, 3, 32, 48)
        assert rel_err(yt, O.deconv(fi, r(wt), bt, 2)) < 1e-4
        x6 = rnd("ic_x6", (2, 6, 16, 16), 0, 1)
        w6 = rnd("ic_w6", (3, 6, 5, 5)) * 0.1
        assert rel_err(Fn.conv2d(x6.to(DEV), w6.to(DEV), None, kernel_size=5, stride=1, padding=2), O.conv(x6, w6, None, 1)) < 1e-4
        w6t = rnd("ic_w6t", (6, 3, 5, 5)) * 0.1
        assert rel_err(Fn.conv2d(x6.to(DEV), w6t.to(DEV), None, kernel_size=5, stride=1, padding=2, transposed=True),
                       O.deconv(x6, w6t, None, 1)) < 1e-4
    finally:
        Fn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("C", [3, 128])
@pytest.mark.parametrize("inv", [False, True])
def test_gdn_golden(ops_golden, C, inv):
    Fn, _ = _imp()
    g, t = ops_golden, f"gdn_C{C}_{'inv' if inv else 'fwd'}_"
    x = T(g[t + "x"]).to(DEV).requires_grad_()
    beta, gamma = T(g[t + "beta"]).to(DEV).requires_grad_(), T(g[t + "gamma"]).to(DEV).requires_grad_()
    y = Fn.gdn(x, beta, gamma, inv)
    assert rel_err(y, T(g[t + "y"])) < 1e-5
    y.backward(T(g[t + "gy"]).to(DEV))
    assert rel_err(x.grad, T(g[t + "dx"])) < 1e-4
    assert rel_err(beta.grad, T(g[t + "dbeta"])) < 1e-4
    assert rel_err(gamma.grad, T(g[t + "dgamma"])) < 1e-4


@pytest.mark.parametrize("inv", [False, True])
def test_gdn128_large_bf16_and_f32(inv):
    Fn, O = _imp()
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=3)
    x = rnd("gdnL", (2, 128, 24, 20), -3, 3)          # 960 pixels: not a multiple of the 128-pixel tile
    ref = O.gdn(x, sd["g.beta"], sd["g.gamma"], inv)
    y = Fn.gdn(x.to(DEV), sd["g.beta"].to(DEV), sd["g.gamma"].to(DEV), inv)
    assert rel_err(y, ref) < 1e-5
    xb = bf(x)
    refb = O.gdn(xb, sd["g.beta"], sd["g.gamma"], inv)
    yb = Fn.gdn(xb.to(DEV, torch.bfloat16), sd["g.beta"].to(DEV), sd["g.gamma"].to(DEV), inv)
    assert yb.dtype == torch.bfloat16 and rel_err(yb, refb) < 1.5e-2


@pytest.mark.parametrize("ac", [True, False])
def test_warp_golden(warp_golden, ac):
    Fn, _ = _imp()
    g = warp_golden
    src = T(g["src"]).to(DEV).requires_grad_()
    out = Fn.warp_perspective(src, T(g["H"]).to(DEV), (24, 32), align_corners=ac)
    assert rel_err(out, T(g[f"out_ac{int(ac)}"])) < 1e-4
    out.backward(T(g["g"]).to(DEV))
    assert rel_err(src.grad, T(g[f"dsrc_ac{int(ac)}"])) < 1e-3


def test_warp_identity_and_layouts():
    Fn, O = _imp()
    x = rnd("wi", (2, 3, 40, 56), 0, 1)
    eye = torch.eye(3).repeat(2, 1, 1)
    assert torch.equal(Fn.warp_perspective(x.to(DEV), eye.to(DEV), (40, 56)).cpu(), x)        # identity is exact
    Hm = torch.from_numpy(np.stack([synthetic.homography(i) for i in range(2)])).float()
    ref = O.warp_perspective(x, Hm, (40, 56))
    for src in (x.to(DEV), x.to(DEV).contiguous(memory_format=torch.channels_last)):
        assert rel_err(Fn.warp_perspective(src, Hm.to(DEV), (40, 56)), ref) < 1e-4


@pytest.mark.parametrize("C", [8, 128])
def test_entropy_bottleneck_golden(ops_golden, C):
    Fn, _ = _imp()
    g, t = ops_golden, f"eb_C{C}_"
    def params():
        m = [T(g[f"{t}p__matrices.{i}"]).to(DEV).requires_grad_() for i in range(5)]
        b = [T(g[f"{t}p__biases.{i}"]).to(DEV).requires_grad_() for i in range(5)]
        f = [T(g[f"{t}p__factors.{i}"]).to(DEV).requires_grad_() for i in range(4)]
        q = T(g[t + "p_quantiles"]).to(DEV).requires_grad_()
        return m, b, f, q
    m, b, f, q = params()
    x = T(g[t + "x"]).to(DEV).requires_grad_()
    zh, lik = Fn.entropy_bottleneck(x, m, b, f, q)
    assert torch.equal(zh.detach().cpu(), T(g[t + "eval_xhat"]))                       # round(z-med)+med: bit-exact
    torch.testing.assert_close(lik.detach().cpu().contiguous(), T(g[t + "eval_lik"]), rtol=2e-4, atol=1e-9)
    (lik * T(g[t + "g_lik"]).to(DEV)).sum().backward()
    names = [f"_matrices.{i}" for i in range(5)] + [f"_biases.{i}" for i in range(5)] + [f"_factors.{i}" for i in range(4)]
    for n, p in zip(names, m + b + f):
        assert rel_err(p.grad, T(g[t + "eval_d_" + n])) < 2e-3, n
    assert rel_err(q.grad, T(g[t + "eval_d_quantiles"])) < 2e-3
    # training mode with the injected noise
    m, b, f, q = params()
    x = T(g[t + "x"]).to(DEV).requires_grad_()
    B, _, H, W = x.shape
    noise = T(g[t + "noise"]).reshape(C, H, W, B).permute(3, 0, 1, 2).contiguous().to(DEV)
    zt, lt = Fn.entropy_bottleneck(x, m, b, f, q, noise=noise)
    torch.testing.assert_close(zt.detach().cpu().contiguous(), T(g[t + "train_xhat"]), rtol=0, atol=1e-6)
    torch.testing.assert_close(lt.detach().cpu().contiguous(), T(g[t + "train_lik"]), rtol=2e-4, atol=1e-9)
    ((lt * T(g[t + "g_lik"]).to(DEV)).sum() + (zt * T(g[t + "g_xhat"]).to(DEV)).sum()).backward()
    assert rel_err(x.grad, T(g[t + "train_dx"])) < 2e-3
    for n, p in zip(names, m + b + f):
        assert rel_err(p.grad, T(g[t + "train_d_" + n])) < 2e-3, n


def test_gmm_golden(ops_golden):
    Fn, _ = _imp()
    g = ops_golden
    ins = [T(g[k]).to(DEV).requires_grad_() for k in ("gmm_y", "gmm_scales", "gmm_means", "gmm_weights")]
    yh, lik = Fn.gaussian_mixture(*ins, K=5)
    assert torch.equal(yh.detach().cpu(), T(g["gmm_eval_yhat"]))
    assert torch.equal(Fn.quantize_symbols(ins[0].detach()).cpu(), T(g["gmm_symbols"]))      # int32, bit-exact
    torch.testing.assert_close(lik.detach().cpu().contiguous(), T(g["gmm_eval_lik"]), rtol=1e-4, atol=1e-9)
    (lik * T(g["gmm_g_lik"]).to(DEV)).sum().backward()
    assert rel_err(ins[1].grad, T(g["gmm_eval_dscales"])) < 1e-3
    assert rel_err(ins[2].grad, T(g["gmm_eval_dmeans"])) < 1e-3
    assert rel_err(ins[3].grad, T(g["gmm_eval_dweights"])) < 1e-3
    ins = [T(g[k]).to(DEV).requires_grad_() for k in ("gmm_y", "gmm_scales", "gmm_means", "gmm_weights")]
    yh, lik = Fn.gaussian_mixture(*ins, K=5, noise=T(g["gmm_noise"]).to(DEV))
    torch.testing.assert_close(lik.detach().cpu().contiguous(), T(g["gmm_train_lik"]), rtol=1e-4, atol=1e-9)
    ((lik * T(g["gmm_g_lik"]).to(DEV)).sum() + (yh * T(g["gmm_g_yhat"]).to(DEV)).sum()).backward()
    for t_, k in zip(ins, ("dy", "dscales", "dmeans", "dweights")):
        assert rel_err(t_.grad, T(g["gmm_train_" + k])) < 1e-3, k


def test_gaussian_conditional_golden(ops_golden):
    Fn, _ = _imp()
    g = ops_golden
    y = T(g["gmm_y"]).to(DEV)
    sc, mu = T(g["gmm_scales"])[:, :16].contiguous().to(DEV), T(g["gmm_means"])[:, :16].contiguous().to(DEV)
    ins = [y.clone().requires_grad_(), sc.clone().requires_grad_(), mu.clone().requires_grad_()]
    yh, lik = Fn.gaussian_conditional(*ins)
    torch.testing.assert_close(yh.detach().cpu().contiguous(), T(g["gc_eval_yhat"]), rtol=0, atol=0)
    assert torch.equal(Fn.quantize_symbols(y, mu).cpu(), T(g["gc_symbols"]))
    torch.testing.assert_close(lik.detach().cpu().contiguous(), T(g["gc_eval_lik"]), rtol=1e-4, atol=1e-9)
    (lik * T(g["gmm_g_lik"]).to(DEV)).sum().backward()
    assert rel_err(ins[1].grad, T(g["gc_eval_dscales"])) < 1e-3
    assert float((ins[2].grad.cpu() - T(g["gc_eval_dmeans"])).abs().max()) < 1e-6
    ins = [y.clone().requires_grad_(), sc.clone().requires_grad_(), mu.clone().requires_grad_()]
    yh, lik = Fn.gaussian_conditional(*ins, noise=T(g["gmm_noise"]).to(DEV))
    ((lik * T(g["gmm_g_lik"]).to(DEV)).sum() + (yh * T(g["gmm_g_yhat"]).to(DEV)).sum()).backward()
    torch.testing.assert_close(lik.detach().cpu().contiguous(), T(g["gc_train_lik"]), rtol=1e-4, atol=1e-9)
    for t_, k in zip(ins, ("dy", "dscales", "dmeans")):
        assert rel_err(t_.grad, T(g["gc_train_" + k])) < 1e-3, k
    # scales / means as the two halves of one tensor (the chunk(2,1) of HESIC+)
    both = torch.cat((sc, mu), 1).contiguous(memory_format=torch.channels_last)
    s2, m2 = both.chunk(2, 1)
    _, lik2 = Fn.gaussian_conditional(y, s2, m2)
    torch.testing.assert_close(lik2.cpu().contiguous(), T(g["gc_eval_lik"]), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("cz,cy", [(128, 192), (12, 20)], ids=["vector", "scalar"])
def test_hyper_glue_bf16_vector_and_scalar_forms(cz, cy):
    """upsample4 / upsample4_cat / round_to in bf16: channel counts that are multiples of 8 take the 16-byte vector kernels (the inference
    schedule's case), others the element-wise ones; both against the oracle on the bf16-rounded input, the copied half and the rounding exact."""
    Fn, O = _imp()
    z, y1 = bf(rnd("hgv_z", (2, cz, 6, 5), -2, 2)), bf(rnd("hgv_y", (2, cy, 24, 20), -8, 8))
    ref = O.upsample_bilinear_x4(z)
    with torch.no_grad():
        up = Fn.upsample4(z.to(DEV, torch.bfloat16))
        cat = Fn.upsample4_cat(z.to(DEV, torch.bfloat16), y1.to(DEV, torch.bfloat16))
    assert up.dtype == torch.bfloat16 and rel_err(up, ref) < 4e-3                       # one bf16 rounding of the interpolated value
    assert torch.equal(cat[:, :cz].float().cpu(), up.float().cpu()) and torch.equal(cat[:, cz:].float().cpu(), y1)
    v = rnd("hgv_r", (2, cy, 8, 8 if cy % 8 == 0 else 7), -6, 6)
    r = Fn.round_to(v.to(DEV), torch.bfloat16)
    assert r.dtype == torch.bfloat16 and torch.equal(r.float().cpu(), torch.round(v))


def test_hyper_glue(ops_golden):
    Fn, O = _imp()
    g = ops_golden
    z = T(g["hy1_z"])
    up = Fn.upsample4(z.to(DEV))
    assert rel_err(up, T(g["hy2_up"])) < 1e-5
    y1 = T(g["hy2_y1"])
    zc = z.to(DEV).requires_grad_()
    yc = y1.to(DEV).requires_grad_()
    cat = Fn.upsample4_cat(zc, yc)
    ref = torch.cat((T(g["hy2_up"]), y1), 1)
    assert rel_err(cat, ref) < 1e-5
    gg = rnd("hg_g", ref.shape)
    cat.backward(gg.to(DEV))
    zo = z.clone().requires_grad_()
    O.upsample_bilinear_x4(zo).backward(gg[:, :z.shape[1]])
    assert rel_err(zc.grad, zo.grad) < 1e-5 and rel_err(yc.grad, gg[:, z.shape[1]:]) == 0
    x = rnd("hg_p", (3, 960, 9, 7), -2, 2)
    ref = torch.nn.functional.leaky_relu(torch.amax(x, dim=(2, 3), keepdim=True), 0.01)
    xd = x.to(DEV).requires_grad_()
    out = Fn.spatial_max(xd, leaky=True)
    assert torch.equal(out.cpu(), ref)
    out.backward(torch.ones_like(out))
    xo = x.clone().requires_grad_()
    torch.nn.functional.leaky_relu(torch.amax(xo, dim=(2, 3), keepdim=True), 0.01).sum().backward()
    assert rel_err(xd.grad, xo.grad) < 1e-6
    # all-negative maxima (the leaky slope in the one-launch backward) under a random gradient, 16-bit storage included
    xn = -x.abs() - 0.1
    gsm = rnd("hg_gsm", (3, 960, 1, 1), -1, 1)
    xo = xn.clone().requires_grad_()
    torch.nn.functional.leaky_relu(torch.amax(xo, dim=(2, 3), keepdim=True), 0.01).backward(gsm)
    for dt_ in (torch.float32, Fn.compute_dtype()):
        xd = xn.to(DEV, dt_).detach().requires_grad_()
        Fn.spatial_max(xd, leaky=True).backward(gsm.to(DEV))
        xr = xn.to(dt_).float().detach().requires_grad_()
        torch.nn.functional.leaky_relu(torch.amax(xr, dim=(2, 3), keepdim=True), 0.01).backward(gsm)
        assert rel_err(xd.grad.float(), xr.grad.to(dt_).float()) < 1e-6, dt_
    K, M = 5, 192
    w = rnd("hg_w", (K * M, K * M, 1, 1)) * 0.05
    b = rnd("hg_b", (K * M,))
    pooled = rnd("hg_pool", (3, K * M, 1, 1), -2, 2)
    refw = O._mix_weights(torch.nn.functional.conv2d(pooled, w, b), K, M)
    assert rel_err(Fn.mix_weights(pooled.to(DEV), w.to(DEV), b.to(DEV), K, M), refw) < 1e-5
    lg = torch.nn.functional.conv2d(pooled, w, b).to(DEV).requires_grad_()
    sw = Fn.softmax_k(lg, K, M)
    assert rel_err(sw, refw) < 1e-5
    gw = rnd("hg_gw", refw.shape)
    sw.backward(gw.to(DEV))
    lo = torch.nn.functional.conv2d(pooled, w, b).requires_grad_()
    O._mix_weights(lo, K, M).backward(gw)
    assert rel_err(lg.grad, lo.grad) < 1e-5


def test_reductions():
    Fn, _ = _imp()
    lik = rnd("red_l", (2, 192, 16, 16), 1e-9, 1.0)
    ref = float(torch.log2(lik.double()).sum())
    assert abs(float(Fn.sum_log2(lik.to(DEV))) - ref) < 1e-3 * abs(ref) * 1e-3 + 1e-2
    a, b = rnd("red_a", (2, 3, 64, 48)), rnd("red_b", (2, 3, 64, 48))
    ref = float(((a.double() - b.double()) ** 2).sum())
    got = float(Fn.sum_sq_diff(a.to(DEV), b.to(DEV).contiguous(memory_format=torch.channels_last)))
    assert abs(got - ref) < 1e-5 * ref


def test_rd_sums_takes_its_maximum_of_eight_maps_and_two_pairs():
    """``hesic_rd_sums`` at its declared limit -- 8 likelihood maps + 2 image pairs = 10 jobs + the end marker (the job table held
    9 first-block indices and the last two ran into the first map's pointer)."""
    Fn, _ = _imp()
    liks = [rnd(f"rd8_l{i}", (1, 16 + 8 * i, 8, 8 + i), 1e-9, 1.0) for i in range(8)]
    pairs = [(rnd(f"rd8_a{i}", (2, 3, 32, 48)), rnd(f"rd8_b{i}", (2, 3, 32, 48))) for i in range(2)]
    lik_acc = torch.zeros(8, dtype=torch.float64, device=DEV)
    sq_acc = torch.zeros(2, dtype=torch.float64, device=DEV)
    Fn.rd_sums([l.to(DEV) for l in liks], [lik_acc[i:i + 1] for i in range(8)], [(a.to(DEV), b.to(DEV)) for a, b in pairs],
               [sq_acc[i:i + 1] for i in range(2)])
    for i, l in enumerate(liks):
        ref = float(torch.log2(l.double()).sum())
        assert abs(float(lik_acc[i]) - ref) < 1e-6 * abs(ref) + 1e-2, i
    for i, (a, b) in enumerate(pairs):
        ref = float(((a.double() - b.double()) ** 2).sum())
        assert abs(float(sq_acc[i]) - ref) < 1e-5 * ref, i


@pytest.mark.parametrize("tr,inv", [(0, False), (1, True)], ids=["conv+gdn", "deconv+igdn"])
def test_fused_conv_gdn_matches_the_two_ops(tr, inv):
    """The inference-only fused epilogue equals conv() followed by GDN.forward (bf16 storage) and the fp32 oracle."""
    Fn, O = _imp()
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=9)
    beta, gamma = sd["g.beta"], sd["g.gamma"]
    Cin = 192 if tr else 128
    w = rnd("fg_w", (Cin, 128, 5, 5) if tr else (128, Cin, 5, 5)) * (0.06 if tr else 0.03)
    b = rnd("fg_b", (128,), -0.1, 0.1)
    x = bf(rnd("fg_x", (2, Cin, 12, 20) if tr else (2, Cin, 40, 24), -2, 2))
    ref = O.gdn((O.deconv if tr else O.conv)(x, bf(w), b, 2), beta, gamma, inv)
    xd = x.to(DEV, torch.bfloat16)
    args = dict(kernel_size=5, stride=2, padding=2, transposed=bool(tr))
    with torch.no_grad():
        two = Fn.gdn(Fn.conv2d(xd, w.to(DEV), b.to(DEV), **args), beta.to(DEV), gamma.to(DEV), inv)
        one = Fn.conv2d_gdn(xd, w.to(DEV), b.to(DEV), beta.to(DEV), gamma.to(DEV), inverse=inv, beta_min=1e-6,
                            packer=Fn.PackedWeight(), gdn_packer=Fn.PackedGdn(), **args)
    assert one.shape == two.shape and one.dtype == torch.bfloat16
    assert rel_err(one, ref) < 1.5e-2 and rel_err(two, ref) < 1.5e-2
    assert rel_err(one, two) < 1.5e-2


@pytest.mark.parametrize("tr,inv", [(0, False), (1, True)], ids=["conv+gdn", "deconv+igdn"])
def test_fused_conv_gdn_under_autograd_matches_the_two_ops(tr, inv):
    """With autograd on the fused kernel also stores the conv output; its backward (GDN backward on that tensor, then the
    conv gradients) must agree with differentiating the two separate ops, and with the fp32 oracle."""
    Fn, O = _imp()
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=9)
    Cin = 192 if tr else 128
    w = bf(rnd("fa_w", (Cin, 128, 5, 5) if tr else (128, Cin, 5, 5)) * (0.06 if tr else 0.03))
    b = rnd("fa_bb", (128,), -0.1, 0.1)
    x = bf(rnd("fa_xx", (2, Cin, 12, 20) if tr else (2, Cin, 40, 24), -2, 2))
    args = dict(kernel_size=5, stride=2, padding=2, transposed=bool(tr))

    def leaves(dev, dt=None):
        ts = [x.clone(), w.clone(), b.clone(), sd["g.beta"].clone(), sd["g.gamma"].clone()]
        ts = [t.to(dev) for t in ts]
        if dt is not None:
            ts[0] = ts[0].to(dt).contiguous(memory_format=torch.channels_last)
        return [t.requires_grad_() for t in ts]

    ro = leaves("cpu")
    yo = O.gdn((O.deconv if tr else O.conv)(ro[0], ro[1], ro[2], 2), ro[3], ro[4], inv)
    gy = bf(rnd("fa_gy", yo.shape))
    yo.backward(gy)
    outs = []
    for fused in (True, False):
        t = leaves(DEV, torch.bfloat16)
        if fused:
            y = Fn.conv2d_gdn(t[0], t[1], t[2], t[3], t[4], inverse=inv, beta_min=1e-6, packer=Fn.PackedWeight(),
                              gdn_packer=Fn.PackedGdn(), **args)
        else:
            y = Fn.gdn(Fn.conv2d(t[0], t[1], t[2], **args), t[3], t[4], inv)
        assert y.requires_grad and y.dtype == torch.bfloat16
        y.backward(gy.to(DEV, torch.bfloat16))
        outs.append((y, [p.grad for p in t]))
        assert rel_err(y, yo) < 1.5e-2
        for g, r in zip(outs[-1][1], ro):
            assert g is not None and rel_err(g, r.grad) < 4e-2
    for g1, g2 in zip(outs[0][1], outs[1][1]):
        assert rel_err(g1, g2) < 3e-2


def test_image_side_conv_gradients_bf16():
    """Backward of the 3-channel-side convs in bf16 storage (MFMA dgrad kernels + the narrow weight-gradient kernels)."""
    Fn, O = _imp()
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        # g_a_conv1: 3 -> 128, 5x5 s2
        x = rnd("ig_x", (2, 3, 64, 96), 0, 1)
        w = rnd("ig_w", (128, 3, 5, 5)) * 0.2
        b = rnd("ig_b", (128,), -0.1, 0.1)
        xo, wo, bo = x.clone().requires_grad_(), bf(w).requires_grad_(), b.clone().requires_grad_()
        yo = O.conv(bf(xo), wo, bo, 2)
        gy = bf(rnd("ig_g", yo.shape))
        yo.backward(gy)
        xd, wd, bd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        y = Fn.conv2d(xd, wd, bd, kernel_size=5, stride=2, padding=2)
        y.backward(gy.to(DEV, torch.bfloat16))
        assert rel_err(xd.grad, xo.grad) < 2e-2 and rel_err(wd.grad, wo.grad) < 2e-2 and rel_err(bd.grad, bo.grad) < 2e-2
        # g_s_conv4: 128 -> 3 transposed
        f = bf(rnd("ig_f", (2, 128, 24, 40)))
        wt = rnd("ig_wt", (128, 3, 5, 5)) * 0.05
        fo, wto = f.clone().requires_grad_(), bf(wt).requires_grad_()
        yo = O.deconv(fo, wto, None, 2)
        g2 = rnd("ig_g2", yo.shape)
        yo.backward(g2)
        fd, wtd = f.to(DEV, torch.bfloat16).requires_grad_(), wt.to(DEV).requires_grad_()
        yt = Fn.conv2d(fd, wtd, None, kernel_size=5, stride=2, padding=2, transposed=True)
        yt.backward(g2.to(DEV))
        assert rel_err(fd.grad, fo.grad) < 2e-2 and rel_err(wtd.grad, wto.grad) < 2e-2
        # pre_conv / after_conv: 6 -> 3 stride 1 (fp32 on both sides)
        for tr in (False, True):
            x6 = rnd("ig_x6", (2, 6, 40, 72), 0, 1)
            w6 = rnd("ig_w6" + str(tr), (6, 3, 5, 5) if tr else (3, 6, 5, 5)) * 0.1
            b6 = rnd("ig_b6", (3,), -0.1, 0.1)
            xo, wo, bo = x6.clone().requires_grad_(), w6.clone().requires_grad_(), b6.clone().requires_grad_()
            yo = (O.deconv if tr else O.conv)(xo, wo, bo, 1)
            g6 = rnd("ig_g6", yo.shape)
            yo.backward(g6)
            xd, wd, bd = x6.to(DEV).requires_grad_(), w6.to(DEV).requires_grad_(), b6.to(DEV).requires_grad_()
            Fn.conv2d(xd, wd, bd, kernel_size=5, stride=1, padding=2, transposed=tr).backward(g6.to(DEV))
            assert rel_err(xd.grad, xo.grad) < 2e-4 and rel_err(wd.grad, wo.grad) < 2e-4 and rel_err(bd.grad, bo.grad) < 2e-4, tr
    finally:
        Fn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("inv", [False, True])
def test_gdn128_backward_bf16(inv):
    """The one-pass MFMA GDN backward (bf16 storage) against the oracle's autograd."""
    Fn, O = _imp()
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=4)
    x = bf(rnd("gb_x", (2, 128, 20, 24), -3, 3))
    g = bf(rnd("gb_g", x.shape))
    xo, bo, go = x.clone().requires_grad_(), sd["g.beta"].clone().requires_grad_(), sd["g.gamma"].clone().requires_grad_()
    O.gdn(xo, bo, go, inv).backward(g)
    xd = x.to(DEV, torch.bfloat16).requires_grad_()
    bd, gd = sd["g.beta"].to(DEV).requires_grad_(), sd["g.gamma"].to(DEV).requires_grad_()
    Fn.gdn(xd, bd, gd, inv).backward(g.to(DEV, torch.bfloat16))
    assert rel_err(xd.grad, xo.grad) < 2e-2
    assert rel_err(bd.grad, bo.grad) < 2e-2
    assert rel_err(gd.grad, go.grad) < 3e-2


def test_fused_image_conv_gdn():
    """g_a_gdn1(g_a_conv1(image)) in one kernel (bf16 storage) vs the oracle on bf16-rounded operands."""
    Fn, O = _imp()
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=11)
    x = rnd("fi_x", (2, 3, 72, 104), 0, 1)                 # not a multiple of the 2x16 wave tile
    w = rnd("fi_w", (128, 3, 5, 5)) * 0.25
    b = rnd("fi_b", (128,), -0.1, 0.1)
    ref = O.gdn(O.conv(bf(x), bf(w), b, 2), sd["g.beta"], sd["g.gamma"], False)
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            y = Fn.conv2d_gdn(x.to(DEV), w.to(DEV), b.to(DEV), sd["g.beta"].to(DEV), sd["g.gamma"].to(DEV), kernel_size=5, stride=2,
                              padding=2, transposed=False, inverse=False, beta_min=1e-6, packer=None, gdn_packer=Fn.PackedGdn())
    finally:
        Fn.set_compute_dtype(torch.float32)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    assert rel_err(y, ref) < 1.5e-2


@pytest.mark.parametrize("hw", [(50, 70), (51, 71), (2, 2), (130, 34)], ids=["even_w", "odd_w_generic", "tiny", "tall"])
@pytest.mark.parametrize("inv", [False, True], ids=["gdn", "igdn"])
def test_fused_image_conv_gdn_shapes(hw, inv):
    """conv1 + GDN in one kernel on sizes that end in partial wave tiles: even widths take the 8-byte buffer-load form
    (poisoned offsets for the padding), odd widths the generic gather -- both against the oracle on bf16-rounded operands."""
    Fn, O = _imp()
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=13)
    x = rnd(f"fis_x{hw}", (3, 3) + hw, 0, 1)
    w = rnd("fis_w", (128, 3, 5, 5)) * 0.25
    b = rnd("fis_b", (128,), -0.1, 0.1)
    ref = O.gdn(O.conv(bf(x), bf(w), b, 2), sd["g.beta"], sd["g.gamma"], inv)
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            y = Fn.conv2d_gdn(x.to(DEV), w.to(DEV), b.to(DEV), sd["g.beta"].to(DEV), sd["g.gamma"].to(DEV), kernel_size=5, stride=2,
                              padding=2, transposed=False, inverse=inv, beta_min=1e-6, packer=None, gdn_packer=Fn.PackedGdn())
    finally:
        Fn.set_compute_dtype(torch.float32)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 1.5e-2


@pytest.mark.parametrize("hw", [(23, 37), (6, 14), (7, 15), (1, 1), (64, 5)])
def test_wide_to_narrow_deconv_shapes(hw):
    """g_s_conv4 (128 -> 3 transposed, splat GEMM + col2im): input maps that end in partial 6x14 patches."""
    Fn, O = _imp()
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        f = bf(rnd(f"w2n_f{hw}", (2, 128) + hw))
        wt = rnd("w2n_w", (128, 3, 5, 5)) * 0.05
        bt = rnd("w2n_b", (3,), -0.1, 0.1)
        keep_sh, Fn.SHAPED_WEIGHTS = Fn.SHAPED_WEIGHTS, False      # round-to-nearest weights, like the oracle call below
        try:
            with torch.no_grad():
                y = Fn.conv2d(f.to(DEV, torch.bfloat16), wt.to(DEV), bt.to(DEV), kernel_size=5, stride=2, padding=2, transposed=True)
        finally:
            Fn.SHAPED_WEIGHTS = keep_sh
        ref = O.deconv(f, bf(wt), bt, 2)
        assert y.shape == ref.shape and y.dtype == torch.float32
        # the MFMA kernel multiplies bf16-rounded weights exactly like the oracle call; a 1x1 map (ambiguous NHWC strides) takes
        # the scalar kernel, which keeps the fp32 weights: bf16 bar
        assert rel_err(y, ref) < (6e-3 if hw == (1, 1) else 1e-4)
    finally:
        Fn.set_compute_dtype(torch.float32)


def test_gmm_bf16_fast_path_matches_fp32_path():
    """bf16 storage takes the pair-per-thread forward and the compile-time-K backward with the branch-free erfc: the
    likelihoods stay within 2e-5 of the fp32 (ocml erfc) kernels run on the same bf16-rounded operands, rounded latents
    are identical, gradients agree to bf16 resolution."""
    Fn, _ = _imp()
    B, M, H, W, K = 2, 192, 6, 10, 5
    y = bf(rnd("gb_y", (B, M, H, W), -6, 6))
    sc = bf(rnd("gb_s", (B, K * M, H, W), 0.02, 3))
    mu = bf(rnd("gb_m", (B, K * M, H, W), -2, 2))
    wt = torch.softmax(rnd("gb_w", (B, K, M), -1, 1), 1).reshape(B, K * M, 1, 1)
    gl = rnd("gb_gl", (B, M, H, W), -1, 1)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        Fn.set_compute_dtype(dt)
        try:
            ins = [t.to(DEV, dt).contiguous(memory_format=torch.channels_last).requires_grad_() for t in (y, sc, mu)]
            wd = wt.to(DEV).requires_grad_()
            yh, lik = Fn.gaussian_mixture(ins[0], ins[1], ins[2], wd, K=K)
            (lik * gl.to(DEV)).sum().backward()
            res[dt] = (yh.detach().float().cpu(), lik.detach().cpu(), [t.grad.float().cpu() for t in ins[1:]], wd.grad.cpu())
            # K == 1 with the mean in the quantiser (GaussianConditional)
            i1 = [t.to(DEV, dt).contiguous(memory_format=torch.channels_last) for t in (y, sc[:, :M], mu[:, :M])]
            yh1, lik1 = Fn.gaussian_conditional(i1[0], i1[1], means=i1[2])
            res[dt] += (yh1.float().cpu(), lik1.cpu())
        finally:
            Fn.set_compute_dtype(torch.float32)
    f, h = res[torch.float32], res[torch.bfloat16]
    assert torch.equal(f[0], h[0])
    torch.testing.assert_close(h[1], f[1], rtol=2e-5, atol=1e-9)
    for a, b_ in zip(h[2], f[2]):
        assert rel_err(a, b_) < 1e-2
    assert rel_err(h[3], f[3]) < 1e-3
    assert torch.equal(bf(f[4]), h[4])          # round(y - mu) + mu, stored in bf16
    torch.testing.assert_close(h[5], f[5], rtol=2e-5, atol=1e-9)


def test_pooled_linear_head_forward_backward():
    """The 1x1 conv on the pooled (B, K*M, 1, 1) vector under autograd (newnet1.py:500): value and the three gradients
    against torch's fp32 linear on the CPU."""
    Fn, _ = _imp()
    B, N = 3, 960
    p = rnd("pl_p", (B, N, 1, 1), -2, 2)
    w = rnd("pl_w", (N, N, 1, 1)) * 0.05
    b = rnd("pl_b", (N,), -0.1, 0.1)
    g = rnd("pl_g", (B, N, 1, 1))
    pr, wr, br = (t.clone().double().requires_grad_() for t in (p, w, b))
    ref = torch.nn.functional.conv2d(pr, wr, br)
    (ref * g.double()).sum().backward()
    pd, wd, bd = (t.to(DEV).requires_grad_() for t in (p, w, b))
    y = Fn.pooled_linear(pd, wd, bd)
    (y * g.to(DEV)).sum().backward()
    assert y.shape == (B, N, 1, 1)
    assert rel_err(y, ref) < 1e-5
    assert rel_err(pd.grad, pr.grad) < 1e-5 and rel_err(wd.grad, wr.grad) < 1e-5 and rel_err(bd.grad, br.grad) < 1e-5


@pytest.mark.parametrize("hw", [(64, 64), (37, 45), (16, 32), (5, 3), (50, 130)])
def test_conv3x3_c32_enhancement_kernel(hw):
    """hesic_conv3x3_c32_forward (32-channel 3x3 convs of the enhancement stage): plain, LeakyReLU + one / two residuals,
    the 32 -> 3 planar output form, and the packed 6 -> 32 input conv, on sizes that end in partial 16x32 tiles; against the
    oracle conv on bf16-rounded operands (fp32 accumulation on both sides, one final bf16 rounding on the device)."""
    Fn, O = _imp()
    H, W = hw
    x = bf(rnd(f"c32_x{hw}", (2, 32, H, W), -2, 2))
    r1, r2 = bf(rnd(f"c32_r1{hw}", (2, 32, H, W))), bf(rnd(f"c32_r2{hw}", (2, 32, H, W)))
    w = rnd("c32_w", (32, 32, 3, 3)) * 0.1
    b = rnd("c32_b", (32,), -0.2, 0.2)
    w3, b3 = rnd("c32_w3", (3, 32, 3, 3)) * 0.1, rnd("c32_b3", (3,), -0.2, 0.2)
    img = rnd(f"c32_img{hw}", (2, 3, H, W), 0, 1)
    leaky = torch.nn.functional.leaky_relu
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        xd = x.to(DEV, torch.bfloat16)
        with torch.no_grad():
            assert Fn.conv3x3_c32_ok(xd, w.to(DEV))
            y0 = Fn.conv3x3_c32(xd, w.to(DEV), b.to(DEV))
            y1 = Fn.conv3x3_c32(xd, w.to(DEV), b.to(DEV), act=2, res1=r1.to(DEV, torch.bfloat16))
            y2 = Fn.conv3x3_c32(xd, w.to(DEV), None, act=2, res1=r1.to(DEV, torch.bfloat16), res2=r2.to(DEV, torch.bfloat16))
            y3 = Fn.conv3x3_c32(xd, w3.to(DEV), b3.to(DEV), res1=img.to(DEV))
            a6, b6 = rnd(f"c32_a{hw}", (2, 3, H, W), 0, 1), rnd(f"c32_bb{hw}", (2, 3, H, W), 0, 1)
            w6 = rnd("c32_w6", (32, 6, 3, 3)) * 0.2
            wpad = torch.zeros(32, 32, 3, 3)
            wpad[:, :6] = w6
            y6 = Fn.conv3x3_c32(Fn.pack_images_c32(a6.to(DEV), b6.to(DEV)), wpad.to(DEV), b.to(DEV))
    finally:
        Fn.set_compute_dtype(torch.float32)
    c = O.conv(x, bf(w), b, 1)
    assert y0.dtype == torch.bfloat16 and y0.shape == c.shape and rel_err(y0, c) < 1e-2
    assert rel_err(y1, leaky(c, 0.01) + r1) < 1e-2
    assert rel_err(y2, leaky(O.conv(x, bf(w), None, 1), 0.01) + r1 + r2) < 1e-2
    assert y3.dtype == torch.float32 and y3.shape == (2, 3, H, W)
    assert rel_err(y3, O.conv(x, bf(w3), b3, 1) + img) < 1e-4
    assert rel_err(y6, O.conv(bf(torch.cat((a6, b6), 1)), bf(w6), b, 1)) < 1e-2


@pytest.mark.parametrize("hw", [(64, 64), (37, 45), (5, 3), (50, 130)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_conv3x3_c32_img6_is_pack_plus_conv_bit_for_bit(hw, dtype):
    """hesic_conv3x3_c32_forward_img6 (round 6: Enhancement.conv1 on the two planar images in one launch, newnet1.py:300-301) against the two-launch
    route of rounds 3 - 5 (pack_images_c32 + the 32-channel kernel on a weight zero-padded along Cin): the same products in the same order."""
    Fn, O = _imp()
    H, W = hw
    xa, xb = rnd(f"i6_a{hw}", (2, 3, H, W)), rnd(f"i6_b{hw}", (2, 3, H, W))
    w, b = rnd("i6_w", (32, 6, 3, 3)) * 0.2, rnd("i6_bias", (32,), -0.2, 0.2)
    Fn.set_compute_dtype(dtype)
    try:
        with torch.no_grad():
            one = Fn.conv3x3_c32_img6(xa.to(DEV), xb.to(DEV), w.to(DEV), b.to(DEV))
            w32 = torch.zeros(32, 32, 3, 3)
            w32[:, :6] = w
            two = Fn.conv3x3_c32(Fn.pack_images_c32(xa.to(DEV), xb.to(DEV)), w32.to(DEV), b.to(DEV))
    finally:
        Fn.set_compute_dtype(torch.float32)
    assert one.dtype == dtype and one.shape == (2, 32, H, W) and torch.equal(one, two)
    assert rel_err(one, O.conv(torch.cat((xa, xb), 1), w, b, 1)) < (1e-2 if dtype == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("hw", [(64, 64), (37, 45), (16, 32), (5, 3), (50, 130), (96, 160)])
@pytest.mark.parametrize("skip", [False, True])
def test_resblock_c32_agrees_with_the_two_launch_path(hw, skip):
    """hesic_resblock_c32_forward (a whole ResidualBlock of the enhancement stage per launch, layers.py:125-147) against two
    hesic_conv3x3_c32_forward launches: same rounding points (16-bit intermediate, fp32 accumulation), on sizes that end in partial tiles and on
    single-tile images (every intermediate pixel of the ring outside the image must be ZERO, not conv1 evaluated there), and close to the oracle
    composition.  Rounds 3 - 5: bit-identical.  Round 6: the one-launch kernel sums a tap's 32 channels in ONE 16 x 16 x 32 MFMA, the single-conv kernel
    in two K = 16 halves on two chains -- another fp32 summation order, so a few of the 16-bit results land on the neighbouring value (and an
    intermediate value that rounds the other way moves the outputs around it).  Bars: every value within ONE bf16 ulp of the tensor's largest value
    (2^-8 of it; measured <= 3.4e-3), fewer than 2e-3 of the values different at all (measured <= 7.3e-4)."""
    Fn, O = _imp()
    H, W = hw
    x = bf(rnd(f"rb_x{hw}", (2, 32, H, W), -2, 2))
    r2 = bf(rnd(f"rb_r2{hw}", (2, 32, H, W))) if skip else None
    w1, w2 = rnd("rb_w1", (32, 32, 3, 3)) * 0.1, rnd("rb_w2", (32, 32, 3, 3)) * 0.1
    b1, b2 = rnd("rb_b1", (32,), -0.2, 0.2), rnd("rb_b2", (32,), -0.2, 0.2)
    leaky = torch.nn.functional.leaky_relu
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        xd = x.to(DEV, torch.bfloat16)
        r2d = None if r2 is None else r2.to(DEV, torch.bfloat16)
        with torch.no_grad():
            mid = Fn.conv3x3_c32(xd, w1.to(DEV), b1.to(DEV), act=2)
            two = Fn.conv3x3_c32(mid, w2.to(DEV), b2.to(DEV), act=2, res1=xd, res2=r2d)
            one = Fn.resblock_c32(xd, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), act=2, res2=r2d)
            nob = Fn.resblock_c32(xd, w1.to(DEV), None, w2.to(DEV), None, act=2, res2=r2d)
            two_nob = Fn.conv3x3_c32(Fn.conv3x3_c32(xd, w1.to(DEV), None, act=2), w2.to(DEV), None, act=2, res1=xd, res2=r2d)
    finally:
        Fn.set_compute_dtype(torch.float32)
    assert one.dtype == torch.bfloat16 and one.shape == two.shape
    for got, want in ((one, two), (nob, two_nob)):
        d = (got.float() - want.float()).abs()
        scale = float(want.float().abs().max())
        meas = {"max_abs_over_scale": float(d.max()) / scale, "share_different": float((d > 0).float().mean()),
                "share_beyond_one_ulp_of_scale": float((d > scale * 2.0 ** -8).float().mean())}
        print("measured:", {k: float("%.3g" % v) for k, v in meas.items()})
        assert meas["max_abs_over_scale"] <= 2.0 ** -8 and meas["share_different"] < 2e-3, meas          # measured: <= 3.4e-3, <= 7.3e-4
    ref = leaky(O.conv(bf(leaky(O.conv(x, bf(w1), b1, 1), 0.01)), bf(w2), b2, 1), 0.01) + x + (0 if r2 is None else r2)
    assert rel_err(one, ref) < 1e-2


# ------------------------------------------------------------------ fp32 latents of the bf16 mode (round 2)
@pytest.mark.parametrize("shape,split", [((2, 128, 192, 5, 2, 64), True), ((2, 128, 128, 5, 2, 16), True), ((1, 128, 960, 5, 1, 32), False),
                                         ((4, 128, 128, 5, 2, 128), False)],
                         ids=["conv4_splitk_per_image", "hyper_z_splitk", "sigma960", "full_size_plain"])
def test_conv_f32out_is_the_unrounded_accumulator(shape, split):
    """hesic_conv2d_forward_f32out: the fp32 copy equals the oracle on the bf16-rounded operands to fp32 accuracy (not bf16
    accuracy), the bf16 copy is its rounding, `y == NULL` writes only the fp32 tensor; plain and split-K launches."""
    Fn, O = _imp()
    from hesic_amd import _lib as L
    import ctypes as C
    B, Cin, Cout, k, s, H = shape
    x = bf(rnd("f32o_x", (B, Cin, H, H), -2, 2))
    w = bf(rnd("f32o_w", (Cout, Cin, k, k)) * 0.03)
    b = rnd("f32o_b", (Cout,), -0.1, 0.1)
    ref = torch.relu(O.conv(x, w, b, s))
    Ho = ref.shape[-1]
    xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wp = Fn.PackedWeight().get(w.to(DEV), None, Cout, Cin, k, k, False, False, torch.bfloat16)
    d = L.ConvDesc(B, H, H, Cin, Ho, Ho, Cout, k, k, s, k // 2, 0, L.BF16, L.ACT_RELU, 0, Cin, 0, Cout, 0, 0)
    need = int(L.lib().hesic_conv2d_f32out_ws_bytes(C.byref(d)))      # the latent-grade launch's own query: K split decided per image
    assert (need > 0) == split
    if split:        # ... so the scratch per image does not depend on the batch
        d1 = L.ConvDesc(1, H, H, Cin, Ho, Ho, Cout, k, k, s, k // 2, 0, L.BF16, L.ACT_RELU, 0, Cin, 0, Cout, 0, 0)
        assert int(L.lib().hesic_conv2d_f32out_ws_bytes(C.byref(d1))) * B == need
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
    lo = torch.zeros((B, Cout, Ho, Ho), device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    hi = torch.full((B, Cout + 8, Ho, Ho), 7.0, device=DEV).contiguous(memory_format=torch.channels_last)
    L.call("hesic_conv2d_forward_f32out", C.byref(d), L.ptr(xd), L.ptr(wp), L.ptr(b.to(DEV)), L.ptr(lo), L.ptr(hi), Cout + 8, 4,
           L.ptr(ws), need, L.stream())
    assert rel_err(hi[:, 4:4 + Cout], ref) < 2e-5
    assert torch.equal(hi[:, :4], torch.full_like(hi[:, :4], 7.0)) and torch.equal(hi[:, 4 + Cout:], torch.full_like(hi[:, 4 + Cout:], 7.0))
    assert torch.equal(lo, hi[:, 4:4 + Cout].to(torch.bfloat16))
    only = torch.zeros((B, Cout, Ho, Ho), device=DEV).contiguous(memory_format=torch.channels_last)
    L.call("hesic_conv2d_forward_f32out", C.byref(d), L.ptr(xd), L.ptr(wp), L.ptr(b.to(DEV)), None, L.ptr(only), Cout, 0, L.ptr(ws), need, L.stream())
    assert torch.equal(only, hi[:, 4:4 + Cout])
    df = L.ConvDesc(B, H, H, Cin, Ho, Ho, Cout, k, k, s, k // 2, 0, L.F32, L.ACT_RELU, 0, Cin, 0, Cout, 0, 0)
    with pytest.raises(RuntimeError):                            # fp32 storage has no separate fp32 copy
        L.call("hesic_conv2d_forward_f32out", C.byref(df), L.ptr(xd), L.ptr(wp), None, None, L.ptr(only), Cout, 0, None, 0, L.stream())


def test_entropy_kernels_f32in_match_the_oracle_bit_exact_indices():
    """hesic_gmm_forward_f32in / hesic_eb_forward_f32in / hesic_round (bf16 mode with fp32 latents): rounded latents are
    bit-exact against the oracle on the same fp32 inputs, likelihoods within 1e-5 (branch-free erfc), y_hat is stored bf16."""
    Fn, O = _imp()
    B, M, K, H = 2, 192, 5, 8
    y = rnd("f32in_y", (B, M, H, H), -12, 12)
    sc = rnd("f32in_s", (B, M * K, H, H), 0.05, 4.0)
    mu = rnd("f32in_m", (B, M * K, H, H), -3, 3)
    wt = torch.softmax(rnd("f32in_w", (B, K, M, 1, 1), -1, 1), 1).reshape(B, K * M, 1, 1)
    yh_o, lik_o = O.gmm_forward(y, sc, mu, wt, K)
    cl = torch.channels_last
    torch.set_grad_enabled(False)            # the f32in kernels are the inference forms
    yh, lik = Fn.gaussian_mixture(y.to(DEV).contiguous(memory_format=cl), sc.to(DEV).contiguous(memory_format=cl),
                                  mu.to(DEV).contiguous(memory_format=cl), wt.to(DEV), K, out_dtype=torch.bfloat16)
    assert yh.dtype == torch.bfloat16 and torch.equal(yh.float().cpu(), yh_o)
    assert float((lik.cpu() - lik_o).abs().max()) < 1e-5
    # single Gaussian with means in the quantiser, scale / mean = the two halves of one tensor (HESIC+)
    gp = rnd("f32in_gp", (B, 2 * M, H, H), -2, 2)
    gp[:, :M] = gp[:, :M].abs() + 0.05
    yh_o, lik_o = O.gc_forward(y, gp[:, :M], gp[:, M:])
    gpd = gp.to(DEV).contiguous(memory_format=cl)
    s_, m_ = gpd.chunk(2, 1)
    yh, lik = Fn.gaussian_conditional(y.to(DEV).contiguous(memory_format=cl), s_, m_, out_dtype=torch.float32)
    assert torch.equal(yh.cpu(), yh_o) and float((lik.cpu() - lik_o).abs().max()) < 1e-5
    # round-half-even into bf16
    r = Fn.round_to(y.to(DEV), torch.bfloat16)
    assert r.dtype == torch.bfloat16 and torch.equal(r.float().cpu(), torch.round(y))
    half = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 3.4999998], device=DEV).reshape(1, 6, 1, 1)
    assert Fn.round_to(half, torch.bfloat16).float().flatten().tolist() == [0.0, 2.0, 2.0, -0.0, -2.0, 3.0]
    torch.set_grad_enabled(True)


def test_entropy_bottleneck_f32in():
    Fn, O = _imp()
    from compressai.entropy_models import EntropyBottleneck
    torch.manual_seed(3)
    eb = EntropyBottleneck(128)
    with torch.no_grad():
        eb.quantiles[:, 0, 1] = rnd("ebq", (128,), -0.4, 0.4)
    z = rnd("eb32_z", (2, 128, 4, 4), -6, 6)
    P = {k: v.detach() for k, v in eb.state_dict().items()}
    zh_o, lik_o = O.eb_forward(P, "", z)
    eb = eb.to(DEV).eval()
    with torch.no_grad():
        zh, lik = eb.forward_with_noise(z.to(DEV), None, out_dtype=torch.bfloat16)
        zh32, lik32 = eb.forward_with_noise(z.to(DEV), None)
    assert zh.dtype == torch.bfloat16 and torch.equal(zh, zh32.to(torch.bfloat16)) and torch.equal(lik, lik32)
    assert torch.equal(zh32.cpu(), zh_o) and rel_err(lik, lik_o) < 2e-4


def test_entropy_bottleneck_honours_its_likelihood_bound():
    """EntropyModel(likelihood_bound=...) reaches the kernel (slot 60 of the parameter table), forward and backward."""
    from compressai.entropy_models import EntropyBottleneck
    z = rnd("lb_z", (2, 128, 4, 4), -3000, 3000)                  # far tails: raw likelihoods underflow to 0
    for bound in (1e-3, 0.0):
        eb = EntropyBottleneck(128, likelihood_bound=bound).to(DEV).eval()
        zd = z.to(DEV).requires_grad_()
        _, lik = eb(zd)
        if bound:
            assert float(lik.min()) == pytest.approx(bound, rel=1e-6)
        else:
            assert float(lik.min()) < 1e-9
        with torch.no_grad():
            _, lik2 = eb(z.to(DEV))                               # cached-table inference path
        assert torch.equal(lik2, lik.detach())


def test_warp_inverse_map_is_the_warp_by_the_inverse():
    """warp(x, inverse(H)) == warp(x, H, inverse_map=True): Independent_EN (newnet1.py:1290-1291) without 3x3 inversion launches."""
    Fn, O = _imp()
    x1, _, Hm = synthetic.stereo_batch(3, 2, 96, 128)
    a = Fn.warp_perspective(x1.to(DEV), torch.inverse(Hm).to(DEV), (96, 128))
    b = Fn.warp_perspective(x1.to(DEV), Hm.to(DEV), (96, 128), inverse_map=True)
    ref = O.warp_perspective(x1, torch.inverse(Hm), (96, 128))
    # torch.inverse is fp32: its rounding moves the sample positions by ~1e-4 px against the exact inverse map
    assert float((a - b).abs().max()) < 2e-4 and float((b.cpu() - ref).abs().max()) < 2e-4


def test_image_side_conv_gdn_is_bit_stable_across_launches():
    """Regression (round 2): the fused 3 -> 128 conv + GDN kernel stored its rows with `buffer_store_dwordx4 ... soffset=SGPR`;
    the compiler skips the "no VALU write of the store data within one wait state" rule for that form and had scheduled the next
    store's address select into data dword 0 right behind the store.  On gfx950 that corrupted two channels of a few 16-byte
    chunks in roughly one forward out of 25.  600 launches interleaved with allocator churn must all be bit-identical."""
    Fn, O = _imp()
    from compressai.layers import GDN
    from compressai.models.utils import conv
    prev = Fn.compute_dtype()
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(5)
        c1, g1 = conv(3, 128).to(DEV), GDN(128).to(DEV)
        x = synthetic.stereo_batch(2, 2, 256, 256)[0].to(DEV)
        with torch.no_grad():
            ref = c1.run_gdn(x, g1).clone()
            bad = 0
            for it in range(600):
                if it % 7 == 0:
                    junk = torch.full((1 << (10 + it % 13),), 1e30, device=DEV)
                    del junk
                bad += int(not torch.equal(c1.run_gdn(x, g1), ref))
        assert bad == 0, bad
    finally:
        Fn.set_compute_dtype(prev)


@pytest.mark.parametrize("transposed", [False, True], ids=["conv", "deconv"])
def test_grouped_conv_matches_separate_convs(transposed):
    """hesic_conv2d_forward_grouped (the hyper-synthesis branches as one launch): shared input with an activation split, and two
    groups reading their own input slices with fp32 output, against the oracle per branch and against the one-conv-at-a-time path."""
    Fn, O = _imp()
    from hesic_amd import _lib as L
    B, Cin, H = 2, 128, 8
    cl = torch.channels_last
    ws = [bf(rnd(f"g{transposed}w{i}", ((Cin, co, 5, 5) if transposed else (co, Cin, 5, 5))) * 0.03) for i, co in enumerate((128, 128, 128))]
    bs = [rnd(f"g{transposed}b{i}", (128,), -0.1, 0.1) for i in range(3)]
    x = bf(rnd(f"g{transposed}x", (B, Cin, H, H), -2, 2))
    op = (lambda x_, w_, b_: O.deconv(x_, w_, b_, 2)) if transposed else (lambda x_, w_, b_: O.conv(x_, w_, b_, 1))
    acts = [torch.relu, lambda t: torch.nn.functional.leaky_relu(t, 0.01), lambda t: torch.nn.functional.leaky_relu(t, 0.01)]
    ref = torch.cat([a(op(x, w, b)) for a, w, b in zip(acts, ws, bs)], 1)
    torch.set_grad_enabled(False)
    try:
        xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=cl)
        wd, bd = [w.to(DEV) for w in ws], [b.to(DEV) for b in bs]
        kw = dict(kernel_size=5, stride=2 if transposed else 1, padding=2, transposed=transposed)
        y, offs = Fn.conv2d_grouped(xd, wd, bd, Fn.PackedGroup(), shared_input=True, acts=[L.ACT_RELU, L.ACT_LEAKY, L.ACT_LEAKY], **kw)
        assert offs == [0, 128, 256] and y.shape == ref.shape and rel_err(y, ref) < 2e-2
        one = torch.cat([Fn.conv2d(xd, w, b, act=a, **kw) for w, b, a in zip(wd, bd, (L.ACT_RELU, L.ACT_LEAKY, L.ACT_LEAKY))], 1)
        assert rel_err(y, one) < 4e-3                                 # split-K vs one block per tile: one bf16 ulp
        # second layer: two groups, each reading its own 128-channel slice of y, fp32 output, second branch without activation
        ref2 = torch.cat([torch.relu(op(bf(ref[:, :128]), ws[0], bs[0])), op(bf(ref[:, 128:256]), ws[1], bs[1])], 1)
        y2, _ = Fn.conv2d_grouped(y, wd[:2], bd[:2], Fn.PackedGroup(), shared_input=False, acts=[L.ACT_RELU, L.ACT_NONE], f32_out="only", **kw)
        assert y2.dtype == torch.float32 and y2.shape == ref2.shape and rel_err(y2, ref2) < 2e-2
        sl = Fn.conv2d_slice(y, 128, wd[1], bd[1], **kw)               # in-place channel slice == the second group
        assert torch.equal(sl, y2[:, 128:].to(torch.bfloat16)) or rel_err(sl, y2[:, 128:]) < 4e-3
        if not transposed:      # branches whose Cout is not a multiple of the 128-cout tile start on padded offsets (960 -> 1024)
            w9 = [bf(rnd(f"g9w{i}", (192, Cin, 5, 5)) * 0.03).to(DEV) for i in range(2)]
            b9 = [rnd(f"g9b{i}", (192,), -0.1, 0.1).to(DEV) for i in range(2)]
            y9, o9 = Fn.conv2d_grouped(y, w9, b9, Fn.PackedGroup(), shared_input=False, acts=[L.ACT_RELU, L.ACT_NONE], f32_out="only", **kw)
            assert o9 == [0, 256] and y9.shape[1] == 512
            for g in range(2):
                r9 = O.conv(bf(ref[:, 128 * g:128 * (g + 1)]), w9[g].cpu(), b9[g].cpu(), 1)
                assert rel_err(y9[:, o9[g]:o9[g] + 192], torch.relu(r9) if g == 0 else r9) < 2e-2
            assert float(y9[:, 192:256].abs().max()) == 0            # pad couts: zero weights, zero bias, ReLU
    finally:
        torch.set_grad_enabled(True)


@pytest.mark.parametrize("shape", [(2, 40, 96), (1, 7, 50), (3, 16, 64)])
def test_conv3x3_c32_weight_gradient_matches_autograd(shape):
    """hesic_conv3x3_c32_wgrad (the stage-2 enhancement layers' weight / bias gradient) against torch autograd on the same bf16-rounded
    operands, incl. strips that end inside an image row, an odd height and a narrower conv (6 input / 3 output channels)."""
    from hesic_amd import functional as Fn
    B, H, W = shape
    g_ = torch.Generator().manual_seed(3)
    x = (torch.rand(B, 32, H, W, generator=g_) - 0.5).bfloat16()
    gy = (torch.rand(B, 32, H, W, generator=g_) - 0.5).bfloat16()
    for cout, cin in ((32, 32), (32, 6), (3, 32)):
        w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
        b = torch.zeros(cout, requires_grad=True)
        y = torch.nn.functional.conv2d(x.float()[:, :cin], w, b, padding=1)
        (y * gy.float()[:, :cout]).sum().backward()
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
        gd = gy.to(DEV).contiguous(memory_format=torch.channels_last)
        dw, db = Fn.conv3x3_c32_wgrad(xd, gd, w.detach().to(DEV), b.detach().to(DEV))
        scale = float(w.grad.abs().max())
        assert float((dw.cpu() - w.grad).abs().max()) <= 2e-3 * scale, (cout, cin)
        torch.testing.assert_close(db.cpu(), b.grad, rtol=2e-3, atol=2e-3 * float(b.grad.abs().max()))


def test_hilo_analysis_kernels_are_bit_stable_across_launches():
    """The bf16x3 analysis stack (fused hi/lo conv1 + GDN with wave-private LDS staging, hi/lo implicit-GEMM layers with their GDN
    epilogue, the fp32 + |y| pair outputs) launched 300 times with allocator churn in between: every launch bit-identical to the
    first (the class of bug of the round-2 store hazard and of the staging reorder found while writing the hi/lo conv1 kernel)."""
    Fn, O = _imp()
    from hesic_amd import models
    prev = Fn.compute_dtype()
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        enc = models.Encoder1(128, 192).to(DEV).eval()
        synthetic.fill_state_dict_(enc.state_dict())
        x = synthetic.stereo_batch(2, 2, 256, 320)[0].to(DEV)
        with torch.no_grad():
            lo0, y0 = enc.latent_hilo(x, True, True)
            lo0, y0 = lo0.t.clone(), y0.clone()
            bad = 0
            for it in range(300):
                if it % 7 == 0:
                    junk = torch.full((1 << (10 + it % 13),), 1e30, device=DEV)
                    del junk
                lo, y = enc.latent_hilo(x, True, True)
                bad += int(not (torch.equal(y, y0) and torch.equal(lo.t, lo0)))
        assert bad == 0, bad
        assert torch.equal(lo0[:, :192].float() + lo0[:, 192:].float() >= 0, torch.ones_like(y0, dtype=torch.bool))      # |y| pairs
    finally:
        Fn.set_compute_dtype(prev)


def test_gdn_packs_of_a_training_step_are_refreshed_in_one_launch():
    """Under ``train_pack_cache`` (the Trainer's step) every GDN keeps a persistent parameter pack and ``repack_all`` refreshes all of
    them with ONE launch (``hesic_gdn_pack_params_batched``): after the parameters moved, the refreshed packs equal what the
    single-GDN launch writes, and a later ``get`` in grad mode returns them without packing again."""
    Fn, _ = _imp()
    from compressai.layers import GDN
    prev_dt = Fn.compute_dtype()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    prev = Fn.train_pack_cache(True)
    try:
        gdns = [GDN(128, inverse=bool(i & 1)).to(DEV) for i in range(3)]
        with torch.enable_grad():
            first = [g.packer().get(g.beta, g.gamma, g.beta_min) for g in gdns]
            with torch.no_grad():                        # an optimiser step that does not bump the version counters
                for i, g in enumerate(gdns):
                    g.gamma.data.add_(0.01 * (i + 1) * torch.rand(128, 128, device=DEV))
                    g.beta.data.add_(0.02 * (i + 1))
            assert Fn._repack_gdns() >= 3
            again = [g.packer().get(g.beta, g.gamma, g.beta_min) for g in gdns]
        for (gp0, bp0), (gp1, bp1), g in zip(first, again, gdns):
            assert gp0.data_ptr() == gp1.data_ptr() and bp0.data_ptr() == bp1.data_ptr()        # the persistent buffers, refreshed in place
            with torch.no_grad():
                gp_ref, bp_ref = Fn.PackedGdn().get(g.beta, g.gamma, g.beta_min)
            assert torch.equal(gp1.view(torch.int16), gp_ref.view(torch.int16)) and torch.equal(bp1, bp_ref)
    finally:
        Fn.train_pack_cache(prev)
        Fn._gdn_registry[:] = [g for g in Fn._gdn_registry if g not in [x.packer() for x in gdns]]
        hesic_amd.set_compute_dtype(prev_dt)


def test_error_feedback_packs_are_for_plain_inference_calls_only():
    """grad mode is off inside an autograd.Function's forward and backward as well as under ``torch.no_grad()``.  A grad-enabled
    call (a training forward outside ``Trainer.step``) and its backward use the plainly rounded weights -- the forward output equals the
    SHAPED_WEIGHTS-off result bit for bit, through one packer that has already cached its error-feedback pack from an inference call -- and
    the inference call after it gets the error-feedback pack back."""
    Fn, O = _imp()
    dt = torch.bfloat16          # float16 is an inference-only format here
    Fn.set_compute_dtype(dt)
    try:
        x = rnd("efi_x", (2, 128, 24, 40)).to(DEV, dt)
        w = (rnd("efi_w", (128, 128, 5, 5)) * 0.05).to(DEV).requires_grad_(True)
        pk = Fn.PackedWeight(shaped=True)
        kw = dict(kernel_size=5, stride=2, padding=2, packer=pk)
        with torch.no_grad():
            y_inf = Fn.conv2d(x, w, None, **kw).clone()
            keep, Fn.SHAPED_WEIGHTS = Fn.SHAPED_WEIGHTS, False
            try:
                y_plain = Fn.conv2d(x, w, None, packer=Fn.PackedWeight(shaped=True), kernel_size=5, stride=2, padding=2).clone()
            finally:
                Fn.SHAPED_WEIGHTS = keep
        assert not torch.equal(y_inf, y_plain)                      # the two packs differ
        seen = []
        class Probe(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                seen.append(("forward", Fn.inference_call()))
                return t.clone()
            @staticmethod
            def backward(ctx, g):
                seen.append(("backward", Fn.inference_call()))
                return g
        y_tr = Fn.conv2d(x, w, None, **kw)                          # grad-enabled call through the SAME packer
        assert y_tr.requires_grad and torch.equal(y_tr.detach(), y_plain)
        Probe.apply(y_tr).float().sum().backward()
        assert w.grad is not None and bool(torch.isfinite(w.grad).all())
        assert ("backward", False) in seen
        with torch.no_grad():
            assert Fn.inference_call()
            assert torch.equal(Fn.conv2d(x, w, None, **kw), y_inf)
    finally:
        Fn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("fmt", ["bf16", "f16"])
def test_synthesis_weights_rounded_with_error_feedback_per_output_phase(fmt):
    """Round 5: at 16-bit inference the weights of g_s_conv2 / g_s_conv3 (``hesic_pack_conv_weight_shaped_tr``) and g_s_conv4's LDS panel
    (``hesic_sconv_pack_weight_image`` kind 2) are rounded with error feedback inside each output phase's tap class.  On a spatially SMOOTH
    input (what an IGDN output is) the result is closer to the transposed conv with the fp32 weights than with round-to-nearest weights, by
    at least 2x in rms; with the switch off the kernels reproduce the round-to-nearest oracle call as before."""
    Fn, O = _imp()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[fmt]
    Fn.set_compute_dtype(dt)
    try:
        g = torch.Generator().manual_seed(7)
        coarse = torch.randn(2, 128, 5, 7, generator=g)
        f = torch.nn.functional.interpolate(coarse, size=(40, 56), mode="bicubic", align_corners=True)     # smooth over a 5 x 5 window
        f16 = f.to(dt).float()
        for cout, name in ((3, "g_s_conv4"), (128, "g_s_conv3")):
            wt = rnd(f"ef_w{cout}", (128, cout, 5, 5)) * 0.05
            exact = O.deconv(f16, wt, None, 2)
            outs = {}
            for sh in (True, False):
                keep_sh, Fn.SHAPED_WEIGHTS = Fn.SHAPED_WEIGHTS, sh
                try:
                    with torch.no_grad():
                        if cout == 3:
                            y = Fn.conv2d(f16.to(DEV, dt), wt.to(DEV), None, kernel_size=5, stride=2, padding=2, transposed=True)
                        else:
                            y = Fn.conv2d(f16.to(DEV, dt), wt.to(DEV), None, kernel_size=5, stride=2, padding=2, transposed=True,
                                          packer=Fn.PackedWeight(shaped=True, tr_stride=2))
                finally:
                    Fn.SHAPED_WEIGHTS = keep_sh
                outs[sh] = (y.float().cpu() - exact).double()
            # interior pixels (the border sees fewer taps of a class); the 128-channel output is itself stored in 16 bits: compare in fp64 rms
            e_ef, e_rn = (float(outs[k][..., 4:-4, 4:-4].pow(2).mean().sqrt()) for k in (True, False))
            print(name, fmt, "rms error vs fp32 weights: error feedback %.3g, round to nearest %.3g" % (e_ef, e_rn))
            if cout == 3:
                assert e_ef < 0.5 * e_rn, (name, e_ef, e_rn)
            else:
                assert e_ef < e_rn, (name, e_ef, e_rn)          # the 16-bit output rounding is a floor under both
    finally:
        Fn.set_compute_dtype(torch.float32)
