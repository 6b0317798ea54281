"""deconv() (compressai/models/utils.py:112-118) on the implicit-GEMM kernels: the form that runs the four output phases of a tile in
ONE block (igemm_tr4_kernel, csrc/conv_igemm.hip) against the one-block-per-phase form and the CPU oracle.

The two forms walk the taps of an output value in the same order, so the bar between them is bit-for-bit; the oracle bar is the
bf16-storage bar of test_gpu_ops.py (2e-2 of the output scale on bf16-rounded operands).  Shapes are chosen for what the fused
block does differently: tiles that hang over the q-grid (rows and columns), odd / single channel-chunk counts (the epilogue buffer
then alternates between the two ring buffers), several cout tiles, other kernel sizes, every epilogue (plain with each
activation, GDN, IGDN, the training form that also stores the conv output)."""
import ctypes as C

import pytest
import torch

import hesic_amd
from hesic_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _imp():
    from hesic_amd import functional as Fn
    from hesic_amd import _lib as L
    from oracle import hesic_oracle as O
    return Fn, L, O


def rnd(name, shape, lo=-1.0, hi=1.0):
    return synthetic._uniform("pf." + name, shape, lo, hi)


def bf(x):
    return x.to(torch.bfloat16).float()


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture
def fusion_modes():
    Fn, _, _ = _imp()
    prev = Fn.set_phase_fusion(1)

    def run(fn):
        outs = []
        for mode in (0, 2):
            Fn.set_phase_fusion(mode)
            outs.append(fn())
        return outs
    yield run
    Fn.set_phase_fusion(prev)


def _variant(L, B, H, W, Cin, Cout, k, pad):
    d = L.ConvDesc(B, H, W, Cin, 2 * H, 2 * W, Cout, k, k, 2, pad, 1, L.BF16, 0, 0, Cin, 0, Cout, 0, 0)
    v = (C.c_int32 * 4)()
    L.call("hesic_conv2d_variant", C.byref(d), v)
    return list(v)


PLAIN_CASES = [
    # tag, Cin, Cout, k, pad, (B, H, W), act
    ("5x5_128", 128, 128, 5, 2, (2, 16, 16), "none"),
    ("5x5_ragged_rows_cols", 128, 128, 5, 2, (1, 9, 13), "relu"),
    ("5x5_wide_row", 128, 128, 5, 2, (3, 5, 40), "leaky"),
    ("5x5_192_odd_stage_counts", 192, 128, 5, 2, (1, 12, 20), "none"),
    ("5x5_64_one_chunk", 64, 128, 5, 2, (2, 8, 24), "relu"),
    ("5x5_two_cout_tiles", 128, 256, 5, 2, (1, 11, 17), "none"),
    ("3x3", 128, 128, 3, 1, (2, 10, 18), "none"),
    ("3x3_leaky_tall", 128, 128, 3, 1, (1, 16, 9), "leaky"),
    ("5x5_1x1_map", 128, 128, 5, 2, (2, 1, 1), "none"),
]


@pytest.mark.parametrize("tag,Cin,Cout,k,pad,bhw,act", PLAIN_CASES, ids=[c[0] for c in PLAIN_CASES])
def test_fused_phases_equal_the_per_phase_blocks_bit_for_bit(tag, Cin, Cout, k, pad, bhw, act, fusion_modes):
    Fn, L, O = _imp()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    B, H, W = bhw
    w = bf(rnd(tag + "_w", (Cin, Cout, k, k)) * 0.05)
    b = rnd(tag + "_b", (Cout,), -0.2, 0.2)
    x = bf(rnd(tag + "_x", (B, Cin, H, W), -2, 2))
    xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    acts = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "leaky": L.ACT_LEAKY}
    with torch.no_grad():
        per_phase, fused = fusion_modes(lambda: Fn.conv2d(xd, w.to(DEV), b.to(DEV), kernel_size=k, stride=2, padding=pad, transposed=True, act=acts[act]))
    Fn.set_phase_fusion(2)
    assert _variant(L, B, H, W, Cin, Cout, k, pad)[3] == 2, "the fused kernel was not selected: the comparison would be vacuous"
    Fn.set_phase_fusion(0)
    assert _variant(L, B, H, W, Cin, Cout, k, pad)[3] == 1
    assert fused.shape == (B, Cout, 2 * H, 2 * W) and torch.equal(fused, per_phase)
    ref = torch.nn.functional.conv_transpose2d(x, w, b, stride=2, padding=pad, output_padding=2 * pad + 2 - k)
    ref = {"none": lambda t: t, "relu": torch.relu, "leaky": lambda t: torch.nn.functional.leaky_relu(t, 0.01)}[act](ref)
    assert rel_err(fused, ref) < 1.5e-2


@pytest.mark.parametrize("inv", [False, True], ids=["gdn", "igdn"])
@pytest.mark.parametrize("Cin,bhw", [(128, (2, 16, 16)), (192, (1, 9, 21)), (64, (3, 7, 8))], ids=["128", "192_ragged", "64_ragged"])
def test_fused_phases_with_the_gdn_epilogue(inv, Cin, bhw, fusion_modes):
    Fn, L, O = _imp()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    B, H, W = bhw
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=21)
    beta, gamma = sd["g.beta"], sd["g.gamma"]
    w = bf(rnd("g_w%d" % Cin, (Cin, 128, 5, 5)) * 0.05)
    b = rnd("g_b", (128,), -0.1, 0.1)
    x = bf(rnd("g_x%d" % Cin, (B, Cin, H, W), -2, 2))
    xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    args = dict(kernel_size=5, stride=2, padding=2, transposed=True)
    with torch.no_grad():
        per_phase, fused = fusion_modes(lambda: Fn.conv2d_gdn(xd, w.to(DEV), b.to(DEV), beta.to(DEV), gamma.to(DEV), inverse=inv, beta_min=1e-6,
                                                                 packer=Fn.PackedWeight(), gdn_packer=Fn.PackedGdn(), **args))
    assert torch.equal(fused, per_phase)
    ref = O.gdn(O.deconv(x, w, b, 2), beta, gamma, inv)
    assert rel_err(fused, ref) < 1.5e-2


def test_fused_phases_training_form_stores_the_same_conv_output_and_gradients(fusion_modes):
    """Autograd on: the kernel also stores v = conv + bias for GDN's backward (first, through the same epilogue tile); the data
    gradient of a stride-2 conv is the plain transposed launch.  Output, data and weight gradients are bit-identical between the forms (the bias / GDN
    parameter gradients are sums of identical values in atomics order)."""
    Fn, L, O = _imp()
    hesic_amd.set_compute_dtype(torch.bfloat16)
    sd = {"g.beta": torch.zeros(128), "g.gamma": torch.zeros(128, 128)}
    synthetic.fill_state_dict_(sd, salt=22)
    w = bf(rnd("t_w", (128, 128, 5, 5)) * 0.05)
    b = rnd("t_b", (128,), -0.1, 0.1)
    x = bf(rnd("t_x", (2, 128, 12, 20), -2, 2))
    gy = bf(rnd("t_gy", (2, 128, 24, 40)))
    wc = bf(rnd("t_wc", (128, 128, 5, 5)) * 0.03)
    gyc = bf(rnd("t_gyc", (2, 128, 6, 10)))

    def run():
        t = [x.clone().to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last), w.clone().to(DEV), b.clone().to(DEV),
             sd["g.beta"].clone().to(DEV), sd["g.gamma"].clone().to(DEV)]
        t = [p.requires_grad_() for p in t]
        y = Fn.conv2d_gdn(t[0], t[1], t[2], t[3], t[4], inverse=True, beta_min=1e-6, packer=Fn.PackedWeight(), gdn_packer=Fn.PackedGdn(),
                          kernel_size=5, stride=2, padding=2, transposed=True)
        y.backward(gy.to(DEV, torch.bfloat16))
        # stride-2 conv: its data gradient is a transposed launch without GDN
        xc = x.clone().to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
        yc = Fn.conv2d(xc, wc.to(DEV), None, kernel_size=5, stride=2, padding=2)
        yc.backward(gyc.to(DEV, torch.bfloat16))
        return [y.detach()] + [p.grad for p in t] + [xc.grad]
    per_phase, fused = fusion_modes(run)
    names = ["y", "dx", "dw", "dbias", "dbeta", "dgamma", "dx of the stride-2 conv"]
    for name, a, c in zip(names, fused, per_phase):
        assert a is not None, name
        if name in ("dbias", "dbeta", "dgamma"):      # column sums that end in fp32 atomics (any two runs differ in the last bits)
            assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max()), name
        else:
            assert torch.equal(a, c), name


def test_set_phase_fusion_rejects_bad_modes():
    Fn, L, _ = _imp()
    prev = Fn.set_phase_fusion(1)
    with pytest.raises(ValueError):
        Fn.set_phase_fusion(3)
    assert Fn.set_phase_fusion(prev) == 1


@pytest.mark.parametrize("fmt", ["bf16", "f16"])
@pytest.mark.parametrize("size", [128, 200])
def test_fused_phases_at_full_size_repeat_bit_for_bit(fmt, size):
    """Round 6: at the benchmark's size (8 x 128 -> 128 on a 128^2 map: 1024 fused blocks, two co-resident per CU, every CU's memory pipe
    backed up) ten launches of g_s_conv3 (+ IGDN) on the same input must be identical to each other and to the one-block-per-phase form.
    The small shapes above did not catch a store hazard that only shows under that load: a 16-byte buffer store with a REGISTER soffset followed
    by a VALU write into its data registers (the next store's address) wrote the new contents -- 4 - 17 % wrong pixels, different on every
    launch (profiles/scripts/tr4_determinism.py).  Also covers the input patch staged once per block (HALO) at a ragged size."""
    Fn, L, O = _imp()
    from compressai.layers import GDN
    from compressai.models.utils import deconv
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[fmt]
    prev_dt = Fn.compute_dtype() if hasattr(Fn, "compute_dtype") else None
    hesic_amd.set_compute_dtype(dt)
    prev = Fn.set_phase_fusion(1)
    try:
        x = (rnd(f"full_x{size}", (8, 128, size, size)) * 0.5).to(DEV, dt).contiguous(memory_format=torch.channels_last)
        layer, g = deconv(128, 128).to(DEV), GDN(128, inverse=True).to(DEV)
        with torch.no_grad():
            for f in ((lambda: layer.run_gdn(x, g)), (lambda: layer.run(x))):
                Fn.set_phase_fusion(1)
                outs = [f().clone() for _ in range(10)]
                Fn.set_phase_fusion(0)
                ref = f().clone()
                torch.cuda.synchronize()
                assert all(torch.equal(outs[0], o) for o in outs[1:])
                assert torch.equal(outs[0], ref)
    finally:
        Fn.set_phase_fusion(prev)
        hesic_amd.set_compute_dtype(prev_dt if prev_dt is not None else torch.float32)
