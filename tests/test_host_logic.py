"""CPU-only checks of the host side: the drop-in ``compressai`` surface (names, signatures, state-dict keys,
error conventions of the reference's own unit tests, tests/test_entropy_models.py / test_layers.py /
test_coder.py in the reference), the C++ rANS coder and CDF builder against reference-generated vectors."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, T, load_golden

import hesic_amd  # noqa: F401  (puts the drop-in `compressai` on the path)
import compressai
from compressai.entropy_models import EntropyBottleneck, EntropyModel, GaussianConditional, GaussianMixtureConditional
from compressai.layers import GDN, GDN1, MaskedConv2d
from compressai.models.utils import conv, deconv


def _keys(kind):
    with open(os.path.join(GOLDEN, f"{kind}_state_keys.txt")) as f:
        return {l.split()[0]: tuple(int(s) for s in l.split()[1:]) for l in f if l.strip()}


@pytest.mark.parametrize("kind", ["hsic", "joint"])
def test_state_dict_keys_and_shapes_match_reference(kind):
    from hesic_amd import models
    net = (models.HSIC if kind == "hsic" else models.HSICJoint)()
    ours = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = _keys(kind)
    assert set(ours) == set(ref), (sorted(set(ours) ^ set(ref))[:10])
    assert all(ours[k] == ref[k] for k in ref)
    n_main = sum(p.numel() for p in net.parameters())
    n_aux = sum(p.numel() for p in net.aux_parameters())
    assert n_aux == 15616 and n_main == (35065128 if kind == "hsic" else 19886056)


def test_enhancement_stage_state_dict_matches_reference():
    from hesic_amd import models
    ours = {k: tuple(v.shape) for k, v in models.Independent_EN().state_dict().items()}
    assert ours == _keys("en")
    both = models.GMM_together().state_dict()
    assert "m1.encoder1.g_a_conv1.weight" in both and "m2.EH2.EB3.RB3.conv2.bias" in both


def test_public_names_of_the_reference_import_block():
    from compressai.ans import BufferedRansEncoder, RansDecoder, RansEncoder  # noqa: F401
    from compressai._CXX import pmf_to_quantized_cdf  # noqa: F401
    from compressai.datasets import ImageFolder  # noqa: F401
    from compressai.layers import (AttentionBlock, ResidualBlock, ResidualBlockUpsample, ResidualBlockWithStride,  # noqa: F401
                                   conv3x3, subpel_conv3x3)
    from compressai.models import CompressionModel  # noqa: F401
    from compressai.models.utils import update_registered_buffers  # noqa: F401
    from compressai.ops import LowerBound, NonNegativeParametrizer, ste_round  # noqa: F401
    assert isinstance(conv(3, 8), torch.nn.Conv2d) and isinstance(deconv(8, 3), torch.nn.ConvTranspose2d)
    c = conv(4, 6, kernel_size=5, stride=2)
    assert c.weight.shape == (6, 4, 5, 5) and c.padding == (2, 2) and c.stride == (2, 2)
    d = deconv(6, 4)
    assert d.weight.shape == (6, 4, 5, 5) and d.output_padding == (1, 1)


def test_entropy_coder_registry():
    assert compressai.get_entropy_coder() == "ans" and "ans" in compressai.available_entropy_coders()
    with pytest.raises(ValueError):
        compressai.set_entropy_coder("cabac")
    with pytest.raises(ValueError):
        EntropyModel(entropy_coder="huffman")
    with pytest.raises(ValueError):
        EntropyModel(entropy_coder=0xFF)


def test_quantize_modes_and_errors():
    em = EntropyModel()
    x = torch.rand(1, 3, 4, 4) * 8 - 4
    assert ((em._quantize(x, "noise") - x).abs() <= 0.5).all()
    assert torch.equal(em._quantize(x, "symbols"), torch.round(x).int())
    mu = torch.rand(1, 3, 4, 4)
    assert torch.equal(em._quantize(x, "dequantize", mu), torch.round(x - mu) + mu)
    with pytest.raises(ValueError):
        em._quantize(x, mode="toto")
    with pytest.raises(NotImplementedError):
        em()


def test_gaussian_conditional_argument_validation():
    with pytest.raises(ValueError):
        GaussianConditional(scale_table=1)
    with pytest.raises(ValueError):
        GaussianConditional(scale_table=[])
    with pytest.raises(ValueError):
        GaussianConditional(scale_table=[1, 0.5])
    with pytest.raises(ValueError):
        GaussianConditional(scale_table=[0, 1])
    with pytest.raises(ValueError):
        GaussianConditional(scale_table=None, scale_bound=None)
    gc = GaussianConditional(None)
    assert float(gc.lower_bound_scale.bound) == pytest.approx(0.11)
    gm = GaussianMixtureConditional(K=5)
    assert gm.K == 5 and "scale_table" in dict(gm.named_buffers())


def test_masked_conv_masks_match_reference_patterns():
    a = MaskedConv2d(1, 1, 5, mask_type="A", padding=2).mask[0, 0]
    b = MaskedConv2d(1, 1, 5, mask_type="B", padding=2).mask[0, 0]
    assert a.sum() == 12 and b.sum() == 13 and a[2, 2] == 0 and b[2, 2] == 1 and a[3:].sum() == 0
    with pytest.raises(ValueError):
        MaskedConv2d(1, 3, 3, mask_type="C")
    g = load_golden("ops.npz")
    m = MaskedConv2d(8, 16, kernel_size=5, padding=2, stride=1, mask_type="A")
    assert torch.equal(m.mask, T(g["mcA_mask"]))


def test_gdn_init_and_parametrizer():
    g = GDN(8)
    assert torch.allclose(g.beta_reparam(g.beta), torch.ones(8), atol=1e-6)
    assert torch.allclose(g.gamma_reparam(g.gamma), 0.1 * torch.eye(8), atol=1e-6)
    x = torch.randn(2, 8, 4, 4)
    y = GDN1(8)(x)                      # GDN1 is tensor-op only (not on the path)
    assert y.shape == x.shape
    from compressai.ops import LowerBound
    lb = LowerBound(0.5)
    t = torch.tensor([0.2, 0.7], requires_grad=True)
    lb(t).backward(torch.tensor([1.0, 1.0]))
    assert t.grad.tolist() == [0.0, 1.0]
    t2 = torch.tensor([0.2], requires_grad=True)
    lb(t2).backward(torch.tensor([-1.0]))
    assert t2.grad.tolist() == [-1.0]


def test_pmf_to_quantized_cdf_and_rans_are_byte_exact_with_the_reference():
    from compressai._CXX import pmf_to_quantized_cdf
    from compressai import ans
    g = load_golden("codec.npz")
    for i in range(4):
        assert pmf_to_quantized_cdf(g[f"pmf{i}"].tolist(), 16) == g[f"cdf{i}"].tolist()
    table, sizes, offs = g["rans_cdfs"].tolist(), g["rans_sizes"].tolist(), g["rans_offsets"].tolist()
    idx, sym = g["rans_indexes"].tolist(), g["rans_symbols"].tolist()
    stream = ans.RansEncoder().encode_with_indexes(sym, idx, table, sizes, offs)
    assert stream == g["rans_bytes"].tobytes()
    assert ans.RansDecoder().decode_with_indexes(stream, idx, table, sizes, offs) == sym
    enc = ans.BufferedRansEncoder()
    enc.encode_with_indexes(sym[:200], idx[:200], table, sizes, offs)
    enc.encode_with_indexes(sym[200:], idx[200:], table, sizes, offs)
    assert enc.flush() == stream
    dec = ans.RansDecoder()
    dec.set_stream(stream)
    assert dec.decode_stream(idx[:123], table, sizes, offs) + dec.decode_stream(idx[123:], table, sizes, offs) == sym


@pytest.mark.parametrize("C", [8, 128])
def test_entropy_bottleneck_update_and_codec_match_reference(C):
    g = load_golden("ops.npz")
    t = f"eb_C{C}_"
    eb = EntropyBottleneck(C)
    sd = eb.state_dict()
    for k in list(sd):
        if t + "p_" + k in g and sd[k].numel():
            sd[k].copy_(T(g[t + "p_" + k]))
    with pytest.raises(ValueError):
        eb.compress(torch.zeros(1, C, 2, 2))          # "Uninitialized CDFs. Run update() first"
    eb.update(force=True)
    assert torch.equal(eb._offset, T(g[t + "offset"])) and torch.equal(eb._cdf_length, T(g[t + "cdf_length"]))
    assert torch.equal(eb._quantized_cdf, T(g[t + "quantized_cdf"]))
    x = T(g[t + "x"])[:1]
    strings = eb.compress(x)
    assert strings[0] == g[t + "string0"].tobytes()
    assert torch.equal(eb.decompress(strings, x.shape[-2:]), T(g[t + "decompressed0"]))
    assert float(eb.loss()) == pytest.approx(float(g[t + "aux_loss"]), rel=1e-5)


def test_update_registered_buffers_resizes_for_strict_loading():
    from compressai.models.utils import update_registered_buffers
    src = EntropyBottleneck(4)
    src.update()
    dst = EntropyBottleneck(4)
    sd = {"eb." + k: v for k, v in src.state_dict().items()}
    with pytest.raises(ValueError):
        update_registered_buffers(dst, "eb", ["_nope"], sd)
    update_registered_buffers(dst, "eb", ["_quantized_cdf", "_offset", "_cdf_length"], sd)
    dst.load_state_dict(src.state_dict(), strict=True)
    assert torch.equal(dst._quantized_cdf, src._quantized_cdf)


def test_geometry_point_transform():
    from hesic_amd.geometry import get_perspective_transform
    src = torch.tensor([[[0., 0.], [255., 0.], [255., 255.], [0., 255.]]])
    dst = src + torch.tensor([[[3., -2.], [1., 4.], [-5., 2.], [2., 2.]]])
    M = get_perspective_transform(src, dst)
    p = torch.cat([src[0], torch.ones(4, 1)], 1) @ M[0].T
    assert torch.allclose(p[:, :2] / p[:, 2:], dst[0], atol=1e-3)


def test_loading_a_checkpoint_mentions_the_warp_convention_once():
    """A checkpoint does not say which kornia release it was trained with; the two warp conventions differ.  While none has been chosen
    explicitly, ``load_state_dict`` says so (once); choosing one silences it."""
    import warnings
    from hesic_amd import geometry, models
    keep = (geometry.DEFAULT_ALIGN_CORNERS, geometry._CONVENTION_CHOSEN, models.StereoCompressionModel._warned_warp)
    try:
        geometry._CONVENTION_CHOSEN, models.StereoCompressionModel._warned_warp = False, False
        net = models.HSIC()
        sd = net.state_dict()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            net.load_state_dict(sd)
            net.load_state_dict(sd)
        assert sum("warp convention" in str(x.message) for x in w) == 1
        prev = geometry.use_reference_era_warp()
        assert prev is True and geometry.DEFAULT_ALIGN_CORNERS is False and "0.4" in geometry.warp_convention()
        models.StereoCompressionModel._warned_warp = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            net.load_state_dict(sd)
        assert not any("warp convention" in str(x.message) for x in w)
    finally:
        geometry.DEFAULT_ALIGN_CORNERS, geometry._CONVENTION_CHOSEN, models.StereoCompressionModel._warned_warp = keep


def test_smooth_synthetic_pairs_are_deterministic_smooth_and_warp_consistent():
    """``synthetic.smooth_stereo_pair`` (round 5: the content the >= 30 dB trained-point parity test trains on): seeded (same pair twice, other
    seed differs), inside [0, 1], far smoother than ``stereo_pair``'s low-passed noise (mean |horizontal difference| 10x smaller), and view 2
    is view 1 warped by the pair's homography (the oracle's warp of x1 reproduces x2 up to its N(0, 0.004) noise where the warp is inside)."""
    import numpy as np
    import torch
    from hesic_amd import synthetic
    from oracle import hesic_oracle as O
    a = synthetic.smooth_stereo_pair(3, 96, 128)
    b = synthetic.smooth_stereo_pair(3, 96, 128)
    c = synthetic.smooth_stereo_pair(4, 96, 128)
    assert all(np.array_equal(u, v) for u, v in zip(a, b)) and not np.array_equal(a[0], c[0])
    x1, x2, H = a
    assert x1.dtype == np.float32 and x1.shape == (3, 96, 128) and 0.0 <= x1.min() and x1.max() <= 1.0 and 0.0 <= x2.min() and x2.max() <= 1.0
    n1 = synthetic.stereo_pair(3, 96, 128)[0]
    assert np.abs(np.diff(x1, axis=2)).mean() < 0.1 * np.abs(np.diff(n1, axis=2)).mean()
    w = O.warp_perspective(torch.from_numpy(x1)[None], torch.from_numpy(H)[None], (96, 128), True)[0].numpy()
    inside = w.sum(0) > 0
    assert inside.mean() > 0.7
    assert np.abs(w - x2)[:, inside].mean() < 0.01
    xb = synthetic.smooth_stereo_batch(3, 2, 96, 128)
    assert tuple(xb[0].shape) == (2, 3, 96, 128) and np.array_equal(xb[0][0].numpy(), x1) and tuple(xb[2].shape) == (2, 3, 3)


def test_call_hook_sees_only_the_recording_threads_launches(monkeypatch):
    """Launch tapes and the segment recorder listen through ``_lib.call_hook``: thread-local, so a launch another thread makes while
    a recording is open neither lands on the tape nor marks a segment; hooks nest and are removed on exit."""
    import threading
    from hesic_amd import _lib as L

    class Stub:
        def __getattr__(self, name):
            return lambda *a: 0
    monkeypatch.setattr(L, "_lib", Stub())
    seen, inner = [], []
    started, done = threading.Event(), threading.Event()

    def other():
        started.wait(5)
        L.call("from_other_thread", 1)
        done.set()
    th = threading.Thread(target=other)
    th.start()
    with L.call_hook(lambda name, args: seen.append((name, args))):
        L.call("a", 1, 2)
        started.set()
        assert done.wait(5)
        with L.call_hook(lambda name, args: inner.append(name)):
            L.call("b")
        L.call("c")
    th.join()
    L.call("after")
    assert [n for n, _ in seen] == ["a", "b", "c"] and seen[0][1] == (1, 2)
    assert inner == ["b"]
    assert not getattr(L._tls, "hooks")


def test_wgrad_flush_empties_its_queues_before_the_calls(monkeypatch):
    """``flush_wgrad_finish`` / ``flush_gdn_finish``: the queued jobs are taken off the queue BEFORE the C calls, so a call that fails (a bad
    launch, an out-of-memory workspace) cannot leave its jobs behind for the next step; deferred split-K launches (jobs that carry their conv
    input) go down as ONE ``hesic_conv2d_wgrad_partial_batched`` call ahead of the finishing call, with the same K-slice counts."""
    import types
    from hesic_amd import _lib as L
    from hesic_amd import functional as Fn
    calls = []

    def fake_call(name, *args):
        calls.append((name, args[0]))
        if name == "hesic_conv2d_wgrad_finish_batched_n" and fail["on"]:
            raise RuntimeError("boom")
    monkeypatch.setattr(L, "call", fake_call)
    monkeypatch.setattr(L, "stream", lambda: None)
    fail = {"on": False}
    done = []
    monkeypatch.setattr(Fn, "_slot_done", lambda s: done.append(s))
    slot = lambda: types.SimpleNamespace(grad=torch.zeros(4))
    t = lambda: torch.zeros(16)
    d = L.ConvDesc()
    prev = Fn.defer_wgrad_finish(True)
    try:
        q = Fn._finish_queue
        assert q == []
        w1, w2, b2 = slot(), slot(), slot()
        q.append(Fn._WgJob(d, t(), t(), w1, None, 1, t(), 64, 3))          # split-K launch deferred (carries x)
        q.append(Fn._WgJob(d, t(), t(), w2, b2, 2, None, 64, 0))            # split-K launch already issued
        Fn.flush_wgrad_finish()
        assert [c for c in calls] == [("hesic_conv2d_wgrad_partial_batched", 1), ("hesic_conv2d_wgrad_finish_batched_n", 2)]
        assert done == [w1, w2, b2] and Fn._finish_queue == []
        calls.clear(); done.clear()
        fail["on"] = True
        q.append(Fn._WgJob(d, t(), t(), w1, None, 1, t(), 64, 3))
        with pytest.raises(RuntimeError, match="boom"):
            Fn.flush_wgrad_finish()
        assert Fn._finish_queue == [] and done == []           # nothing left behind, no slot reported as written
        fail["on"] = False
        Fn.flush_wgrad_finish()
        assert len(calls) == 2                                    # the failed flush's two calls; the empty one made none
    finally:
        Fn._finish_queue.clear()
        Fn._gdn_finish_queue.clear()
        monkeypatch.setattr(Fn, "_finish_queue", None if not prev else [])
