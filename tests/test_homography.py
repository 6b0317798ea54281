"""HomographyNet + h_matrix derivation (ywz/mywork/model.py:73-111, newtrain1_real.py:47-57,113-123).

CPU part: the oracle restatement against the golden outputs of the reference's own ``model.Net`` (tests/golden/
make_golden.py homo), state-dict compatibility of the drop-in module, properties of the DLT restatement (kornia is not
vendored: that piece is pinned by properties only).  GPU part: the HIP path against golden / oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, T, load_golden
from hesic_amd import synthetic
from oracle import hesic_oracle as O


def _params():
    shapes = [l.split() for l in open(os.path.join(GOLDEN, "homo_state_keys.txt"))]
    sd = {s[0]: torch.empty([int(v) for v in s[1:]]) for s in shapes}
    return synthetic.fill_homography_state_dict_(sd)


def test_oracle_matches_reference_net():
    g = load_golden("homo.npz")
    a, b, _ = synthetic.homography_batch(0, 2)
    delta, feat = O.homography_net(_params(), a, b, return_features=True)
    assert float((feat[:, ::8, ::2, ::2] - T(g["cnn_sub"])).abs().max()) < 1e-4
    assert abs(float(feat.abs().mean()) - float(g["cnn_absmean"])) < 1e-5
    assert float((delta - T(g["delta"])).abs().max()) < 2e-4          # deltas are a few pixels


def test_dropin_state_dict_keys_and_shapes():
    from hesic_amd import homography
    net = homography.Net()
    want = {l.split()[0]: tuple(int(v) for v in l.split()[1:]) for l in open(os.path.join(GOLDEN, "homo_state_keys.txt"))}
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want
    net.load_state_dict(_params(), strict=True)
    with pytest.raises(RuntimeError):           # no CPU fallback
        with torch.no_grad():
            net.eval()(torch.zeros(1, 1, 128, 128), torch.zeros(1, 1, 128, 128))


def test_dlt_properties():
    r = np.random.Generator(np.random.PCG64(3))
    src = torch.tensor([[0.0, 0.0], [128, 0], [128, 128], [0, 128]]).repeat(5, 1, 1)
    dst = src + T(r.uniform(-20, 20, (5, 4, 2)).astype(np.float32))
    H = O.get_perspective_transform(src, dst).double()
    p = torch.cat((src.double(), torch.ones(5, 4, 1, dtype=torch.float64)), -1) @ H.transpose(1, 2)
    assert float((p[..., :2] / p[..., 2:] - dst.double()).abs().max()) < 1e-4     # maps every corner onto its target
    assert torch.equal(H[:, 2, 2], torch.ones(5, dtype=torch.float64))
    I = O.get_perspective_transform(src, src)
    assert float((I - torch.eye(3)).abs().max()) < 1e-6
    # h_adjust with equal frames is the identity; the derivation composes the three steps
    Hm = O.h_matrix_from_delta(src + 40.0, dst - src, 256, 256, 256)
    assert float((Hm - torch.inverse(H.float())).abs().max()) < 1e-4
    Hs = O.h_matrix_from_delta(src + 40.0, dst - src, 512, 384, 256)
    a, b = 2.0, 1.5
    S = torch.tensor([[1.0, a / b, a], [b / a, 1.0, b], [1 / a, 1 / b, 1.0]])      # row0*a, col0/a, row1*b, col1/b
    assert float((Hs - Hm * S).abs().max()) < 1e-5


# ----------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 0.25)], ids=["f32", "bf16"])
def test_gpu_net_matches_golden(dtype, tol):
    from hesic_amd import homography
    g = load_golden("homo.npz")
    net = homography.Net(dtype=dtype)
    net.load_state_dict(_params())
    net = net.cuda().eval()
    a, b, _ = synthetic.homography_batch(0, 2)
    with torch.no_grad():
        delta = net(a.cuda(), b.cuda())
        delta8 = net(a.repeat(4, 1, 1, 1).cuda(), b.repeat(4, 1, 1, 1).cuda())     # batch 8: other tile shapes
    assert delta.shape == (2, 4, 2) and delta.dtype == torch.float32
    assert float((delta.cpu() - T(g["delta"])).abs().max()) < tol
    assert float((delta8.cpu() - T(g["delta"]).repeat(4, 1, 1)).abs().max()) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_gpu_maxpool(dtype):
    from hesic_amd import homography
    x = synthetic._uniform("mp.x", (3, 64, 10, 12), -2, 2).to(dtype)
    y = homography.max_pool2(x.cuda().contiguous(memory_format=torch.channels_last))
    assert torch.equal(y.float().cpu(), torch.nn.functional.max_pool2d(x.float(), 2, 2))


@pytest.mark.gpu
def test_gpu_h_matrix_derivation():
    from hesic_amd import homography
    r = np.random.Generator(np.random.PCG64(5))
    _, _, corners = synthetic.homography_batch(1, 6)
    delta = T(r.uniform(-24, 24, (6, 4, 2)).astype(np.float32))
    c0 = corners - corners[:, :1]
    Hg = homography.get_perspective_transform(c0.cuda(), (c0 + delta).cuda()).cpu()
    assert float((Hg - O.get_perspective_transform(c0, c0 + delta)).abs().max()) < 1e-5
    for (ih, iw, pic) in [(256, 256, 256), (512, 512, 256), (860, 1080, 256)]:
        want = O.h_matrix_from_delta(corners, delta, ih, iw, pic)
        got = homography.h_matrix_from_delta(corners.cuda(), delta.cuda(), ih, iw, pic).cpu()
        assert float(((got - want).abs() / (want.abs() + 1e-3)).max()) < 1e-4
    # Net.get_h keeps the corners as given (model.py:99-111)
    want = torch.inverse(O.get_perspective_transform(corners, corners + delta))
    got = homography.h_matrix_from_delta(corners.cuda(), delta.cuda(), 1, 1, 1, subtract_origin=False).cpu()
    assert float(((got - want).abs() / (want.abs() + 1e-3)).max()) < 2e-3
