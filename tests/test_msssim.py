"""MS-SSIM, the second quality metric of the reference's evaluation (ywz/mywork/test3real.py:107-109: pytorch_msssim.ms_ssim(x_hat, x,
data_range=1, size_average=False)).  ``pytorch_msssim`` is third party, absent from the reference tree and from this image, and not
pinned by the reference -- parity UNPINNED for this metric, like kornia's warp: the published algorithm is restated twice, independently
(oracle/hesic_oracle.py::ms_ssim with torch convolutions; tests/golden/make_golden.py::fx_msssim in fp64 numpy with scipy's correlate1d
and a reshape-mean pool), and the device path (csrc/msssim.hip) is checked against both."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def _images():
    g = load_golden("msssim.npz")
    x = torch.from_numpy(g["x_u8"].astype(np.float32) / np.float32(255))
    y = torch.from_numpy(g["y_u8"].astype(np.float32) / np.float32(255))
    return g, x, y


def test_oracle_ms_ssim_matches_the_independent_restatement():
    from oracle import hesic_oracle as O
    g, x, y = _images()
    v = O.ms_ssim(x, y)
    assert v.shape == (2,)
    np.testing.assert_allclose(v.numpy(), g["ms_ssim"], rtol=0, atol=2e-7)
    assert float(O.ms_ssim(x, x).min()) == pytest.approx(1.0, abs=1e-12)                  # identical images
    assert 0.9 < g["ms_ssim"][1] < g["ms_ssim"][0] < 1.0                                # more noise, lower score
    with pytest.raises(ValueError):
        O.ms_ssim(x[..., :160, :], y[..., :160, :])                                      # five scales need > 160 pixels


@pytest.mark.gpu
def test_device_ms_ssim_matches_oracle_and_golden():
    from hesic_amd import models
    from oracle import hesic_oracle as O
    g, x, y = _images()
    v = models.ms_ssim(y.cuda(), x.cuda())
    assert v.dtype == torch.float64 and v.shape == (2,)
    np.testing.assert_allclose(v.cpu().numpy(), g["ms_ssim"], rtol=0, atol=5e-6)          # fp32 window sums on the device, fp64 in the golden
    # strided inputs (a crop of a larger NCHW tensor and a channels-last tensor), odd sizes at several scales
    big = torch.rand(2, 3, 300, 333, generator=torch.Generator().manual_seed(3))
    a = big[:, :, 7:7 + 251, 11:11 + 309]
    b = (a + 0.05 * torch.randn(a.shape, generator=torch.Generator().manual_seed(4))).clamp(0, 1)
    want = O.ms_ssim(b, a)
    got = models.ms_ssim(b.cuda().contiguous(memory_format=torch.channels_last), big.cuda()[:, :, 7:7 + 251, 11:11 + 309])
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=5e-6)
    assert float(models.ms_ssim(a.cuda(), a.cuda()).min()) == pytest.approx(1.0, abs=1e-6)
    db = models.ms_ssim_db(got)
    assert torch.allclose(db.cpu(), -10 * torch.log10(1 - want), atol=1e-3)
    with pytest.raises(ValueError):
        models.ms_ssim(a.cuda()[..., :150, :], a.cuda()[..., :150, :])


@pytest.mark.gpu
def test_device_ms_ssim_on_model_outputs():
    """The metric on what it is for: reconstructions of the default (float16) forward against the inputs, both views, vs the oracle
    metric on the same tensors."""
    import hesic_amd
    from hesic_amd import models, synthetic
    from oracle import hesic_oracle as O
    hesic_amd.set_compute_dtype(torch.float16)
    net = models.HSIC()
    synthetic.fill_state_dict_(net.state_dict())
    net = net.cuda().eval()
    x1, x2, Hm = (t.cuda() for t in synthetic.stereo_batch(0, 2, 256, 320))
    with torch.no_grad():
        out = net(x1, x2, Hm)
    for xh, x in ((out["x1_hat"], x1), (out["x2_hat"], x2)):
        got = models.ms_ssim(xh, x)
        want = O.ms_ssim(xh.float().cpu(), x.cpu())
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5)
