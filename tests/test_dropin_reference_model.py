"""Drop-in check (build container only: needs /root/reference): the REFERENCE's own model files
ywz/mywork/newnet1.py and newnet1_joint.py import and construct against this repository's `compressai`
package + the kornia shim, with identical state-dict keys, and strict-load a state dict of our model.
Third-party modules the reference imports at module load but the image lacks (cv2, torchvision, range_coder)
are stubbed here exactly as in tests/golden/make_golden.py -- they are not on the path."""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference/ywz/mywork"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this machine")


@pytest.fixture(scope="module")
def ref_models():
    import hesic_amd
    from hesic_amd import geometry
    import compressai
    assert compressai.__file__.startswith(os.path.dirname(hesic_amd.__file__)), "a foreign compressai is active"
    saved = {k: sys.modules.get(k) for k in ("cv2", "torchvision", "torchvision.transforms", "range_coder", "kornia")}
    sys.modules["cv2"] = types.ModuleType("cv2")
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    rc = types.ModuleType("range_coder")
    rc.RangeEncoder = rc.RangeDecoder = rc.prob_to_cum_freq = object
    sys.modules["range_coder"] = rc
    kn = types.ModuleType("kornia")
    kn.warp_perspective, kn.get_perspective_transform = geometry.warp_perspective, geometry.get_perspective_transform
    sys.modules["kornia"] = kn
    sys.path.insert(0, REF)
    try:
        import newnet1
        import newnet1_joint
        yield newnet1, newnet1_joint
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("newnet1", None)
        sys.modules.pop("newnet1_joint", None)


@pytest.mark.parametrize("which", [0, 1], ids=["newnet1.HSIC", "newnet1_joint.HSIC"])
def test_reference_model_file_runs_on_our_package(ref_models, which):
    from hesic_amd import models
    from compressai.models.utils import HipConv2d, HipConvTranspose2d
    from compressai.layers import GDN
    ref = ref_models[which].HSIC()
    ours = (models.HSIC, models.HSICJoint)[which]()
    assert list(ref.state_dict().keys()) and set(ref.state_dict()) == set(ours.state_dict())
    ref.load_state_dict(ours.state_dict(), strict=True)
    # the reference file's layers are OUR HIP modules
    assert isinstance(ref.encoder1.g_a_conv2, HipConv2d) and isinstance(ref.decoder1.g_s_conv2, HipConvTranspose2d)
    assert isinstance(ref.encoder1.g_a_gdn1, GDN) and type(ref.gaussian1).__module__.startswith("compressai.entropy_models")
    assert sum(p.numel() for p in ref.parameters()) == sum(p.numel() for p in ours.parameters())
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # and they refuse to run without the GPU
        ref.eval()(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64), torch.eye(3)[None])
