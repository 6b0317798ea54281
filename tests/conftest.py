import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


@pytest.fixture(scope="session")
def ops_golden():
    return load_golden("ops.npz")


@pytest.fixture(scope="session")
def warp_golden():
    return load_golden("warp.npz")


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(autouse=True)
def _default_modes_between_tests():
    """Every test starts from the package defaults: fp32 compute dtype, the bfloat16 library as the active 16-bit format (tests that
    hand bf16 tensors straight to an operator rely on it), analysis precision "auto" -- whatever the previous test selected."""
    yield
    mod = sys.modules.get("hesic_amd.functional")
    if mod is not None and os.path.exists(os.path.join(ROOT, "hesic_amd", "libhesic_hip.so")):
        if mod.L.h16_dtype() != torch.bfloat16:
            mod.set_compute_dtype(torch.bfloat16)
        mod.set_compute_dtype(torch.float32)
        mod.set_analysis_precision("auto")
