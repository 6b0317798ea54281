"""hesic_amd -- MI355X-native (gfx950) implementation of the HESIC / HESIC+ stereo-compression
forward/backward hot path (reference: ywz978020607/HESIC, ``ywz/mywork/newnet1{,_joint}.py``).

Layout
  csrc/            HIP kernels + C ABI (include/hesic_hip.h) -> libhesic_hip.so; host C++ coder -> libhesic_host.so
  functional.py    autograd.Functions over the C ABI
  compressai/      drop-in for the reference's ``compressai`` operator surface (import name: ``compressai``)
  geometry.py      warp_perspective / get_perspective_transform (the kornia calls of the path)
  models.py        HSIC / HSICJoint with the reference's module tree and state-dict keys
  train.py         R-D loss, two-optimiser train step, data-parallel wrapper (RCCL)
  synthetic.py     deterministic weights / stereo pairs

Importing this package makes ``import compressai`` resolve to ``hesic_amd/compressai`` (unless another
``compressai`` was imported first, which is reported loudly).
"""
import os
import sys
import warnings

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def _expose_compressai():
    mod = sys.modules.get("compressai")
    if mod is not None:
        if not os.path.abspath(getattr(mod, "__file__", "")).startswith(_PKG_DIR):
            warnings.warn("hesic_amd: a different `compressai` is already imported; the MI355X drop-in is NOT active")
        return
    if _PKG_DIR not in sys.path:
        sys.path.insert(0, _PKG_DIR)


_expose_compressai()

from .functional import compute_dtype, set_compute_dtype  # noqa: E402,F401

__version__ = "0.1.0"
