"""MI355X-native drop-in for the ``compressai`` surface that HESIC / HESIC+ consume
(reference: compressai/__init__.py:15-60; consumers: ywz/mywork/newnet1.py:14-34).

Same names, constructor signatures, parameter / buffer names and exceptions as the reference
package; every ``forward`` on the hot path dispatches to a HIP kernel of ``libhesic_hip.so``
(see ``hesic_amd/functional.py``).  There is no CPU fallback.
"""
from compressai import datasets, entropy_models, layers, models, ops  # noqa: F401

_entropy_coder = "ans"
_available_entropy_coders = [_entropy_coder]

try:
    import range_coder  # noqa: F401
    _available_entropy_coders.append("rangecoder")
except ImportError:
    pass


def set_entropy_coder(entropy_coder):
    """Specifies the default entropy coder used to encode the bit-streams."""
    global _entropy_coder
    if entropy_coder not in _available_entropy_coders:
        raise ValueError(f'Invalid entropy coder "{entropy_coder}", choose from'
                         f'({", ".join(_available_entropy_coders)}).')
    _entropy_coder = entropy_coder


def get_entropy_coder():
    """Return the name of the default entropy coder used to encode the bit-streams."""
    return _entropy_coder


def available_entropy_coders():
    """Return the list of available entropy coders."""
    return _available_entropy_coders


__version__ = "1.0.0+hesic.amd"
