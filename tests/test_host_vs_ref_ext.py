"""The C++ host coder (hesic_amd/csrc/host) against the REFERENCE's own extensions built into oracle/_ref
(oracle/Makefile) on randomised inputs: CDF tables equal, rANS streams byte-identical, cross-decoding works."""
import glob
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT

import hesic_amd  # noqa: F401
from compressai import ans as my_ans
from compressai._CXX import pmf_to_quantized_cdf as my_cdf


def _load(name):
    hits = glob.glob(os.path.join(ROOT, "oracle", "_ref", name + "*.so"))
    if not hits:
        pytest.skip("oracle/_ref not built (needs /root/reference); golden vectors in test_host_logic.py still pin the coder")
    spec = importlib.util.spec_from_file_location(name, hits[0])
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmf_to_quantized_cdf_random():
    ref = _load("_CXX")
    r = np.random.Generator(np.random.PCG64(5))
    for trial in range(200):
        n = int(r.integers(2, 300))
        p = r.dirichlet(np.full(n, r.choice([0.05, 0.5, 5.0])))
        if trial % 3 == 0:
            p[r.integers(0, n, n // 2)] = 0.0            # forces the zero-width repair path
            if p.sum() == 0:
                continue
            p /= p.sum()
        p = p.astype(np.float32).tolist()
        assert my_cdf(p, 16) == list(ref.pmf_to_quantized_cdf(p, 16)), trial


def test_rans_streams_random():
    ref_cxx, ref_ans = _load("_CXX"), _load("ans")
    r = np.random.Generator(np.random.PCG64(6))
    for trial in range(40):
        ntab = int(r.integers(1, 6))
        cdfs = [list(ref_cxx.pmf_to_quantized_cdf(r.dirichlet(np.ones(int(r.integers(2, 40)))).astype(np.float32).tolist(), 16))
                for _ in range(ntab)]
        sizes = [len(c) for c in cdfs]
        L = max(sizes)
        table = [c + [0] * (L - len(c)) for c in cdfs]
        offsets = [int(r.integers(-20, 5)) for _ in range(ntab)]
        n = int(r.integers(1, 3000))
        idx = r.integers(0, ntab, n).tolist()
        spread = int(r.choice([0, 3, 300]))                # 300: deep into the bypass escape
        sym = [int(r.integers(offsets[i] - spread, offsets[i] + sizes[i] + spread + 1)) for i in idx]
        a = my_ans.RansEncoder().encode_with_indexes(sym, idx, table, sizes, offsets)
        b = ref_ans.RansEncoder().encode_with_indexes(sym, idx, table, sizes, offsets)
        assert a == b, trial
        assert my_ans.RansDecoder().decode_with_indexes(b, idx, table, sizes, offsets) == sym
        assert ref_ans.RansDecoder().decode_with_indexes(a, idx, table, sizes, offsets) == sym
