"""The C-ABI library loads on a GPU-less host and exports every symbol include/hesic_hip.h declares;
argument validation returns HESIC_EINVAL with a message (no compute calls here)."""
import ctypes as C
import os
import subprocess

import pytest


def _lib():
    from hesic_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return L


def test_library_exports_every_declared_symbol():
    L = _lib()
    l = L.lib()
    assert l.hesic_abi_version() == 1
    declared = L.declared_symbols()
    assert len(declared) >= 30
    exported = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH], text=True)
    missing = [s for s in declared if f" T {s}\n" not in exported]
    assert not missing, missing
    assert set(declared) == set(L._SIGS), (set(declared) ^ set(L._SIGS))


def test_struct_layouts_match_header():
    L = _lib()
    assert C.sizeof(L.ConvDesc) == 20 * 4
    assert C.sizeof(L.SConvDesc) == 16 * 4 + 8 * 8 and L.SConvDesc.xs_b.offset == 64
    assert C.sizeof(L.WarpDesc) == 10 * 4 + 8 * 8 and L.WarpDesc.ss_b.offset == 40
    assert C.sizeof(L.GmmDesc) == 11 * 4


def test_bad_arguments_are_rejected_with_a_message():
    L = _lib()
    l = L.lib()
    d = L.ConvDesc(1, 8, 8, 30, 8, 8, 64, 5, 5, 1, 2, 0, L.F32, 0, 0, 30, 0, 64, 0, 0)
    rc = l.hesic_conv2d_forward(C.byref(d), C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), None)
    assert rc == -1 and b"multiple of 32" in l.hesic_last_error()
    assert l.hesic_gdn_forward(None, None, None, None, 1, 3, 0, 1e-6, 0, None) == -1
    g = L.GmmDesc(1, 4, 8, 9, L.F32, 0, 72, 0, 0, 0.11, 1e-9)
    assert l.hesic_gmm_forward(C.byref(g), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), None,
                               C.c_void_p(16), C.c_void_p(16), None, None) == -1


def test_product_path_refuses_cpu_tensors():
    import torch
    from hesic_amd import functional as Fn
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Fn.gdn(torch.zeros(1, 3, 4, 4), torch.ones(3), torch.eye(3))
