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


def test_phase_fusion_selection_rule():
    """hesic_conv2d_variant is host logic: which transposed launches take the kernel that runs the four output phases of a tile in one
    block (igemm_tr4_kernel; deconv(), compressai/models/utils.py:112-118).  Auto mode: from 384 fused blocks on, when the rounds of the
    512 block slots are >= 70 % full; mode 0 never; mode 2 whenever the shape is eligible; convs and other channel counts never."""
    L = _lib()
    l = L.lib()

    def variant(B, H, W, Cin=128, Cout=128, transposed=1, k=5):
        Ho, Wo = (2 * H, 2 * W) if transposed else (H // 2, W // 2)
        d = L.ConvDesc(B, H, W, Cin, Ho, Wo, Cout, k, k, 2, k // 2, transposed, L.BF16, 0, 0, Cin, 0, Cout, 0, 0)
        v = (C.c_int32 * 4)()
        assert l.hesic_conv2d_variant(C.byref(d), v) == 0, l.hesic_last_error()
        return list(v)

    prev = l.hesic_conv2d_set_phase_fusion(1)
    try:
        assert variant(8, 128, 128) == [128, 128, 64, 2]          # g_s_conv3 at B=8 512^2: 1024 fused blocks
        assert variant(4, 128, 128)[3] == 2                       # 512
        assert variant(1, 224, 272)[3] == 2                       # config C5 (896 x 1088 padded): 476 blocks, one round 93 % full
        assert variant(8, 96, 96)[3] == 1                         # 576: the second round would be 12 % full
        assert variant(2, 128, 128)[3] == 1                       # 256
        assert variant(8, 128, 128, transposed=0)[3] == 1         # a conv
        assert variant(8, 128, 128, Cout=192)[3] == 1             # cout tiles of 64
        assert l.hesic_conv2d_set_phase_fusion(0) == 1
        assert variant(8, 128, 128)[3] == 1
        assert l.hesic_conv2d_set_phase_fusion(2) == 0
        assert variant(2, 16, 16) == [128, 128, 64, 2]
        assert l.hesic_conv2d_set_phase_fusion(7) == -1 and b"mode" in l.hesic_last_error()
    finally:
        l.hesic_conv2d_set_phase_fusion(prev if prev in (0, 1, 2) else 1)
