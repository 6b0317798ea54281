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


@pytest.mark.parametrize("fmt", ["bf16", "f16"])
def test_library_exports_every_declared_symbol(fmt):
    """Both builds of the library (16-bit format bfloat16 / IEEE half: same sources, same entry points) load without a GPU, say which
    format they were built for and export every symbol of the header."""
    import torch
    L = _lib()
    h16 = torch.float16 if fmt == "f16" else torch.bfloat16
    path = L.LIB_PATH_F16 if fmt == "f16" else L.LIB_PATH
    l = L.lib(h16)
    assert l.hesic_abi_version() == L.ABI_VERSION == 2            # include/hesic_hip.h HESIC_ABI_VERSION
    assert l.hesic_h16_format() == (1 if fmt == "f16" else 0)
    declared = L.declared_symbols()
    assert len(declared) >= 30
    exported = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    missing = [s for s in declared if f" T {s}\n" not in exported]
    assert not missing, missing
    assert set(declared) == set(L._SIGS), (set(declared) ^ set(L._SIGS))


def test_compute_dtype_selects_the_library():
    """set_compute_dtype(float16 / bfloat16) binds the library built for that format; a tensor of the OTHER 16-bit format is refused
    loudly (no silent reinterpretation of its bits); fp32 goes to whichever library is active."""
    import torch
    import hesic_amd
    L = _lib()
    try:
        hesic_amd.set_compute_dtype(torch.float16)
        assert L.h16_dtype() == torch.float16 and L.lib().hesic_h16_format() == 1
        assert L.dt(torch.float16) == L.H16 and L.dt(torch.float32) == L.F32
        with pytest.raises(TypeError, match="active 16-bit format"):
            L.dt(torch.bfloat16)
        hesic_amd.set_compute_dtype(torch.float32)          # fp32 keeps the active library
        assert L.h16_dtype() == torch.float16
        hesic_amd.set_compute_dtype(torch.bfloat16)
        assert L.h16_dtype() == torch.bfloat16 and L.lib().hesic_h16_format() == 0
        with pytest.raises(TypeError, match="active 16-bit format"):
            L.dt(torch.float16)
        with pytest.raises(ValueError):
            hesic_amd.set_compute_dtype(torch.float64)
    finally:
        hesic_amd.set_compute_dtype(torch.bfloat16)
        hesic_amd.set_compute_dtype(torch.float32)


def test_analysis_precision_modes():
    """Host logic of the analysis-precision switch: "auto" resolves to the pair mode, the fast modes are explicit, round-3 names stay valid."""
    import torch
    import hesic_amd
    from hesic_amd import functional as Fn
    prev = Fn.set_analysis_precision("auto")
    try:
        hesic_amd.set_compute_dtype(torch.float16)
        assert Fn.analysis_precision() == "x3"          # round 5: the parity-strict pair mode for both 16-bit formats
        hesic_amd.set_compute_dtype(torch.bfloat16)
        assert Fn.analysis_precision() == "x3"
        Fn.set_analysis_precision("x3c2")
        assert Fn.analysis_precision() == "x3c2" and Fn.analysis_conv2_single()
        Fn.set_analysis_precision("bf16x3")
        assert Fn.analysis_precision() == "x3"
        Fn.set_analysis_precision("bf16")
        assert Fn.analysis_precision() == "x1"
        with pytest.raises(ValueError):
            Fn.set_analysis_precision("x4")
    finally:
        Fn.set_analysis_precision(prev)
        hesic_amd.set_compute_dtype(torch.float32)


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


def test_tape_from_calls_packs_pointers_and_integers():
    """A recorded launch becomes (entry point id, its arguments as 64-bit words without the trailing stream): byref(struct) -> the struct's
    address (kept alive), c_void_p -> its value, None -> 0, negative integers two's complement; unknown entry points are refused."""
    import ctypes as C
    from hesic_amd import _lib as L
    d = L.ConvDesc()
    arr, keep = L.tape_from_calls([("hesic_conv2d_forward", (C.byref(d), C.c_void_p(0x1000), C.c_void_p(None), None, C.c_void_p(0x2000), C.c_void_p(7))),
                                   ("hesic_joint_step", (C.c_void_p(16), 1, 192, 12, -1) + (0,) * 14 + (C.c_void_p(7),))])
    assert len(arr) == 2 and arr[0].fn == L.TAPE_IDS["hesic_conv2d_forward"] and arr[0].nargs == 5
    assert arr[0].a[0] == C.addressof(d) and keep[0] is d and arr[0].a[1] == 0x1000 and arr[0].a[2] == 0 and arr[0].a[3] == 0 and arr[0].a[4] == 0x2000
    assert arr[1].fn == L.TAPE_IDS["hesic_joint_step"] and arr[1].nargs == 19 and arr[1].a[4] == 0xFFFFFFFFFFFFFFFF
    with pytest.raises(KeyError):
        L.tape_from_calls([("hesic_gmm_cdf", (None, None))])
    with pytest.raises(TypeError):
        L.tape_from_calls([("hesic_conv2d_forward", (1.5, None))])
