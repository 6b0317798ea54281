"""ctypes binding of ``libhesic_hip.so`` / ``libhesic_hip_f16.so`` (C ABI: ``include/hesic_hip.h``).

The libraries are built in-tree by ``__graft_entry__.build()`` / ``make -C hesic_amd/csrc`` from the same sources: one per
16-bit storage / matrix-core operand format (bfloat16: training + inference; IEEE float16: inference).  ``use_h16()`` selects
which one ``call()`` goes to; fp32 tensors are served by either.
There is NO fallback: if the shared object is missing or a kernel call fails, the caller gets
an exception -- the product path never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhesic_hip.so")
LIB_PATH_F16 = os.path.join(_HERE, "libhesic_hip_f16.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "hesic_hip.h")

F32, H16 = 0, 1
ABI_VERSION = 2      # include/hesic_hip.h HESIC_ABI_VERSION
BF16 = H16       # historical name: the library's 16-bit format (hesic_h16_format())
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
EB_PARAM_STRIDE = 64

_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [(n, _i32) for n in (
        "B", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad", "transposed", "dtype", "act",
        "in_abs", "x_pix_stride", "x_c_off", "y_pix_stride", "y_c_off", "tap_mask_lo")]


class SConvDesc(C.Structure):
    _fields_ = [(n, _i32) for n in ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad",
                                    "transposed", "x_dtype", "y_dtype", "act")] + \
               [("_pad", _i32)] + \
               [(n, _i64) for n in ("xs_b", "xs_c", "xs_y", "xs_x", "ys_b", "ys_c", "ys_y", "ys_x")]


class WarpDesc(C.Structure):
    _fields_ = [(n, _i32) for n in ("B", "C", "H", "W", "Ho", "Wo", "align_corners", "src_dtype", "dst_dtype", "m_is_dst_to_src")] + \
               [(n, _i64) for n in ("ss_b", "ss_c", "ss_y", "ss_x", "ds_b", "ds_c", "ds_y", "ds_x")]


ADAM_MAX_TENSORS = 24


class AdamChunk(C.Structure):
    _fields_ = [("p", _vp * ADAM_MAX_TENSORS), ("g", _vp * ADAM_MAX_TENSORS), ("m", _vp * ADAM_MAX_TENSORS),
                ("v", _vp * ADAM_MAX_TENSORS), ("step", _vp * ADAM_MAX_TENSORS), ("numel", _i64 * ADAM_MAX_TENSORS),
                ("block0", _i32 * (ADAM_MAX_TENSORS + 1)), ("n", _i32), ("lr", _f32), ("beta1", _f32), ("beta2", _f32), ("eps", _f32)]


EB_MAX_TENSORS = 16


class EbLayout(C.Structure):
    _fields_ = [("ptr", _vp * EB_MAX_TENSORS), ("width", _i32 * EB_MAX_TENSORS), ("stride", _i32 * EB_MAX_TENSORS),
                ("first", _i32 * EB_MAX_TENSORS), ("col", _i32 * EB_MAX_TENSORS), ("n", _i32), ("lik_bound", _f32)]


class GmmDesc(C.Structure):
    _fields_ = [(n, _i32) for n in ("B", "HW", "M", "K", "dtype", "use_means_in_quant", "sm_pix_stride",
                                    "s_c_off", "m_c_off")] + [("scale_bound", _f32), ("lik_bound", _f32)]


class TapeCall(C.Structure):
    """hesic_tape_call: a recorded launch (entry point id, its arguments as 64-bit words, the stream argument left out)."""
    _fields_ = [("fn", _i32), ("nargs", _i32), ("a", C.c_uint64 * 20)]


TAPE_IDS = {"hesic_joint_step": 0, "hesic_conv2d_forward": 1, "hesic_conv2d_forward_f32out": 2}


def tape_from_calls(calls):
    """``calls``: [(entry point name, ctypes arguments as passed to ``call``)] -> (TapeCall array, objects to keep alive).  Pointers and
    integers only; the trailing stream argument is dropped (the replaying C function supplies it)."""
    arr = (TapeCall * len(calls))()
    keep = []
    for t, (name, args) in zip(arr, calls):
        t.fn, t.nargs = TAPE_IDS[name], len(args) - 1
        for i, v in enumerate(args[:-1]):
            if v is None:
                w = 0
            elif isinstance(v, int):
                w = v
            elif isinstance(v, C.c_void_p):
                w = v.value or 0
            elif hasattr(v, "_obj"):                    # byref(struct)
                keep.append(v._obj)
                w = C.addressof(v._obj)
            elif isinstance(v, (C.Array, C.Structure)):
                keep.append(v)
                w = C.addressof(v)
            else:
                raise TypeError(f"tape_from_calls: {name} argument {i} of type {type(v).__name__}")
            t.a[i] = w & 0xFFFFFFFFFFFFFFFF
    return arr, keep


_P = C.POINTER
_SIGS = {
    "hesic_abi_version": ([], _i32),
    "hesic_h16_format": ([], _i32),
    "hesic_last_error": ([], C.c_char_p),
    "hesic_pack_conv_weight": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_joint_step": ([_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp], _i32),
    "hesic_memcpy_async": ([_vp, _vp, C.c_size_t, _i32, _vp], _i32),
    "hesic_stream_synchronize": ([_vp], _i32),
    "hesic_probe_mfma_loop": ([_vp, _vp, _i32, _P(C.c_double), _vp], _i32),
    "hesic_joint_decode_groups": ([_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp], _i32),
    "hesic_joint_decode_groups_tape": ([_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp], _i32),
    "hesic_ssim_scale": ([_vp, _P(_i64), _vp, _P(_i64), _i32, _i32, _i32, _i32, _f32, _vp, _vp], _i32),
    "hesic_avgpool2_pad": ([_vp, _P(_i64), _vp, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_pack_conv_weight_shaped": ([_vp, _vp, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_pack_conv_weight_shaped_tr": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_pack_conv_weights_batched": ([_vp, _i32, _i32, _vp], _i32),
    "hesic_conv2d_forward": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp], _i32),
    "hesic_gdn_forward_planar": ([_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _f32, _i32, _vp], _i32),
    "hesic_eb_prepare_params": ([_vp, _vp, _i32, _vp], _i32),
    "hesic_gmm_cdf": ([_P(GmmDesc), _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp], _i32),
    "hesic_gmm_cdf_rows": ([_P(GmmDesc), _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp], _i32),
    "hesic_gmm_cdf_dyn": ([_P(GmmDesc), _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp], _i32),
    "hesic_maxpool2_forward": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_perspective_transform": ([_vp, _vp, _vp, _i32, _vp], _i32),
    "hesic_h_from_delta": ([_vp, _vp, _f32, _f32, _i32, _vp, _i32, _vp], _i32),
    "hesic_conv2d_ws_bytes": ([_P(ConvDesc)], C.c_size_t),
    "hesic_conv2d_f32out_ws_bytes": ([_P(ConvDesc)], C.c_size_t),
    "hesic_conv2d_forward_ws": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp], _i32),
    "hesic_gdn_pack_params": ([_vp, _vp, _f32, _vp, _vp, _i32, _vp], _i32),
    "hesic_conv2d_gdn_forward": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp], _i32),
    "hesic_conv2d_gdn_forward_train": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp], _i32),
    "hesic_conv2d_variant": ([_P(ConvDesc), _P(_i32)], _i32),
    "hesic_conv2d_set_phase_fusion": ([_i32], _i32),
    "hesic_conv2d_wgrad_ws_bytes": ([_P(ConvDesc)], _i64),
    "hesic_conv2d_wgrad": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i64, _vp], _i32),
    "hesic_unpack_conv_wgrad": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_sconv2d_forward": ([_P(SConvDesc), _vp, _vp, _vp, _vp, _vp], _i32),
    "hesic_sconv2d_forward_cat": ([_P(SConvDesc), _vp, _vp, C.POINTER(C.c_int64), _i32, _i32, _vp, _vp, _vp, _vp], _i32),
    "hesic_sconv2d_forward_cat_gdn": ([_P(SConvDesc), _vp, _vp, C.POINTER(C.c_int64), _i32, _i32, _vp, _vp, _vp, _vp, C.c_float, _i32, _i32,
                                       _vp, _vp], _i32),
    "hesic_sconv2d_gdn_forward": ([_P(SConvDesc), _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp], _i32),
    "hesic_sconv2d_gdn_forward_train": ([_P(SConvDesc), _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp], _i32),
    "hesic_sconv_pack_weight_image": ([_i32, _vp, _vp, _vp, _vp], _i32),
    "hesic_sconv2d_forward_prepacked": ([_P(SConvDesc), _vp, _vp, _vp, _vp, _vp, _vp], _i32),
    "hesic_sconv2d_gdn_forward_prepacked": ([_P(SConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp], _i32),
    "hesic_sconv2d_dgrad": ([_P(SConvDesc), _vp, _vp, _vp, _vp], _i32),
    "hesic_sconv2d_wgrad_ws_bytes": ([_P(SConvDesc)], _i64),
    "hesic_sconv2d_wgrad": ([_P(SConvDesc), _vp, _vp, _vp, _vp, _vp, _i64, _vp], _i32),
    "hesic_gdn_forward": ([_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _i32, _vp], _i32),
    "hesic_gdn_backward_ws_bytes": ([_i64, _i32], _i64),
    "hesic_gdn_backward": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _i32, _vp], _i32),
    "hesic_warp_perspective_forward": ([_P(WarpDesc), _vp, _vp, _vp, _vp], _i32),
    "hesic_warp_perspective_backward": ([_P(WarpDesc), _vp, _vp, _vp, _vp], _i32),
    "hesic_eb_forward": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    "hesic_eb_backward": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    "hesic_gmm_forward": ([_P(GmmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i32),
    "hesic_gmm_backward": ([_P(GmmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i32),
    "hesic_upsample4_forward": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_upsample4_backward": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_copy_channels": ([_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_spatial_max": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_mix_weights_forward": ([_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
    "hesic_conv3x3_c32_forward": ([_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
    "hesic_pack_images_c32": ([_vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
    "hesic_conv3x3_c32_forward_img6": ([_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp], _i32),
    "hesic_resblock_c32_forward": ([_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
    "hesic_adam_step": ([_vp, _vp], _i32),
    "hesic_pooled_linear_forward": ([_vp, _vp, _vp, _vp, _i32, _i32, _vp], _i32),
    "hesic_pooled_linear_backward": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp], _i32),
    "hesic_softmax_k_forward": ([_vp, _vp, _i32, _i32, _i32, _vp], _i32),
    "hesic_softmax_k_backward": ([_vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
    "hesic_sum_log2": ([_vp, _i64, _vp, _vp], _i32),
    "hesic_sum_sq_diff": ([_vp, _i32, _P(_i64), _vp, _i32, _P(_i64), _i32, _i32, _i32, _i32, _vp, _vp], _i32),
    "hesic_rd_sums": ([_i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i32),
    "hesic_log_backward": ([_vp, _f32, _vp, _i64, _vp], _i32),
    "hesic_sq_diff_backward": ([_vp, _i32, _P(_i64), _vp, _i32, _P(_i64), _i32, _i32, _i32, _i32, _f32, _vp, _vp], _i32),
    "hesic_act_backward": ([_vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
    "hesic_cast": ([_vp, _i32, _vp, _i32, _i64, _vp], _i32),
    "hesic_round": ([_vp, _i32, _vp, _i32, _i64, _vp], _i32),
    "hesic_conv2d_wgrad_direct": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp], _i32),
    "hesic_conv2d_forward_grouped": ([_P(ConvDesc), _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp], _i32),
    "hesic_pack_conv_weight_slice": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_conv2d_wgrad_partial": ([_P(ConvDesc), _vp, _vp, _vp, _i64, _vp], _i32),
    "hesic_conv2d_hilo_set_acc_scale": ([_f32], _i32),
    "hesic_sconv_pack_weight_image_hilo_scaled": ([_vp, _vp, _f32, _vp, _vp], _i32),
    "hesic_conv2d_wgrad_partial_batched": ([_i32, _P(ConvDesc), _P(_vp), _P(_vp), _P(_vp), _P(_i64), _P(_i32), _vp], _i32),
    "hesic_conv2d_wgrad_nsplit": ([_P(ConvDesc), _i32], _i32),
    "hesic_conv2d_wgrad_ws_bytes_n": ([_P(ConvDesc), _i32], _i64),
    "hesic_conv2d_wgrad_finish_batched_n": ([_i32, _P(ConvDesc), _P(_vp), _P(_vp), _P(_vp), _P(_vp), _i32, _P(_i32), _vp], _i32),
    "hesic_conv2d_wgrad_finish_batched": ([_i32, _P(ConvDesc), _P(_vp), _P(_vp), _P(_vp), _P(_vp), _i32, _vp], _i32),
    "hesic_gdn_backward_acc": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f32, _i32, _vp], _i32),
    "hesic_gdn_backward_partial_ok": ([_i64, _i32, _i32], _i32),
    "hesic_gdn_backward_partial": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _i32, _vp], _i32),
    "hesic_gdn_param_finish_batched": ([_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp], _i32),
    "hesic_gdn_backward_planar_acc": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _i32, _f32, _i32, _vp], _i32),
    "hesic_eb_pack_table": ([_P(EbLayout), _vp, _i32, _vp], _i32),
    "hesic_eb_scatter_grads": ([_P(EbLayout), _vp, _i32, _i32, _vp], _i32),
    "hesic_eb_aux_loss": ([_vp, _vp, _f32, _vp, _vp, _i32, _i32, _vp], _i32),
    "hesic_conv2d_forward_f32out": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, C.c_size_t, _vp], _i32),
    "hesic_eb_forward_f32in": ([_vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp], _i32),
    "hesic_gmm_forward_f32in": ([_P(GmmDesc), _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp], _i32),
    "hesic_conv2d_forward_hilo": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, C.c_size_t, _vp], _i32),
    "hesic_conv2d_forward_hilo_w1": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, C.c_size_t, _vp], _i32),
    "hesic_conv2d_hilo_ws_bytes": ([_P(ConvDesc)], C.c_size_t),
    "hesic_gdn_pack_params_lo": ([_vp, _vp, _i32, _vp], _i32),
    "hesic_gdn_pack_params_batched": ([_vp, _i32, _vp], _i32),
    "hesic_spatial_max_backward": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "hesic_rd_loss_combine": ([_vp, C.c_double, C.c_int64, C.c_int64, _vp, _vp], _i32),
    "hesic_sconv_pack_weight_image_hilo": ([_vp, _vp, _vp, _vp], _i32),
    "hesic_sconv_pack_weight_image_hilo_out1": ([_vp, _vp, _vp, _vp], _i32),
    "hesic_sconv2d_gdn_forward_hilo": ([_P(SConvDesc), _vp, _vp, _vp, _vp, _i32, _vp, _vp], _i32),
    "hesic_sconv2d_gdn_forward_hilo_out1": ([_P(SConvDesc), _vp, _vp, _vp, _vp, _i32, _vp, _vp], _i32),
    "hesic_conv2d_gdn_forward_hilo_out": ([_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp], _i32),
    "hesic_conv3x3_c32_wgrad_ws_bytes": ([], _i64),
    "hesic_conv3x3_c32_wgrad": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _vp], _i32),
    "hesic_im2col_hilo": ([_vp, _P(_i64), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp], _i32),
}

_libs = {}                      # torch 16-bit dtype -> CDLL
_h16 = torch.bfloat16           # the active 16-bit format
_lib = None                     # the active library (None until first use)


def declared_symbols():
    """Every ``hesic_*`` function declared in include/hesic_hip.h (used by the ABI test)."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(hesic_[a-z0-9_]+)\s*\(", text)))


def _load(h16):
    l = _libs.get(h16)
    if l is None:
        path = LIB_PATH_F16 if h16 == torch.float16 else LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"hesic_amd: {path} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C hesic_amd/csrc`. There is no CPU fallback.")
        l = C.CDLL(path)
        # the version check comes BEFORE the signatures are bound: a stale build then fails with the rebuild hint instead of an AttributeError
        # on the first entry point it lacks
        rebuild = "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C hesic_amd/csrc`"
        try:
            l.hesic_abi_version.argtypes, l.hesic_abi_version.restype = [], _i32
            ver = l.hesic_abi_version()
        except AttributeError:
            raise RuntimeError(f"hesic_amd: {os.path.basename(path)} exports no hesic_abi_version -- not this package's library; {rebuild}") from None
        if ver != ABI_VERSION:
            raise RuntimeError(f"hesic_amd: {os.path.basename(path)} ABI version mismatch (library {ver}, package {ABI_VERSION}): {rebuild}")
        for name, (args, res) in _SIGS.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                raise RuntimeError(f"hesic_amd: {os.path.basename(path)} lacks the entry point {name} although it reports ABI {ver}: {rebuild}") from None
            fn.argtypes, fn.restype = args, res
        if l.hesic_h16_format() != (1 if h16 == torch.float16 else 0):
            raise RuntimeError(f"hesic_amd: {os.path.basename(path)} was built for the other 16-bit format")
        _libs[h16] = l
    return l


def lib(h16=None):
    """The active library (or the one built for ``h16`` = torch.bfloat16 / torch.float16)."""
    global _lib
    if h16 is not None:
        return _load(h16)
    if _lib is None:
        _lib = _load(_h16)
    return _lib


def use_h16(dtype):
    """Select the 16-bit format (torch.bfloat16 or torch.float16) -- i.e. the library -- every following call goes to."""
    global _h16, _lib
    if dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("the 16-bit storage format is torch.bfloat16 or torch.float16")
    if dtype != _h16 or _lib is None:
        _lib = _load(dtype)
        _h16 = dtype


def h16_dtype():
    """torch dtype of HESIC_H16 storage in the active library."""
    return _h16


_tls = threading.local()


class call_hook:
    """``with call_hook(fn):`` -- ``fn(name, args)`` sees every ``call`` the CURRENT THREAD makes inside the block (launch tapes, the
    segment recorder).  Thread-local: a launch from another thread (a second model, a prefetch or metrics thread) during a recording
    neither lands on the tape nor marks a segment.  Hooks nest (the inner one runs first)."""

    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        hooks = getattr(_tls, "hooks", None)
        if hooks is None:
            hooks = _tls.hooks = []
        hooks.append(self.fn)
        return self

    def __exit__(self, *exc):
        _tls.hooks.remove(self.fn)
        return False


def call(name, *args):
    hooks = getattr(_tls, "hooks", None)
    if hooks:
        for h in reversed(hooks):
            h(name, args)
    rc = getattr(_lib or lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib().hesic_last_error().decode()}")


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if torch.is_tensor(t_or_dtype) else t_or_dtype
    if d == torch.float32:
        return F32
    if d == _h16:
        return H16
    if d in (torch.bfloat16, torch.float16):
        raise TypeError(f"hesic_amd: a {d} tensor while the active 16-bit format is {_h16} (set_compute_dtype selects it)")
    raise TypeError(f"hesic_amd: unsupported dtype {d} (float32, bfloat16 or float16)")


def ptr(t):
    return None if t is None else _vp(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current device's current HIP stream as a void*.  ``torch.cuda.current_stream()`` costs ~5 us per call (device-index
    normalisation, ``is_available()`` with its environment look-ups, a Stream object): at ~70 launches per forward that was a
    quarter of the host time of an eager forward; the raw accessors are two C calls."""
    if _raw_stream is not None:
        return _vp(_raw_stream(_cur_device()))
    return _vp(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    """Every kernel is launched on the CURRENT device's current stream: refuse CPU tensors (no fallback) and tensors of
    another GPU (the launch would run on the wrong device against foreign pointers)."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("hesic_amd: the HIP path needs tensors on a ROCm device (no CPU fallback); "
                               f"got a tensor on {t.device}")
        if cur is None:
            cur = _cur_device() if _cur_device is not None else torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(f"hesic_amd: tensor on {t.device} but the current device is cuda:{cur} -- kernels launch on the current "
                               "device's stream; wrap the call in `with torch.cuda.device(tensor.device):`")
