"""ctypes binding of ``libhesic_host.so`` (csrc/host/hesic_host.h): pmf->CDF and the rANS coder."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhesic_host.so")
_lib = None
_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
_pi32 = C.POINTER(C.c_int32)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing -- run `make -C hesic_amd/csrc` (or __graft_entry__.build())")
        l = C.CDLL(LIB_PATH)
        l.hesic_pmf_to_quantized_cdf.argtypes = [C.POINTER(C.c_float), _i32, _i32, C.POINTER(C.c_uint32)]
        l.hesic_rans_encoder_new.restype = _vp
        l.hesic_rans_encoder_free.argtypes = [_vp]
        l.hesic_rans_encoder_push.argtypes = [_vp, _pi32, _pi32, _i64, _pi32, _i32, _i32, _pi32, _pi32]
        l.hesic_rans_encoder_flush.argtypes = [_vp, C.c_char_p, _i64]
        l.hesic_rans_encoder_flush.restype = _i64
        l.hesic_rans_decoder_new.restype = _vp
        l.hesic_rans_decoder_free.argtypes = [_vp]
        l.hesic_rans_decoder_set_stream.argtypes = [_vp, C.c_char_p, _i64]
        l.hesic_rans_decoder_decode.argtypes = [_vp, _pi32, _i64, _pi32, _i32, _i32, _pi32, _pi32, _pi32]
        _lib = l
    return _lib


def i32_array(seq):
    return (C.c_int32 * len(seq))(*seq)


def cdf_table(cdfs):
    """list of (ragged) int lists -> (flat row-major int32 array, nrows, stride)"""
    stride = max(len(r) for r in cdfs)
    flat = (C.c_int32 * (len(cdfs) * stride))()
    for i, r in enumerate(cdfs):
        flat[i * stride:i * stride + len(r)] = list(r)
    return flat, len(cdfs), stride
