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
        pu32, pu8 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        l.hesic_rc_encoder_new.restype = _vp
        l.hesic_rc_encoder_free.argtypes = [_vp]
        l.hesic_rc_encoder_encode.argtypes = [_vp, _pi32, pu32, _i64, _i32]
        l.hesic_rc_encoder_finish.argtypes = [_vp, pu8, _i64]
        l.hesic_rc_encoder_finish.restype = _i64
        l.hesic_rc_decoder_new.argtypes = [C.c_char_p, _i64]
        l.hesic_rc_decoder_new.restype = _vp
        l.hesic_rc_decoder_free.argtypes = [_vp]
        l.hesic_rc_decoder_decode.argtypes = [_vp, pu32, _i64, _i32, _pi32]
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


class RangeEncoder:
    """Adaptive range coder with one cumulative-frequency table per symbol (the role ``range_coder.RangeEncoder`` plays
    in ``HSIC.compress``, ywz/mywork/newnet1.py:905-1040).  ``encode(symbols, cdf)``: symbols (n,) int32,
    cdf (n, A+1) uint32 numpy arrays."""

    def __init__(self):
        self._h = lib().hesic_rc_encoder_new()

    def encode(self, symbols, cdf):
        import numpy as np
        symbols = np.ascontiguousarray(symbols, dtype=np.int32).reshape(-1)
        cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
        if cdf.ndim != 2 or cdf.shape[0] != symbols.size:
            raise ValueError("RangeEncoder.encode: cdf must be (n_symbols, alphabet + 1)")
        rc = lib().hesic_rc_encoder_encode(self._h, symbols.ctypes.data_as(_pi32), cdf.ctypes.data_as(C.POINTER(C.c_uint32)),
                                           symbols.size, cdf.shape[1])
        if rc:
            raise ValueError("RangeEncoder.encode: symbol outside its table or with zero frequency" if rc == -2 else "bad argument")

    def finish(self) -> bytes:
        n = lib().hesic_rc_encoder_finish(self._h, None, 0)
        buf = (C.c_uint8 * max(int(n), 1))()
        lib().hesic_rc_encoder_finish(self._h, buf, n)
        return bytes(buf[:n])

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().hesic_rc_encoder_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown
            pass


class RangeDecoder:
    def __init__(self, data: bytes):
        self._h = lib().hesic_rc_decoder_new(data, len(data))
        if not self._h:
            raise ValueError("RangeDecoder: bad stream")

    def decode(self, cdf):
        import numpy as np
        cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
        out = np.empty(cdf.shape[0], dtype=np.int32)
        rc = lib().hesic_rc_decoder_decode(self._h, cdf.ctypes.data_as(C.POINTER(C.c_uint32)), cdf.shape[0], cdf.shape[1],
                                           out.ctypes.data_as(_pi32))
        if rc:
            raise ValueError("RangeDecoder.decode: bad table")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().hesic_rc_decoder_free(self._h)
                self._h = None
        except Exception:
            pass
