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
        l.hesic_pmf_rows_to_quantized_cdfs.argtypes = [C.POINTER(C.c_float), _i32, _i32, _pi32, C.POINTER(C.c_float), _i32, _pi32, _i32]
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
        l.hesic_rc_decoder_decode_grid.argtypes = [_vp, pu32, _i64, _i64, _i64, _i64, _i32, _pi32]
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


def _np_i32(a):
    import numpy as np
    return np.ascontiguousarray(a, dtype=np.int32)


def quantized_cdf_rows(pmf, lengths, tail_mass, precision, cdf_stride):
    """(rows, stride) float32 pmf matrix -> (rows, cdf_stride) int32 quantised CDFs in ONE native call (row r codes its first
    ``lengths[r]`` bins plus the tail-mass escape bin)."""
    import numpy as np
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    lengths, tail = _np_i32(lengths).reshape(-1), np.ascontiguousarray(tail_mass, dtype=np.float32).reshape(-1)
    if pmf.ndim != 2 or lengths.size != pmf.shape[0] or tail.size != pmf.shape[0]:
        raise ValueError("quantized_cdf_rows: one length and one tail mass per pmf row")
    out = np.zeros((pmf.shape[0], int(cdf_stride)), dtype=np.int32)
    rc = lib().hesic_pmf_rows_to_quantized_cdfs(pmf.ctypes.data_as(C.POINTER(C.c_float)), pmf.shape[0], pmf.shape[1],
                                                lengths.ctypes.data_as(_pi32), tail.ctypes.data_as(C.POINTER(C.c_float)), int(precision),
                                                out.ctypes.data_as(_pi32), out.shape[1])
    if rc:
        raise ValueError("quantized_cdf_rows: invalid pmf / lengths / precision")
    return out


def rans_encode_arrays(symbols, indexes, cdf_table, cdf_sizes, offsets) -> bytes:
    """One rANS stream from numpy buffers (no Python lists): symbols / indexes (n,), cdf_table (ncdf, stride) int32."""
    symbols, indexes, table = _np_i32(symbols).reshape(-1), _np_i32(indexes).reshape(-1), _np_i32(cdf_table)
    sizes, offsets = _np_i32(cdf_sizes).reshape(-1), _np_i32(offsets).reshape(-1)
    if symbols.size != indexes.size or table.ndim != 2 or sizes.size != table.shape[0] or offsets.size != table.shape[0]:
        raise ValueError("rans_encode_arrays: inconsistent shapes")
    l = lib()
    h = l.hesic_rans_encoder_new()
    try:
        p = lambda a: a.ctypes.data_as(_pi32)
        if l.hesic_rans_encoder_push(h, p(symbols), p(indexes), symbols.size, p(table), table.shape[0], table.shape[1], p(sizes), p(offsets)):
            raise ValueError("encode_with_indexes: invalid indexes / cdfs")
        n = l.hesic_rans_encoder_flush(h, None, 0)
        buf = C.create_string_buffer(int(n))
        l.hesic_rans_encoder_flush(h, buf, n)
        return buf.raw
    finally:
        l.hesic_rans_encoder_free(h)


def rans_decode_arrays(stream, indexes, cdf_table, cdf_sizes, offsets):
    """Inverse of ``rans_encode_arrays``: int32 numpy array of ``indexes.size`` symbols."""
    import numpy as np
    indexes, table = _np_i32(indexes).reshape(-1), _np_i32(cdf_table)
    sizes, offsets = _np_i32(cdf_sizes).reshape(-1), _np_i32(offsets).reshape(-1)
    out = np.empty(indexes.size, dtype=np.int32)
    l = lib()
    h = l.hesic_rans_decoder_new()
    try:
        p = lambda a: a.ctypes.data_as(_pi32)
        stream = bytes(stream)
        if l.hesic_rans_decoder_set_stream(h, stream, len(stream)):
            raise ValueError("set_stream: invalid stream")
        if l.hesic_rans_decoder_decode(h, p(indexes), indexes.size, p(table), table.shape[0], table.shape[1], p(sizes), p(offsets), p(out)):
            raise ValueError("decode_stream: invalid indexes / cdfs / stream")
        return out
    finally:
        l.hesic_rans_decoder_free(h)


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

    def decode_grid_raw(self, cdf_ptr, stride, n_outer, n_inner, row_step_outer, row_step_inner, out_ptr):
        """``decode_grid`` on raw host addresses (table rows of ``stride`` uint32 at ``cdf_ptr``, int32 symbols to ``out_ptr``): no numpy
        objects per call -- the HESIC+ wavefront decode calls this once per group."""
        rc = lib().hesic_rc_decoder_decode_grid(self._h, C.cast(cdf_ptr, C.POINTER(C.c_uint32)), n_outer, n_inner, row_step_outer, row_step_inner,
                                                stride, C.cast(out_ptr, _pi32))
        if rc:
            raise ValueError("RangeDecoder.decode_grid: bad table")

    def grid_callback(self):
        """(function address, decoder handle) of ``hesic_rc_decoder_decode_grid`` for a C caller that drives the decoder itself
        (``hesic_joint_decode_groups`` of the HIP library: the HESIC+ wavefront walk without a Python step per group)."""
        return C.cast(lib().hesic_rc_decoder_decode_grid, C.c_void_p), self._h

    def decode_grid(self, cdf, n_outer, n_inner, row_step_outer, row_step_inner):
        """Symbols (p, q), p outer, under table row ``p * row_step_outer + q * row_step_inner`` of ``cdf`` (rows, n): (n_outer, n_inner) int32."""
        import numpy as np
        cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
        out = np.empty(n_outer * n_inner, dtype=np.int32)
        rc = lib().hesic_rc_decoder_decode_grid(self._h, cdf.ctypes.data_as(C.POINTER(C.c_uint32)), n_outer, n_inner, row_step_outer,
                                                row_step_inner, cdf.shape[1], out.ctypes.data_as(_pi32))
        if rc:
            raise ValueError("RangeDecoder.decode_grid: bad table")
        return out.reshape(n_outer, n_inner)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().hesic_rc_decoder_free(self._h)
                self._h = None
        except Exception:
            pass
