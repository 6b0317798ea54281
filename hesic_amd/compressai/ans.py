"""``compressai.ans`` (reference: compressai/cpp_exts/rans/rans_interface.cpp:352-372): the range-ANS
coder classes, implemented in C++ (csrc/host/hesic_host.cpp) and bound with ctypes.  Streams are
byte-identical to the reference coder's."""
from hesic_amd import _host


class BufferedRansEncoder:
    def __init__(self):
        self._h = _host.lib().hesic_rans_encoder_new()

    def __del__(self):
        if getattr(self, "_h", None) and _host is not None and _host._lib is not None:
            _host._lib.hesic_rans_encoder_free(self._h)
            self._h = None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        flat, n, stride = _host.cdf_table(cdfs)
        rc = _host.lib().hesic_rans_encoder_push(self._h, _host.i32_array(symbols), _host.i32_array(indexes), len(symbols),
                                                 flat, n, stride, _host.i32_array(cdfs_sizes), _host.i32_array(offsets))
        if rc != 0:
            raise ValueError("encode_with_indexes: invalid indexes / cdfs")

    def flush(self):
        import ctypes as C
        l = _host.lib()
        n = l.hesic_rans_encoder_flush(self._h, None, 0)
        buf = C.create_string_buffer(n)
        l.hesic_rans_encoder_flush(self._h, buf, n)
        return buf.raw


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        enc = BufferedRansEncoder()
        enc.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)
        return enc.flush()


class RansDecoder:
    def __init__(self):
        self._h = _host.lib().hesic_rans_decoder_new()

    def __del__(self):
        if getattr(self, "_h", None) and _host is not None and _host._lib is not None:
            _host._lib.hesic_rans_decoder_free(self._h)
            self._h = None

    def set_stream(self, encoded):
        if _host.lib().hesic_rans_decoder_set_stream(self._h, bytes(encoded), len(encoded)) != 0:
            raise ValueError("set_stream: invalid stream")

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        import ctypes as C
        flat, n, stride = _host.cdf_table(cdfs)
        out = (C.c_int32 * len(indexes))()
        rc = _host.lib().hesic_rans_decoder_decode(self._h, _host.i32_array(indexes), len(indexes), flat, n, stride,
                                                   _host.i32_array(cdfs_sizes), _host.i32_array(offsets), out)
        if rc != 0:
            raise ValueError("decode_stream: invalid indexes / cdfs / stream")
        return list(out)

    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes, offsets):
        self.set_stream(encoded)
        return self.decode_stream(indexes, cdfs, cdfs_sizes, offsets)
