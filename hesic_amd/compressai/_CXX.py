"""``compressai._CXX`` (reference: compressai/cpp_exts/ops/ops.cpp:24-90) over libhesic_host.so."""
import ctypes as C

from hesic_amd import _host


def pmf_to_quantized_cdf(pmf, precision):
    """List[float] -> List[int]: quantised CDF with strictly positive bin widths summing to 2**precision."""
    n = len(pmf)
    src = (C.c_float * n)(*pmf)
    dst = (C.c_uint32 * (n + 1))()
    if _host.lib().hesic_pmf_to_quantized_cdf(src, n, int(precision), dst) != 0:
        raise ValueError("pmf_to_quantized_cdf: invalid pmf / precision")
    return list(dst)
