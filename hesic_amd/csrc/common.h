// Shared device/host helpers for libhesic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hesic_hip.h"

typedef uint16_t bf16_t;   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- error plumbing (host)
void hesic_set_error(const char* fmt, ...);
#define HESIC_CHECK_ARG(cond, ...)                  \
    do {                                            \
        if (!(cond)) {                              \
            hesic_set_error(__VA_ARGS__);           \
            return HESIC_EINVAL;                    \
        }                                           \
    } while (0)
#define HESIC_LAUNCH_RETURN(name)                                              \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            hesic_set_error("%s: %s", name, hipGetErrorString(e__));           \
            return (int)e__;                                                   \
        }                                                                      \
        return 0;                                                              \
    } while (0)

// ---- bf16 <-> f32 (round to nearest even, NaN kept quiet)
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __bf16 hw_bf2_t __attribute__((ext_vector_type(2)));
typedef float hw_f2_t __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32: hardware round-to-nearest-even, two values per instruction
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(hw_f2_t{lo, hi}, hw_bf2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

template <typename T> struct elem;
template <> struct elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// dtype-erased scalar access for the strided (image-side) kernels
__device__ __forceinline__ float ld_any(const void* p, int64_t i, int dtype) {
    return dtype == HESIC_BF16 ? bf2f(((const bf16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, int64_t i, int dtype, float v) {
    if (dtype == HESIC_BF16) ((bf16_t*)p)[i] = f2bf(v);
    else ((float*)p)[i] = v;
}

// Branch-free on purpose: `act` is a launch parameter, and written as `if (act == ...) return ...` every ELEMENT of an epilogue became
// its own chain of scalar compare-and-branch blocks (three branches per value, no scheduling across values).  Same results bit for
// bit: NONE is v * 1, RELU yields +0 (not 0 * v = -0) for v <= 0 and for NaN, LEAKY 0.01f * v.
__device__ __forceinline__ float apply_act(float v, int act) {
    const float slope = act == HESIC_ACT_LEAKY ? 0.01f : 1.0f;
    const float neg = act == HESIC_ACT_RELU ? 0.0f : v * slope;
    return v > 0.f ? v : neg;
}

// division of a 31-bit unsigned by a launch-time constant without the ~20-instruction rcp sequence: q = (mulhi(n, magic) + n) >> shift
struct FastDiv { uint32_t magic, shift; };
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv d) { return (uint32_t)(((uint64_t)__umulhi(n, d.magic) + n) >> d.shift); }
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t sh = 0;
    while ((1ull << sh) < d) ++sh;
    f.shift = sh;
    f.magic = (uint32_t)((((1ull << sh) - d) << 32) / d + 1);
    return f;
}

// Workgroups are dealt round-robin to the 8 XCDs (hardware block id & 7 = XCD), each with its own L2.  This maps the
// hardware id to a logical id such that every XCD owns one CONTIGUOUS range of logical ids: neighbouring tiles (shared
// halos, shared weights) then meet in the same L2 instead of being fetched once per XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int grid_for(int64_t n, int block, int max_blocks = 256 * 8) {
    int64_t g = cdiv64(n, block);
    if (g > max_blocks) g = max_blocks;
    if (g < 1) g = 1;
    return (int)g;
}
