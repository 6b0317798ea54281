// Shared device/host helpers for libhesic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hesic_hip.h"

// ---- the 16-bit storage / matrix-core operand format of THIS build of the library.
// The same sources build twice: libhesic_hip.so with bfloat16 (HESIC_H16_IS_F16 == 0: training and inference, 8-bit significand, fp32 range)
// and libhesic_hip_f16.so with IEEE binary16 (== 1: inference, 11-bit significand -- v_mfma_f32_32x32x16_f16 issues at the bf16 rate on
// gfx950 and honours subnormal inputs, profiles/scripts/micro/f16_probe.hip).  Every kernel goes through the helpers below (h16_t raw bits,
// h2f / h2f_lo / h2f_hi, pack_h2, mfma_32x32x16_h16), so nothing else knows which format it is; hesic_h16_format() reports it.
#ifndef HESIC_H16_IS_F16
#define HESIC_H16_IS_F16 0
#endif
#ifndef HESIC_NO_DYN_SQ
#define HESIC_NO_DYN_SQ 0     /* A/B builds only (-DHESIC_NO_DYN_SQ=1): the fixed 2^-6 scale of the pair (I)GDN squares of rounds 3-4 */
#endif
typedef uint16_t h16_t;   // raw bits of a 16-bit float (format: see above)
#if HESIC_H16_IS_F16
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
#define mfma_32x32x16_h16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define mfma_16x16x32_h16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define H16_ONE_PAIR 0x3c003c00u          /* two 1.0 */
// squares that go through 16-bit storage inside the fused (I)GDN contractions are scaled by H16_SQ_SCALE (gamma' by its inverse, in
// hesic_gdn_pack_params*): v^2 * 2^-6 stays finite up to |v| = 2047 where fp16 itself ends at 255
#define H16_SQ_SCALE 0.015625f
#define H16_SQ_UNSCALE 64.0f
#define H16_SQ_ROOT 0.125f                /* sqrt(H16_SQ_SCALE) */
#else
typedef __attribute__((ext_vector_type(8))) __bf16 h16x8;
#define mfma_32x32x16_h16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define mfma_16x16x32_h16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define H16_ONE_PAIR 0x3f803f80u
#define H16_SQ_SCALE 1.0f
#define H16_SQ_UNSCALE 1.0f
#define H16_SQ_ROOT 1.0f
#endif
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

// ---- 16-bit <-> f32 (round to nearest even)
typedef float hw_f2_t __attribute__((ext_vector_type(2)));
#if HESIC_H16_IS_F16
typedef _Float16 hw_h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h2f(h16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// the two halves of a packed pair (element 0 in the low 16 bits)
__device__ __forceinline__ float h2f_lo(uint32_t p) { return (float)__builtin_bit_cast(hw_h2_t, p)[0]; }
__device__ __forceinline__ float h2f_hi(uint32_t p) { return (float)__builtin_bit_cast(hw_h2_t, p)[1]; }
// two values per instruction, round to nearest even; finite inputs beyond the fp16 range saturate at +-65504 instead of turning into
// infinities (a too-large activation then costs accuracy at that pixel, not a NaN map)
__device__ __forceinline__ float h16_clamp(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
// v - (float)half of a packed pair in ONE instruction (v_fma_mix_f32: v * 1.0 + (-h), the 16-bit source converted on the way in; exact like
// the two-instruction form).  The compiler finds this fusion only in some contexts, the pair-splitting code relies on it everywhere.
__device__ __forceinline__ float sub_h2_lo(float v, uint32_t p) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(v), "v"(p));
    return d;
}
__device__ __forceinline__ float sub_h2_hi(float v, uint32_t p) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(v), "v"(p));
    return d;
}
// the bare conversion: for values known to be in range (residuals of a clamped value, data bounded by construction)
__device__ __forceinline__ uint32_t pack_h2_raw(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(hw_f2_t{lo, hi}, hw_h2_t));
}
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) { return pack_h2_raw(h16_clamp(lo), h16_clamp(hi)); }
#else
typedef __bf16 hw_h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h2f(h16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float h2f_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float h2f_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
// v_cvt_pk_bf16_f32: hardware round-to-nearest-even, two values per instruction
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(hw_f2_t{lo, hi}, hw_h2_t));
}
__device__ __forceinline__ float h16_clamp(float v) { return v; }
__device__ __forceinline__ uint32_t pack_h2_raw(float lo, float hi) { return pack_h2(lo, hi); }
__device__ __forceinline__ float sub_h2_lo(float v, uint32_t p) { return v - h2f_lo(p); }
__device__ __forceinline__ float sub_h2_hi(float v, uint32_t p) { return v - h2f_hi(p); }
#endif
// (hi, lo) pairs of two values: hi = the 16-bit rounding (saturating), lo = the rounding of what it left -- one clamp per value, the
// residual of a clamped value is in range by construction
__device__ __forceinline__ void split_h2(float p, float q, uint32_t& hi, uint32_t& lo) {
    p = h16_clamp(p); q = h16_clamp(q);
    hi = pack_h2_raw(p, q);
    lo = pack_h2_raw(sub_h2_lo(p, hi), sub_h2_hi(q, hi));
}
__device__ __forceinline__ h16_t f2h(float f) { return (h16_t)(pack_h2(f, 0.f) & 0xffffu); }
// the squares of a fused (I)GDN contraction in 16-bit storage (scaled: see H16_SQ_SCALE; gamma' carries the inverse factor)
__device__ __forceinline__ uint32_t pack_sq2(float a, float b) { return pack_h2(a * a * H16_SQ_SCALE, b * b * H16_SQ_SCALE); }

template <typename T> struct elem;
template <> struct elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct elem<h16_t> {
    static __device__ __forceinline__ float ld(const h16_t* p) { return h2f(*p); }
    static __device__ __forceinline__ void st(h16_t* p, float v) { *p = f2h(v); }
};

// dtype-erased scalar access for the strided (image-side) kernels
__device__ __forceinline__ float ld_any(const void* p, int64_t i, int dtype) {
    return dtype == HESIC_H16 ? h2f(((const h16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, int64_t i, int dtype, float v) {
    if (dtype == HESIC_H16) ((h16_t*)p)[i] = f2h(v);
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
