// GDN / IGDN (compressai/layers/gdn.py:55-70) fused into one pass over the activations:
//   beta' = max(beta, sqrt(beta_min + 2^-36))^2 - 2^-36,  gamma' = max(gamma, 2^-18)^2 - 2^-36
//   (NonNegativeParametrizer, compressai/ops/parametrizers.py:41-44)
//   n[p,i] = beta'[i] + sum_j gamma'[i,j] * x[p,j]^2 ;  y = x * rsqrt(n)  (GDN)  |  x * sqrt(n)  (IGDN)
// The reference runs x**2, a 1x1 conv, rsqrt and mul as four kernels plus four for the reparam; here the
// activation tile is read once, the CxC contraction runs on the matrix cores out of LDS, and y is written
// once: the kernel is HBM bound (2 * C * sizeof(T) bytes per pixel).
#include "common.h"

namespace {

constexpr float kPedestal = 1.0f / 68719476736.0f;   // 2^-36
constexpr float kGammaBound = 1.0f / 262144.0f;      // 2^-18

__device__ __forceinline__ float reparam(float v, float bound) {
    const float t = fmaxf(v, bound);
    return t * t - kPedestal;
}

// ------------------------------------------------------------------ generic (any C), one thread per output
__global__ void gdn_generic_kernel(const void* __restrict__ x, const float* __restrict__ beta, const float* __restrict__ gamma,
                                   void* __restrict__ y, int64_t P, int C, int inverse, float beta_bound, int dtype) {
    const int64_t n = P * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        const int64_t p = i / C;
        float norm = reparam(beta[c], beta_bound);
        for (int j = 0; j < C; ++j) {
            const float xv = ld_any(x, p * C + j, dtype);
            norm += reparam(gamma[(int64_t)c * C + j], kGammaBound) * xv * xv;
        }
        const float xv = ld_any(x, i, dtype);
        st_any(y, i, dtype, xv * (inverse ? sqrtf(norm) : rsqrtf(norm)));
    }
}

// NHWC maps with a handful of channels (the 3-channel GDNs under autograd): one thread per PIXEL, gamma' / beta' in registers -- the
// one-thread-per-output kernel above re-reads the pixel and re-does the reparametrisation for every channel behind 64-bit i % C, i / C (26 us
// on a 512^2 batch-8 image)
template <int C, typename T>
__global__ __launch_bounds__(256) void gdn_small_nhwc_kernel(const T* __restrict__ x, const float* __restrict__ beta, const float* __restrict__ gamma,
                                                             T* __restrict__ y, int64_t P, int inverse, float beta_bound) {
    float g[C][C], bt[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        bt[c] = reparam(beta[c], beta_bound);
#pragma unroll
        for (int j = 0; j < C; ++j) g[c][j] = reparam(gamma[c * C + j], kGammaBound);
    }
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        float xv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = elem<T>::ld(x + p * C + c);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float norm = bt[c];
#pragma unroll
            for (int j = 0; j < C; ++j) norm += g[c][j] * xv[j] * xv[j];
            elem<T>::st(y + p * C + c, xv[c] * (inverse ? sqrtf(norm) : rsqrtf(norm)));
        }
    }
}

// planar (NCHW) images with a handful of channels -- pre_gdn / after_gdn, C = 3 (newnet1.py:630,669): one thread per pixel,
// every plane read and written coalesced, no layout copy on either side
template <int C>
__global__ void gdn_planar_kernel(const void* __restrict__ x, const float* __restrict__ beta, const float* __restrict__ gamma,
                                  void* __restrict__ y, int B, int64_t HW, int inverse, float beta_bound, int dtype) {
    float g[C][C], bt[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        bt[c] = reparam(beta[c], beta_bound);
#pragma unroll
        for (int j = 0; j < C; ++j) g[c][j] = reparam(gamma[c * C + j], kGammaBound);
    }
    const int64_t n = (int64_t)B * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW, base = b * C * HW + p;
        float xv[C], sq[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { xv[c] = ld_any(x, base + c * HW, dtype); sq[c] = xv[c] * xv[c]; }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float norm = bt[c];
#pragma unroll
            for (int j = 0; j < C; ++j) norm += g[c][j] * sq[j];
            st_any(y, base + c * HW, dtype, xv[c] * (inverse ? sqrtf(norm) : rsqrtf(norm)));
        }
    }
}

// ------------------------------------------------------------------ C = 128 on the matrix cores
template <typename T> struct G;
template <> struct G<h16_t> { static constexpr int BP = 128, CE = 8; };   // pixels per tile, elems / 16 B
template <> struct G<float> { static constexpr int BP = 64, CE = 4; };

template <typename T>
__device__ __forceinline__ int g_off(int row, int slot) {
    // rows are 128 channels = 16 (bf16) or 32 (fp32) 16-byte slots; XOR the low 4 slot bits with the row
    return (row * (128 / G<T>::CE) + (slot ^ (row & 15))) * 16;
}

template <typename T>
__global__ __launch_bounds__(256) void gdn128_kernel(const T* __restrict__ x, const float* __restrict__ beta,
                                                     const float* __restrict__ gamma, T* __restrict__ y, int64_t P,
                                                     int inverse, float beta_bound) {
    constexpr int C = 128;
    constexpr int BP = G<T>::BP, CE = G<T>::CE, SPR = C / CE;           // slots per row
    constexpr int ROWB = C * (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* gs = smem;                       // gamma' [128][128] T
    unsigned char* xs = smem + C * ROWB;            // x tile [BP][128] T
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // gamma' -> LDS (once per block)
    for (int c = tid; c < C * SPR; c += 256) {
        const int row = c / SPR, slot = c % SPR;
        const float* gp = gamma + row * C + slot * CE;
        if constexpr (sizeof(T) == 2) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = reparam(gp[e], kGammaBound) * H16_SQ_UNSCALE;
            *(u32x4*)(gs + g_off<T>(row, slot)) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
        } else {
            *(f32x4*)(gs + g_off<T>(row, slot)) = f32x4{reparam(gp[0], kGammaBound), reparam(gp[1], kGammaBound),
                                                         reparam(gp[2], kGammaBound), reparam(gp[3], kGammaBound)};
        }
    }
    // wave tiling: NPT pixel tiles of 32; each pixel tile shared by 4/NPT waves that split the 4 cout tiles
    constexpr int NPT = BP / 32, WPT = 4 / NPT, CT = 4 / WPT;
    const int pt = wave % NPT, cbase = (wave / NPT) * CT;
    const int frow = lane & 31, fh = lane >> 5;
    float bv[CT][4][4];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[i][g][e] = reparam(beta[(cbase + i) * 32 + 8 * g + 4 * fh + e], beta_bound);

    const int64_t ntiles = (P + BP - 1) / BP;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p0 = tile * BP;
        __syncthreads();   // previous tile's readers are done (also orders the gamma stores on the first trip)
        for (int c = tid; c < BP * SPR; c += 256) {
            const int row = c / SPR, slot = c % SPR;
            u32x4 v = u32x4{0, 0, 0, 0};
            if (p0 + row < P) v = *(const u32x4*)(x + (p0 + row) * C + slot * CE);
            *(u32x4*)(xs + g_off<T>(row, slot)) = v;
        }
        __syncthreads();
        f32x16 acc[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks) {
                const u32x4 raw = *(const u32x4*)(xs + g_off<T>(pt * 32 + frow, ks * 2 + fh));
                float f[8] = {h2f_lo(raw.x), h2f_hi(raw.x),
                              h2f_lo(raw.y), h2f_hi(raw.y),
                              h2f_lo(raw.z), h2f_hi(raw.z),
                              h2f_lo(raw.w), h2f_hi(raw.w)};
                u32x4 sq = u32x4{pack_sq2(f[0], f[1]), pack_sq2(f[2], f[3]), pack_sq2(f[4], f[5]), pack_sq2(f[6], f[7])};
                const h16x8 xf = __builtin_bit_cast(h16x8, sq);
#pragma unroll
                for (int i = 0; i < CT; ++i) {
                    const h16x8 gf = *(const h16x8*)(gs + g_off<T>((cbase + i) * 32 + frow, ks * 2 + fh));
                    acc[i] = mfma_32x32x16_h16(gf, xf, acc[i], 0, 0, 0);
                }
            }
        } else {
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {   // lane half h owns k = 64h .. 64h+63, four at a time
                const f32x4 xv = *(const f32x4*)(xs + g_off<T>(pt * 32 + frow, fh * 16 + s));
                const f32x4 xq = f32x4{xv.x * xv.x, xv.y * xv.y, xv.z * xv.z, xv.w * xv.w};
#pragma unroll
                for (int i = 0; i < CT; ++i) {
                    const f32x4 gf = *(const f32x4*)(gs + g_off<T>((cbase + i) * 32 + frow, fh * 16 + s));
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[e], xq[e], acc[i], 0, 0, 0);
                }
            }
        }
        // epilogue: lane owns pixel (pt*32+frow) and channels (cbase+i)*32 + 8g + 4fh + {0..3}
        const int64_t p = p0 + pt * 32 + frow;
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (cbase + i) * 32 + 8 * g + 4 * fh;
                float xv[4], o[4];
                const unsigned char* src = xs + g_off<T>(pt * 32 + frow, ch / CE) + (ch % CE) * (int)sizeof(T);
                if constexpr (sizeof(T) == 2) {
                    const u32x2 raw = *(const u32x2*)src;
                    xv[0] = h2f_lo(raw.x); xv[1] = h2f_hi(raw.x);
                    xv[2] = h2f_lo(raw.y); xv[3] = h2f_hi(raw.y);
                } else {
                    const f32x4 raw = *(const f32x4*)src;
                    xv[0] = raw.x; xv[1] = raw.y; xv[2] = raw.z; xv[3] = raw.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n = acc[i][4 * g + e] + bv[i][g][e];
                    o[e] = xv[e] * (inverse ? (sizeof(T) == 2 ? __builtin_amdgcn_sqrtf(n) : sqrtf(n)) : rsqrtf(n));   // bf16 storage: raw v_sqrt_f32
                }
                if (p < P) {
                    if constexpr (sizeof(T) == 2) *(u32x2*)(y + p * C + ch) = u32x2{pack_h2(o[0], o[1]), pack_h2(o[2], o[3])};
                    else *(f32x4*)(y + p * C + ch) = f32x4{o[0], o[1], o[2], o[3]};
                }
            }
    }
}

}  // namespace

extern "C" int hesic_gdn_forward_planar(const void* x, const float* beta, const float* gamma, void* y, int B, int C, int64_t HW,
                                        int inverse, float beta_min, int dtype, void* stream) {
    HESIC_CHECK_ARG(x && beta && gamma && y && B > 0 && HW > 0, "gdn_forward_planar: bad arguments");
    HESIC_CHECK_ARG(C == 3, "gdn_forward_planar: built for the 3-channel image-side GDNs (got C=%d)", C);
    HESIC_CHECK_ARG(dtype == HESIC_H16 || dtype == HESIC_F32, "gdn_forward_planar: bad dtype");
    const float bound = sqrtf(beta_min + 1.0f / 68719476736.0f);
    hipLaunchKernelGGL(gdn_planar_kernel<3>, dim3(grid_for((int64_t)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, x, beta, gamma, y,
                       B, HW, inverse, bound, dtype);
    HESIC_LAUNCH_RETURN("gdn_forward_planar");
}

extern "C" int hesic_gdn_forward(const void* x, const float* beta, const float* gamma, void* y, int64_t P, int C,
                                 int inverse, float beta_min, int dtype, void* stream) {
    HESIC_CHECK_ARG(x && beta && gamma && y && P > 0 && C > 0, "gdn_forward: bad arguments");
    HESIC_CHECK_ARG(dtype == HESIC_H16 || dtype == HESIC_F32, "gdn_forward: bad dtype");
    const float bound = sqrtf(beta_min + kPedestal);
    hipStream_t st = (hipStream_t)stream;
    if (C == 128) {
        if (dtype == HESIC_H16) {
            const int64_t tiles = cdiv64(P, G<h16_t>::BP);
            const int grid = (int)(tiles < 512 ? tiles : 512);
            hipLaunchKernelGGL(gdn128_kernel<h16_t>, dim3(grid), dim3(256), (128 + G<h16_t>::BP) * 128 * 2, st,
                               (const h16_t*)x, beta, gamma, (h16_t*)y, P, inverse, bound);
        } else {
            const int64_t tiles = cdiv64(P, G<float>::BP);
            const int grid = (int)(tiles < 256 ? tiles : 256);
            hipLaunchKernelGGL(gdn128_kernel<float>, dim3(grid), dim3(256), (128 + G<float>::BP) * 128 * 4, st,
                               (const float*)x, beta, gamma, (float*)y, P, inverse, bound);
        }
    } else if (C == 3) {
        const dim3 g3(grid_for(P, 256, 2048));
        if (dtype == HESIC_H16) hipLaunchKernelGGL((gdn_small_nhwc_kernel<3, h16_t>), g3, dim3(256), 0, st, (const h16_t*)x, beta, gamma, (h16_t*)y, P, inverse, bound);
        else hipLaunchKernelGGL((gdn_small_nhwc_kernel<3, float>), g3, dim3(256), 0, st, (const float*)x, beta, gamma, (float*)y, P, inverse, bound);
    } else {
        hipLaunchKernelGGL(gdn_generic_kernel, dim3(grid_for(P * C, 256)), dim3(256), 0, st, x, beta, gamma, y, P, C, inverse,
                           bound, dtype);
    }
    HESIC_LAUNCH_RETURN("gdn_forward");
}
