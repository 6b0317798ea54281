// Glue of the hyper-synthesis heads and the loss reductions:
// bilinear x4 upsample fused with the concat write, channel-slice copy, global spatial max + LeakyReLU,
// the 1x1 conv + softmax-over-K mixture-weight head, sum(log2 likelihood), sum of squared differences,
// activation backward and dtype cast.  All are single-pass, HBM-bound kernels.
#include <stdarg.h>

#include "common.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

// ------------------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
void hesic_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* hesic_last_error(void) { return g_err; }
extern "C" int hesic_abi_version(void) { return HESIC_ABI_VERSION; }
extern "C" int hesic_h16_format(void) { return HESIC_H16_IS_F16 ? HESIC_H16_FLOAT16 : HESIC_H16_BFLOAT16; }

namespace {

// nn.UpsamplingBilinear2d(scale_factor=4): align_corners=True, src = dst*(in-1)/(out-1) (newnet1.py:524)
template <typename T>
__global__ void upsample4_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int yps, int yco) {
    const int Ho = 4 * H, Wo = 4 * W;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        int64_t r = i / C;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int b = r / Ho;
        const float sy = ry * oy, sx = rx * ox;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = sy - y0, lx = sx - x0;
        const T* xb = x + (int64_t)b * H * W * C + c;
        const float v00 = elem<T>::ld(xb + ((int64_t)y0 * W + x0) * C), v01 = elem<T>::ld(xb + ((int64_t)y0 * W + x1) * C);
        const float v10 = elem<T>::ld(xb + ((int64_t)y1 * W + x0) * C), v11 = elem<T>::ld(xb + ((int64_t)y1 * W + x1) * C);
        const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        elem<T>::st(y + (((int64_t)b * Ho + oy) * Wo + ox) * yps + yco + c, v);
    }
}

// bf16, channels and offsets multiples of 8: one thread = 8 channels of an output pixel (four 16-byte loads, one 16-byte store) -- the
// element-wise form above spends a chain of 64-bit divisions and 2-byte accesses on every value: 43 us for the 5 MB concat buffer of
// gmm_hyper_y2, on the critical path between the third analysis pass and the hyper-synthesis.  Same arithmetic per value.
__global__ void upsample4_fwd_v8_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, int B, int H, int W, int C8, int yps, int yco) {
    const int Ho = 4 * H, Wo = 4 * W;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int total = B * Ho * Wo * C8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c8 = i % C8;
        int r = i / C8;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int b = r / Ho;
        const float sy = ry * oy, sx = rx * ox;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = sy - y0, lx = sx - x0;
        const h16_t* xb = x + ((int64_t)b * H * W * C8 + c8) * 8;
        const int64_t C = (int64_t)C8 * 8;
        const u32x4 q00 = *(const u32x4*)(xb + ((int64_t)y0 * W + x0) * C), q01 = *(const u32x4*)(xb + ((int64_t)y0 * W + x1) * C);
        const u32x4 q10 = *(const u32x4*)(xb + ((int64_t)y1 * W + x0) * C), q11 = *(const u32x4*)(xb + ((int64_t)y1 * W + x1) * C);
        const uint32_t a00[4] = {q00.x, q00.y, q00.z, q00.w}, a01[4] = {q01.x, q01.y, q01.z, q01.w};
        const uint32_t a10[4] = {q10.x, q10.y, q10.z, q10.w}, a11[4] = {q11.x, q11.y, q11.z, q11.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float l = (1.f - ly) * ((1.f - lx) * h2f_lo(a00[e]) + lx * h2f_lo(a01[e])) +
                            ly * ((1.f - lx) * h2f_lo(a10[e]) + lx * h2f_lo(a11[e]));
            const float h = (1.f - ly) * ((1.f - lx) * h2f_hi(a00[e]) + lx * h2f_hi(a01[e])) +
                            ly * ((1.f - lx) * h2f_hi(a10[e]) + lx * h2f_hi(a11[e]));
            o[e] = pack_h2(l, h);
        }
        *(u32x4*)(y + (((int64_t)b * Ho + oy) * Wo + ox) * yps + yco + c8 * 8) = u32x4{o[0], o[1], o[2], o[3]};
    }
}

__global__ void copy_channels_v16_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int64_t P, int chunks, int64_t xps_b,
                                         int64_t xco_b, int64_t yps_b, int64_t yco_b) {
    const int64_t total = P * chunks;          // 16-byte chunks
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int64_t p = i / chunks;
        *(u32x4*)(y + p * yps_b + yco_b + c * 16) = *(const u32x4*)(x + p * xps_b + xco_b + c * 16);
    }
}

// gather form of the transpose: each input cell sums the <= 8x8 outputs that reference it (no atomics)
template <typename T>
__global__ void upsample4_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C, int yps, int yco) {
    const int Ho = 4 * H, Wo = 4 * W;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        int64_t r = i / C;
        const int ix = r % W; r /= W;
        const int iy = r % H;
        const int b = r / H;
        float acc = 0.f;
        // outputs whose y0 or y1 equals iy have src coordinate in (iy-1, iy+1): oy in ((iy-1)/ry, (iy+1)/ry)
        const int oy_lo = ry > 0.f ? max(0, (int)floorf((iy - 1) / ry)) : 0, oy_hi = ry > 0.f ? min(Ho - 1, (int)ceilf((iy + 1) / ry)) : Ho - 1;
        const int ox_lo = rx > 0.f ? max(0, (int)floorf((ix - 1) / rx)) : 0, ox_hi = rx > 0.f ? min(Wo - 1, (int)ceilf((ix + 1) / rx)) : Wo - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float sy = ry * oy;
            const int y0 = (int)sy, y1 = y0 + (y0 < H - 1);
            const float ly = sy - y0;
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - ly;
            if (y1 == iy) wy += ly;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float sx = rx * ox;
                const int x0 = (int)sx, x1 = x0 + (x0 < W - 1);
                const float lx = sx - x0;
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - lx;
                if (x1 == ix) wx += lx;
                if (wx == 0.f) continue;
                acc += wy * wx * elem<T>::ld(dy + (((int64_t)b * Ho + oy) * Wo + ox) * yps + yco + c);
            }
        }
        elem<T>::st(dx + i, acc);
    }
}

template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t P, int C, int xps, int xco, int yps, int yco) {
    const int64_t total = P * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        const int64_t p = i / C;
        y[p * yps + yco + c] = x[p * xps + xco + c];
    }
}

// spatial_pool2d (+ LeakyReLU): block = 64 channels x 4 pixel lanes, grid = (C/64, B)
template <typename T>
__global__ __launch_bounds__(256) void spatial_max_kernel(const T* __restrict__ x, float* __restrict__ out, int32_t* __restrict__ arg,
                                                          int HW, int C, int leaky) {
    // block = 64 channels of one image: 8 groups of 8 channels (one 16-byte load per lane for bf16) x 32 pixel lanes;
    // ties go to the lowest pixel index, like the serial scan of the reference's loop (newnet1.py:441-453)
    constexpr int V = 8, G = 64 / V, PL = 256 / G;
    __shared__ float sv[PL][64];
    __shared__ int si[PL][64];
    const int gl = threadIdx.x % G, pl = threadIdx.x / G;
    const int c0 = blockIdx.x * 64 + gl * V, b = blockIdx.y;
    float best[V];
    int bi[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    const bool vec = sizeof(T) == 2 && (C % V) == 0 && c0 + V <= C;
    for (int p = pl; p < HW; p += PL) {
        const T* px = x + ((int64_t)b * HW + p) * C + c0;
        float v[V];
        if (vec) {
            const u32x4 raw = *(const u32x4*)px;
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = h2f_lo(w[e]); v[2 * e + 1] = h2f_hi(w[e]); }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = c0 + e < C ? elem<T>::ld(px + e) : -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < V; ++e)
            if (v[e] > best[e]) { best[e] = v[e]; bi[e] = p; }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { sv[pl][gl * V + e] = best[e]; si[pl][gl * V + e] = bi[e]; }
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        const int cl = threadIdx.x;
        float bv = sv[0][cl];
        int bp = si[0][cl];
        for (int k = 1; k < PL; ++k)
            if (sv[k][cl] > bv || (sv[k][cl] == bv && si[k][cl] < bp)) { bv = sv[k][cl]; bp = si[k][cl]; }
        const int c = blockIdx.x * 64 + cl;
        out[(int64_t)b * C + c] = leaky ? (bv > 0.f ? bv : 0.01f * bv) : bv;
        if (arg) arg[(int64_t)b * C + c] = bp;
    }
}

// inference variant (no argmax): pixels are split over blockIdx.z as well, partial maxima merge through an
// order-preserving integer atomic; LeakyReLU is monotone, so it is applied to the partials.
__global__ void fill_neg_inf_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = -INFINITY;
}
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
    else atomicMin((unsigned int*)addr, __float_as_uint(v));
}
template <typename T>
__global__ __launch_bounds__(256) void spatial_max_split_kernel(const T* __restrict__ x, float* __restrict__ out, int HW, int C,
                                                                int leaky, int pix_per_block) {
    __shared__ float sv[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
    const int p0 = blockIdx.z * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    float best = -INFINITY;
    if (c < C)
        for (int p = p0 + pl; p < p1; p += 4) best = fmaxf(best, elem<T>::ld(x + ((int64_t)b * HW + p) * C + c));
    sv[pl][cl] = best;
    __syncthreads();
    if (pl == 0 && c < C) {
        best = fmaxf(fmaxf(sv[0][cl], sv[1][cl]), fmaxf(sv[2][cl], sv[3][cl]));
        if (leaky) best = best > 0.f ? best : 0.01f * best;
        atomic_max_float(out + (int64_t)b * C + c, best);
    }
}

// logits[b,n] = sum_j w[n,j] * pooled[b,j] + bias[n]: one wave per output.  (One wave per ROW n for all samples -- W read once instead of
// B times -- leaves 960 waves with 15 serial steps each: 73 - 98 us against 35 - 46 us for this form at B = 8.)
__global__ __launch_bounds__(256) void mix_logits_kernel(const float* __restrict__ pooled, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ logits, int B, int N) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * N) return;
    const int n = wave % N, b = wave / N;
    float acc = 0.f;
    for (int j = lane; j < N; j += 64) acc += w[(int64_t)n * N + j] * pooled[(int64_t)b * N + j];
    acc = wave_sum(acc);
    if (lane == 0) logits[(int64_t)b * N + n] = acc + (bias ? bias[n] : 0.f);
}

// Backward of logits = W pooled + bias on the pooled (B, N) vector (training form of the head; the conv kernels would run
// a 128x128 tile pipeline for 8 "pixels"): dpooled[b, j] = sum_n g[b, n] W[n, j] -- one thread per (b, j), rows of W read
// coalesced; dW[n, j] = sum_b g[b, n] pooled[b, j], dbias[n] = sum_b g[b, n] -- one thread per element, fixed b order.
__global__ __launch_bounds__(256) void pooled_linear_dx_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                               float* __restrict__ dpooled, int B, int N) {
    // block = 32 columns j x 8 slices of the n range, up to 8 samples per thread so every W element is loaded once per
    // block; the samples' g rows sit in LDS (broadcast reads), the W loads of 8 consecutive n are issued together, the
    // slices meet in LDS in a fixed order.  (One thread per (b, j) walking all n was a 960-step serial chain of loads.)
    extern __shared__ float gsm[];                       // [8][N] g rows, then [8][8][32] partial sums
    float* red = gsm + 8 * N;
    const int jl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + jl, b0 = blockIdx.y * 8;
    const int nb = B - b0 < 8 ? B - b0 : 8;
    for (int i = threadIdx.x; i < 8 * N; i += 256) {
        const int bb = i / N;
        gsm[i] = bb < nb ? g[(int64_t)(b0 + bb) * N + (i - bb * N)] : 0.f;
    }
    __syncthreads();
    const int per = (N + 7) / 8, n0 = sl * per, n1 = n0 + per < N ? n0 + per : N;
    float acc[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) acc[bb] = 0.f;
    const int jc = j < N ? j : N - 1;                    // clamped column: loads stay in range, result discarded
    for (int n = n0; n < n1; n += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = n + u < n1 ? w[(int64_t)(n + u) * N + jc] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int nn = n + u < n1 ? n + u : n0;
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) acc[bb] = fmaf(gsm[bb * N + nn], wv[u], acc[bb]);
        }
    }
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) red[(sl * 8 + bb) * 32 + jl] = acc[bb];
    __syncthreads();
    if (j < N && sl < nb) {                              // thread (jl, sl) finishes sample b0 + sl
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += red[(q * 8 + sl) * 32 + jl];
        dpooled[(int64_t)(b0 + sl) * N + j] = v;
    }
}

__global__ __launch_bounds__(256) void pooled_linear_dw_kernel(const float* __restrict__ pooled, const float* __restrict__ g,
                                                               float* __restrict__ dw, float* __restrict__ dbias, int B, int N) {
    const int j = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (j >= N) return;
    float acc = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float gv = g[(int64_t)b * N + n];
        acc = fmaf(gv, pooled[(int64_t)b * N + j], acc);
        sb += gv;
    }
    dw[(int64_t)n * N + j] = acc;
    if (dbias && j == 0) dbias[n] = sb;
}

__global__ void softmax_k_kernel(const float* __restrict__ logits, float* __restrict__ weights, int B, int K, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * M) return;
    const int m = i % M, b = i / M;
    const float* l = logits + (int64_t)b * K * M + m;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, l[k * M]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(l[k * M] - mx);
    for (int k = 0; k < K; ++k) weights[(int64_t)b * K * M + k * M + m] = expf(l[k * M] - mx) / s;
}

__global__ void softmax_k_bwd_kernel(const float* __restrict__ w, const float* __restrict__ g, float* __restrict__ dl, int B, int K, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * M) return;
    const int m = i % M, b = i / M;
    const int64_t o = (int64_t)b * K * M + m;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += g[o + k * M] * w[o + k * M];
    for (int k = 0; k < K; ++k) dl[o + k * M] = w[o + k * M] * (g[o + k * M] - dot);
}

__device__ __forceinline__ void sum_log2_body(const float* __restrict__ lik, int64_t n, double* __restrict__ out, int bid, int nb, double* red) {
    double acc = 0.0;
    const int64_t gt = bid * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)nb * blockDim.x;
    const int64_t n4 = (((uintptr_t)lik & 15) == 0) ? (n >> 2) : 0;           // 16-byte lanes when the tensor allows it
    for (int64_t i = gt; i < n4; i += stride) {
        const f32x4 v = ((const f32x4*)lik)[i];
        acc += (double)log2f(v.x) + (double)log2f(v.y) + (double)log2f(v.z) + (double)log2f(v.w);
    }
    for (int64_t i = n4 * 4 + gt; i < n; i += stride) acc += (double)log2f(lik[i]);
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sum_log2_kernel(const float* __restrict__ lik, int64_t n, double* __restrict__ out) {
    __shared__ double red[4];
    sum_log2_body(lik, n, out, (int)blockIdx.x, (int)gridDim.x, red);
}

struct SqArgs {
    const void* a; const void* b; int a_dt, b_dt; int64_t as[4], bs[4]; int B, C, H, W; double* out;
};
__device__ __forceinline__ void sum_sq_diff_body(const SqArgs& q, int bid, int nb, double* red) {
    // one pixel per thread and iteration (all C channels of it): the index split is 32-bit and done once per pixel, and
    // both an NCHW and an NHWC operand are read with at most a C-element stride between neighbouring lanes
    const int64_t npix = (int64_t)q.B * q.H * q.W;
    double acc = 0.0;
    auto offsets = [&](int64_t p, int64_t& oa, int64_t& ob) {
        int x, y, b;
        if (npix < (1ll << 31)) {
            unsigned r = (unsigned)p;
            x = r % (unsigned)q.W; r /= (unsigned)q.W;
            y = r % (unsigned)q.H; b = r / (unsigned)q.H;
        } else {
            int64_t r = p;
            x = r % q.W; r /= q.W;
            y = r % q.H; b = (int)(r / q.H);
        }
        oa = b * q.as[0] + y * q.as[2] + x * q.as[3];
        ob = b * q.bs[0] + y * q.bs[2] + x * q.bs[3];
    };
    const int64_t stride = (int64_t)nb * blockDim.x;
    int64_t p = bid * (int64_t)blockDim.x + threadIdx.x;
    if (q.C == 3) {
        // the image case: 4 pixels x 3 channels x 2 operands = 24 independent loads in flight per lane
        for (; p + 3 * stride < npix; p += 4 * stride) {
            float va[4][3], vb[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t oa, ob;
                offsets(p + u * stride, oa, ob);
#pragma unroll
                for (int c = 0; c < 3; ++c) { va[u][c] = ld_any(q.a, oa + c * q.as[1], q.a_dt); vb[u][c] = ld_any(q.b, ob + c * q.bs[1], q.b_dt); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float part = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) { const float df = va[u][c] - vb[u][c]; part = fmaf(df, df, part); }
                acc += (double)part;
            }
        }
    }
    for (; p < npix; p += stride) {
        int64_t oa, ob;
        offsets(p, oa, ob);
        float part = 0.f;
        for (int c = 0; c < q.C; ++c) {
            const float df = ld_any(q.a, oa + c * q.as[1], q.a_dt) - ld_any(q.b, ob + c * q.bs[1], q.b_dt);
            part = fmaf(df, df, part);
            if ((c & 7) == 7) { acc += (double)part; part = 0.f; }
        }
        acc += (double)part;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(q.out, red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sum_sq_diff_kernel(const SqArgs q) {
    __shared__ double red[4];
    sum_sq_diff_body(q, (int)blockIdx.x, (int)gridDim.x, red);
}

// The bits / squared-error reductions of one forward in ONE launch (round 5: four sum_log2 + two sum_sq_diff launches before, 66 us per step
// on the metrics stream): block -> job by a scan of <= 10 first-block indices; the bodies are the single-job kernels' (same sums, same order
// per job for a given block count).  *first zero: block 0 of every job does NOT clear its accumulator -- the caller's zero-fill stays.
struct RdBatch {
    int n; int start[11];          // 8 likelihood maps + 2 image pairs + the end marker (9 let start[9..10] run into lik[0])
    const float* lik[8]; int64_t numel[8]; double* lik_out[8];
    int nsq; SqArgs sq[2];
};
__global__ __launch_bounds__(256) void rd_sums_kernel(const RdBatch b) {
    __shared__ double red[4];
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.start[j + 1]) ++j;
    const int bid = (int)blockIdx.x - b.start[j], nb = b.start[j + 1] - b.start[j];
    const int nl = b.n - b.nsq;
    if (j < nl) sum_log2_body(b.lik[j], b.numel[j], b.lik_out[j], bid, nb, red);
    else sum_sq_diff_body(b.sq[j - nl], bid, nb, red);
}

__global__ void log_bwd_kernel(const float* __restrict__ lik, float scale, float* __restrict__ g, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] = scale / lik[i];
}
__global__ void sq_diff_bwd_kernel(const SqArgs q, float scale, float* __restrict__ g) {
    const int64_t n = (int64_t)q.B * q.C * q.H * q.W;
    const int64_t hw = (int64_t)q.H * q.W;
    auto dense = [&](const int64_t* st) { return st[3] == 1 && st[2] == q.W && st[1] == hw && st[0] == hw * q.C; };
    if (q.a_dt == HESIC_F32 && q.b_dt == HESIC_F32 && dense(q.as) && dense(q.bs) && (n & 3) == 0 && !(((uintptr_t)q.a | (uintptr_t)q.b | (uintptr_t)g) & 15)) {
        // both images planar fp32 and dense (x_hat of a 512^2 batch against x): 16-byte lanes, no index arithmetic (the loop below: 26 us on 25 MB)
        const f32x4* a4 = (const f32x4*)q.a;
        const f32x4* b4 = (const f32x4*)q.b;
        f32x4* g4 = (f32x4*)g;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n >> 2); i += (int64_t)gridDim.x * blockDim.x) {
            const f32x4 va = a4[i], vb = b4[i];
            g4[i] = f32x4{scale * (va.x - vb.x), scale * (va.y - vb.y), scale * (va.z - vb.z), scale * (va.w - vb.w)};
        }
        return;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int x = r % q.W; r /= q.W;
        const int y = r % q.H; r /= q.H;
        const int c = r % q.C;
        const int b = r / q.C;
        const float va = ld_any(q.a, b * q.as[0] + c * q.as[1] + y * q.as[2] + x * q.as[3], q.a_dt);
        const float vb = ld_any(q.b, b * q.bs[0] + c * q.bs[1] + y * q.bs[2] + x * q.bs[3], q.b_dt);
        g[i] = scale * (va - vb);
    }
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, int64_t n, int act) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float yv = elem<T>::ld(y + i), g = elem<T>::ld(dy + i);
        float o = g;
        if (act == HESIC_ACT_RELU) o = yv > 0.f ? g : 0.f;
        else if (act == HESIC_ACT_LEAKY) o = yv > 0.f ? g : 0.01f * g;
        elem<T>::st(dx + i, o);
    }
}

__global__ void cast_kernel(const void* __restrict__ x, int xd, void* __restrict__ y, int yd, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        st_any(y, i, yd, ld_any(x, i, xd));
}

// round-half-even of x, stored as y's dtype: EntropyModel._quantize(x, "dequantize") without means (newnet1.py:755) on an
// fp32 latent whose rounded value feeds a bf16 conv
__global__ void round_cast_kernel(const void* __restrict__ x, int xd, void* __restrict__ y, int yd, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        st_any(y, i, yd, rintf(ld_any(x, i, xd)));
}

// the case the inference schedule issues three times per forward (fp32 latent -> rounded bf16 copy): 8 values per thread
__global__ void round_f32_to_bf16_v8_kernel(const float* __restrict__ x, h16_t* __restrict__ y, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4*)(x + i * 8), b = *(const f32x4*)(x + i * 8 + 4);
        *(u32x4*)(y + i * 8) = u32x4{pack_h2(rintf(a.x), rintf(a.y)), pack_h2(rintf(a.z), rintf(a.w)), pack_h2(rintf(b.x), rintf(b.y)),
                                     pack_h2(rintf(b.z), rintf(b.w))};
    }
}

}  // namespace

extern "C" int hesic_round(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, void* stream) {
    HESIC_CHECK_ARG(x && y && n > 0, "round: bad arguments");
    if (x_dtype == HESIC_F32 && y_dtype == HESIC_H16 && n % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
        hipLaunchKernelGGL(round_f32_to_bf16_v8_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (h16_t*)y, n / 8);
        HESIC_LAUNCH_RETURN("round");
    }
    hipLaunchKernelGGL(round_cast_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, n);
    HESIC_LAUNCH_RETURN("round");
}

extern "C" int hesic_upsample4_forward(const void* x, void* y, int B, int H, int W, int C, int yps, int yco, int dtype, void* stream) {
    HESIC_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && yco + C <= yps, "upsample4_forward: bad arguments");
    const int64_t total = (int64_t)B * 16 * H * W * C;
    if (dtype == HESIC_H16 && C % 8 == 0 && yps % 8 == 0 && yco % 8 == 0 && total / 8 < (1ll << 31) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0)
        hipLaunchKernelGGL(upsample4_fwd_v8_kernel, dim3(grid_for(total / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, (h16_t*)y,
                           B, H, W, C / 8, yps, yco);
    else if (dtype == HESIC_H16)
        hipLaunchKernelGGL(upsample4_fwd_kernel<h16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const h16_t*)x, (h16_t*)y, B, H, W, C, yps, yco);
    else
        hipLaunchKernelGGL(upsample4_fwd_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, B, H, W, C, yps, yco);
    HESIC_LAUNCH_RETURN("upsample4_forward");
}

extern "C" int hesic_upsample4_backward(const void* dy, void* dx, int B, int H, int W, int C, int yps, int yco, int dtype, void* stream) {
    HESIC_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && yco + C <= yps, "upsample4_backward: bad arguments");
    const int64_t total = (int64_t)B * H * W * C;
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(upsample4_bwd_kernel<h16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const h16_t*)dy, (h16_t*)dx, B, H, W, C, yps, yco);
    else
        hipLaunchKernelGGL(upsample4_bwd_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dy, (float*)dx, B, H, W, C, yps, yco);
    HESIC_LAUNCH_RETURN("upsample4_backward");
}

extern "C" int hesic_copy_channels(const void* x, void* y, int64_t P, int C, int xps, int xco, int yps, int yco, int dtype, void* stream) {
    HESIC_CHECK_ARG(x && y && P > 0 && C > 0 && xco + C <= xps && yco + C <= yps, "copy_channels: bad arguments");
    const int es = dtype == HESIC_H16 ? 2 : 4;
    if ((C * es) % 16 == 0 && (xps * es) % 16 == 0 && (xco * es) % 16 == 0 && (yps * es) % 16 == 0 && (yco * es) % 16 == 0 && ((uintptr_t)x & 15) == 0 &&
        ((uintptr_t)y & 15) == 0) {
        // whole 16-byte chunks on both sides: one chunk per thread instead of one element (38 us for 3 MB, on the same critical path)
        const int chunks = C * es / 16;
        hipLaunchKernelGGL(copy_channels_v16_kernel, dim3(grid_for(P * chunks, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x,
                           (unsigned char*)y, P, chunks, (int64_t)xps * es, (int64_t)xco * es, (int64_t)yps * es, (int64_t)yco * es);
        HESIC_LAUNCH_RETURN("copy_channels");
    }
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(copy_channels_kernel<h16_t>, dim3(grid_for(P * C, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const h16_t*)x, (h16_t*)y, P, C, xps, xco, yps, yco);
    else
        hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(grid_for(P * C, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, P, C, xps, xco, yps, yco);
    HESIC_LAUNCH_RETURN("copy_channels");
}

// backward of spatial_max: dx[b, p, c] = (p == argmax[b, c]) ? g[b, c] * (leaky && out[b, c] <= 0 ? 0.01 : 1) : 0 -- every element of dx
// written once, coalesced (the tensor-op form was a fill, a compare, a where, a cast and a scatter: seven launches)
template <typename T>
__global__ void spatial_max_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out, const int32_t* __restrict__ arg, T* __restrict__ dx,
                                       int64_t n, int HW, int C, int leaky) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        const int p = (int)(r % HW), b = (int)(r / HW);
        float v = 0.f;
        if (arg[b * C + c] == p) {
            const float gv = g[b * C + c];
            v = (leaky && !(out[b * C + c] > 0.f)) ? 0.01f * gv : gv;
        }
        elem<T>::st(dx + i, v);
    }
}

extern "C" int hesic_spatial_max_backward(const float* g, const float* out, const int32_t* argmax, void* dx, int B, int HW, int C, int dtype, int leaky,
                                          void* stream) {
    HESIC_CHECK_ARG(g && out && argmax && dx && B > 0 && HW > 0 && C > 0 && (dtype == HESIC_H16 || dtype == HESIC_F32), "spatial_max_backward: bad arguments");
    const int64_t n = (int64_t)B * HW * C;
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(spatial_max_bwd_kernel<h16_t>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, g, out, argmax, (h16_t*)dx, n, HW, C, leaky);
    else
        hipLaunchKernelGGL(spatial_max_bwd_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, g, out, argmax, (float*)dx, n, HW, C, leaky);
    HESIC_LAUNCH_RETURN("spatial_max_backward");
}

extern "C" int hesic_spatial_max(const void* x, float* out, int32_t* argmax, int B, int HW, int C, int dtype, int leaky, void* stream) {
    HESIC_CHECK_ARG(x && out && B > 0 && HW > 0 && C > 0, "spatial_max: bad arguments");
    if (!argmax && HW >= 256) {
        const int ppb = 64, nz = (HW + ppb - 1) / ppb;
        hipLaunchKernelGGL(fill_neg_inf_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, B * C);
        const dim3 g3((C + 63) / 64, B, nz);
        if (dtype == HESIC_H16)
            hipLaunchKernelGGL(spatial_max_split_kernel<h16_t>, g3, dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, out, HW, C, leaky, ppb);
        else
            hipLaunchKernelGGL(spatial_max_split_kernel<float>, g3, dim3(256), 0, (hipStream_t)stream, (const float*)x, out, HW, C, leaky, ppb);
        HESIC_LAUNCH_RETURN("spatial_max");
    }
    const dim3 grid((C + 63) / 64, B);
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(spatial_max_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, out, argmax, HW, C, leaky);
    else
        hipLaunchKernelGGL(spatial_max_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, out, argmax, HW, C, leaky);
    HESIC_LAUNCH_RETURN("spatial_max");
}

extern "C" int hesic_mix_weights_forward(const float* pooled, const float* w, const float* bias, float* logits, float* weights,
                                         int B, int K, int M, void* stream) {
    HESIC_CHECK_ARG(pooled && w && logits && weights && B > 0 && K > 0 && M > 0, "mix_weights_forward: bad arguments");
    const int N = K * M;
    hipLaunchKernelGGL(mix_logits_kernel, dim3((unsigned)cdiv64((int64_t)B * N * 64, 256)), dim3(256), 0, (hipStream_t)stream, pooled, w, bias, logits, B, N);
    hipLaunchKernelGGL(softmax_k_kernel, dim3((B * M + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, weights, B, K, M);
    HESIC_LAUNCH_RETURN("mix_weights_forward");
}

// ---------------------------------------------------------------- Adam over many tensors in one launch
// torch.optim.Adam (newtrain1.py:294-295) in its default (no amsgrad / weight decay) form.  The descriptor travels BY VALUE
// in the kernel arguments (no device table: nothing to keep alive, capturable in a HIP graph); every tensor has its own
// fp32 step counter like torch's capturable state, bumped by a first tiny launch so all blocks of a tensor see the same
// count.  Math as the reference's single-tensor path: m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), bias corrections in double.
__global__ void adam_bump_steps_kernel(const hesic_adam_chunk c) {
    const int i = threadIdx.x;
    if (i < c.n) *c.step[i] += 1.f;
}

constexpr int ADAM_EPB = 4096;                   // elements per block: the two double-precision pow() of a block's bias corrections amortise
__global__ __launch_bounds__(256) void adam_update_kernel(const hesic_adam_chunk c) {
    const int bid = blockIdx.x;
    int t = 0;
    while (t + 1 < c.n && c.block0[t + 1] <= bid) ++t;            // uniform: scalar loop over <= 24 entries
    __shared__ float sc[2];
    if (threadIdx.x == 0) {
        const double st = (double)*c.step[t];
        const double bc1 = 1.0 - pow((double)c.beta1, st), bc2 = 1.0 - pow((double)c.beta2, st);
        sc[0] = (float)((double)c.lr / bc1);
        sc[1] = (float)(1.0 / sqrt(bc2));
    }
    __syncthreads();
    const float step_size = sc[0], rbc2s = sc[1];
    float* __restrict__ p = c.p[t];
    const float* __restrict__ g = c.g[t];
    float* __restrict__ m = c.m[t];
    float* __restrict__ v = c.v[t];
    const int64_t n = c.numel[t];
    const int64_t base = (int64_t)(bid - c.block0[t]) * ADAM_EPB;
    const float omb1 = 1.f - c.beta1, omb2 = 1.f - c.beta2;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && base + ADAM_EPB <= n;
    if (vec) {
#pragma unroll
        for (int u = 0; u < ADAM_EPB / 1024; ++u) {
            const int64_t i = base + u * 1024 + threadIdx.x * 4;
            const f32x4 gv = *(const f32x4*)(g + i);
            f32x4 mv = *(const f32x4*)(m + i), vv = *(const f32x4*)(v + i), pv = *(const f32x4*)(p + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mv[e] = mv[e] + omb1 * (gv[e] - mv[e]);
                vv[e] = vv[e] * c.beta2 + omb2 * gv[e] * gv[e];
                pv[e] -= step_size * (mv[e] / (sqrtf(vv[e]) * rbc2s + c.eps));
            }
            *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv; *(f32x4*)(p + i) = pv;
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < base + ADAM_EPB && i < n; i += 256) {
            const float gv = g[i];
            const float mv = m[i] + omb1 * (gv - m[i]);
            const float vv = v[i] * c.beta2 + omb2 * gv * gv;
            m[i] = mv; v[i] = vv;
            p[i] -= step_size * (mv / (sqrtf(vv) * rbc2s + c.eps));
        }
    }
}

extern "C" int hesic_adam_step(const hesic_adam_chunk* chunk_host, void* stream) {
    HESIC_CHECK_ARG(chunk_host && chunk_host->n > 0 && chunk_host->n <= HESIC_ADAM_MAX_TENSORS, "adam_step: bad chunk");
    hesic_adam_chunk c = *chunk_host;
    int blk = 0;
    for (int i = 0; i < c.n; ++i) {
        HESIC_CHECK_ARG(c.p[i] && c.g[i] && c.m[i] && c.v[i] && c.step[i] && c.numel[i] > 0, "adam_step: null tensor");
        c.block0[i] = blk;
        blk += (int)((c.numel[i] + ADAM_EPB - 1) / ADAM_EPB);
    }
    c.block0[c.n] = blk;
    hipLaunchKernelGGL(adam_bump_steps_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c);
    hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, c);
    HESIC_LAUNCH_RETURN("adam_step");
}

extern "C" int hesic_pooled_linear_forward(const float* pooled, const float* w, const float* bias, float* logits, int B, int N,
                                           void* stream) {
    HESIC_CHECK_ARG(pooled && w && logits && B > 0 && N > 0, "pooled_linear_forward: bad arguments");
    hipLaunchKernelGGL(mix_logits_kernel, dim3((unsigned)cdiv64((int64_t)B * N * 64, 256)), dim3(256), 0, (hipStream_t)stream, pooled, w, bias,
                       logits, B, N);
    HESIC_LAUNCH_RETURN("pooled_linear_forward");
}

extern "C" int hesic_pooled_linear_backward(const float* pooled, const float* w, const float* g, float* dpooled, float* dw, float* dbias,
                                            int B, int N, void* stream) {
    HESIC_CHECK_ARG(pooled && w && g && B > 0 && N > 0 && N <= 4096 && B <= 65535, "pooled_linear_backward: bad arguments (N <= 4096)");
    if (dpooled)
        hipLaunchKernelGGL(pooled_linear_dx_kernel, dim3((N + 31) / 32, (B + 7) / 8), dim3(256), (size_t)(8 * N + 8 * 8 * 32) * sizeof(float), (hipStream_t)stream, w, g,
                           dpooled, B, N);
    if (dw)
        hipLaunchKernelGGL(pooled_linear_dw_kernel, dim3((N + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, pooled, g, dw, dbias, B, N);
    HESIC_LAUNCH_RETURN("pooled_linear_backward");
}

extern "C" int hesic_softmax_k_forward(const float* logits, float* weights, int B, int K, int M, void* stream) {
    HESIC_CHECK_ARG(logits && weights && B > 0 && K > 0 && M > 0, "softmax_k_forward: bad arguments");
    hipLaunchKernelGGL(softmax_k_kernel, dim3((B * M + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, weights, B, K, M);
    HESIC_LAUNCH_RETURN("softmax_k_forward");
}

extern "C" int hesic_softmax_k_backward(const float* weights, const float* g, float* dlogits, int B, int K, int M, void* stream) {
    HESIC_CHECK_ARG(weights && g && dlogits && B > 0 && K > 0 && M > 0, "softmax_k_backward: bad arguments");
    hipLaunchKernelGGL(softmax_k_bwd_kernel, dim3((B * M + 255) / 256), dim3(256), 0, (hipStream_t)stream, weights, g, dlogits, B, K, M);
    HESIC_LAUNCH_RETURN("softmax_k_backward");
}

// RateDistortionLoss from its three device-side sums (newtrain1.py:37-56): bpp = -acc[0] / npix, mse = (acc[1] + acc[2]) / numel,
// loss = lambda_255sq * mse + bpp -- one single-thread launch instead of nine one-element tensor ops (each a ~5 us graph node)
__global__ void rd_loss_combine_kernel(const double* __restrict__ acc, double lambda_255sq, double inv_npix, double inv_numel, float* __restrict__ out) {
    const double bpp = -acc[0] * inv_npix, mse = (acc[1] + acc[2]) * inv_numel;
    out[0] = (float)(lambda_255sq * mse + bpp);
    out[1] = (float)bpp;
    out[2] = (float)mse;
}

extern "C" int hesic_rd_loss_combine(const double* acc, double lambda_255sq, int64_t npix, int64_t numel, float* out3, void* stream) {
    HESIC_CHECK_ARG(acc && out3 && npix > 0 && numel > 0, "rd_loss_combine: bad arguments");
    hipLaunchKernelGGL(rd_loss_combine_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, lambda_255sq, 1.0 / (double)npix, 1.0 / (double)numel, out3);
    HESIC_LAUNCH_RETURN("rd_loss_combine");
}

extern "C" int hesic_sum_log2(const float* lik, int64_t n, double* out, void* stream) {
    HESIC_CHECK_ARG(lik && out && n > 0, "sum_log2: bad arguments");
    // one fp64 atomic per block on ONE address: they serialise in L2.  Measured (round 4, 1.57 M likelihoods, back-to-back launches):
    // 128 blocks 6.0 us, 512 blocks 8.9 us, 2048 blocks 21.3 us -- so few, fat blocks (HESIC_SUM_LOG2_BLOCKS = A/B switch)
    constexpr int max_blocks = 128;
    hipLaunchKernelGGL(sum_log2_kernel, dim3(grid_for(n / 4 + 1, 256, max_blocks < 1 ? 1 : max_blocks)), dim3(256), 0, (hipStream_t)stream, lik, n, out);
    HESIC_LAUNCH_RETURN("sum_log2");
}

extern "C" int hesic_sum_sq_diff(const void* a, int a_dtype, const int64_t a_strides[4], const void* b, int b_dtype,
                                 const int64_t b_strides[4], int B, int C, int H, int W, double* out, void* stream) {
    HESIC_CHECK_ARG(a && b && out && a_strides && b_strides && B > 0 && C > 0 && H > 0 && W > 0, "sum_sq_diff: bad arguments");
    SqArgs q;
    q.a = a; q.b = b; q.a_dt = a_dtype; q.b_dt = b_dtype; q.B = B; q.C = C; q.H = H; q.W = W; q.out = out;
    for (int i = 0; i < 4; ++i) { q.as[i] = a_strides[i]; q.bs[i] = b_strides[i]; }
    // few blocks: every block ends in one fp64 atomic on the same address, and those serialise in L2
    hipLaunchKernelGGL(sum_sq_diff_kernel, dim3(grid_for((int64_t)B * H * W, 256 * 4, 512)), dim3(256), 0, (hipStream_t)stream, q);
    HESIC_LAUNCH_RETURN("sum_sq_diff");
}

extern "C" int hesic_rd_sums(int n_lik, const float* const* lik, const int64_t* numel, double* const* lik_out, int n_sq, const void* const* a,
                             const int* a_dtype, const int64_t* a_strides, const void* const* b, const int* b_dtype, const int64_t* b_strides,
                             const int* dims, double* const* sq_out, void* stream) {
    HESIC_CHECK_ARG(n_lik >= 0 && n_lik <= 8 && n_sq >= 0 && n_sq <= 2 && n_lik + n_sq > 0, "rd_sums: at most 8 likelihood maps and 2 image pairs");
    HESIC_CHECK_ARG((n_lik == 0 || (lik && numel && lik_out)) && (n_sq == 0 || (a && b && a_dtype && b_dtype && a_strides && b_strides && dims && sq_out)), "rd_sums: null pointer");
    RdBatch rb;
    memset(&rb, 0, sizeof(rb));
    constexpr int max_blocks = 128;
    int blk = 0;
    for (int i = 0; i < n_lik; ++i) {
        HESIC_CHECK_ARG(lik[i] && lik_out[i] && numel[i] > 0, "rd_sums: likelihood map %d: bad arguments", i);
        rb.lik[i] = lik[i]; rb.numel[i] = numel[i]; rb.lik_out[i] = lik_out[i];
        rb.start[i] = blk;
        blk += grid_for(numel[i] / 4 + 1, 256, max_blocks < 1 ? 1 : max_blocks);
    }
    for (int i = 0; i < n_sq; ++i) {
        HESIC_CHECK_ARG(a[i] && b[i] && sq_out[i] && dims[4 * i] > 0 && dims[4 * i + 1] > 0 && dims[4 * i + 2] > 0 && dims[4 * i + 3] > 0, "rd_sums: image pair %d: bad arguments", i);
        SqArgs& q = rb.sq[i];
        q.a = a[i]; q.b = b[i]; q.a_dt = a_dtype[i]; q.b_dt = b_dtype[i]; q.B = dims[4 * i]; q.C = dims[4 * i + 1]; q.H = dims[4 * i + 2]; q.W = dims[4 * i + 3];
        q.out = sq_out[i];
        for (int k = 0; k < 4; ++k) { q.as[k] = a_strides[4 * i + k]; q.bs[k] = b_strides[4 * i + k]; }
        rb.start[n_lik + i] = blk;
        blk += grid_for((int64_t)q.B * q.H * q.W, 256 * 4, 512);
    }
    rb.n = n_lik + n_sq; rb.nsq = n_sq;
    rb.start[rb.n] = blk;
    hipLaunchKernelGGL(rd_sums_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, rb);
    HESIC_LAUNCH_RETURN("rd_sums");
}

extern "C" int hesic_log_backward(const float* lik, float scale, float* g_lik, int64_t n, void* stream) {
    HESIC_CHECK_ARG(lik && g_lik && n > 0, "log_backward: bad arguments");
    hipLaunchKernelGGL(log_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, lik, scale, g_lik, n);
    HESIC_LAUNCH_RETURN("log_backward");
}

extern "C" int hesic_sq_diff_backward(const void* a, int a_dtype, const int64_t a_strides[4], const void* b, int b_dtype,
                                      const int64_t b_strides[4], int B, int C, int H, int W, float scale, float* g_a, void* stream) {
    HESIC_CHECK_ARG(a && b && g_a && a_strides && b_strides && B > 0 && C > 0 && H > 0 && W > 0, "sq_diff_backward: bad arguments");
    SqArgs q;
    q.a = a; q.b = b; q.a_dt = a_dtype; q.b_dt = b_dtype; q.B = B; q.C = C; q.H = H; q.W = W; q.out = nullptr;
    for (int i = 0; i < 4; ++i) { q.as[i] = a_strides[i]; q.bs[i] = b_strides[i]; }
    hipLaunchKernelGGL(sq_diff_bwd_kernel, dim3(grid_for((int64_t)B * C * H * W, 256)), dim3(256), 0, (hipStream_t)stream, q, scale, g_a);
    HESIC_LAUNCH_RETURN("sq_diff_backward");
}

extern "C" int hesic_act_backward(const void* y, const void* dy, void* dx, int64_t n, int act, int dtype, void* stream) {
    HESIC_CHECK_ARG(y && dy && dx && n > 0, "act_backward: bad arguments");
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(act_bwd_kernel<h16_t>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)y,
                           (const h16_t*)dy, (h16_t*)dx, n, act);
    else
        hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)y,
                           (const float*)dy, (float*)dx, n, act);
    HESIC_LAUNCH_RETURN("act_backward");
}

extern "C" int hesic_cast(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, void* stream) {
    HESIC_CHECK_ARG(x && y && n > 0, "cast: bad arguments");
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, n);
    HESIC_LAUNCH_RETURN("cast");
}

// ------------------------------------------------------------------------------ im2col of the image side as hi/lo bf16
// First analysis layer of the bf16x3 mode (g_a_conv1, newnet1.py:583 / :633): the 3-channel fp32 image becomes the column
// matrix P[pixel][k], k = (ci*KH + ky)*KW + kx (the row-major flattening of a PyTorch conv weight (Cout, Cin, KH, KW)), zero
// beyond Cin*KH*KW up to KP, written as [hi(KP) | lo(KP)] bf16 per output pixel -- so the layer runs as a 1x1 implicit GEMM on
// hi/lo operands (hesic_conv2d_forward_hilo) with the hi/lo GDN epilogue, and no value passes through a single bf16.
namespace {
__global__ __launch_bounds__(256) void im2col_hilo_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int C,
                                                          int H, int W, int KH, int KW, int stride, int pad, int Ho, int Wo, int KP,
                                                          h16_t* __restrict__ cols) {
    const int chunks = KP >> 3, kk = KH * KW, kmax = C * kk;
    const int64_t total = (int64_t)B * Ho * Wo * chunks;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % chunks);
        int64_t p = i / chunks;
        const int ox = (int)(p % Wo);
        int64_t r = p / Wo;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = q * 8 + e;
            const int ci = k / kk, t = k - ci * kk, ky = t / KW, kx = t - ky * KW;
            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
            const bool ok = k < kmax && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            v[e] = ok ? x[b * sb + ci * sc + iy * sy + ix * sx] : 0.f;
        }
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = pack_h2(v[2 * e], v[2 * e + 1]);
            lo[e] = pack_h2(v[2 * e] - h2f_lo(hi[e]), v[2 * e + 1] - h2f_hi(hi[e]));
        }
        h16_t* dst = cols + p * (2 * KP) + q * 8;
        *(u32x4*)dst = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *(u32x4*)(dst + KP) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
}
}  // namespace

extern "C" int hesic_im2col_hilo(const float* x, const int64_t x_strides[4], int B, int C, int H, int W, int KH, int KW, int stride, int pad,
                                 int Ho, int Wo, int KP, void* cols, void* stream) {
    HESIC_CHECK_ARG(x && x_strides && cols, "im2col_hilo: null pointer");
    HESIC_CHECK_ARG(B > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && KP % 8 == 0 && KP >= C * KH * KW, "im2col_hilo: bad geometry (KP must be a multiple of 8 >= C*KH*KW)");
    HESIC_CHECK_ARG(Ho == (H + 2 * pad - KH) / stride + 1 && Wo == (W + 2 * pad - KW) / stride + 1, "im2col_hilo: output size does not match");
    const int64_t total = (int64_t)B * Ho * Wo * (KP / 8);
    hipLaunchKernelGGL(im2col_hilo_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, x, x_strides[0], x_strides[1],
                       x_strides[2], x_strides[3], B, C, H, W, KH, KW, stride, pad, Ho, Wo, KP, (h16_t*)cols);
    HESIC_LAUNCH_RETURN("im2col_hilo");
}

// ---- HESIC+ wavefront decode (ywz/mywork/newnet1_joint.py:1190-1260 walks the map pixel by pixel; models.HSICJoint.decompress walks it
// group by group of mutually independent pixels, t = w + 3h).  The device work of one group is a HIP graph replayed per step; this kernel
// opens it and is what makes the graph step-independent: every per-step quantity (position in the index tables, the previous group's
// size and rows) lives in device memory.  ONE block (a group is <= ~W/3 pixels: ~100 KB to move), so the phases order themselves with
// block barriers:
//   1. the symbols the host decoded for the PREVIOUS group (sym: [nprev][C] int32 pixel-major, copied up in front of the replay) go
//      into the padded latent map as values sym - minmax in the map's storage type;
//   2. this group's P pixels at offset *pos of the whole-map tables: their 5 x 5 crops of the padded map -> crops[P][25][M] (the masked
//      conv's input batch), their hyper-decoder rows par[row] -> feat[p][0, c_par), view 2's extra rows ext[row] -> feat[p][e_off, +M)
//      (the masked conv writes its own slice of feat in between, the 1x1 entropy-parameter net reads feat);
//   3. *pos += P, state[0] = P, prev_centre = this group's padded rows.
// state = {nprev, C, minmax}; channels = the C coded channels (header bitmap); both filled by the host once per view.  P == 0: phase 1 only.
namespace {
struct JointStep {
    unsigned char* y_rows; int es, M, Wp;
    const int32_t* sym; int64_t* prev_centre; int32_t* state; const int32_t* channels;
    const int64_t* all_centre; const int64_t* all_rows; int64_t* pos; int P, dtype;
    unsigned char* crops; const unsigned char* par; int c_par; const unsigned char* ext; int e_off; unsigned char* feat; int c_feat;
};
__global__ __launch_bounds__(1024) void joint_step_kernel(const JointStep a) {
    const int tid = threadIdx.x;
    const int nprev = a.state[0], C = a.state[1], minmax = a.state[2];
    const int64_t p0 = *a.pos;
    for (int i = tid; i < nprev * C; i += 1024) {
        const int p = i / C, c = i - p * C;
        st_any(a.y_rows, a.prev_centre[p] * a.M + a.channels[c], a.dtype, (float)(a.sym[i] - minmax));
    }
    __threadfence();
    __syncthreads();
    const int rch = a.M * a.es / 16;                                  // 16-byte chunks per map row
    for (int i = tid; i < a.P * 25 * rch; i += 1024) {
        const int r = i / rch, c = i - r * rch, p = r / 25, t = r - p * 25;
        const int64_t src = a.all_centre[p0 + p] + (t / 5 - 2) * a.Wp + (t % 5 - 2);
        ((u32x4*)a.crops)[i] = ((const u32x4*)(a.y_rows + src * a.M * a.es))[c];
    }
    const int pch = a.c_par * a.es / 16, fch = a.c_feat * a.es / 16;
    for (int i = tid; i < a.P * pch; i += 1024) {
        const int p = i / pch, c = i - p * pch;
        ((u32x4*)a.feat)[p * fch + c] = ((const u32x4*)(a.par + a.all_rows[p0 + p] * a.c_par * a.es))[c];
    }
    if (a.ext)
        for (int i = tid; i < a.P * rch; i += 1024) {
            const int p = i / rch, c = i - p * rch;
            ((u32x4*)a.feat)[p * fch + a.e_off * a.es / 16 + c] = ((const u32x4*)(a.ext + a.all_rows[p0 + p] * a.M * a.es))[c];
        }
    __syncthreads();                                                  // every thread has read the previous group's rows and *pos
    if (tid < a.P) a.prev_centre[tid] = a.all_centre[p0 + tid];
    if (tid == 0) { *a.pos = p0 + a.P; a.state[0] = a.P; }
}
}  // namespace

extern "C" int hesic_joint_step(void* y_rows, int dtype, int M, int Wp, const int32_t* sym, int64_t* prev_centre, int32_t* state,
                                const int32_t* channels, const int64_t* all_centre, const int64_t* all_rows, int64_t* pos, int P, void* crops,
                                const void* par, int c_par, const void* ext, int e_off, void* feat, int c_feat, void* stream) {
    HESIC_CHECK_ARG(y_rows && sym && prev_centre && state && channels && pos && M > 0 && P >= 0 && P <= 1024, "joint_step: bad arguments");
    HESIC_CHECK_ARG(dtype == HESIC_H16 || dtype == HESIC_F32, "joint_step: bad dtype");
    const int es = dtype == HESIC_H16 ? 2 : 4;
    HESIC_CHECK_ARG(P == 0 || (all_centre && all_rows && crops && par && feat && (M * es) % 16 == 0 && (c_par * es) % 16 == 0 && (c_feat * es) % 16 == 0 &&
                               (e_off * es) % 16 == 0 && c_par <= c_feat && (!ext || e_off + M <= c_feat)),
                    "joint_step: rows must be whole 16-byte chunks and the feature slices must fit");
    JointStep a;
    a.y_rows = (unsigned char*)y_rows; a.es = es; a.M = M; a.Wp = Wp; a.sym = sym; a.prev_centre = prev_centre; a.state = state; a.channels = channels;
    a.all_centre = all_centre; a.all_rows = all_rows; a.pos = pos; a.P = P; a.dtype = dtype; a.crops = (unsigned char*)crops;
    a.par = (const unsigned char*)par; a.c_par = c_par; a.ext = (const unsigned char*)ext; a.e_off = e_off; a.feat = (unsigned char*)feat; a.c_feat = c_feat;
    hipLaunchKernelGGL(joint_step_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    HESIC_LAUNCH_RETURN("joint_step");
}

// Plain copies / a stream wait for host loops that sit between launches (the HESIC+ wavefront decode makes ~750 of them per pair): the
// torch route costs ~10 us of dispatcher per call.  kind: 1 = host -> device, 2 = device -> host (pinned host memory for async behaviour).
extern "C" int hesic_memcpy_async(void* dst, const void* src, size_t bytes, int kind, void* stream) {
    HESIC_CHECK_ARG(dst && src && (kind == 1 || kind == 2), "memcpy_async: bad arguments");
    if (bytes == 0) return 0;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) { hesic_set_error("memcpy_async: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
// A recorded launch of one of the entry points a HESIC+ group step is made of, replayed by address: the arguments as 64-bit words
// (pointers and integers only; the trailing stream argument is supplied at replay).
static int run_tape(const hesic_tape_call* t, int n, void* stream) {
    for (int i = 0; i < n; ++i) {
        const uint64_t* a = t[i].a;
        int rc;
        // argument counts of the three entry points (stream left out): a tape packed against another revision of a signature is refused
        const int want = t[i].fn == HESIC_TAPE_JOINT_STEP ? 19 : t[i].fn == HESIC_TAPE_CONV2D_FORWARD ? 5 : t[i].fn == HESIC_TAPE_CONV2D_FORWARD_F32OUT ? 10 : -1;
        if (want >= 0 && t[i].nargs != want) {
            hesic_set_error("joint_decode_groups: tape entry %d (entry point %d) carries %d arguments, %d expected", i, t[i].fn, t[i].nargs, want);
            return HESIC_EINVAL;
        }
        switch (t[i].fn) {
        case HESIC_TAPE_JOINT_STEP:
            rc = hesic_joint_step((void*)a[0], (int)a[1], (int)a[2], (int)a[3], (const int32_t*)a[4], (int64_t*)a[5], (int32_t*)a[6], (const int32_t*)a[7],
                                  (const int64_t*)a[8], (const int64_t*)a[9], (int64_t*)a[10], (int)a[11], (void*)a[12], (const void*)a[13], (int)a[14],
                                  (const void*)a[15], (int)a[16], (void*)a[17], (int)a[18], stream);
            break;
        case HESIC_TAPE_CONV2D_FORWARD:
            rc = hesic_conv2d_forward((const hesic_conv_desc*)a[0], (const void*)a[1], (const void*)a[2], (const float*)a[3], (void*)a[4], stream);
            break;
        case HESIC_TAPE_CONV2D_FORWARD_F32OUT:
            rc = hesic_conv2d_forward_f32out((const hesic_conv_desc*)a[0], (const void*)a[1], (const void*)a[2], (const float*)a[3], (void*)a[4], (float*)a[5],
                                             (int)a[6], (int)a[7], (void*)a[8], (size_t)a[9], stream);
            break;
        default:
            hesic_set_error("joint_decode_groups: unknown entry point %d on the tape", t[i].fn);
            return HESIC_EINVAL;
        }
        if (rc) return rc;
    }
    return 0;
}

// The whole decode walk of one view behind one call (models.HSICJoint._decode_view_graphed): per group -- previous symbols up, the group's
// device step (a captured hipGraphExec_t of torch's, or the recorded launches of the step replayed one by one: back-to-back launches
// of a dependent chain of small kernels start closer together than the nodes of a graph), table launch, tables down, wait, range-decode
// through the caller's decoder (libhesic_host.so: hesic_rc_decoder_decode_grid) into the pinned symbol buffer.  Six Python -> C crossings
// per group became none; the wait polls hipStreamQuery (a blocked hipStreamSynchronize wakes up through an interrupt).
static int joint_decode_groups(int n_groups, const int32_t* group_size, void* const* graph_exec, const hesic_tape_call* const* tapes, const int32_t* tape_len,
                               const hesic_gmm_desc* descs, void* const* scale_mean, const int32_t* channels, int n_channels, int minmax,
                               uint32_t* tab_dev, uint32_t* tab_host, int32_t* sym_dev, int32_t* sym_host, hesic_decode_grid_fn decode, void* decoder,
                               int spin, void* stream) {
    HESIC_CHECK_ARG(n_groups >= 0 && group_size && (graph_exec || (tapes && tape_len)) && (scale_mean || !descs) && channels && n_channels > 0 && minmax >= 0 &&
                        tab_dev && tab_host && sym_dev && sym_host && decode && decoder,
                    "joint_decode_groups: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int n_tab = 2 * minmax + 2;
    int nprev = 0;
    auto fail = [](const char* what, hipError_t e) { hesic_set_error("joint_decode_groups: %s: %s", what, hipGetErrorString(e)); return (int)e; };
    static const bool timing = getenv("HESIC_JOINT_TIMING") != nullptr;      // stderr: where the wall time of the walk went
    double t_submit = 0, t_wait = 0, t_decode = 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int g = 0; g < n_groups; ++g) {
        const int P = group_size[g];
        hipError_t e;
        const double t0 = timing ? now() : 0;
        if (nprev && sym_dev != sym_host && (e = hipMemcpyAsync(sym_dev, sym_host, (size_t)nprev * n_channels * 4, hipMemcpyHostToDevice, st)) != hipSuccess)
            return fail("symbols up", e);
        if (graph_exec) {
            if ((e = hipGraphLaunch((hipGraphExec_t)graph_exec[g], st)) != hipSuccess) return fail("graph launch", e);
        } else if (int rc = run_tape(tapes[g], tape_len[g], stream)) {
            return rc;
        }
        if (descs)
            if (int rc = hesic_gmm_cdf_rows(&descs[g], 0, scale_mean[g], scale_mean[g], nullptr, channels, n_channels, minmax, 1, tab_dev, stream)) return rc;
        if (tab_dev != tab_host && (e = hipMemcpyAsync(tab_host, tab_dev, (size_t)n_channels * P * n_tab * 4, hipMemcpyDeviceToHost, st)) != hipSuccess)
            return fail("tables down", e);
        const double t1 = timing ? now() : 0;
        if (spin) {
            while ((e = hipStreamQuery(st)) == hipErrorNotReady) {}
        } else {
            e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess) return fail("wait", e);
        const double t2 = timing ? now() : 0;
        if (int rc = decode(decoder, tab_host, P, n_channels, n_channels, 1, n_tab, sym_host)) { hesic_set_error("joint_decode_groups: range decoder failed (%d) in group %d", rc, g); return HESIC_EINVAL; }
        if (timing) { const double t3 = now(); t_submit += t1 - t0; t_wait += t2 - t1; t_decode += t3 - t2; }
        nprev = P;
    }
    if (nprev && sym_dev != sym_host) {
        const hipError_t e = hipMemcpyAsync(sym_dev, sym_host, (size_t)nprev * n_channels * 4, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail("symbols up", e);
    }
    if (timing)
        fprintf(stderr, "joint_decode_groups: %d groups, %d channels, tables of %d: submit %.0f us, wait %.0f us, host decode %.0f us\n", n_groups, n_channels,
                n_tab, t_submit, t_wait, t_decode);
    return 0;
}

extern "C" int hesic_joint_decode_groups(int n_groups, const int32_t* group_size, void* const* graph_exec, const hesic_gmm_desc* descs,
                                         void* const* scale_mean, const int32_t* channels, int n_channels, int minmax, uint32_t* tab_dev,
                                         uint32_t* tab_host, int32_t* sym_dev, int32_t* sym_host, hesic_decode_grid_fn decode, void* decoder,
                                         int spin, void* stream) {
    HESIC_CHECK_ARG(graph_exec, "joint_decode_groups: no graphs");
    return joint_decode_groups(n_groups, group_size, graph_exec, nullptr, nullptr, descs, scale_mean, channels, n_channels, minmax, tab_dev, tab_host, sym_dev,
                               sym_host, decode, decoder, spin, stream);
}

extern "C" int hesic_joint_decode_groups_tape(int n_groups, const int32_t* group_size, const hesic_tape_call* const* tapes, const int32_t* tape_len,
                                              const hesic_gmm_desc* descs, void* const* scale_mean, const int32_t* channels, int n_channels, int minmax,
                                              uint32_t* tab_dev, uint32_t* tab_host, int32_t* sym_dev, int32_t* sym_host, hesic_decode_grid_fn decode,
                                              void* decoder, int spin, void* stream) {
    HESIC_CHECK_ARG(tapes && tape_len, "joint_decode_groups_tape: no tapes");
    return joint_decode_groups(n_groups, group_size, nullptr, tapes, tape_len, descs, scale_mean, channels, n_channels, minmax, tab_dev, tab_host, sym_dev,
                               sym_host, decode, decoder, spin, stream);
}

extern "C" int hesic_stream_synchronize(void* stream) {
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { hesic_set_error("stream_synchronize: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

// ---- measurement aid (bench.py `roofline.power_state`; no product path calls it): a register-only loop of independent
// v_mfma_f32_32x32x16 on 16-bit operands, every SIMD of the chip busy, no LDS and no memory traffic -- what the matrix pipe sustains under the
// board's power management on operands of the given kind.  `src` holds >= 64 KB of 16-bit values (random, or zeros); `sink` takes nothing unless
// the sums hit a magic value (keeps the loop alive).  2048 blocks x 4 waves x iters x 16 MFMAs of 32768 flops.
namespace {
__global__ __launch_bounds__(256, 2) void probe_mfma_loop_kernel(const h16x8* __restrict__ src, float* __restrict__ sink, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const h16x8 a0 = src[t & 4095], a1 = src[(t + 64) & 4095], b0 = src[(t + 128) & 4095], b1 = src[(t + 192) & 4095];
    f32x16 c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[u][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c[0] = mfma_32x32x16_h16(a0, b0, c[0], 0, 0, 0);
            c[1] = mfma_32x32x16_h16(a0, b1, c[1], 0, 0, 0);
            c[2] = mfma_32x32x16_h16(a1, b0, c[2], 0, 0, 0);
            c[3] = mfma_32x32x16_h16(a1, b1, c[3], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[u][r];
    if (s == 12345.678f) sink[t & 1023] = s;
}
}  // namespace

extern "C" int hesic_probe_mfma_loop(const void* src_64k, float* sink_4k, int iters, double* flops_out, void* stream) {
    HESIC_CHECK_ARG(src_64k && sink_4k && iters > 0, "probe_mfma_loop: null pointer or no iterations");
    constexpr int BLOCKS = 2048;
    hipLaunchKernelGGL(probe_mfma_loop_kernel, dim3(BLOCKS), dim3(256), 0, (hipStream_t)stream, (const h16x8*)src_64k, sink_4k, iters);
    if (flops_out) *flops_out = (double)BLOCKS * 4 * iters * 16 * 32768.0;
    HESIC_LAUNCH_RETURN("probe_mfma_loop");
}
