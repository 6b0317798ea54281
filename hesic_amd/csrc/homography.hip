// The step in front of the stereo path that produces h_matrix: the pieces of HomographyNet that
// are not convolutions -- MaxPool2d(2,2) of Block (ywz/mywork/model.py:62-63) -- and the corner-delta -> homography
// derivation of the `_real` scripts (newtrain1_real.py:113-123 with h_adjust :47-57; model.py:99-111):
//     corners0 = corners - corners[:,0]; dst = corners0 + delta
//     h = kornia.get_perspective_transform(corners0, dst)   (4-point DLT, h33 = 1)
//     h_matrix = h_adjust(H_img, W_img, pic, pic, torch.inverse(h))
// kornia is not vendored by the reference (unpinned): the DLT is restated from its published definition
// (8x8 linear system, rows [x y 1 0 0 0 -xu -yu | u], [0 0 0 x y 1 -xv -yv | v]).  Nine numbers per pair: one thread
// per pair, fp64 elimination with partial pivoting, fp32 in / out.
#include "common.h"

namespace {

template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    constexpr int V = 16 / sizeof(T);                 // channels per thread (one 16-byte access)
    const int Ho = H >> 1, Wo = W >> 1, cg = C / V;
    const int64_t total = (int64_t)B * Ho * Wo * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cg) * V;
        int64_t r = i / cg;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int b = r / Ho;
        const T* p = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
        float m[V];
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const u32x4 raw = *(const u32x4*)(p + ((int64_t)dy * W + dx) * C);
                const T* v = (const T*)&raw;
#pragma unroll
                for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], elem<T>::ld(v + e));
            }
        u32x4 out;
        T* o = (T*)&out;
#pragma unroll
        for (int e = 0; e < V; ++e) elem<T>::st(o + e, m[e]);
        *(u32x4*)(y + (((int64_t)b * Ho + oy) * Wo + ox) * C + c) = out;
    }
}

// solve the 4-point DLT for dst ~ H src; returns false for a singular configuration
__device__ bool dlt4(const double sx[4], const double sy[4], const double dx[4], const double dy[4], double h[9]) {
    double A[8][9];
    for (int i = 0; i < 4; ++i) {
        const double x = sx[i], y = sy[i], u = dx[i], v = dy[i];
        double* r0 = A[2 * i];
        double* r1 = A[2 * i + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; r1[8] = v;
    }
    for (int c = 0; c < 8; ++c) {
        int piv = c;
        for (int r = c + 1; r < 8; ++r)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (fabs(A[piv][c]) < 1e-300) return false;
        if (piv != c)
            for (int k = 0; k < 9; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        const double inv = 1.0 / A[c][c];
        for (int r = c + 1; r < 8; ++r) {
            const double f = A[r][c] * inv;
            for (int k = c; k < 9; ++k) A[r][k] -= f * A[c][k];
        }
    }
    for (int c = 7; c >= 0; --c) {
        double s = A[c][8];
        for (int k = c + 1; k < 8; ++k) s -= A[c][k] * h[k];
        h[c] = s / A[c][c];
    }
    h[8] = 1.0;
    return true;
}

__device__ void inv3(const double m[9], double o[9]) {
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double id = 1.0 / (m[0] * c0 + m[1] * c1 + m[2] * c2);
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__global__ void perspective_transform_kernel(const float* __restrict__ src, const float* __restrict__ dst, float* __restrict__ H, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double sx[4], sy[4], dx[4], dy[4], h[9];
    for (int i = 0; i < 4; ++i) {
        sx[i] = src[b * 8 + 2 * i]; sy[i] = src[b * 8 + 2 * i + 1];
        dx[i] = dst[b * 8 + 2 * i]; dy[i] = dst[b * 8 + 2 * i + 1];
    }
    if (!dlt4(sx, sy, dx, dy, h))
        for (int k = 0; k < 9; ++k) h[k] = NAN;
    for (int k = 0; k < 9; ++k) H[b * 9 + k] = (float)h[k];
}

__global__ void h_from_delta_kernel(const float* __restrict__ corners, const float* __restrict__ delta, float ra, float rb,
                                    int subtract_origin, float* __restrict__ H, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double sx[4], sy[4], dx[4], dy[4], h[9], hi[9];
    const float x0 = subtract_origin ? corners[b * 8] : 0.f, y0 = subtract_origin ? corners[b * 8 + 1] : 0.f;
    for (int i = 0; i < 4; ++i) {
        // fp32 like the reference's tensor arithmetic (corners - corners[:,0], + delta_hat)
        const float cx = corners[b * 8 + 2 * i] - x0, cy = corners[b * 8 + 2 * i + 1] - y0;
        sx[i] = cx; sy[i] = cy;
        dx[i] = cx + delta[b * 8 + 2 * i]; dy[i] = cy + delta[b * 8 + 2 * i + 1];
    }
    if (!dlt4(sx, sy, dx, dy, h)) {
        for (int k = 0; k < 9; ++k) H[b * 9 + k] = NAN;
        return;
    }
    inv3(h, hi);
    // h_adjust (newtrain1_real.py:47-57): row 0 *= a, column 0 /= a, row 1 *= b, column 1 /= b, in that order
    const double a = ra, bb = rb;
    for (int k = 0; k < 3; ++k) hi[k] *= a;
    for (int r = 0; r < 3; ++r) hi[3 * r] *= 1.0 / a;
    for (int k = 0; k < 3; ++k) hi[3 + k] *= bb;
    for (int r = 0; r < 3; ++r) hi[3 * r + 1] *= 1.0 / bb;
    for (int k = 0; k < 9; ++k) H[b * 9 + k] = (float)hi[k];
}

}  // namespace

extern "C" int hesic_maxpool2_forward(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    HESIC_CHECK_ARG(x && y && B > 0 && H > 1 && W > 1 && C > 0, "maxpool2_forward: bad arguments");
    HESIC_CHECK_ARG(dtype == HESIC_H16 || dtype == HESIC_F32, "maxpool2_forward: bad dtype");
    const int V = dtype == HESIC_H16 ? 8 : 4;
    HESIC_CHECK_ARG(C % V == 0, "maxpool2_forward: C=%d must be a multiple of %d", C, V);
    const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / V);
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(maxpool2_kernel<h16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const h16_t*)x, (h16_t*)y, B, H, W, C);
    else
        hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, H, W, C);
    HESIC_LAUNCH_RETURN("maxpool2_forward");
}

extern "C" int hesic_perspective_transform(const float* src, const float* dst, float* H, int B, void* stream) {
    HESIC_CHECK_ARG(src && dst && H && B > 0, "perspective_transform: bad arguments");
    hipLaunchKernelGGL(perspective_transform_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, src, dst, H, B);
    HESIC_LAUNCH_RETURN("perspective_transform");
}

extern "C" int hesic_h_from_delta(const float* corners, const float* delta, float ratio_a, float ratio_b, int subtract_origin,
                                  float* H, int B, void* stream) {
    HESIC_CHECK_ARG(corners && delta && H && B > 0 && ratio_a > 0.f && ratio_b > 0.f, "h_from_delta: bad arguments");
    hipLaunchKernelGGL(h_from_delta_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, corners, delta, ratio_a, ratio_b, subtract_origin, H, B);
    HESIC_LAUNCH_RETURN("h_from_delta");
}
