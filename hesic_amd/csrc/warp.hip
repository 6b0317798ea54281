// warp_perspective: projective bilinear resample with zero padding, the build's replacement for
// kornia.warp_perspective(src, M, dsize) (third party; call sites ywz/mywork/newnet1.py:746,753,767).
//
// One thread per destination pixel (grid.y = image): M is inverted once per block (3x3, fp64), (x',y') mapped back to the
// source, gather the 4 neighbours of every channel.  Neighbouring lanes hit neighbouring source pixels,
// so the gather is wavefront-coalesced; the kernel moves 2*C*H*W elements and is HBM / latency bound.
#include "common.h"

namespace {

struct WArgs {
    hesic_warp_desc d;
    const void* src; const float* M; void* dst; float* dsrc;
};

// inverse of image b's 3x3 (fp64, adjugate / det) -- once per block, shared through LDS
__device__ __forceinline__ void invert_h(const float* M, int b, double inv9[9], int already_inverse = 0) {
    const float* m = M + b * 9;
    if (already_inverse) {          // the caller's matrix maps destination -> source pixels (warp by H^-1 given H)
#pragma unroll
        for (int j = 0; j < 9; ++j) inv9[j] = m[j];
        return;
    }
    const double a = m[0], bb = m[1], c = m[2], dd = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double A = e * i - f * h, B = -(dd * i - f * g), Cc = dd * h - e * g;
    const double det = a * A + bb * B + c * Cc;
    const double inv = 1.0 / det;
    inv9[0] = A * inv; inv9[1] = -(bb * i - c * h) * inv; inv9[2] = (bb * f - c * e) * inv;
    inv9[3] = B * inv; inv9[4] = (a * i - c * g) * inv; inv9[5] = -(a * f - c * dd) * inv;
    inv9[6] = Cc * inv; inv9[7] = -(a * h - bb * g) * inv; inv9[8] = (a * e - bb * dd) * inv;
}

__device__ __forceinline__ bool src_coords(const hesic_warp_desc& d, const double* iv, int ox, int oy, float& sx, float& sy) {
    const double X = iv[0] * ox + iv[1] * oy + iv[2], Y = iv[3] * ox + iv[4] * oy + iv[5], Z = iv[6] * ox + iv[7] * oy + iv[8];
    const double rz = 1.0 / Z;                 // one fp64 division per pixel
    double x = X * rz, y = Y * rz;
    if (!d.align_corners) {   // kornia <= 0.4: normalised with (W-1), sampled with align_corners=False
        x = x * d.W / (double)(d.W - 1) - 0.5;
        y = y * d.H / (double)(d.H - 1) - 0.5;
    }
    sx = (float)x; sy = (float)y;
    return isfinite(sx) && isfinite(sy);
}

__global__ void warp_fwd_kernel(const WArgs a) {
    const hesic_warp_desc& d = a.d;
    __shared__ double iv[9];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) invert_h(a.M, b, iv, a.d.m_is_dst_to_src);
    __syncthreads();
    const int64_t total = (int64_t)d.Ho * d.Wo;
    for (int64_t i = xcd_remap(blockIdx.x, gridDim.x) * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = i % d.Wo, oy = i / d.Wo;
        float sx, sy;
        const bool fin = src_coords(d, iv, ox, oy, sx, sy);
        const float fx0 = floorf(sx), fy0 = floorf(sy);
        const float wx1 = sx - fx0, wy1 = sy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool big = !fin || fabsf(fx0) > 1e8f || fabsf(fy0) > 1e8f;
        const int x0 = big ? -10 : (int)fx0, y0 = big ? -10 : (int)fy0;
        const bool vx0 = x0 >= 0 && x0 < d.W, vx1 = x0 + 1 >= 0 && x0 + 1 < d.W;
        const bool vy0 = y0 >= 0 && y0 < d.H, vy1 = y0 + 1 >= 0 && y0 + 1 < d.H;
        const int64_t sb = b * d.ss_b + y0 * d.ss_y + x0 * d.ss_x;
        const int64_t db = b * d.ds_b + oy * d.ds_y + ox * d.ds_x;
        for (int c = 0; c < d.C; ++c) {
            const int64_t s = sb + c * d.ss_c;
            float v = 0.f;
            if (vy0 && vx0) v += ld_any(a.src, s, d.src_dtype) * (wx0 * wy0);
            if (vy0 && vx1) v += ld_any(a.src, s + d.ss_x, d.src_dtype) * (wx1 * wy0);
            if (vy1 && vx0) v += ld_any(a.src, s + d.ss_y, d.src_dtype) * (wx0 * wy1);
            if (vy1 && vx1) v += ld_any(a.src, s + d.ss_y + d.ss_x, d.src_dtype) * (wx1 * wy1);
            st_any(a.dst, db + c * d.ds_c, d.dst_dtype, v);
        }
    }
}

// Same result, bit for bit, for the layout the path feeds it (fp32 in and out, unit pixel stride, C == 3): the twelve taps
// of a pixel are branch-free buffer loads (an out-of-image tap is a poisoned offset that reads 0, its weight is forced to
// 0 as well so non-finite coordinates stay silent) and all in flight together; 32-bit indexing, one row of output per
// block row so the pixel index needs no division.  The generic kernel took a branch and a 64-bit address per tap.
constexpr int WARP_ROWS = 4;     // output rows per block: one 3x3 inversion and one block start-up per 4 x 256 pixels, 48 taps in flight per lane
__global__ __launch_bounds__(256) void warp_fwd_f32c3_kernel(const WArgs a) {
    const hesic_warp_desc& d = a.d;
    __shared__ double iv[9];
    const int b = blockIdx.z;
    if (threadIdx.x == 0) invert_h(a.M, b, iv, a.d.m_is_dst_to_src);
    __syncthreads();
    const int ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= d.Wo) return;
    constexpr uint32_t POISON = 0x80000000u;
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)a.src + (int64_t)b * d.ss_b), 0, (int)POISON, 0x00020000);
    const int sy_ = (int)d.ss_y, sc_ = (int)d.ss_c;
    float t[WARP_ROWS][3][4], wt[WARP_ROWS][4];
    bool ok[WARP_ROWS][4];
#pragma unroll
    for (int r = 0; r < WARP_ROWS; ++r) {
        const int oy = blockIdx.y * WARP_ROWS + r;
        float sx, sy;
        const bool fin = src_coords(d, iv, ox, oy, sx, sy) && oy < d.Ho;
        const float fx0 = floorf(sx), fy0 = floorf(sy);
        const float wx1 = sx - fx0, wy1 = sy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool big = !fin || fabsf(fx0) > 1e8f || fabsf(fy0) > 1e8f;
        const int x0 = big ? -10 : (int)fx0, y0 = big ? -10 : (int)fy0;
        const bool vx0 = x0 >= 0 && x0 < d.W, vx1 = x0 + 1 >= 0 && x0 + 1 < d.W;
        const bool vy0 = y0 >= 0 && y0 < d.H, vy1 = y0 + 1 >= 0 && y0 + 1 < d.H;
        const uint32_t o00 = (uint32_t)((y0 * sy_ + x0) * 4);
        ok[r][0] = vy0 && vx0; ok[r][1] = vy0 && vx1; ok[r][2] = vy1 && vx0; ok[r][3] = vy1 && vx1;
        const uint32_t off[4] = {ok[r][0] ? o00 : POISON, ok[r][1] ? o00 + 4u : POISON, ok[r][2] ? o00 + (uint32_t)(sy_ * 4) : POISON,
                                 ok[r][3] ? o00 + (uint32_t)(sy_ * 4) + 4u : POISON};
        wt[r][0] = ok[r][0] ? wx0 * wy0 : 0.f; wt[r][1] = ok[r][1] ? wx1 * wy0 : 0.f;
        wt[r][2] = ok[r][2] ? wx0 * wy1 : 0.f; wt[r][3] = ok[r][3] ? wx1 * wy1 : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[r][c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, (int)off[k], c * sc_ * 4, 0));
    }
#pragma unroll
    for (int r = 0; r < WARP_ROWS; ++r) {
        const int oy = blockIdx.y * WARP_ROWS + r;
        if (oy >= d.Ho) break;
        float* dp = (float*)a.dst + (int64_t)b * d.ds_b + (int64_t)oy * d.ds_y + ox;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[r][k]) v += t[r][c][k] * wt[r][k];           // the generic kernel skips invalid taps: keep its sums (no 0 * x terms)
            dp[(int64_t)c * d.ds_c] = v;
        }
    }
}

// (Round 5 built the variant asked for twice -- a lane owns FOUR consecutive output pixels of two rows, 16-byte stores: bit-identical and 2x SLOWER,
// 25.1 vs 11.7 us back to back, the gathers lose their coalescing.  Removed from the library in round 6: profiles/experiments/r05_warp_v4_four_pixels_per_lane.patch.)

// transpose of the gather: scatter-add of the same four weights into d_src (fp32)
__global__ void warp_bwd_kernel(const WArgs a) {
    const hesic_warp_desc& d = a.d;
    __shared__ double iv[9];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) invert_h(a.M, b, iv, a.d.m_is_dst_to_src);
    __syncthreads();
    const int64_t total = (int64_t)d.Ho * d.Wo;
    for (int64_t i = xcd_remap(blockIdx.x, gridDim.x) * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = i % d.Wo, oy = i / d.Wo;
        float sx, sy;
        const bool fin = src_coords(d, iv, ox, oy, sx, sy);
        const float fx0 = floorf(sx), fy0 = floorf(sy);
        const float wx1 = sx - fx0, wy1 = sy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool big = !fin || fabsf(fx0) > 1e8f || fabsf(fy0) > 1e8f;
        const int x0 = big ? -10 : (int)fx0, y0 = big ? -10 : (int)fy0;
        const bool vx0 = x0 >= 0 && x0 < d.W, vx1 = x0 + 1 >= 0 && x0 + 1 < d.W;
        const bool vy0 = y0 >= 0 && y0 < d.H, vy1 = y0 + 1 >= 0 && y0 + 1 < d.H;
        const int64_t sb = b * d.ss_b + y0 * d.ss_y + x0 * d.ss_x;
        const int64_t db = b * d.ds_b + oy * d.ds_y + ox * d.ds_x;
        for (int c = 0; c < d.C; ++c) {
            const float g = ld_any(a.dst, db + c * d.ds_c, d.dst_dtype);
            const int64_t s = sb + c * d.ss_c;
            if (vy0 && vx0) atomicAdd(a.dsrc + s, g * (wx0 * wy0));
            if (vy0 && vx1) atomicAdd(a.dsrc + s + d.ss_x, g * (wx1 * wy0));
            if (vy1 && vx0) atomicAdd(a.dsrc + s + d.ss_y, g * (wx0 * wy1));
            if (vy1 && vx1) atomicAdd(a.dsrc + s + d.ss_y + d.ss_x, g * (wx1 * wy1));
        }
    }
}

int check(const hesic_warp_desc* d, const char* who) {
    HESIC_CHECK_ARG(d && d->B > 0 && d->C > 0 && d->H > 1 && d->W > 1 && d->Ho > 0 && d->Wo > 0, "%s: bad geometry", who);
    return 0;
}

}  // namespace

extern "C" int hesic_warp_perspective_forward(const hesic_warp_desc* d, const void* src, const float* M, void* dst,
                                              void* stream) {
    if (int e = check(d, "warp_perspective_forward")) return e;
    HESIC_CHECK_ARG(src && M && dst, "warp_perspective_forward: null pointer");
    WArgs a; a.d = *d; a.src = src; a.M = M; a.dst = dst; a.dsrc = nullptr;
    const bool fast = d->C == 3 && d->src_dtype == HESIC_F32 && d->dst_dtype == HESIC_F32 && d->ss_x == 1 && d->ds_x == 1 &&
                      d->ss_y > 0 && d->ss_c > 0 && (2 * d->ss_c + (int64_t)(d->H + 1) * d->ss_y + d->W + 2) * 4 < (1ll << 31) &&
                      d->Ho < 65536 && d->B < 65536;
    if (fast)
        hipLaunchKernelGGL(warp_fwd_f32c3_kernel, dim3((d->Wo + 255) / 256, (d->Ho + WARP_ROWS - 1) / WARP_ROWS, d->B), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(warp_fwd_kernel, dim3(grid_for((int64_t)d->Ho * d->Wo, 256), d->B), dim3(256), 0, (hipStream_t)stream, a);
    HESIC_LAUNCH_RETURN("warp_perspective_forward");
}

extern "C" int hesic_warp_perspective_backward(const hesic_warp_desc* d, const void* d_dst, const float* M, float* d_src,
                                               void* stream) {
    if (int e = check(d, "warp_perspective_backward")) return e;
    HESIC_CHECK_ARG(d_dst && M && d_src, "warp_perspective_backward: null pointer");
    WArgs a; a.d = *d; a.src = nullptr; a.M = M; a.dst = (void*)d_dst; a.dsrc = d_src;
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(grid_for((int64_t)d->Ho * d->Wo, 256), d->B), dim3(256), 0, (hipStream_t)stream, a);
    HESIC_LAUNCH_RETURN("warp_perspective_backward");
}
