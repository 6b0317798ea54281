// 3x3 stride-1 convolutions of the stage-2 enhancement net (Independent_EN / Enhancement / ResidualBlock,
// ywz/mywork/newnet1.py:272-311, compressai/layers/layers.py:125-147): 32 channels at FULL image resolution, nineteen of
// them per view.  On the generic implicit-GEMM kernel a 32 -> 32 layer wastes half of every 64-cout MFMA tile and runs a
// 9-stage K loop per block (136 us), the residual adds are separate passes (61 us each) and the 32 -> 3 output conv fell to
// the scalar kernel (1.7 ms): 10.1 ms per forward for B=8 512x512, 4.6x the whole HESIC forward.
//
// This kernel is built for exactly that shape.  NHWC bf16 in, the 9 x 32 x 32 weights live in REGISTERS as 18 MFMA A
// fragments (32 couts x 16 channels each), a block stages the 18 x 34-pixel halo of a 16 x 32-pixel tile in LDS once and
// every wave walks four image rows of it: 18 ds_read_b128 + 18 v_mfma_f32_32x32x16_bf16 per 32 pixels, all 32 couts live.
// Epilogue: bias, LeakyReLU, up to two residual tensors (the block's identity and the Enhancement_Block's outer skip),
// bf16 NHWC out -- or, for the last layer (32 -> 3), fp32 planar out plus the fp32 planar residual image.
// HBM-bound by construction: 64 B in + 64 B out per pixel (134 + 134 MB per layer at B=8 512x512).
#include "common.h"

namespace {

struct C32Args {
    const bf16_t* x; const float* w; const float* bias; const void* res1; const void* res2; void* y;
    int B, H, W, Cout, act, tiles_x, tiles_y;
    FastDiv fd_tx, fd_ty;
};

constexpr int TH = 16, TW = 32, HW_ = TW + 2, HH = TH + 2, HPIX = HH * HW_;     // halo: 18 x 34 pixels of 64 bytes

template <int MODE>      // 0: 32 couts, bf16 NHWC out (+ bf16 NHWC residuals); 1: <= 4 couts, fp32 planar out (+ fp32 planar residual)
__global__ __launch_bounds__(256, 2) void c32_conv3x3_kernel(const C32Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char halo[HPIX * 64];
    constexpr uint32_t POISON = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    // weights -> 18 A fragments: lane (cout = p, half h) holds channels k*16 + h*8 + [0,8) of tap t
    bf16x8 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = p < a.Cout ? a.w[((int64_t)p * 32 + k * 16 + h * 8 + e) * 9 + t] : 0.f;
            wf[t][k] = __builtin_bit_cast(bf16x8, u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])});
        }
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 8 * j + 4 * h + e;
            bv[j][e] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;
        }
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t q = fdiv((uint32_t)tile, a.fd_tx);
        const int tx = tile - (int)q * a.tiles_x;
        const int b = (int)fdiv(q, a.fd_ty);
        const int ty = (int)q - b * a.tiles_y;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        // halo -> LDS: 16-byte piece i = (pixel i >> 2, slot i & 3) holds channel chunk slot ^ ((pixel >> 2) & 3); pixels outside
        // the image are poisoned offsets (zeros = the conv's zero padding)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)b * a.H * a.W * 32), 0, (int)POISON, 0x00020000);
        u32x4 pc[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int i = tid + 256 * u;
            const int hp = i >> 2, slot = i & 3;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = hp < HPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int chunk = slot ^ ((hp >> 2) & 3);
            pc[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (int)(((iy * a.W + ix) * 32 + chunk * 8) * 2) : (int)POISON, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int i = tid + 256 * u;
            if (i < HPIX * 4) *(u32x4*)(halo + i * 16) = pc[u];
        }
        __syncthreads();
#pragma unroll 1
        for (int rq = 0; rq < TH / 4; ++rq) {
            const int yl = wave + 4 * rq;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hp = (yl + t / 3) * HW_ + p + t % 3;
                const unsigned char* row = halo + hp * 64;
                const int sw = (hp >> 2) & 3;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const bf16x8 xf = *(const bf16x8*)(row + (((k * 2 + h) ^ sw) << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][k], xf, acc, 0, 0, 0);
                }
            }
            // D[cout][pixel]: lane holds pixel p, couts 8 j + 4 h + e (j = r >> 2, e = r & 3)
            const int y = ty * TH + yl, x = tx * TW + p;
            if (y < a.H && x < a.W) {
                if constexpr (MODE == 0) {
                    const int64_t pix = (((int64_t)b * a.H + y) * a.W + x) * 32;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c0 = 8 * j + 4 * h;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[4 * j + e] + bv[j][e], a.act);
                        if (a.res1) {
                            const u32x2 r = *(const u32x2*)((const bf16_t*)a.res1 + pix + c0);
                            v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                            v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                        }
                        if (a.res2) {
                            const u32x2 r = *(const u32x2*)((const bf16_t*)a.res2 + pix + c0);
                            v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                            v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                        }
                        *(u32x2*)((bf16_t*)a.y + pix + c0) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    }
                } else {
                    if (h == 0) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < a.Cout) {
                                const int64_t o = (((int64_t)b * a.Cout + c) * a.H + y) * a.W + x;
                                float v = apply_act(acc[c] + bv[0][c], a.act);
                                if (a.res1) v += ((const float*)a.res1)[o];
                                ((float*)a.y)[o] = v;
                            }
                    }
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int hesic_conv3x3_c32_forward(const void* x, const float* w, const float* bias, int Cout, int act, const void* res1,
                                         const void* res2, void* y, int B, int H, int W, void* stream) {
    HESIC_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0, "conv3x3_c32_forward: bad arguments");
    HESIC_CHECK_ARG(Cout == 32 || (Cout >= 1 && Cout <= 4), "conv3x3_c32_forward: Cout must be 32 (bf16 NHWC out) or <= 4 (fp32 planar out)");
    HESIC_CHECK_ARG(Cout == 32 || !res2, "conv3x3_c32_forward: the planar form takes one residual");
    HESIC_CHECK_ARG((int64_t)H * W * 64 < (1ll << 31), "conv3x3_c32_forward: image too large for 32-bit offsets");
    C32Args a;
    a.x = (const bf16_t*)x; a.w = w; a.bias = bias; a.res1 = res1; a.res2 = res2; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
    a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
    a.fd_tx = make_fastdiv((uint32_t)a.tiles_x); a.fd_ty = make_fastdiv((uint32_t)a.tiles_y);
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * B;
    HESIC_CHECK_ARG(ntiles < (1ll << 31), "conv3x3_c32_forward: too many tiles");
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);          // persistent: two blocks per CU, weights packed once per block
    if (Cout == 32) hipLaunchKernelGGL(c32_conv3x3_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(c32_conv3x3_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    HESIC_LAUNCH_RETURN("conv3x3_c32_forward");
}
