// Convolutions whose image side has only a few channels (3 or 6): g_a_conv1 (3->N, newnet1.py:583),
// pre_conv (6->3, :629), g_s_conv4 (N->3 transposed, :612), after_conv (6->3 transposed s1, :670).
// They are HBM / VALU bound (SURVEY.md 7 "hard parts"), so they run on the vector ALU with the
// weights broadcast from LDS; the image side is addressed through explicit element strides so NCHW
// planar fp32 images are consumed / produced without a layout copy.
//
//   narrow_to_wide : Conv2d,           Cin <= 8, Cout % 32 == 0, y channels contiguous (NHWC)
//   wide_to_narrow : ConvTranspose2d,  5x5 s2,   Cout <= 4, Cin % 8 == 0, x channels contiguous (NHWC)
//   generic        : anything else (also the fallback used by the odd-shape parity tests)
#include <stdlib.h>

#include "common.h"

namespace {

struct SArgs {
    const void* x; const float* w; const float* bias; void* y;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed, x_dtype, y_dtype, act;
    int64_t xs_b, xs_c, xs_y, xs_x, ys_b, ys_c, ys_y, ys_x;
    // optional second input (hesic_sconv2d_forward_cat): channels [c_split, Cin) come from x2 -- the torch.cat in front of
    // pre_conv / after_conv (newnet1.py:643,686) never materialises
    const void* x2; int64_t x2s_b, x2s_c, x2s_y, x2s_x; int x2_dtype, c_split;
    // optional 3-channel (I)GDN fused into the 6 -> 3 stages (hesic_sconv2d_forward_cat_gdn): gdn_mode 1 = GDN / IGDN (gdn_inverse) on the
    // three OUTPUT channels (pre_conv -> GDN(3), newnet1.py:643-644), 2 = on the first three INPUT channels before the conv
    // (IGDN(3) -> cat -> after_conv, newnet1.py:684-686); raw beta / gamma, reparametrised here like gdn_planar_kernel does
    const float* gdn_beta; const float* gdn_gamma; float gdn_bound; int gdn_mode, gdn_inverse;
    const void* w_img;   // optional: the kernel's LDS weight image, pre-packed once per weight update (hesic_sconv_pack_weight_image)
    int dbg;     // HESIC_N2W_DBG profiling ablations of the fused 3 -> 128 kernel (garbage results): 1 no loads, 2 no conv MFMAs, 4 no GDN MFMAs, 8 no global stores
};

// weight element for (co, ci, ky, kx) in either PyTorch layout
__device__ __forceinline__ float w_at(const float* w, int co, int ci, int ky, int kx, int Cout, int Cin, int KH, int KW,
                                      int transposed) {
    return transposed ? w[(((int64_t)ci * Cout + co) * KH + ky) * KW + kx] : w[(((int64_t)co * Cin + ci) * KH + ky) * KW + kx];
}

// ---------------------------------------------------------------- generic: one thread = one pixel x 4 couts
__global__ void sconv_generic_kernel(const SArgs a) {
    const int cog = (a.Cout + 3) / 4;
    const int64_t total = (int64_t)a.B * a.Ho * a.Wo * cog;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int ox = r % a.Wo; r /= a.Wo;
        const int oy = r % a.Ho; r /= a.Ho;
        const int g = r % cog;
        const int b = r / cog;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < a.KH; ++ky) {
            int iy;
            if (!a.transposed) {
                iy = oy * a.stride - a.pad + ky;
            } else {
                const int t = oy + a.pad - ky;
                if (t % a.stride) continue;
                iy = t / a.stride;
            }
            if (iy < 0 || iy >= a.H) continue;
            for (int kx = 0; kx < a.KW; ++kx) {
                int ix;
                if (!a.transposed) {
                    ix = ox * a.stride - a.pad + kx;
                } else {
                    const int t = ox + a.pad - kx;
                    if (t % a.stride) continue;
                    ix = t / a.stride;
                }
                if (ix < 0 || ix >= a.W) continue;
                const int64_t xb = b * a.xs_b + iy * a.xs_y + ix * a.xs_x;
                for (int ci = 0; ci < a.Cin; ++ci) {
                    const float xv = ld_any(a.x, xb + ci * a.xs_c, a.x_dtype);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = g * 4 + e;
                        if (co < a.Cout) acc[e] += xv * w_at(a.w, co, ci, ky, kx, a.Cout, a.Cin, a.KH, a.KW, a.transposed);
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = g * 4 + e;
            if (co < a.Cout) {
                const float v = apply_act(acc[e] + (a.bias ? a.bias[co] : 0.f), a.act);
                st_any(a.y, b * a.ys_b + co * a.ys_c + oy * a.ys_y + ox * a.ys_x, a.y_dtype, v);
            }
        }
    }
}

// ---------------------------------------------------------------- narrow -> wide (conv1 class)
// block: 64 output pixels (8x8) x Cout; thread = pixel (tid%64) x 32-cout group (tid/64 + 4*pass).
// weights live in LDS as [tap*Cin+ci][Cout] fp32: a wave shares its cout group -> broadcast reads.
constexpr int NW_MAXK = 8 * 25;
__global__ __launch_bounds__(256) void sconv_narrow_to_wide_kernel(const SArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [KK][Cout]
    const int KK = a.KH * a.KW * a.Cin;
    for (int i = threadIdx.x; i < KK * a.Cout; i += 256) {
        const int co = i % a.Cout, k = i / a.Cout;
        const int ci = k % a.Cin, tap = k / a.Cin;
        wl[i] = w_at(a.w, co, ci, tap / a.KW, tap % a.KW, a.Cout, a.Cin, a.KH, a.KW, a.transposed);
    }
    __syncthreads();
    const int tiles_x = (a.Wo + 7) / 8, tiles_y = (a.Ho + 7) / 8;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int p = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int oy = ty * 8 + (p >> 3), ox = tx * 8 + (p & 7);
    const bool ok = oy < a.Ho && ox < a.Wo;
    // gather the receptive field once (<= 200 values would not fit registers: keep per-row reuse instead)
    for (int cg = grp; cg * 32 < a.Cout; cg += 4) {
        float acc[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) acc[e] = 0.f;
        if (ok) {
            for (int ky = 0; ky < a.KH; ++ky) {
                const int iy = oy * a.stride - a.pad + ky;
                if (iy < 0 || iy >= a.H) continue;
                for (int kx = 0; kx < a.KW; ++kx) {
                    const int ix = ox * a.stride - a.pad + kx;
                    if (ix < 0 || ix >= a.W) continue;
                    const int64_t xb = b * a.xs_b + iy * a.xs_y + ix * a.xs_x;
                    for (int ci = 0; ci < a.Cin; ++ci) {
                        const float xv = ld_any(a.x, xb + ci * a.xs_c, a.x_dtype);
                        const float* wr = wl + ((ky * a.KW + kx) * a.Cin + ci) * a.Cout + cg * 32;
#pragma unroll
                        for (int e = 0; e < 32; e += 4) {
                            const f32x4 wv = *(const f32x4*)(wr + e);
                            acc[e] += xv * wv.x; acc[e + 1] += xv * wv.y; acc[e + 2] += xv * wv.z; acc[e + 3] += xv * wv.w;
                        }
                    }
                }
            }
            const int64_t yb = b * a.ys_b + oy * a.ys_y + ox * a.ys_x + cg * 32;   // ys_c == 1
            if (a.y_dtype == HESIC_H16) {
                h16_t* yp = (h16_t*)a.y + yb;
#pragma unroll
                for (int e = 0; e < 32; e += 8) {
                    u32x4 o;
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = apply_act(acc[e + k] + (a.bias ? a.bias[cg * 32 + e + k] : 0.f), a.act);
                    o.x = pack_h2(v[0], v[1]); o.y = pack_h2(v[2], v[3]); o.z = pack_h2(v[4], v[5]); o.w = pack_h2(v[6], v[7]);
                    *(u32x4*)(yp + e) = o;
                }
            } else {
                float* yp = (float*)a.y + yb;
#pragma unroll
                for (int e = 0; e < 32; e += 4) {
                    f32x4 o;
                    o.x = apply_act(acc[e] + (a.bias ? a.bias[cg * 32 + e] : 0.f), a.act);
                    o.y = apply_act(acc[e + 1] + (a.bias ? a.bias[cg * 32 + e + 1] : 0.f), a.act);
                    o.z = apply_act(acc[e + 2] + (a.bias ? a.bias[cg * 32 + e + 2] : 0.f), a.act);
                    o.w = apply_act(acc[e + 3] + (a.bias ? a.bias[cg * 32 + e + 3] : 0.f), a.act);
                    *(f32x4*)(yp + e) = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- wide -> narrow (g_s_conv4 class)
// ConvTranspose2d 5x5 s2 p2 op1, Cout <= 4.  thread = one input-grid cell q -> its 2x2 output quad.
// out(2q+r) gets x[q + d] * w[k] with k = r + 2 - 2d, d in {-1,0,1} (k in range).  Weights in LDS as
// [tap][ci][4] fp32 (zero padded couts), read as one broadcast float4 per (tap, ci).
template <typename T>
__global__ __launch_bounds__(256) void sconv_wide_to_narrow_kernel(const SArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [25][Cin][4]
    for (int i = threadIdx.x; i < 25 * a.Cin * 4; i += 256) {
        const int co = i & 3, ci = (i >> 2) % a.Cin, tap = (i >> 2) / a.Cin;
        wl[i] = co < a.Cout ? w_at(a.w, co, ci, tap / 5, tap % 5, a.Cout, a.Cin, 5, 5, 1) : 0.f;
    }
    __syncthreads();
    const int64_t total = (int64_t)a.B * a.H * a.W;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int qx = i % a.W, qy = (i / a.W) % a.H, b = i / ((int64_t)a.W * a.H);
    float acc[2][2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][s][c] = 0.f;
    const T* xg = (const T*)a.x;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = qy + dy;
        if (iy < 0 || iy >= a.H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int ix = qx + dx;
            if (ix < 0 || ix >= a.W) continue;
            const T* xp = xg + b * a.xs_b + iy * a.xs_y + ix * a.xs_x;   // xs_c == 1
            for (int c0 = 0; c0 < a.Cin; c0 += 8) {
                float xv[8];
                if constexpr (sizeof(T) == 2) {
                    const u32x4 raw = *(const u32x4*)(xp + c0);
                    xv[0] = h2f_lo(raw.x); xv[1] = h2f_hi(raw.x);
                    xv[2] = h2f_lo(raw.y); xv[3] = h2f_hi(raw.y);
                    xv[4] = h2f_lo(raw.z); xv[5] = h2f_hi(raw.z);
                    xv[6] = h2f_lo(raw.w); xv[7] = h2f_hi(raw.w);
                } else {
                    const f32x4 r0 = *(const f32x4*)(xp + c0), r1 = *(const f32x4*)(xp + c0 + 4);
                    xv[0] = r0.x; xv[1] = r0.y; xv[2] = r0.z; xv[3] = r0.w; xv[4] = r1.x; xv[5] = r1.y; xv[6] = r1.z; xv[7] = r1.w;
                }
#pragma unroll
                for (int ry = 0; ry < 2; ++ry) {
                    const int ky = ry + 2 - 2 * dy;
                    if (ky < 0 || ky > 4) continue;
#pragma unroll
                    for (int rx = 0; rx < 2; ++rx) {
                        const int kx = rx + 2 - 2 * dx;
                        if (kx < 0 || kx > 4) continue;
                        const float* wr = wl + ((ky * 5 + kx) * a.Cin + c0) * 4;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const f32x4 wv = *(const f32x4*)(wr + e * 4);
                            acc[ry][rx][0] += xv[e] * wv.x; acc[ry][rx][1] += xv[e] * wv.y;
                            acc[ry][rx][2] += xv[e] * wv.z; acc[ry][rx][3] += xv[e] * wv.w;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int ry = 0; ry < 2; ++ry)
#pragma unroll
        for (int rx = 0; rx < 2; ++rx)
            for (int co = 0; co < a.Cout; ++co) {
                const float v = apply_act(acc[ry][rx][co] + (a.bias ? a.bias[co] : 0.f), a.act);
                st_any(a.y, b * a.ys_b + co * a.ys_c + (2 * qy + ry) * a.ys_y + (2 * qx + rx) * a.ys_x, a.y_dtype, v);
            }
}

// ---------------------------------------------------------------- narrow -> wide on the matrix cores (bf16 output)
// g_a_conv1 (3 -> 128, 5x5 s2): K = Cin*KH*KW = 75 is re-ordered as (ci, ky) rows of 8 (5 real taps + 3 zeros) so that one
// lane-half of an MFMA B fragment is exactly one contiguous image row segment: K_pad = 16 rows * 8 = 128.
// A = weights [cout][K_pad] (bf16, LDS, loaded once per block), B = im2col fragment gathered straight from the image
// (any strides, fp32 or bf16), D -> bias/act -> bf16 -> LDS -> full NHWC rows.  HBM bound on the 128-channel output.
// Geometry is a template parameter so every loop unrolls and the index arithmetic folds (the first, fully run-time
// version spent 2250 VALU + 1570 SALU instructions per 32-pixel tile next to 32 MFMAs).
__device__ __forceinline__ uint32_t pack_h2_fast(float lo, float hi) { return pack_h2(lo, hi); }   // v_cvt_pk_bf16_f32

constexpr int N2W_KPAD = 128;
template <int CIN, int KS, int ST, typename XT>
__global__ __launch_bounds__(256) void sconv_n2w_mfma_kernel(const SArgs a) {
    constexpr int OROW = 128 * 2 + 16;
    constexpr int R = CIN * KS, PAD = KS / 2;
    static_assert(R <= 16 && KS <= 8, "K_pad = 128");
    __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 256 + 128 * OROW + 512];
    unsigned char* wl = smem;                    // [128 cout][128 k] bf16, 16-byte slots XOR (row & 15)
    unsigned char* os = smem + 128 * 256;        // output staging [128 px][OROW]
    float* bl = (float*)(smem + 128 * 256 + 128 * OROW);   // bias[128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + 7) / 8;
    const int64_t ntiles = (int64_t)tiles_x * tiles_y * a.B;
    const int frow = lane & 31, fh = lane >> 5;
    const XT* xg = (const XT*)a.x;

    for (int n0 = 0; n0 < a.Cout; n0 += 128) {
        __syncthreads();
        {   // weights: thread owns (ci,ky) row r = tid & 15 of couts tid>>4, +16, ...
            const int r = tid & 15, ci = r / KS, ky = r % KS;
            for (int co = tid >> 4; co < 128; co += 16) {
                float v[8];
#pragma unroll
                for (int kx = 0; kx < 8; ++kx)
                    v[kx] = (r < R && kx < KS && n0 + co < a.Cout) ? a.w[(((int64_t)(n0 + co) * CIN + ci) * KS + ky) * KS + kx] : 0.f;
                *(u32x4*)(wl + (co * 16 + (r ^ (co & 15))) * 16) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
            }
            if (tid < 128) bl[tid] = (a.bias && n0 + tid < a.Cout) ? a.bias[n0 + tid] : 0.f;
        }
        __syncthreads();
        for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / ((int64_t)tiles_x * tiles_y);
            const int pl = wave * 32 + frow;                 // pixel index in the 8x16 patch
            const int oy = ty * 8 + (pl >> 4), ox = tx * 16 + (pl & 15);
            const bool pok = oy < a.Ho && ox < a.Wo;
            const int ix0 = ox * ST - PAD;
            const XT* xb = xg + b * a.xs_b;
            // gather the whole im2col fragment first (all loads in flight together), then run the MFMAs
            u32x4 frag[N2W_KPAD / 16];
#pragma unroll
            for (int ks = 0; ks < N2W_KPAD / 16; ++ks) {
                // lane half fh owns (ci,ky) row r = 2*ks + fh; both candidates are compile-time constants
                const int r = 2 * ks + fh;
                const int ci = fh ? (2 * ks + 1) / KS : (2 * ks) / KS, ky = fh ? (2 * ks + 1) % KS : (2 * ks) % KS;
                const int iy = oy * ST - PAD + ky;
                const bool rok = r < R && pok && (unsigned)iy < (unsigned)a.H;
                const XT* rp = xb + ci * a.xs_c + (int64_t)iy * a.xs_y;
                float v[8];
#pragma unroll
                for (int kx = 0; kx < 8; ++kx) {
                    v[kx] = 0.f;
                    if (kx < KS) {
                        const bool ok = rok && (unsigned)(ix0 + kx) < (unsigned)a.W;
                        if (ok) v[kx] = elem<XT>::ld(rp + (int64_t)(ix0 + kx) * a.xs_x);
                    }
                }
                frag[ks] = u32x4{pack_h2_fast(v[0], v[1]), pack_h2_fast(v[2], v[3]), pack_h2_fast(v[4], v[5]), pack_h2_fast(v[6], v[7])};
            }
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < (R + 1) / 2; ++ks) {        // k-steps beyond the real rows are all zero
                const h16x8 xf = __builtin_bit_cast(h16x8, frag[ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 32 + frow;
                    const h16x8 wf = *(const h16x8*)(wl + (row * 16 + ((ks * 2 + fh) ^ (row & 15))) * 16);
                    acc[i] = mfma_32x32x16_h16(wf, xf, acc[i], 0, 0, 0);
                }
            }
            // D[i = cout][j = pixel]: lane holds pixel frow of this wave, couts i*32 + 8g + 4fh + {0..3}
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = i * 32 + 8 * g + 4 * fh;
                    const f32x4 bv = *(const f32x4*)(bl + cl);
                    const float o0 = apply_act(acc[i][4 * g] + bv.x, a.act), o1 = apply_act(acc[i][4 * g + 1] + bv.y, a.act);
                    const float o2 = apply_act(acc[i][4 * g + 2] + bv.z, a.act), o3 = apply_act(acc[i][4 * g + 3] + bv.w, a.act);
                    *(u32x2*)(os + pl * OROW + cl * 2) = u32x2{pack_h2_fast(o0, o1), pack_h2_fast(o2, o3)};
                }
            __syncthreads();
            h16_t* yg = (h16_t*)a.y;
#pragma unroll
            for (int c = tid; c < 128 * 16; c += 256) {
                const int pr = c >> 4, cc = c & 15;
                const int y2 = ty * 8 + (pr >> 4), x2 = tx * 16 + (pr & 15);
                if (y2 < a.Ho && x2 < a.Wo && n0 + cc * 8 < a.Cout)
                    *(u32x4*)(yg + b * a.ys_b + y2 * a.ys_y + x2 * a.ys_x + n0 + cc * 8) = *(const u32x4*)(os + pr * OROW + cc * 16);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- conv1 + GDN1 in one kernel (inference, bf16 output)
// g_a_gdn1(g_a_conv1(image)) (newnet1.py:594-595): the 3 -> 128 conv above followed, per pixel, by the 128x128 GDN
// contraction -- both on the matrix cores, the 128-channel activation never visits HBM in between.  512 threads = 8 waves,
// each wave owns a 32-pixel tile end to end (gather -> conv MFMAs -> bf16 rows in its private LDS slice -> squared rows
// back as the B operand of the GDN MFMAs -> normalise -> rows out), so the tile loop needs no block barrier at all.
template <int CIN, int KS, int ST, typename XT>
__global__ __launch_bounds__(512) void sconv_n2w_gdn_kernel(const SArgs a, const h16_t* __restrict__ gamma_packed,
                                                            const float* __restrict__ beta_packed, int inverse,
                                                            h16_t* __restrict__ y_pre) {
    constexpr int OROW = 128 * 2 + 16, R = CIN * KS, PAD = KS / 2, NW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                          // conv weights [128][128] bf16, slot ^ (row & 15)
    unsigned char* gl = smem + 32768;                  // gamma' image, same layout
    float* bl = (float*)(smem + 65536);                // conv bias[128], beta'[128]
    unsigned char* osb = smem + 65536 + 1024;          // per-wave 32 x OROW slices
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    unsigned char* os = osb + wave * 32 * OROW;
    const XT* xg = (const XT*)a.x;
    {
        const int r = tid & 15, ci = r / KS, ky = r % KS;
        for (int co = tid >> 4; co < 128; co += 32) {
            float v[8];
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
                v[kx] = (r < R && kx < KS && co < a.Cout) ? a.w[(((int64_t)co * CIN + ci) * KS + ky) * KS + kx] : 0.f;
            *(u32x4*)(wl + (co * 16 + (r ^ (co & 15))) * 16) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
        }
        // gamma' image for the GDN contraction.  Its K (input-channel) order is permuted so that the squares a lane needs as
        // MFMA B operand are the accumulator registers it already holds: slab ks = (i, gp) covers channels 32i+16gp+[0,16),
        // and 16-byte chunk 2ks+h of a row carries channels {32i+16gp+4h+[0,4)} ++ {32i+16gp+8+4h+[0,4)} -- exactly
        // acc[i][8gp .. 8gp+7] of lane-half h.  Source: the LDS-image half of hesic_gdn_pack_params (slot ^ (row & 15)).
        for (int idx = tid; idx < 2048; idx += 512) {
            const int row = idx >> 4, q = idx & 15, ks = q >> 1, h = q & 1;
            const unsigned char* srow = (const unsigned char*)gamma_packed + row * 256;
            const u32x2 lo = *(const u32x2*)(srow + (((2 * ks) ^ (row & 15)) << 4) + 8 * h);
            const u32x2 hi = *(const u32x2*)(srow + (((2 * ks + 1) ^ (row & 15)) << 4) + 8 * h);
            *(u32x4*)(gl + row * 256 + ((q ^ (row & 15)) << 4)) = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
        if (tid < 128) { bl[tid] = a.bias ? a.bias[tid] : 0.f; bl[128 + tid] = beta_packed[tid]; }
    }
    __syncthreads();
    const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + 1) / 2;       // wave tile = 2 output rows x 16 columns
    const int64_t ntiles = (int64_t)tiles_x * tiles_y * a.B;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);                    // neighbouring tiles (shared input rows) on one XCD
    for (int64_t tile = (int64_t)lb * NW + wave; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / ((int64_t)tiles_x * tiles_y);
        const int oy = ty * 2 + (frow >> 4), ox = tx * 16 + (frow & 15);
        const bool pok = oy < a.Ho && ox < a.Wo;
        const int ix0 = ox * ST - PAD;
        const XT* xb = xg + b * a.xs_b;
        u32x4 frag[(R + 1) / 2];
#pragma unroll
        for (int ks = 0; ks < (R + 1) / 2; ++ks) {
            const int r = 2 * ks + fh;
            const int ci = fh ? (2 * ks + 1) / KS : (2 * ks) / KS, ky = fh ? (2 * ks + 1) % KS : (2 * ks) % KS;
            const int iy = oy * ST - PAD + ky;
            const bool rok = r < R && pok && (unsigned)iy < (unsigned)a.H;
            const XT* rp = xb + ci * a.xs_c + (int64_t)iy * a.xs_y;
            float v[8];
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) {
                v[kx] = 0.f;
                if (kx < KS) {
                    const bool ok = rok && (unsigned)(ix0 + kx) < (unsigned)a.W;
                    if (ok) v[kx] = elem<XT>::ld(rp + (int64_t)(ix0 + kx) * a.xs_x);
                }
            }
            frag[ks] = u32x4{pack_h2_fast(v[0], v[1]), pack_h2_fast(v[2], v[3]), pack_h2_fast(v[4], v[5]), pack_h2_fast(v[6], v[7])};
        }
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < (R + 1) / 2; ++ks) {
            const h16x8 xf = __builtin_bit_cast(h16x8, frag[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a.dbg & 2) break;
                const int row = i * 32 + frow;
                const h16x8 wf = *(const h16x8*)(wl + (row * 16 + ((ks * 2 + fh) ^ (row & 15))) * 16);
                acc[i] = mfma_32x32x16_h16(wf, xf, acc[i], 0, 0, 0);
            }
        }
        // bias in registers; the squares of lane-half h's own accumulators are the B operand of the GDN contraction (see
        // the permuted gamma' image above): no LDS exchange between the two GEMMs
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = i * 32 + 8 * g + 4 * fh;
                const f32x4 bv = *(const f32x4*)(bl + cl);
                acc[i][4 * g] += bv.x; acc[i][4 * g + 1] += bv.y; acc[i][4 * g + 2] += bv.z; acc[i][4 * g + 3] += bv.w;
            }
        f32x16 nrm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int si = ks >> 1, so = (ks & 1) * 8;
            const u32x4 sq = u32x4{pack_sq2(acc[si][so], acc[si][so + 1]),
                                   pack_sq2(acc[si][so + 2], acc[si][so + 3]),
                                   pack_sq2(acc[si][so + 4], acc[si][so + 5]),
                                   pack_sq2(acc[si][so + 6], acc[si][so + 7])};
            const h16x8 qf = __builtin_bit_cast(h16x8, sq);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a.dbg & 4) break;
                const int row = i * 32 + frow;
                const h16x8 gf = *(const h16x8*)(gl + (row * 16 + ((ks * 2 + fh) ^ (row & 15))) * 16);
                nrm[i] = mfma_32x32x16_h16(gf, qf, nrm[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = i * 32 + 8 * g + 4 * fh;
                const f32x4 be = *(const f32x4*)(bl + 128 + cl);
                float o[4];
                const float bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n = nrm[i][4 * g + e] + bb[e];
                    o[e] = acc[i][4 * g + e] * (inverse ? __builtin_amdgcn_sqrtf(n) : rsqrtf(n));
                }
                *(u32x2*)(os + frow * OROW + cl * 2) = u32x2{pack_h2_fast(o[0], o[1]), pack_h2_fast(o[2], o[3])};
            }
        auto store_rows = [&](h16_t* dstp) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = it * 64 + lane, pr = c >> 4, cc = c & 15;
                const int y2 = ty * 2 + (pr >> 4), x2 = tx * 16 + (pr & 15);
                if (y2 < a.Ho && x2 < a.Wo)
                    *(u32x4*)(dstp + b * a.ys_b + y2 * a.ys_y + x2 * a.ys_x + cc * 8) = *(const u32x4*)(os + pr * OROW + cc * 16);
            }
        };
        store_rows((h16_t*)a.y);
        if (y_pre) {
            // training form: the conv output v (still in acc) goes out through the same wave-private rows
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = i * 32 + 8 * g + 4 * fh;
                    *(u32x2*)(os + frow * OROW + cl * 2) = u32x2{pack_h2(acc[i][4 * g], acc[i][4 * g + 1]), pack_h2(acc[i][4 * g + 2], acc[i][4 * g + 3])};
                }
            store_rows(y_pre);
        }
    }
}

// Fast form of the kernel above for the layout the hot path actually feeds it: fp32 planes with unit pixel stride and an
// even width.  The generic gather costs ~1500 VALU instructions per 32-pixel tile (64-bit addressing and a predicate per
// element, both transcendentals of the GDN/IGDN select evaluated, 64-bit tile decode) against 64 MFMAs -- the kernel was
// VALU-bound at 2.6x its HBM time.  Here
//   * the 5 taps of an input row are three 8-byte buffer loads (ix0 = 2 ox - 2 is even); out-of-image rows / pairs are a
//     poisoned offset (>= 2^31: the buffer unit returns zeros), one add + three selects per row;
//   * per-lane row constants (plane/row byte offset, ky) are computed once, the tile decode is scalar (fast division);
//   * the next tile's rows are requested as soon as this tile's conv MFMAs have consumed the fragments, so the HBM
//     latency hides behind the GDN contraction, the epilogue and the stores;
//   * accumulators start from bias / beta' (no separate adds), INV is a template parameter, rows leave through buffer
//     stores with scalar tile offsets.
template <int INV>
__global__ __launch_bounds__(512) void sconv_n2w_gdn_fast_kernel(const SArgs a, const h16_t* __restrict__ gamma_packed,
                                                                 const float* __restrict__ beta_packed, h16_t* __restrict__ y_pre,
                                                                 FastDiv fd_tx, FastDiv fd_ty) {
    constexpr int OROW = 128 * 2 + 16, CIN = 3, KS = 5, R = CIN * KS, NW = 8;
    constexpr uint32_t POISON = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                          // conv weights [128][128] bf16, slot ^ (row & 15)
    unsigned char* gl = smem + 32768;                  // gamma' image (K order permuted, see the kernel above)
    float* bl = (float*)(smem + 65536);                // conv bias[128], beta'[128]
    unsigned char* osb = smem + 65536 + 1024;          // per-wave 32 x OROW slices
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    unsigned char* os = osb + wave * 32 * OROW;

    const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + 1) / 2;       // wave tile = 2 output rows x 16 columns
    const int ntiles = tiles_x * tiles_y * a.B;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tstride = (int)gridDim.x * NW;
    u32x2 raw[8][3];
    int tb = 0, tty = 0, ttx = 0;
    auto decode = [&](int tile) {
        const uint32_t q = fdiv((uint32_t)tile, fd_tx);
        ttx = tile - (int)q * tiles_x;
        tb = (int)fdiv(q, fd_ty);
        tty = (int)q - tb * tiles_y;
    };
    auto request = [&](int tile) {                    // issue the 24 row-pair loads of a tile
        decode(tile);
        if (a.dbg & 1) return;
        // the resource starts 8 bytes in front of the image so that every in-image pair has a non-negative lane offset
        // (row 0, ox = 0: the second pair sits at byte 0, its lane offset without the shift would be -8 = out of range)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)a.x + (int64_t)tb * a.xs_b - 2), 0, (int)POISON, 0x00020000);
        const int oy = tty * 2 + (frow >> 4), ox = ttx * 16 + (frow & 15);
        const bool pok = oy < a.Ho && ox < a.Wo;
        const int iy0 = oy * 2 - 2, ix0 = ox * 2 - 2;
        const uint32_t base = (uint32_t)((iy0 * (int)a.xs_y + ix0) * 4 + 8);
        const bool ok0 = pok && ox > 0, ok2 = pok && ix0 + 4 < a.W;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            // k-step ks of lane-half fh carries input row r = 2 ks + fh = (plane ci, tap row ky); row 15 does not exist
            const int ra = 2 * ks, rb = 2 * ks + 1;
            const int kya = ra % KS, kyb = rb < R ? rb % KS : 0x40000000, cia = ra / KS, cib = rb / KS;
            const uint32_t offa = (uint32_t)((cia * (int)a.xs_c + kya * (int)a.xs_y) * 4), offb = (uint32_t)((cib * (int)a.xs_c + (rb % KS) * (int)a.xs_y) * 4);
            const bool okr = (unsigned)(iy0 + (fh ? kyb : kya)) < (unsigned)a.H;
            const uint32_t v = base + (fh ? offb : offa);
            const uint32_t v0 = (okr && ok0) ? v : POISON, v1 = (okr && pok) ? v : POISON, v2 = (okr && ok2) ? v : POISON;
            raw[ks][0] = __builtin_amdgcn_raw_buffer_load_b64(xr, (int)v0, 0, 0);
            raw[ks][1] = __builtin_amdgcn_raw_buffer_load_b64(xr, (int)v1, 8, 0);
            raw[ks][2] = __builtin_amdgcn_raw_buffer_load_b64(xr, (int)v2, 16, 0);
        }
    };
    int tile = lb * NW + wave;
    if (tile < ntiles) request(tile);                 // in flight while the block packs its weights

    if (a.w_img) {
        // both LDS images (conv weights, K-permuted gamma') were laid out once per weight update by sconv_pack_n2w_image_kernel:
        // the block start-up is a straight 64 KB copy, 8 independent 16-byte loads per thread (one memory round trip).  The
        // in-kernel gather below (4 + 4 dependent round trips of scalar loads) cost ~14 us of a 49 us launch.
        const u32x4* src = (const u32x4*)a.w_img;
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[tid + j * 512];
#pragma unroll
        for (int j = 0; j < 8; ++j) *(u32x4*)(smem + (tid + j * 512) * 16) = v[j];
        if (tid < 128) { bl[tid] = a.bias ? a.bias[tid] : 0.f; bl[128 + tid] = beta_packed[tid]; }
    } else {
        const int r = tid & 15, ci = r / KS, ky = r % KS;
        for (int co = tid >> 4; co < 128; co += 32) {
            float v[8];
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
                v[kx] = (r < R && kx < KS && co < a.Cout) ? a.w[(((int64_t)co * CIN + ci) * KS + ky) * KS + kx] : 0.f;
            *(u32x4*)(wl + (co * 16 + (r ^ (co & 15))) * 16) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
        }
        for (int idx = tid; idx < 2048; idx += 512) {
            const int row = idx >> 4, q = idx & 15, ks = q >> 1, h = q & 1;
            const unsigned char* srow = (const unsigned char*)gamma_packed + row * 256;
            const u32x2 lo = *(const u32x2*)(srow + (((2 * ks) ^ (row & 15)) << 4) + 8 * h);
            const u32x2 hi = *(const u32x2*)(srow + (((2 * ks + 1) ^ (row & 15)) << 4) + 8 * h);
            *(u32x4*)(gl + row * 256 + ((q ^ (row & 15)) << 4)) = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
        if (tid < 128) { bl[tid] = a.bias ? a.bias[tid] : 0.f; bl[128 + tid] = beta_packed[tid]; }
    }
    __syncthreads();
    if (a.dbg & 16) return;          // prologue only

    // lane part of the output row addresses: store instruction `it` writes pixel it*4 + (lane >> 4), 16-byte chunk lane & 15
    const uint32_t st_lane = (uint32_t)(((lane >> 4) * (int)a.ys_x + (lane & 15) * 8) * 2);
    for (; tile < ntiles; tile += tstride) {
        const int b = tb, ty = tty, tx = ttx;        // of the tile whose rows are in `raw`
        u32x4 frag[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x2 p0 = __builtin_bit_cast(f32x2, raw[ks][0]), p1 = __builtin_bit_cast(f32x2, raw[ks][1]), p2 = __builtin_bit_cast(f32x2, raw[ks][2]);
            frag[ks] = u32x4{pack_h2_fast(p0.x, p0.y), pack_h2_fast(p1.x, p1.y), pack_h2_fast(p2.x, 0.f), 0u};
        }
        // the rows of the NEXT tile are requested now: `raw` is free (its values live on in `frag`), and the loads then have
        // both MFMA phases (~2000 matrix-pipe cycles) to come back instead of the epilogue alone
        __builtin_amdgcn_sched_barrier(0);
        if (tile + tstride < ntiles) request(tile + tstride);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(bl + i * 32 + 8 * g + 4 * fh);
                acc[i][4 * g] = bv.x; acc[i][4 * g + 1] = bv.y; acc[i][4 * g + 2] = bv.z; acc[i][4 * g + 3] = bv.w;
            }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const h16x8 xf = __builtin_bit_cast(h16x8, frag[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a.dbg & 2) break;
                const int row = i * 32 + frow;
                const h16x8 wf = *(const h16x8*)(wl + (row * 16 + ((ks * 2 + fh) ^ (row & 15))) * 16);
                acc[i] = mfma_32x32x16_h16(wf, xf, acc[i], 0, 0, 0);
            }
        }
        f32x16 nrm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 be = *(const f32x4*)(bl + 128 + i * 32 + 8 * g + 4 * fh);
                nrm[i][4 * g] = be.x; nrm[i][4 * g + 1] = be.y; nrm[i][4 * g + 2] = be.z; nrm[i][4 * g + 3] = be.w;
            }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int si = ks >> 1, so = (ks & 1) * 8;
            const u32x4 sq = u32x4{pack_sq2(acc[si][so], acc[si][so + 1]),
                                   pack_sq2(acc[si][so + 2], acc[si][so + 3]),
                                   pack_sq2(acc[si][so + 4], acc[si][so + 5]),
                                   pack_sq2(acc[si][so + 6], acc[si][so + 7])};
            const h16x8 qf = __builtin_bit_cast(h16x8, sq);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a.dbg & 4) break;
                const int row = i * 32 + frow;
                const h16x8 gf = *(const h16x8*)(gl + (row * 16 + ((ks * 2 + fh) ^ (row & 15))) * 16);
                nrm[i] = mfma_32x32x16_h16(gf, qf, nrm[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = i * 32 + 8 * g + 4 * fh;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n = nrm[i][4 * g + e];
                    o[e] = acc[i][4 * g + e] * (INV ? __builtin_amdgcn_sqrtf(n) : __builtin_amdgcn_rsqf(n));
                }
                *(u32x2*)(os + frow * OROW + cl * 2) = u32x2{pack_h2_fast(o[0], o[1]), pack_h2_fast(o[2], o[3])};
            }
        const bool full = ty * 2 + 2 <= a.Ho && tx * 16 + 16 <= a.Wo;     // wave-uniform
        auto store_rows = [&](h16_t* dstp) {
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(dstp + (int64_t)b * a.ys_b), 0, (int)POISON, 0x00020000);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int y2 = ty * 2 + (it >> 2), x2 = tx * 16 + (it & 3) * 4;
                const int so = (y2 * (int)a.ys_y + x2 * (int)a.ys_x) * 2;                       // scalar
                const bool ok = full || (y2 < a.Ho && x2 + (lane >> 4) < a.Wo);
                const u32x4 v = *(const u32x4*)(os + (it * 4 + (lane >> 4)) * OROW + (lane & 15) * 16);
                // The tile offset is ADDED INTO THE VGPR OFFSET, not passed as the SGPR soffset.  A buffer store of more than 8
                // bytes must not be followed by a VALU write of its data VGPRs within one wait state; the compiler's hazard
                // recognizer skips that rule when soffset is a register (the ISA manual exempts that form), and with the SGPR
                // form it scheduled the next store's v_cndmask into data dword 0 right behind the store -- on gfx950 that is NOT
                // safe: ~1 forward in 25 came back with the first two channels of a few 16-byte chunks replaced by offset bits
                // (found in round 2 by a bit-identity stress test).  With an immediate soffset the compiler inserts the s_nop.
                __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)((ok && !(a.dbg & 8)) ? st_lane + (uint32_t)so : POISON), 0, 0);
            }
        };
        store_rows((h16_t*)a.y);
        if (y_pre) {
            // training form: the conv output v (still in acc) goes out through the same wave-private rows
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = i * 32 + 8 * g + 4 * fh;
                    *(u32x2*)(os + frow * OROW + cl * 2) = u32x2{pack_h2(acc[i][4 * g], acc[i][4 * g + 1]), pack_h2(acc[i][4 * g + 2], acc[i][4 * g + 3])};
                }
            store_rows(y_pre);
        }
    }
}

// ---------------------------------------------------------------- wide -> narrow on the matrix cores (bf16 input)
// g_s_conv4 (ConvTranspose2d 128 -> 3, 5x5 s2 p2 op1).  With only 3 output channels the GEMM is turned round: every
// INPUT pixel is multiplied by the whole [Cin x (25*Cout)] weight panel (N = 75 -> 96), giving its 5x5xCout "splat"
// G[pixel][tap*Cout+co]; the output pixel then gathers the <= 9 splats that land on it (col2im) from LDS.
// Block = a 6x14 input patch + 1-pixel halo (8x16 = 128 pixels = one 32-pixel MFMA tile per wave) -> a 12x28 output
// patch; 75 KB of LDS -> two blocks per CU, and the next tile's pixel fragments are fetched while this tile's
// col2im runs, so the HBM latency is hidden both by the co-resident block and by the prefetch.
constexpr int W2N_TH = 6, W2N_TW = 14;
template <int COUT>
__global__ __launch_bounds__(256) void sconv_w2n_mfma_kernel(const SArgs a, FastDiv fd_tx, FastDiv fd_ty) {
    constexpr int N = 25 * COUT, NT = (N + 31) / 32, NP = NT * 32;
    constexpr int GROW = NP * 4 + 16;                         // G row stride (bytes)
    constexpr int CIN = 128, SPR = CIN / 8;
    constexpr uint32_t POISON = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char dsm[NP * CIN * 2 + 128 * GROW];
    unsigned char* wl = dsm;                                  // [NP][128] bf16, 16-byte slots XOR (row & 15)
    unsigned char* G = dsm + NP * CIN * 2;                    // [128 px][GROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int tiles_x = (a.W + W2N_TW - 1) / W2N_TW, tiles_y = (a.H + W2N_TH - 1) / W2N_TH;
    const int ntiles = tiles_x * tiles_y * a.B;               // < 2^31 (launcher)
    const int pl = wave * 32 + frow;                          // this lane's pixel in the 8x16 halo patch
    // two tiles of pixel fragments in flight (A / B): a fetch is issued as soon as the MFMAs have consumed its registers and
    // is not needed before the block has worked through the other tile, so the HBM latency never sits in front of a barrier
    struct Slot { u32x4 raws[8]; int tb, tty, ttx; };
    Slot sa, sb;
    auto fetch = [&](Slot& sl, int tile) {
        u32x4 (&raws)[8] = sl.raws;
        const uint32_t q = fdiv((uint32_t)tile, fd_tx);
        const int ttx = tile - (int)q * tiles_x;
        const int tb = (int)fdiv(q, fd_ty);
        const int tty = (int)q - tb * tiles_y;
        sl.tb = tb; sl.tty = tty; sl.ttx = ttx;
        // one buffer resource per image: a halo pixel outside the image is a poisoned offset and reads zeros
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const h16_t*)a.x + (int64_t)tb * a.xs_b), 0, (int)POISON, 0x00020000);
        const int iy = tty * W2N_TH - 1 + (pl >> 4), ix = ttx * W2N_TW - 1 + (pl & 15);
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const uint32_t off = ok ? (uint32_t)((iy * (int)a.xs_y + ix * (int)a.xs_x + fh * 8) * 2) : POISON;    // xs_c == 1
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) raws[ks] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, ks * 32, 0);
    };
    const int lb = xcd_remap(blockIdx.x, gridDim.x);                    // neighbouring patches share their 1-pixel halo in one L2
    const int gstep = (int)gridDim.x;
    if (lb < ntiles) fetch(sa, lb);                           // in flight while the weight panel is packed
    if (lb + gstep < ntiles) fetch(sb, lb + gstep);
    if (a.w_img) {
        // pre-packed weight panel (sconv_pack_w2n_image_kernel): straight copy, NP * SPR / 256 independent 16-byte loads per thread
        constexpr int PER = NP * SPR / 256;
        static_assert(NP * SPR % 256 == 0, "panel size");
        const u32x4* src = (const u32x4*)a.w_img;
        u32x4 v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = src[tid + j * 256];
#pragma unroll
        for (int j = 0; j < PER; ++j) *(u32x4*)(wl + (tid + j * 256) * 16) = v[j];
    } else {
        for (int i = tid; i < NP * SPR; i += 256) {
            const int n = i / SPR, sl = i % SPR;
            const int co = n % COUT, tap = n / COUT;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = n < N ? a.w[((int64_t)(sl * 8 + e) * COUT + co) * 25 + tap] : 0.f;   // w[ci][co][ky][kx]
            *(u32x4*)(wl + (n * SPR + (sl ^ (n & 15))) * 16) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
        }
    }
    // col2im role of this thread: output channel cq of the 2x2 output quad around input pixel (qy, qx) of the 6x14 patch --
    // 252 of the 256 threads busy, every lane the same 25 taps (no divergence between output parities)
    const int qi = tid / COUT, cq = tid - qi * COUT;
    const int qy = qi / W2N_TW, qx = qi - qy * W2N_TW;
    const bool qlive = qi < W2N_TH * W2N_TW;
    const float bq = (a.bias && qlive) ? a.bias[cq] : 0.f;
    const unsigned char* gq = G + ((qy + 2) * 16 + qx + 2) * GROW + cq * 4;
    const bool pair_st = a.y_dtype == HESIC_F32 && a.ys_x == 1 && !((a.ys_b | a.ys_c | a.ys_y) & 1) && !((uintptr_t)a.y & 7);
    __syncthreads();
    auto process = [&](Slot& sl, int next_tile) {
        u32x4 (&raws)[8] = sl.raws;
        const int tx = sl.ttx, ty = sl.tty, b = sl.tb;        // of the tile whose pixels are in `raws`
        f32x16 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const h16x8 xf = __builtin_bit_cast(h16x8, raws[ks]);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int row = i * 32 + frow;
                const h16x8 wf = *(const h16x8*)(wl + (row * SPR + ((ks * 2 + fh) ^ (row & 15))) * 16);
                acc[i] = mfma_32x32x16_h16(wf, xf, acc[i], 0, 0, 0);
            }
        }
        if (next_tile < ntiles) fetch(sl, next_tile);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(G + pl * GROW + (i * 32 + 8 * g + 4 * fh) * 4) =
                    f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        __syncthreads();
        // col2im: out(2 q + p) = bias + sum over taps k = p + 2 j (<= 4) of G[q + 1 - j (+1 halo)][k]
        const int gy = ty * W2N_TH + qy, gx = tx * W2N_TW + qx;
        if (qlive && gy < a.H && gx < a.W) {
            float o[2][2] = {{bq, bq}, {bq, bq}};
#pragma unroll
            for (int jy = 0; jy < 3; ++jy)
#pragma unroll
                for (int jx = 0; jx < 3; ++jx) {
                    const unsigned char* gp = gq - (jy * 16 + jx) * GROW;
#pragma unroll
                    for (int py = 0; py < 2; ++py)
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            const int ky = py + 2 * jy, kx = px + 2 * jx;
                            if (ky < 5 && kx < 5) o[py][px] += *(const float*)(gp + (ky * 5 + kx) * COUT * 4);
                        }
                }
            const int64_t ob = b * a.ys_b + cq * a.ys_c + (int64_t)(2 * gy) * a.ys_y + (int64_t)(2 * gx) * a.ys_x;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const float o0 = apply_act(o[py][0], a.act), o1 = apply_act(o[py][1], a.act);
                if (pair_st) *(f32x2*)((float*)a.y + ob + py * a.ys_y) = f32x2{o0, o1};
                else {
                    st_any(a.y, ob + py * a.ys_y, a.y_dtype, o0);
                    st_any(a.y, ob + py * a.ys_y + a.ys_x, a.y_dtype, o1);
                }
            }
        }
        __syncthreads();
    };
    for (int tile = lb; tile < ntiles; tile += 2 * gstep) {
        process(sa, tile + 2 * gstep);
        if (tile + gstep < ntiles) process(sb, tile + 3 * gstep);
    }
}

// ---------------------------------------------------------------- narrow -> narrow, stride 1 (pre_conv / after_conv)
// Conv2d or (stride-1) ConvTranspose2d with few channels on both sides: y[co][o] = sum x[ci][o + k - p] w'[k][ci][co] where
// w' is the kernel (mirrored for the transposed op).  One thread = PX consecutive output pixels of one row; per (ci, ky)
// it loads the K+PX-1 inputs once and feeds PX*K*COUT FMAs.  Weights sit in LDS as [ky][kx][ci][4] (broadcast float4).
// LDS-tiled variant for the full-resolution 6 -> 3 stage (pre_conv / after_conv at 512x512): a block stages a
// (16+K-1) x (64+K-1) x CIN input patch in LDS with row-contiguous loads (every input element leaves L2 once), then each
// thread produces 4 consecutive pixels x COUT from two aligned ds_read_b128 per (ci, ky); weights are broadcast float4.
template <int CIN, int COUT, int K>
__global__ __launch_bounds__(256) void sconv_small_s1_lds_kernel(const SArgs a) {
    constexpr int TH = 16, TW = 64, PH = TH + K - 1, PW = TW + 8, PAD = K / 2;   // PW: halo rounded up to keep rows 16-B aligned
    constexpr int CP = COUT <= 4 ? 4 : 8;                                         // padded cout count of the weight rows
    static_assert(COUT <= 8, "at most 8 output channels");
    __shared__ __attribute__((aligned(16))) float xs[CIN * PH * PW];
    __shared__ __attribute__((aligned(16))) float wl[K * K * CIN * CP];
    const int tid = threadIdx.x;
    for (int i = tid; i < K * K * CIN * CP; i += 256) {
        const int co = i % CP, ci = (i / CP) % CIN, tap = (i / CP) / CIN;
        int ky = tap / K, kx = tap % K;
        if (a.transposed) { ky = K - 1 - ky; kx = K - 1 - kx; }
        wl[i] = co < COUT ? w_at(a.w, co, ci, ky, kx, COUT, CIN, K, K, a.transposed) : 0.f;
    }
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = lb % tiles_x, ty = (lb / tiles_x) % tiles_y, b = lb / (tiles_x * tiles_y);
    const int y0 = ty * TH - PAD, x0 = tx * TW - PAD;
    for (int i = tid; i < CIN * PH * PW; i += 256) {
        const int px = i % PW, py = (i / PW) % PH, ci = i / (PW * PH);
        const int iy = y0 + py, ix = x0 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
            v = ld_any(a.x, b * a.xs_b + ci * a.xs_c + (int64_t)iy * a.xs_y + (int64_t)ix * a.xs_x, a.x_dtype);
        xs[i] = v;
    }
    __syncthreads();
    const int ly = tid >> 4, lx = (tid & 15) * 4;
    float acc[4][COUT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[p][c] = 0.f;
#pragma unroll 1
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const float* row = xs + (ci * PH + ly + ky) * PW + lx;
            const f32x4 r0 = *(const f32x4*)row, r1 = *(const f32x4*)(row + 4);
            const float xin[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float* wp = wl + ((ky * K + kx) * CIN + ci) * CP;
                float wv[CP];
                *(f32x4*)wv = *(const f32x4*)wp;
                if (CP == 8) *(f32x4*)(wv + 4) = *(const f32x4*)(wp + 4);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[p][co] += xin[p + kx] * wv[co];
            }
        }
    }
    const int oy = ty * TH + ly;
    if (oy < a.Ho)
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int ox = tx * TW + lx + p;
                if (ox < a.Wo) st_any(a.y, b * a.ys_b + co * a.ys_c + oy * a.ys_y + ox * a.ys_x, a.y_dtype, apply_act(acc[p][co] + bv, a.act));
            }
        }
}

// 3-channel (I)GDN of one pixel, the arithmetic of gdn_planar_kernel<3> (csrc/gdn.hip) statement for statement
struct Gdn3 {
    float g[3][3], bt[3];
    int inverse;
    __device__ __forceinline__ void load(const float* beta, const float* gamma, float bound, int inv) {
        constexpr float kPed = 1.0f / 68719476736.0f, kGb = 1.0f / 262144.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = fmaxf(beta[c], bound);
            bt[c] = t * t - kPed;
#pragma unroll
            for (int j = 0; j < 3; ++j) { t = fmaxf(gamma[c * 3 + j], kGb); g[c][j] = t * t - kPed; }
        }
        inverse = inv;
    }
    __device__ __forceinline__ void apply(float (&v)[3]) const {
        float sq[3], o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) sq[c] = v[c] * v[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float norm = bt[c];
#pragma unroll
            for (int j = 0; j < 3; ++j) norm += g[c][j] * sq[j];
            o[c] = v[c] * (inverse ? sqrtf(norm) : rsqrtf(norm));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = o[c];
    }
};


// ---------------------------------------------------------------- 6 -> 3, 5x5, stride 1 (pre_conv / after_conv)
// 16 x 128 output tile per block, all 6 input planes of the tile (+halo) in LDS as fp32, 8 consecutive pixels x 3 couts
// per thread: per (ci, ky) three ds_read_b128 of pixels and five broadcast weight reads feed 120 FMAs.  The 132-float
// row pitch keeps the 16-lane groups of those reads on disjoint banks.  The two 3-channel halves of the input may come
// from different tensors (any strides / fp32 or bf16): planar fp32 rows are fetched as 8-byte pairs.
template <int K, int PX>
__global__ __launch_bounds__(256, PX == 4 ? 4 : 2) void sconv_6to3_s1_kernel(const SArgs a) {
    // 16 x 64 pixel tile, 4 pixels per thread: 35 KB of LDS and < 128 VGPRs -> four blocks (16 waves) per CU, so the staging
    // of one block overlaps the FMA phase of the others
    // (PX = 8: 16 x 128 tile, 66 KB of LDS, two blocks per CU, 1.5 instead of 2.1 VALU instructions per packed FMA)
    constexpr int CIN = 6, COUT = 3, TH = 16, TW = 16 * PX, PAD = K / 2, PH = TH + K - 1, PW = TW + 4, NP = PW / 2;
    __shared__ __attribute__((aligned(16))) float xs[CIN * PH * PW];
    __shared__ __attribute__((aligned(16))) float wl[K * K * CIN * 4];
    const int tid = threadIdx.x;
    for (int i = tid; i < K * K * CIN * 4; i += 256) {
        const int co = i & 3, ci = (i >> 2) % CIN, tap = (i >> 2) / CIN;
        int ky = tap / K, kx = tap % K;
        if (a.transposed) { ky = K - 1 - ky; kx = K - 1 - kx; }
        wl[i] = co < COUT ? w_at(a.w, co, ci, ky, kx, COUT, CIN, K, K, a.transposed) : 0.f;
    }
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = lb % tiles_x, ty = (lb / tiles_x) % tiles_y, b = lb / (tiles_x * tiles_y);
    const int y0 = ty * TH - PAD, x0 = tx * TW - PAD;
    for (int half = 0; half < 2; ++half) {
        const int c_lo = half ? a.c_split : 0, c_hi = half ? CIN : a.c_split;
        if (c_hi <= c_lo) continue;
        const void* src = half ? a.x2 : a.x;
        const int64_t sb = half ? a.x2s_b : a.xs_b, sc = half ? a.x2s_c : a.xs_c, sy = half ? a.x2s_y : a.xs_y, sx = half ? a.x2s_x : a.xs_x;
        const int dt = half ? a.x2_dtype : a.x_dtype;
        const int rows = (c_hi - c_lo) * PH;
        const bool pairs = dt == HESIC_F32 && sx == 1 && !(a.W & 1) && !((sb | sc | sy) & 1) && !((uintptr_t)src & 7);
        if (pairs && ((int64_t)CIN * sc + (int64_t)a.H * sy) * 4 < (1ll << 31)) {
            // 8-byte buffer loads, eight in flight per thread before the first LDS write (the one-load-per-iteration form
            // serialised ~31 HBM round trips per block and set the kernel's time); out-of-image pairs read zeros through a
            // poisoned offset, so the loop body has no branch
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)src + b * sb), 0, (int)0x80000000u, 0x00020000);
            const int total = rows * NP;
            for (int i0 = tid; i0 < total; i0 += 256 * 8) {
                u32x2 v[8];
                int dst[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * 256;
                    const int r = i / NP, j = i - r * NP;
                    const int ci = r / PH, py = r - ci * PH;
                    const int iy = y0 + py, ix = x0 + 2 * j;
                    const bool ok = i < total && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    const uint32_t off = ok ? (uint32_t)((ci * (int)sc + iy * (int)sy + ix) * 4) : 0x80000000u;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b64(xr, (int)off, 0, 0);
                    dst[u] = i < total ? ((c_lo + ci) * PH + py) * PW + 2 * j : -1;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (dst[u] >= 0) *(u32x2*)(xs + dst[u]) = v[u];
            }
        } else {
            for (int i = tid; i < rows * PW; i += 256) {
                const int r = i / PW, px = i - r * PW;
                const int ci = r / PH, py = r - ci * PH;
                const int iy = y0 + py, ix = x0 + px;
                float v = 0.f;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    v = ld_any(src, b * sb + ci * sc + (int64_t)iy * sy + (int64_t)ix * sx, dt);
                xs[((c_lo + ci) * PH + py) * PW + px] = v;
            }
        }
    }
    __syncthreads();
    if (a.gdn_mode == 2) {
        // (I)GDN of the first three staged channels in place, halo included (padding pixels are zeros and stay zeros)
        Gdn3 gd;
        gd.load(a.gdn_beta, a.gdn_gamma, a.gdn_bound, a.gdn_inverse);
        for (int i = tid; i < PH * PW; i += 256) {
            float v[3] = {xs[i], xs[PH * PW + i], xs[2 * PH * PW + i]};
            gd.apply(v);
            xs[i] = v[0]; xs[PH * PW + i] = v[1]; xs[2 * PH * PW + i] = v[2];
        }
        __syncthreads();
    }
    const int ly = tid >> 4, lx = (tid & 15) * PX;
    float acc[PX][COUT];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[p][c] = 0.f;
#pragma unroll 1
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            const float* row = xs + (ci * PH + ly + ky) * PW + lx;
            float xin[PX + 4];
#pragma unroll
            for (int q = 0; q < (PX + 4) / 4; ++q) *(f32x4*)(xin + 4 * q) = *(const f32x4*)(row + 4 * q);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 wv = *(const f32x4*)(wl + ((ky * K + kx) * CIN + ci) * 4);
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    acc[p][0] = fmaf(xin[p + kx], wv.x, acc[p][0]);
                    acc[p][1] = fmaf(xin[p + kx], wv.y, acc[p][1]);
                    acc[p][2] = fmaf(xin[p + kx], wv.z, acc[p][2]);
                }
            }
        }
    }
    const int oy = ty * TH + ly, ox0 = tx * TW + lx;
    if (oy >= a.Ho) return;
    const bool vec = a.y_dtype == HESIC_F32 && a.ys_x == 1 && !(a.Wo & 3) && !((a.ys_b | a.ys_c | a.ys_y) & 3) && !((uintptr_t)a.y & 15);
    if (a.gdn_mode == 1) {
        // conv + bias (+ act) of a pixel's three channels are all in this thread: the 3-channel (I)GDN needs nothing else
        Gdn3 gd;
        gd.load(a.gdn_beta, a.gdn_gamma, a.gdn_bound, a.gdn_inverse);
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            float v[3];
#pragma unroll
            for (int co = 0; co < COUT; ++co) v[co] = apply_act(acc[p][co] + (a.bias ? a.bias[co] : 0.f), a.act);
            gd.apply(v);
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[p][co] = v[co];
        }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = (a.bias && a.gdn_mode != 1) ? a.bias[co] : 0.f;
        float o[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) o[p] = a.gdn_mode == 1 ? acc[p][co] : apply_act(acc[p][co] + bv, a.act);
        const int64_t base = b * a.ys_b + co * a.ys_c + oy * a.ys_y;
        if (vec && ox0 + PX <= a.Wo) {
            float* yp = (float*)a.y + base + ox0;
#pragma unroll
            for (int q = 0; q < PX / 4; ++q) *(f32x4*)(yp + 4 * q) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
        } else {
#pragma unroll
            for (int p = 0; p < PX; ++p)
                if (ox0 + p < a.Wo) st_any(a.y, base + (ox0 + p) * a.ys_x, a.y_dtype, o[p]);
        }
    }
}

constexpr int SS_PX = 4;
__global__ __launch_bounds__(256) void sconv_small_s1_kernel(const SArgs a) {
    __shared__ __attribute__((aligned(16))) float wl[7 * 7 * 8 * 4];
    const int nW = a.KH * a.KW * a.Cin * 4;
    for (int i = threadIdx.x; i < nW; i += 256) {
        const int co = i & 3, ci = (i >> 2) % a.Cin, tap = (i >> 2) / a.Cin;
        int ky = tap / a.KW, kx = tap % a.KW;
        if (a.transposed) { ky = a.KH - 1 - ky; kx = a.KW - 1 - kx; }
        wl[i] = co < a.Cout ? w_at(a.w, co, ci, ky, kx, a.Cout, a.Cin, a.KH, a.KW, a.transposed) : 0.f;
    }
    __syncthreads();
    const int pad_y = a.transposed ? a.KH - 1 - a.pad : a.pad, pad_x = a.transposed ? a.KW - 1 - a.pad : a.pad;
    const int segs = (a.Wo + SS_PX - 1) / SS_PX;
    const int64_t total = (int64_t)a.B * a.Ho * segs;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int sg = i % segs, oy = (i / segs) % a.Ho, b = i / ((int64_t)segs * a.Ho);
        const int ox0 = sg * SS_PX;
        float acc[SS_PX][4];
#pragma unroll
        for (int p = 0; p < SS_PX; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[p][c] = 0.f;
        for (int ci = 0; ci < a.Cin; ++ci) {
            for (int ky = 0; ky < a.KH; ++ky) {
                const int iy = oy - pad_y + ky;
                if ((unsigned)iy >= (unsigned)a.H) continue;
                const int64_t rb = b * a.xs_b + ci * a.xs_c + iy * a.xs_y;
                float xin[SS_PX + 6];
#pragma unroll
                for (int j = 0; j < SS_PX + 6; ++j) {
                    const int ix = ox0 - pad_x + j;
                    xin[j] = (j < SS_PX + a.KW - 1 && (unsigned)ix < (unsigned)a.W) ? ld_any(a.x, rb + ix * a.xs_x, a.x_dtype) : 0.f;
                }
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    if (kx < a.KW) {
                        const f32x4 wv = *(const f32x4*)(wl + (((ky * a.KW + kx) * a.Cin) + ci) * 4);
#pragma unroll
                        for (int p = 0; p < SS_PX; ++p) {
                            acc[p][0] += xin[p + kx] * wv.x; acc[p][1] += xin[p + kx] * wv.y;
                            acc[p][2] += xin[p + kx] * wv.z; acc[p][3] += xin[p + kx] * wv.w;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < SS_PX; ++p) {
            const int ox = ox0 + p;
            if (ox < a.Wo)
                for (int co = 0; co < a.Cout; ++co)
                    st_any(a.y, b * a.ys_b + co * a.ys_c + oy * a.ys_y + ox * a.ys_x, a.y_dtype,
                           apply_act(acc[p][co] + (a.bias ? a.bias[co] : 0.f), a.act));
        }
    }
}

int launch_forward(const SArgs& a, hipStream_t st) {
    constexpr bool legacy = false;   // A/B switch for profiling
    if (!legacy && !a.transposed && a.y_dtype == HESIC_H16 && a.Cin == 3 && a.KH == 5 && a.KW == 5 && a.stride == 2 && a.pad == 2 &&
        a.Cout % 8 == 0 && a.ys_c == 1 && (a.ys_x % 8) == 0 && (a.ys_y % 8) == 0 && (a.ys_b % 8) == 0) {
        const int64_t tiles = (int64_t)((a.Wo + 15) / 16) * ((a.Ho + 7) / 8) * a.B;
        const dim3 grid((unsigned)(tiles < 512 ? tiles : 512));      // 2 resident blocks per CU, each loops over tiles
        if (a.x_dtype == HESIC_H16) hipLaunchKernelGGL((sconv_n2w_mfma_kernel<3, 5, 2, h16_t>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((sconv_n2w_mfma_kernel<3, 5, 2, float>), grid, dim3(256), 0, st, a);
    } else if (!a.transposed && a.Cin <= 8 && a.Cout % 32 == 0 && a.ys_c == 1 && (a.ys_x % 8) == 0 && (a.ys_y % 8) == 0 &&
        (a.ys_b % 8) == 0 && a.KH * a.KW * a.Cin * a.Cout * 4 <= 60 * 1024) {
        const int tiles = ((a.Wo + 7) / 8) * ((a.Ho + 7) / 8) * a.B;
        const size_t lds = (size_t)a.KH * a.KW * a.Cin * a.Cout * 4;
        hipLaunchKernelGGL(sconv_narrow_to_wide_kernel, dim3(tiles), dim3(256), lds, st, a);
    } else if (!legacy && a.transposed && a.x_dtype == HESIC_H16 && a.stride == 2 && a.KH == 5 && a.KW == 5 && a.pad == 2 &&
               a.Cout == 3 && a.Cin == 128 && a.xs_c == 1 && (a.xs_x % 8) == 0 && (a.xs_y % 8) == 0 && (a.xs_b % 8) == 0 &&
               a.Ho == 2 * a.H && a.Wo == 2 * a.W && (int64_t)a.H * a.xs_y * 2 < (1ll << 31) && (int64_t)a.B * a.H * a.W < (1ll << 31)) {
        const int64_t tiles = (int64_t)((a.W + W2N_TW - 1) / W2N_TW) * ((a.H + W2N_TH - 1) / W2N_TH) * a.B;
        hipLaunchKernelGGL((sconv_w2n_mfma_kernel<3>), dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(256), 0, st, a,
                           make_fastdiv((uint32_t)((a.W + W2N_TW - 1) / W2N_TW)), make_fastdiv((uint32_t)((a.H + W2N_TH - 1) / W2N_TH)));
    } else if (a.transposed && a.stride == 2 && a.KH == 5 && a.KW == 5 && a.pad == 2 && a.Cout <= 4 && a.Cin % 8 == 0 &&
               a.xs_c == 1 && (a.xs_x % 8) == 0 && (a.xs_y % 8) == 0 && (a.xs_b % 8) == 0 && a.Cin <= 128) {
        const int64_t total = (int64_t)a.B * a.H * a.W;
        const size_t lds = (size_t)25 * a.Cin * 16;
        if (a.x_dtype == HESIC_H16)
            hipLaunchKernelGGL(sconv_wide_to_narrow_kernel<h16_t>, dim3((unsigned)cdiv64(total, 256)), dim3(256), lds, st, a);
        else
            hipLaunchKernelGGL(sconv_wide_to_narrow_kernel<float>, dim3((unsigned)cdiv64(total, 256)), dim3(256), lds, st, a);
    } else if (!legacy && a.stride == 1 && a.Cin == 6 && a.Cout == 3 && a.KH == 5 && a.KW == 5 && a.pad == 2 && a.Ho == a.H &&
               a.Wo == a.W && a.Wo >= 128) {
        constexpr int px = 4;       // A/B switch
        if (px == 8) {
            const int tiles = ((a.Wo + 127) / 128) * ((a.Ho + 15) / 16) * a.B;
            hipLaunchKernelGGL((sconv_6to3_s1_kernel<5, 8>), dim3(tiles), dim3(256), 0, st, a);
        } else {
            const int tiles = ((a.Wo + 63) / 64) * ((a.Ho + 15) / 16) * a.B;
            hipLaunchKernelGGL((sconv_6to3_s1_kernel<5, 4>), dim3(tiles), dim3(256), 0, st, a);
        }
    } else if (!legacy && a.stride == 1 && ((a.Cin == 6 && a.Cout == 3) || (a.Cin == 3 && a.Cout == 6)) && a.KH == 5 && a.KW == 5 &&
               a.pad == 2 && a.Ho == a.H && a.Wo == a.W && a.Wo >= 64) {
        const int tiles = ((a.Wo + 63) / 64) * ((a.Ho + 15) / 16) * a.B;
        if (a.Cin == 6) hipLaunchKernelGGL((sconv_small_s1_lds_kernel<6, 3, 5>), dim3(tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((sconv_small_s1_lds_kernel<3, 6, 5>), dim3(tiles), dim3(256), 0, st, a);      // their data gradients
    } else if (a.stride == 1 && a.Cin <= 8 && a.Cout <= 4 && a.KH <= 7 && a.KW <= 7 && a.Ho == a.H && a.Wo == a.W) {
        const int64_t total = (int64_t)a.B * a.Ho * ((a.Wo + SS_PX - 1) / SS_PX);
        hipLaunchKernelGGL(sconv_small_s1_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, st, a);
    } else {
        const int64_t total = (int64_t)a.B * a.Ho * a.Wo * ((a.Cout + 3) / 4);
        hipLaunchKernelGGL(sconv_generic_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, st, a);
    }
    return 0;
}

int check_desc(const hesic_sconv_desc* d, const char* who) {
    HESIC_CHECK_ARG(d, "%s: null descriptor", who);
    HESIC_CHECK_ARG(d->stride >= 1 && d->KH > 0 && d->KW > 0 && d->B > 0, "%s: bad geometry", who);
    if (!d->transposed)
        HESIC_CHECK_ARG(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                        "%s: conv output size mismatch", who);
    else
        HESIC_CHECK_ARG(d->Ho == (d->H - 1) * d->stride - 2 * d->pad + d->KH + d->stride - 1 &&
                            d->Wo == (d->W - 1) * d->stride - 2 * d->pad + d->KW + d->stride - 1,
                        "%s: transposed output size mismatch", who);
    return 0;
}

SArgs make_args(const hesic_sconv_desc* d) {
    SArgs a;
    memset(&a, 0, sizeof(a));
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.transposed = d->transposed;
    a.x_dtype = d->x_dtype; a.y_dtype = d->y_dtype; a.act = d->act;
    a.xs_b = d->xs_b; a.xs_c = d->xs_c; a.xs_y = d->xs_y; a.xs_x = d->xs_x;
    a.ys_b = d->ys_b; a.ys_c = d->ys_c; a.ys_y = d->ys_y; a.ys_x = d->ys_x;
    a.c_split = d->Cin;          // single input tensor
    return a;
}

}  // namespace

static thread_local const void* g_w_img = nullptr;       // set by the *_prepacked entry points around the ordinary launchers

static thread_local const float* g_cat_beta = nullptr;    // set by hesic_sconv2d_forward_cat_gdn around the ordinary launcher
static thread_local const float* g_cat_gamma = nullptr;
static thread_local float g_cat_bound = 0.f;
static thread_local int g_cat_mode = 0, g_cat_inverse = 0;

extern "C" int hesic_sconv2d_forward_cat(const hesic_sconv_desc* d, const void* xa, const void* xb, const int64_t xb_strides[4],
                                         int xb_dtype, int ca, const float* w, const float* bias, void* y, void* stream);

extern "C" int hesic_sconv2d_forward_cat_gdn(const hesic_sconv_desc* d, const void* xa, const void* xb, const int64_t xb_strides[4],
                                             int xb_dtype, int ca, const float* w, const float* bias, const float* gdn_beta,
                                             const float* gdn_gamma, float beta_min, int inverse, int gdn_on_input, void* y, void* stream) {
    HESIC_CHECK_ARG(gdn_beta && gdn_gamma, "sconv2d_forward_cat_gdn: null pointer");
    HESIC_CHECK_ARG(!gdn_on_input || ca == 3, "sconv2d_forward_cat_gdn: the input-side (I)GDN covers the first tensor's three channels");
    g_cat_beta = gdn_beta; g_cat_gamma = gdn_gamma; g_cat_bound = sqrtf(beta_min + 1.0f / 68719476736.0f);
    g_cat_mode = gdn_on_input ? 2 : 1; g_cat_inverse = inverse ? 1 : 0;
    const int rc = hesic_sconv2d_forward_cat(d, xa, xb, xb_strides, xb_dtype, ca, w, bias, y, stream);
    g_cat_beta = g_cat_gamma = nullptr; g_cat_mode = 0;
    return rc;
}

extern "C" int hesic_sconv2d_forward_cat(const hesic_sconv_desc* d, const void* xa, const void* xb, const int64_t xb_strides[4],
                                         int xb_dtype, int ca, const float* w, const float* bias, void* y, void* stream) {
    if (int e = check_desc(d, "sconv2d_forward_cat")) return e;
    HESIC_CHECK_ARG(xa && xb && xb_strides && w && y, "sconv2d_forward_cat: null pointer");
    HESIC_CHECK_ARG(d->Cin == 6 && d->Cout == 3 && d->KH == 5 && d->KW == 5 && d->stride == 1 && d->pad == 2 && d->Wo >= 128 &&
                        ca > 0 && ca < d->Cin,
                    "sconv2d_forward_cat: built for the 6 -> 3 5x5 stride-1 stages (pre_conv / after_conv) at width >= 128");
    HESIC_CHECK_ARG(xb_dtype == HESIC_F32 || xb_dtype == HESIC_H16, "sconv2d_forward_cat: bad dtype");
    SArgs a = make_args(d);
    a.x = xa; a.w = w; a.bias = bias; a.y = y;
    a.x2 = xb; a.x2s_b = xb_strides[0]; a.x2s_c = xb_strides[1]; a.x2s_y = xb_strides[2]; a.x2s_x = xb_strides[3];
    a.x2_dtype = xb_dtype; a.c_split = ca;
    a.gdn_beta = g_cat_beta; a.gdn_gamma = g_cat_gamma; a.gdn_bound = g_cat_bound; a.gdn_mode = g_cat_mode; a.gdn_inverse = g_cat_inverse;
    constexpr int px = 4;           // A/B switch
    if (px == 8) {
        const int tiles = ((a.Wo + 127) / 128) * ((a.Ho + 15) / 16) * a.B;
        hipLaunchKernelGGL((sconv_6to3_s1_kernel<5, 8>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        const int tiles = ((a.Wo + 63) / 64) * ((a.Ho + 15) / 16) * a.B;
        hipLaunchKernelGGL((sconv_6to3_s1_kernel<5, 4>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    }
    HESIC_LAUNCH_RETURN("sconv2d_forward_cat");
}

extern "C" int hesic_sconv2d_forward(const hesic_sconv_desc* d, const void* x, const float* w, const float* bias, void* y,
                                     void* stream) {
    if (int e = check_desc(d, "sconv2d_forward")) return e;
    HESIC_CHECK_ARG(x && w && y, "sconv2d_forward: null pointer");
    SArgs a = make_args(d);
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.w_img = g_w_img;
    launch_forward(a, (hipStream_t)stream);
    HESIC_LAUNCH_RETURN("sconv2d_forward");
}

extern "C" int hesic_sconv2d_forward_prepacked(const hesic_sconv_desc* d, const void* x, const float* w, const void* w_image, const float* bias,
                                               void* y, void* stream) {
    HESIC_CHECK_ARG(d && d->transposed && d->Cin == 128 && d->Cout == 3 && d->KH == 5 && d->KW == 5 && d->stride == 2,
                    "sconv2d_forward_prepacked: the image is the weight panel of the 128 -> 3 transposed 5x5 stride-2 stage");
    g_w_img = w_image;
    const int rc = hesic_sconv2d_forward(d, x, w, bias, y, stream);
    g_w_img = nullptr;
    return rc;
}

// conv (3 -> 128, 5x5 s2) + GDN fused; gamma_packed / beta_packed from hesic_gdn_pack_params.
// LDS weight images of the two image-side MFMA kernels, built once per weight update instead of by every block of every launch.
//   kind 0 (g_a_conv1 + GDN, 3 -> 128 5x5 s2): 64 KB = conv weights [128 co][16 slots ^ (co & 15)] of 8 bf16 (row r = ci*5+ky, 8 kx slots)
//                                              followed by gamma' in the K-permuted order the GDN contraction reads it
//   kind 1 (g_s_conv4, 128 -> 3 transposed):   24 KB = [96 (tap*3+co, padded)][16 slots ^ (n & 15)] of 8 cin
__global__ void sconv_pack_n2w_image_kernel(const float* __restrict__ w, const unsigned char* __restrict__ gamma_packed, unsigned char* __restrict__ img) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // 2048 weight slots + 2048 gamma slots
    if (idx < 2048) {
        const int co = idx >> 4, r = idx & 15, ci = r / 5, ky = r % 5;
        float v[8];
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) v[kx] = (r < 15 && kx < 5) ? w[((co * 3 + ci) * 5 + ky) * 5 + kx] : 0.f;
        *(u32x4*)(img + (co * 16 + (r ^ (co & 15))) * 16) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
    } else if (idx < 4096) {
        const int j = idx - 2048, row = j >> 4, q = j & 15, ks = q >> 1, h = q & 1;
        const unsigned char* srow = gamma_packed + row * 256;
        const u32x2 lo = *(const u32x2*)(srow + (((2 * ks) ^ (row & 15)) << 4) + 8 * h);
        const u32x2 hi = *(const u32x2*)(srow + (((2 * ks + 1) ^ (row & 15)) << 4) + 8 * h);
        *(u32x4*)(img + 32768 + row * 256 + ((q ^ (row & 15)) << 4)) = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
}

__global__ void sconv_pack_w2n_image_kernel(const float* __restrict__ w, unsigned char* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // 96 rows x 16 slots
    if (i >= 96 * 16) return;
    const int n = i / 16, sl = i % 16, co = n % 3, tap = n / 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = n < 75 ? w[((sl * 8 + e) * 3 + co) * 25 + tap] : 0.f;       // w[ci][co][ky][kx]
    *(u32x4*)(img + (n * 16 + (sl ^ (n & 15))) * 16) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
}

// kind 2 (round 5): the same panel with ERROR-FEEDBACK rounding per output phase.  g_s_conv4 writes the reconstruction itself: at a trained
// operating point (31 dB) rounding its 9600 weights to 16 bits moved the PSNR by +5.9e-4 dB on its own -- more than the other three
// synthesis layers and the 16-bit activation storage together (profiles/scripts/synthesis_precision.py) -- against a 1e-3 bar.  An output pixel
// of phase (py, px) sums the taps (py + 2 jy, px + 2 jx) over NEIGHBOURING input pixels of a spatially smooth IGDN output, so the part of the
// rounding error that matters is the SUM of a phase's tap errors per (cin, cout): each tap is rounded after adding the error the previous tap
// of its phase left (serpentine inside the phase), which keeps that sum below half an ulp of one weight.
__global__ void sconv_pack_w2n_image_shaped_kernel(const float* __restrict__ w, unsigned char* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // one (cin, cout) pair per thread; rows 75 .. 95 of the panel are zero
    if (i < 21 * 16) *(u32x4*)(img + ((75 + i / 16) * 16 + ((i % 16) ^ ((75 + i / 16) & 15))) * 16) = u32x4{0u, 0u, 0u, 0u};
    if (i >= 128 * 3) return;
    const int ci = i / 3, co = i - ci * 3, sl = ci >> 3, e = ci & 7;
    const float* wp = w + (ci * 3 + co) * 25;                       // w[ci][co][ky][kx]
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            float carry = 0.f;
            int row = 0;
            for (int ky = py; ky < 5; ky += 2, ++row)
                for (int s = 0; s < (5 - px + 1) / 2; ++s) {
                    const int nx = (5 - px + 1) / 2;
                    const int kx = px + 2 * ((row & 1) ? nx - 1 - s : s);              // serpentine: consecutive taps are neighbours
                    const int tap = ky * 5 + kx, n = tap * 3 + co;
                    const float tgt = wp[tap] + carry;
                    const h16_t q = f2h(tgt);
                    carry = tgt - h2f(q);
                    *(h16_t*)(img + (n * 16 + (sl ^ (n & 15))) * 16 + e * 2) = q;
                }
        }
}

extern "C" int hesic_sconv_pack_weight_image(int kind, const float* w, const void* gamma_packed, void* image, void* stream) {
    HESIC_CHECK_ARG(w && image && (kind == 0 || kind == 1 || kind == 2) && (kind != 0 || gamma_packed), "sconv_pack_weight_image: bad arguments");
    if (kind == 0)
        hipLaunchKernelGGL(sconv_pack_n2w_image_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, (const unsigned char*)gamma_packed, (unsigned char*)image);
    else if (kind == 2)
        hipLaunchKernelGGL(sconv_pack_w2n_image_shaped_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)image);
    else
        hipLaunchKernelGGL(sconv_pack_w2n_image_kernel, dim3(6), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)image);
    HESIC_LAUNCH_RETURN("sconv_pack_weight_image");
}

static int sconv_gdn_launch(const hesic_sconv_desc* d, const void* x, const float* w, const float* bias, const void* gamma_packed,
                            const float* beta_packed, int inverse, void* y, void* y_pre, void* stream) {
    if (int e = check_desc(d, "sconv2d_gdn_forward")) return e;
    HESIC_CHECK_ARG(x && w && y && gamma_packed && beta_packed, "sconv2d_gdn_forward: null pointer");
    HESIC_CHECK_ARG(!d->transposed && d->Cin == 3 && d->Cout == 128 && d->KH == 5 && d->KW == 5 && d->stride == 2 && d->pad == 2 &&
                        d->y_dtype == HESIC_H16 && d->ys_c == 1 && d->act == HESIC_ACT_NONE && (d->ys_x % 8) == 0 && (d->ys_y % 8) == 0 &&
                        (d->ys_b % 8) == 0,
                    "sconv2d_gdn_forward: built for the 3 -> 128 5x5 stride-2 stage with bf16 NHWC output");
    SArgs a = make_args(d);
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    constexpr int n2w_dbg = 0;
    a.dbg = n2w_dbg;
    a.w_img = g_w_img;
    const size_t lds = 65536 + 1024 + 8 * 32 * (128 * 2 + 16);
    const int64_t tiles = (int64_t)((d->Wo + 15) / 16) * ((d->Ho + 1) / 2) * d->B;
    const unsigned grid = (unsigned)((tiles + 7) / 8 < 256 ? (tiles + 7) / 8 : 256);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)sconv_n2w_gdn_kernel<3, 5, 2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)sconv_n2w_gdn_kernel<3, 5, 2, h16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    // fp32 planes, unit pixel stride, even width, everything addressable with 32-bit byte offsets inside one image
    constexpr bool no_fast = false;              // A/B switch for profiling
    const bool fastx = !no_fast && d->x_dtype == HESIC_F32 && d->xs_x == 1 && d->W % 2 == 0 && tiles < (1ll << 30) &&
                       (2 * d->xs_c + (int64_t)(d->H + 4) * d->xs_y + d->W) * 4 < (1ll << 31) &&
                       ((int64_t)d->Ho * d->ys_y + (int64_t)d->Wo * d->ys_x) * 2 < (1ll << 31) && d->xs_c >= 0 && d->xs_y >= 0;
    if (fastx) {
        static bool attr_f = false;
        if (!attr_f) {
            (void)hipFuncSetAttribute((const void*)sconv_n2w_gdn_fast_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)sconv_n2w_gdn_fast_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_f = true;
        }
        const FastDiv fd_tx = make_fastdiv((uint32_t)((d->Wo + 15) / 16)), fd_ty = make_fastdiv((uint32_t)((d->Ho + 1) / 2));
        if (inverse)
            hipLaunchKernelGGL((sconv_n2w_gdn_fast_kernel<1>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a, (const h16_t*)gamma_packed, beta_packed, (h16_t*)y_pre, fd_tx, fd_ty);
        else
            hipLaunchKernelGGL((sconv_n2w_gdn_fast_kernel<0>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a, (const h16_t*)gamma_packed, beta_packed, (h16_t*)y_pre, fd_tx, fd_ty);
    } else if (d->x_dtype == HESIC_H16)
        hipLaunchKernelGGL((sconv_n2w_gdn_kernel<3, 5, 2, h16_t>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a, (const h16_t*)gamma_packed, beta_packed, inverse, (h16_t*)y_pre);
    else
        hipLaunchKernelGGL((sconv_n2w_gdn_kernel<3, 5, 2, float>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a, (const h16_t*)gamma_packed, beta_packed, inverse, (h16_t*)y_pre);
    HESIC_LAUNCH_RETURN("sconv2d_gdn_forward");
}

extern "C" int hesic_sconv2d_gdn_forward(const hesic_sconv_desc* d, const void* x, const float* w, const float* bias,
                                         const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* stream) {
    return sconv_gdn_launch(d, x, w, bias, gamma_packed, beta_packed, inverse, y, nullptr, stream);
}

extern "C" int hesic_sconv2d_gdn_forward_train(const hesic_sconv_desc* d, const void* x, const float* w, const float* bias,
                                               const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* y_pre,
                                               void* stream) {
    HESIC_CHECK_ARG(y_pre, "sconv2d_gdn_forward_train: null pointer");
    return sconv_gdn_launch(d, x, w, bias, gamma_packed, beta_packed, inverse, y, y_pre, stream);
}

// dx of y = op(x): the opposite op (conv <-> transposed conv) applied to dy with the same weight tensor:
// a Conv2d weight (Cout,Cin,k,k) read as a ConvTranspose2d weight maps Cout -> Cin, and vice versa.
extern "C" int hesic_sconv2d_dgrad(const hesic_sconv_desc* d, const void* dy, const float* w, void* dx, void* stream) {
    if (int e = check_desc(d, "sconv2d_dgrad")) return e;
    HESIC_CHECK_ARG(dy && w && dx, "sconv2d_dgrad: null pointer");
    SArgs a = make_args(d);
    a.transposed = !d->transposed;
    a.Cin = d->Cout; a.Cout = d->Cin; a.H = d->Ho; a.W = d->Wo; a.Ho = d->H; a.Wo = d->W;
    a.x_dtype = d->y_dtype; a.y_dtype = d->x_dtype; a.act = HESIC_ACT_NONE;
    a.xs_b = d->ys_b; a.xs_c = d->ys_c; a.xs_y = d->ys_y; a.xs_x = d->ys_x;
    a.ys_b = d->xs_b; a.ys_c = d->xs_c; a.ys_y = d->xs_y; a.ys_x = d->xs_x;
    a.x = dy; a.w = w; a.bias = nullptr; a.y = dx;
    // when the forward op is a stride-s conv whose input size is not "Ho*s", the transposed op's natural
    // output is smaller than H: the generic kernel handles any (H, Ho) pair by pure index arithmetic.
    const bool natural = a.transposed ? (a.Ho == (a.H - 1) * a.stride - 2 * a.pad + a.KH + a.stride - 1 &&
                                         a.Wo == (a.W - 1) * a.stride - 2 * a.pad + a.KW + a.stride - 1)
                                      : true;
    if (natural) {
        launch_forward(a, (hipStream_t)stream);
    } else {
        const int64_t total = (int64_t)a.B * a.Ho * a.Wo * ((a.Cout + 3) / 4);
        hipLaunchKernelGGL(sconv_generic_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, a);
    }
    HESIC_LAUNCH_RETURN("sconv2d_dgrad");
}

extern "C" int hesic_sconv2d_gdn_forward_prepacked(const hesic_sconv_desc* d, const void* x, const float* w, const void* w_image, const float* bias,
                                                   const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* y_pre,
                                                   void* stream) {
    g_w_img = w_image;
    const int rc = sconv_gdn_launch(d, x, w, bias, gamma_packed, beta_packed, inverse, y, y_pre, stream);
    g_w_img = nullptr;
    return rc;
}
