// Weight gradients of the convolutions (autograd's wgrad of nn.Conv2d / nn.ConvTranspose2d as used by
// conv()/deconv(), compressai/models/utils.py:104-118), bias gradients, and the narrow-channel variants.
//
// Wide layers (Cin % 8 == 0, Cout % 8 == 0):
//   dWp[tap][co][ci] = sum_q DY[pa(q,tap)][co] * X[pb(q,tap)][ci]
//   conv:        q over the output grid, pa = q,           pb = q*s + (k-p)   (zero outside)
//   transposed:  q over the input grid,  pa = q*s + (k-p), pb = q
// The contraction index (pixels) is the strided one in NHWC memory for BOTH operands, so each thread
// loads an 8x8 (bf16) / 4x4 (fp32) pixel x channel block, transposes it in registers and stores
// channel-major rows into LDS; from there the loop is the same MFMA tile loop as the forward kernel.
// Pixels are split over blocks (split-K); partial tiles go to a workspace and are summed in a fixed
// order by a second kernel, so the result is deterministic.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int TC = 128;      // channel tile on both sides

struct WgArgs {
    const void* x; const void* dy; float* out;     // out: workspace [split][tap][Cout][Cin] or dw itself
    int B, H, W, Cin, x_ps, x_co, Ho, Wo, Cout, y_ps, y_co;
    int QH, QW, transposed, stride, pad, KW, in_abs;
    int64_t Q, chunk;
    int co_tiles, ci_tiles, ntaps, nsplit;
    int8_t tap_id[25];
};

template <typename T> struct WC;
template <> struct WC<bf16_t> { static constexpr int BK = 64, PB = 8, CB = 8; };
template <> struct WC<float> { static constexpr int BK = 32, PB = 4, CB = 4; };

template <typename T>
__device__ __forceinline__ int w_off(int row, int slot) {   // 128-byte rows, 8 slots of 16 B
    return (row * 8 + (slot ^ ((row >> 1) & 7))) * 16;
}

// 8 pixels x 8 channels (bf16) -> 8 channel rows of 8 pixels
__device__ __forceinline__ void transpose8(const u32x4 (&in)[8], u32x4 (&out)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = in[2 * j][c >> 1], b = in[2 * j + 1][c >> 1];
            // low half <- pixel 2j, high half <- pixel 2j+1, both of 16-bit lane (c & 1)
            o[j] = (c & 1) ? __builtin_amdgcn_perm(b, a, 0x07060302u) : __builtin_amdgcn_perm(b, a, 0x05040100u);
        }
        out[c] = u32x4{o[0], o[1], o[2], o[3]};
    }
}
__device__ __forceinline__ void transpose4(const u32x4 (&in)[4], u32x4 (&out)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = u32x4{in[0][c], in[1][c], in[2][c], in[3][c]};
}

template <typename T>
__global__ __launch_bounds__(NT) void wgrad_kernel(const WgArgs a) {
    using K = WC<T>;
    constexpr int BK = K::BK, PB = K::PB, CB = K::CB;
    constexpr int OPB = TC * 128;                 // bytes of one operand stage (128 rows x 128 B)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * OPB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int bid = blockIdx.x;
    const int split = bid % a.nsplit; bid /= a.nsplit;
    const int cit = bid % a.ci_tiles; bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles; bid /= a.co_tiles;
    const int tapi = bid;
    const int tap = a.tap_id[tapi];
    const int ky = tap / a.KW, kx = tap % a.KW;
    const int sh_y = ky - a.pad, sh_x = kx - a.pad;
    const int co0 = cot * TC, ci0 = cit * TC;
    const int64_t q_begin = split * a.chunk;
    const int64_t q_end = (q_begin + a.chunk < a.Q) ? q_begin + a.chunk : a.Q;

    const T* xg = (const T*)a.x;
    const T* dg = (const T*)a.dy;

    // staging roles. bf16: threads 0-127 stage DY, 128-255 stage X (one 8x8 block each).
    // fp32: every thread stages one 4x4 block of DY and one of X.
    constexpr int NBLK = (sizeof(T) == 2) ? 1 : 2;
    int opnd[NBLK], pg[NBLK], chg[NBLK];
    if constexpr (sizeof(T) == 2) {
        opnd[0] = tid >> 7; pg[0] = tid & 7; chg[0] = (tid & 127) >> 3;
    } else {
        opnd[0] = 0; opnd[1] = 1; pg[0] = pg[1] = tid & 7; chg[0] = chg[1] = tid >> 3;
    }
    u32x4 regs[NBLK][PB];

    auto load_tile = [&](int64_t q0) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const bool isx = opnd[nb] == 1;
            const int cbase = (isx ? ci0 : co0) + chg[nb] * CB;
            const bool cok = cbase < (isx ? a.Cin : a.Cout);
            // which operand carries the shift: conv -> X shifted; transposed -> DY shifted
            const bool shifted = isx ? !a.transposed : a.transposed;
            const int GH = isx ? a.H : a.Ho, GW = isx ? a.W : a.Wo;
            const int ps = isx ? a.x_ps : a.y_ps, cof = isx ? a.x_co : a.y_co;
            const T* base = isx ? xg : dg;
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int64_t q = q0 + pg[nb] * PB + i;
                u32x4 v = u32x4{0, 0, 0, 0};
                if (q < q_end && cok) {
                    const int qx = q % a.QW;
                    const int64_t r = q / a.QW;
                    const int qy = r % a.QH;
                    const int b = r / a.QH;
                    int py = qy, px = qx;
                    bool ok = true;
                    if (shifted) {
                        py = qy * a.stride + sh_y; px = qx * a.stride + sh_x;
                        ok = (unsigned)py < (unsigned)GH && (unsigned)px < (unsigned)GW;
                    }
                    if (ok) v = *(const u32x4*)(base + (((int64_t)b * GH + py) * GW + px) * ps + cof + cbase);
                }
                regs[nb][i] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            unsigned char* dst = smem + buf * 2 * OPB + opnd[nb] * OPB;
            u32x4 t[PB];
            if constexpr (sizeof(T) == 2) {
                transpose8(regs[nb], t);
                if (a.in_abs && opnd[nb] == 1) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) t[c] = u32x4{t[c].x & 0x7fff7fffu, t[c].y & 0x7fff7fffu, t[c].z & 0x7fff7fffu, t[c].w & 0x7fff7fffu};
                }
            } else {
                transpose4(regs[nb], t);
                if (a.in_abs && opnd[nb] == 1) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) t[c] = u32x4{t[c].x & 0x7fffffffu, t[c].y & 0x7fffffffu, t[c].z & 0x7fffffffu, t[c].w & 0x7fffffffu};
                }
            }
#pragma unroll
            for (int c = 0; c < CB; ++c) *(u32x4*)(dst + w_off<T>(chg[nb] * CB + c, pg[nb])) = t[c];
        }
    };

    const int wm = wave & 1, wn = wave >> 1;        // co half, ci half
    const int frow = lane & 31, fh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int64_t nsteps = (q_end - q_begin + BK - 1) / BK;
    if (nsteps > 0) {
        load_tile(q_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int64_t step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) load_tile(q_begin + (step + 1) * BK);
        const unsigned char* ds = smem + buf * 2 * OPB;
        const unsigned char* xs = ds + OPB;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 df[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) df[i] = *(const bf16x8*)(ds + w_off<T>(wm * 64 + i * 32 + frow, ks * 2 + fh));
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[j] = *(const bf16x8*)(xs + w_off<T>(wn * 64 + j * 32 + frow, ks * 2 + fh));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[i], xf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x4 df[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) df[i] = *(const f32x4*)(ds + w_off<T>(wm * 64 + i * 32 + frow, fh * 4 + s));
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[j] = *(const f32x4*)(xs + w_off<T>(wn * 64 + j * 32 + frow, fh * 4 + s));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(df[i][e], xf[j][e], acc[i][j], 0, 0, 0);
            }
        }
        if (step + 1 < nsteps) store_tile(buf ^ 1);
        __syncthreads();
    }
    // C[i = co][j = ci]: col = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* out = a.out + ((int64_t)split * a.ntaps + tapi) * a.Cout * a.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wn * 64 + j * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (co < a.Cout && ci < a.Cin) out[(int64_t)co * a.Cin + ci] = acc[i][j][r];
            }
        }
}

// dw[tap_id[t]][..] = sum_s ws[s][t][..]; dead taps (masked conv) are zero-filled by the host memset
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int ntaps, int64_t per_tap,
                                    const WgArgs a) {
    const int64_t n = (int64_t)ntaps * per_tap;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += ws[k * n + i];
        const int t = i / per_tap;
        dw[(int64_t)a.tap_id[t] * per_tap + (i - t * per_tap)] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ db, int64_t P, int C, int ps, int co,
                                                     int64_t rows_per_block) {
    // threads over channels (coalesced), block over a row range; one atomic per (block, channel)
    const int64_t r0 = blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < P ? r0 + rows_per_block : P;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int64_t r = r0; r < r1; ++r) acc += elem<T>::ld(dy + r * ps + co + c);
        atomicAdd(db + c, acc);
    }
}

// ------------------------------------------------------------------ narrow-channel weight gradient
struct SWArgs {
    const void* x; const void* dy; float* dw; float* db;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed, x_dtype, y_dtype;
    int64_t xs_b, xs_c, xs_y, xs_x, ys_b, ys_c, ys_y, ys_x;
    int64_t q_per_block;
};

// thread = one weight element (PyTorch layout index), block column = a chunk of the q grid.
__global__ __launch_bounds__(256) void sconv_wgrad_generic_kernel(const SWArgs a) {
    const int64_t nw = (int64_t)a.Cout * a.Cin * a.KH * a.KW;
    const int64_t wi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (wi >= nw) return;
    int64_t r = wi;
    const int kx = r % a.KW; r /= a.KW;
    const int ky = r % a.KH; r /= a.KH;
    int co, ci;
    if (a.transposed) { co = r % a.Cout; ci = r / a.Cout; } else { ci = r % a.Cin; co = r / a.Cin; }
    // q runs over the grid of the un-shifted operand: conv -> output grid, transposed -> input grid
    const int QH = a.transposed ? a.H : a.Ho, QW = a.transposed ? a.W : a.Wo;
    const int64_t Q = (int64_t)a.B * QH * QW;
    const int64_t q0 = blockIdx.y * a.q_per_block;
    const int64_t q1 = q0 + a.q_per_block < Q ? q0 + a.q_per_block : Q;
    float acc = 0.f;
    for (int64_t q = q0; q < q1; ++q) {
        const int qx = q % QW;
        const int64_t t = q / QW;
        const int qy = t % QH;
        const int b = t / QH;
        int iy, ix, oy, ox;
        if (!a.transposed) { oy = qy; ox = qx; iy = qy * a.stride - a.pad + ky; ix = qx * a.stride - a.pad + kx;
            if ((unsigned)iy >= (unsigned)a.H || (unsigned)ix >= (unsigned)a.W) continue;
        } else { iy = qy; ix = qx; oy = qy * a.stride - a.pad + ky; ox = qx * a.stride - a.pad + kx;
            if ((unsigned)oy >= (unsigned)a.Ho || (unsigned)ox >= (unsigned)a.Wo) continue;
        }
        acc += ld_any(a.x, b * a.xs_b + ci * a.xs_c + iy * a.xs_y + ix * a.xs_x, a.x_dtype) *
               ld_any(a.dy, b * a.ys_b + co * a.ys_c + oy * a.ys_y + ox * a.ys_x, a.y_dtype);
    }
    atomicAdd(a.dw + wi, acc);
}

__global__ __launch_bounds__(256) void sconv_dbias_kernel(const SWArgs a) {
    __shared__ float red[4];
    const int co = blockIdx.y;
    const int64_t n = (int64_t)a.B * a.Ho * a.Wo;
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = i % a.Wo, oy = (i / a.Wo) % a.Ho, b = i / ((int64_t)a.Wo * a.Ho);
        acc += ld_any(a.dy, b * a.ys_b + co * a.ys_c + oy * a.ys_y + ox * a.ys_x, a.y_dtype);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.db + co, red[0] + red[1] + red[2] + red[3]);
}

// ------------------------------------------------------------------------------ GDN backward (v1)
// n_i = beta'_i + sum_j gamma'_ij x_j^2.  GDN: y = x n^-1/2, dn_i = -1/2 g_i x_i n_i^-3/2;  IGDN: y = x n^1/2,
// dn_i = +1/2 g_i x_i n_i^-1/2.   dx_j = g_j n_j^(-+1/2) + 2 x_j sum_i gamma'_ij dn_i ;
// dgamma'_ij = sum_p dn_i x_j^2 ; dbeta'_i = sum_p dn_i ; raw-parameter chain: theta' = max(theta,b)^2 - 2^-36.
constexpr float kPedestal = 1.0f / 68719476736.0f;
constexpr float kGammaBound = 1.0f / 262144.0f;
__device__ __forceinline__ float reparam(float v, float bound) { const float t = fmaxf(v, bound); return t * t - kPedestal; }

// pass 1: dn[p][i] (fp32 workspace) and the direct term of dx
__global__ void gdn_bwd_dn_kernel(const void* __restrict__ x, const void* __restrict__ gy, const float* __restrict__ beta,
                                  const float* __restrict__ gamma, float* __restrict__ dn, float* __restrict__ dx0, int64_t P, int C,
                                  int inverse, float beta_bound, int dtype) {
    const int64_t n = P * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        const int64_t p = i / C;
        float norm = reparam(beta[c], beta_bound);
        for (int j = 0; j < C; ++j) {
            const float xv = ld_any(x, p * C + j, dtype);
            norm += reparam(gamma[(int64_t)c * C + j], kGammaBound) * xv * xv;
        }
        const float xv = ld_any(x, i, dtype), g = ld_any(gy, i, dtype);
        if (inverse) {
            const float sq = sqrtf(norm);
            dn[i] = 0.5f * g * xv / sq;
            dx0[i] = g * sq;
        } else {
            const float rs = rsqrtf(norm);
            dn[i] = -0.5f * g * xv * rs * rs * rs;
            dx0[i] = g * rs;
        }
    }
}
// pass 2: dx_j = dx0_j + 2 x_j sum_i gamma'_ij dn_i
__global__ void gdn_bwd_dx_kernel(const void* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dn,
                                  const float* __restrict__ dx0, void* __restrict__ dx, int64_t P, int C, int dtype) {
    const int64_t n = P * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = i % C;
        const int64_t p = i / C;
        float s = 0.f;
        for (int k = 0; k < C; ++k) s += reparam(gamma[(int64_t)k * C + j], kGammaBound) * dn[p * C + k];
        st_any(dx, i, dtype, dx0[i] + 2.f * ld_any(x, i, dtype) * s);
    }
}
// pass 3: dgamma'_ij, dbeta'_i over a pixel chunk; thread = (i, j); then chain to the raw parameters
__global__ __launch_bounds__(256) void gdn_bwd_param_kernel(const void* __restrict__ x, const float* __restrict__ dn,
                                                            float* __restrict__ dgp, float* __restrict__ dbp, int64_t P, int C,
                                                            int dtype, int64_t rows_per_block) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= (int64_t)C * C) return;
    const int j = e % C, i = e / C;
    const int64_t r0 = blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < P ? r0 + rows_per_block : P;
    float acc = 0.f, accb = 0.f;
    for (int64_t p = r0; p < r1; ++p) {
        const float d = dn[p * C + i], xv = ld_any(x, p * C + j, dtype);
        acc += d * xv * xv;
        accb += d;
    }
    atomicAdd(dgp + e, acc);
    if (j == 0) atomicAdd(dbp + i, accb);
}
__global__ void gdn_bwd_chain_kernel(const float* __restrict__ beta, const float* __restrict__ gamma, const float* __restrict__ dgp,
                                     const float* __restrict__ dbp, float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                     float beta_bound) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < C * C) {
        const float th = gamma[e], g = dgp[e] * 2.f * fmaxf(th, kGammaBound);
        dgamma[e] = (th >= kGammaBound || g < 0.f) ? g : 0.f;
    }
    if (e < C) {
        const float th = beta[e], g = dbp[e] * 2.f * fmaxf(th, beta_bound);
        dbeta[e] = (th >= beta_bound || g < 0.f) ? g : 0.f;
    }
}

int pick_splits(int64_t Q, int bk, int tiles) {
    // aim at ~1500 blocks, at least 4 K-steps per block
    int64_t s = (1536 + tiles - 1) / tiles;
    const int64_t maxs = Q / (4 * bk) > 0 ? Q / (4 * bk) : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

int fill_args(const hesic_conv_desc* d, WgArgs& a) {
    memset(&a, 0, sizeof(a));
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_ps = d->x_pix_stride; a.x_co = d->x_c_off;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.y_ps = d->y_pix_stride; a.y_co = d->y_c_off;
    a.transposed = d->transposed; a.stride = d->stride; a.pad = d->pad; a.KW = d->KW; a.in_abs = d->in_abs;
    a.QH = d->transposed ? d->H : d->Ho; a.QW = d->transposed ? d->W : d->Wo;
    a.Q = (int64_t)a.B * a.QH * a.QW;
    a.co_tiles = (d->Cout + TC - 1) / TC; a.ci_tiles = (d->Cin + TC - 1) / TC;
    int n = 0;
    for (int t = 0; t < d->KH * d->KW; ++t)
        if (!d->tap_mask_lo || ((d->tap_mask_lo >> t) & 1)) a.tap_id[n++] = (int8_t)t;
    a.ntaps = n;
    const int bk = d->dtype == HESIC_BF16 ? WC<bf16_t>::BK : WC<float>::BK;
    a.nsplit = pick_splits(a.Q, bk, n * a.co_tiles * a.ci_tiles);
    a.chunk = ((a.Q + a.nsplit - 1) / a.nsplit + bk - 1) / bk * bk;
    a.nsplit = (int)((a.Q + a.chunk - 1) / a.chunk);
    return 0;
}

}  // namespace

extern "C" int64_t hesic_conv2d_wgrad_ws_bytes(const hesic_conv_desc* d) {
    if (!d || d->KH * d->KW > 25) return 0;
    WgArgs a;
    fill_args(d, a);
    return (int64_t)a.nsplit * a.ntaps * d->Cout * d->Cin * 4;
}

extern "C" int hesic_conv2d_wgrad(const hesic_conv_desc* d, const void* x, const void* dy, float* dw_packed, float* dbias,
                                  void* ws, int64_t ws_bytes, void* stream) {
    HESIC_CHECK_ARG(d && x && dy && dw_packed, "conv2d_wgrad: null pointer");
    const int ce = d->dtype == HESIC_BF16 ? 8 : 4;
    HESIC_CHECK_ARG(d->Cin % ce == 0 && d->Cout % ce == 0 && d->x_pix_stride % ce == 0 && d->y_pix_stride % ce == 0 &&
                        d->x_c_off % ce == 0 && d->y_c_off % ce == 0,
                    "conv2d_wgrad: channels must be multiples of %d", ce);
    HESIC_CHECK_ARG(d->KH * d->KW <= 25, "conv2d_wgrad: at most 25 taps");
    WgArgs a;
    fill_args(d, a);
    const int64_t need = (int64_t)a.nsplit * a.ntaps * d->Cout * d->Cin * 4;
    HESIC_CHECK_ARG(ws && ws_bytes >= need, "conv2d_wgrad: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)need);
    hipStream_t st = (hipStream_t)stream;
    a.x = x; a.dy = dy; a.out = (float*)ws;
    const int64_t blocks = (int64_t)a.ntaps * a.co_tiles * a.ci_tiles * a.nsplit;
    if (d->dtype == HESIC_BF16) hipLaunchKernelGGL(wgrad_kernel<bf16_t>, dim3((unsigned)blocks), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(wgrad_kernel<float>, dim3((unsigned)blocks), dim3(NT), 0, st, a);
    const int64_t per_tap = (int64_t)d->Cout * d->Cin;
    if (a.ntaps < d->KH * d->KW) hipMemsetAsync(dw_packed, 0, (size_t)d->KH * d->KW * per_tap * 4, st);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(a.ntaps * per_tap, 256)), dim3(256), 0, st, (const float*)ws, dw_packed,
                       a.nsplit, a.ntaps, per_tap, a);
    if (dbias) {
        hipMemsetAsync(dbias, 0, (size_t)d->Cout * 4, st);
        const int64_t P = (int64_t)d->B * d->Ho * d->Wo;
        const int64_t rpb = P / 1024 > 0 ? (P + 1023) / 1024 : 1;
        const unsigned g = (unsigned)((P + rpb - 1) / rpb);
        if (d->dtype == HESIC_BF16)
            hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)dy, dbias, P, d->Cout, d->y_pix_stride, d->y_c_off, rpb);
        else
            hipLaunchKernelGGL(colsum_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dy, dbias, P, d->Cout, d->y_pix_stride, d->y_c_off, rpb);
    }
    HESIC_LAUNCH_RETURN("conv2d_wgrad");
}

extern "C" int hesic_sconv2d_wgrad(const hesic_sconv_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* stream) {
    HESIC_CHECK_ARG(d && x && dy && dw, "sconv2d_wgrad: null pointer");
    SWArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.dw = dw; a.db = dbias;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad = d->pad; a.transposed = d->transposed; a.x_dtype = d->x_dtype; a.y_dtype = d->y_dtype;
    a.xs_b = d->xs_b; a.xs_c = d->xs_c; a.xs_y = d->xs_y; a.xs_x = d->xs_x;
    a.ys_b = d->ys_b; a.ys_c = d->ys_c; a.ys_y = d->ys_y; a.ys_x = d->ys_x;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nw = (int64_t)d->Cout * d->Cin * d->KH * d->KW;
    const int64_t Q = (int64_t)d->B * (d->transposed ? d->H * d->W : d->Ho * d->Wo);
    const int gx = (int)((nw + 255) / 256);
    int gy = (int)(2048 / gx > 0 ? 2048 / gx : 1);
    if (gy > Q) gy = (int)Q;
    a.q_per_block = (Q + gy - 1) / gy;
    gy = (int)((Q + a.q_per_block - 1) / a.q_per_block);
    hipMemsetAsync(dw, 0, (size_t)nw * 4, st);
    hipLaunchKernelGGL(sconv_wgrad_generic_kernel, dim3(gx, gy), dim3(256), 0, st, a);
    if (dbias) {
        hipMemsetAsync(dbias, 0, (size_t)d->Cout * 4, st);
        hipLaunchKernelGGL(sconv_dbias_kernel, dim3(64, d->Cout), dim3(256), 0, st, a);
    }
    HESIC_LAUNCH_RETURN("sconv2d_wgrad");
}

extern "C" int64_t hesic_gdn_backward_ws_bytes(int64_t P, int C) { return (2 * P * C + (int64_t)C * C + C) * 4; }

extern "C" int hesic_gdn_backward(const void* x, const void* dy, const float* beta, const float* gamma, void* dx, float* dbeta,
                                  float* dgamma, void* ws, int64_t P, int C, int inverse, float beta_min, int dtype, void* stream) {
    HESIC_CHECK_ARG(x && dy && beta && gamma && dx && dbeta && dgamma && ws && P > 0 && C > 0, "gdn_backward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float bound = sqrtf(beta_min + kPedestal);
    float* dn = (float*)ws;
    float* dx0 = dn + P * C;
    float* dgp = dx0 + P * C;
    float* dbp = dgp + (int64_t)C * C;
    hipMemsetAsync(dgp, 0, ((size_t)C * C + C) * 4, st);
    hipLaunchKernelGGL(gdn_bwd_dn_kernel, dim3(grid_for(P * C, 256)), dim3(256), 0, st, x, dy, beta, gamma, dn, dx0, P, C, inverse, bound, dtype);
    hipLaunchKernelGGL(gdn_bwd_dx_kernel, dim3(grid_for(P * C, 256)), dim3(256), 0, st, x, gamma, dn, dx0, dx, P, C, dtype);
    const int gx = (C * C + 255) / 256;
    int gy = 2048 / gx > 0 ? 2048 / gx : 1;
    if (gy > P) gy = (int)P;
    const int64_t rpb = (P + gy - 1) / gy;
    gy = (int)((P + rpb - 1) / rpb);
    hipLaunchKernelGGL(gdn_bwd_param_kernel, dim3(gx, gy), dim3(256), 0, st, x, dn, dgp, dbp, P, C, dtype, rpb);
    hipLaunchKernelGGL(gdn_bwd_chain_kernel, dim3(gx), dim3(256), 0, st, beta, gamma, dgp, dbp, dgamma, dbeta, C, bound);
    HESIC_LAUNCH_RETURN("gdn_backward");
}
