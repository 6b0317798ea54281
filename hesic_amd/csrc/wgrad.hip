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
#include <stdlib.h>

#include <type_traits>
#include <vector>
#include <algorithm>

#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int TC = 128;      // channel tile on both sides

// hipMemsetAsync is NOT used in this library: recorded into a HIP graph (torch.cuda.graph) its memset node did not
// reliably precede the kernels that accumulate into the buffer on replays >= 1 (measured on ROCm 7.2 / MI355X: bias
// gradients summed on top of the previous replay's values).  A plain kernel keeps the stream order in every mode.
__global__ void zero_f32_kernel(float* __restrict__ p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
static inline void zero_async(float* p, int64_t n, hipStream_t st) {
    if (n <= 0) return;
    const int64_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, p, n);
}

struct WgArgs {
    const void* x; const void* dy; float* out;     // out: workspace [split][tap][Cout][Cin] or dw itself
    int B, H, W, Cin, x_ps, x_co, Ho, Wo, Cout, y_ps, y_co;
    int QH, QW, transposed, stride, pad, KW, in_abs, in_sq;
    int64_t Q, chunk;
    int co_tiles, ci_tiles, ntaps, nsplit;
    int rowk;                                      // wgrad_row_kernel takes this layer (fill_args): blocks = KH x tiles x nsplit
    // split-K workspace layout: 0 = [split][tap][Cout][Cin] (rounds 1-4; the reduce kernels of the packed-layout API and the 1x1 GEMM users
    // read it), 1 = [tap][Cout][split][Cin] (round 5, the direct / partial / finish_batched route): the slices of one (tap, cout) row are
    // ADJACENT 512-byte rows, so the finishing pass streams nsplit * Cin * 4 contiguous bytes per row instead of nsplit 128-byte runs that
    // lie ntaps * Cout * Cin * 4 = 1.6 MB apart (365 us per step for ~1 GB of partials = 2.9 TB/s, profiles/r05_d_train_step_timeline.txt)
    int ws_layout;
    int8_t tap_id[25];
    // bias gradient on the matrix cores (wgrad_tr_kernel only): the blocks of the taps listed in b_tap (indices into the live taps)
    // with ci tile 0 also form the column sums of their dY slice -- one more MFMA per cout fragment and k-step against a fragment of
    // ones -- and leave them in bias_part[split][nb_taps][Cout].  A conv reads every dY pixel in every tap (one tap is listed); a
    // transposed conv reads dY at q * stride + k - pad: the stride^2 taps with k - pad in [0, stride) cover every pixel exactly once.
    float* bias_part; int nb_taps; int8_t b_tap[4];
};

// element offset of row (split, live tap index, cout) of the split-K workspace
__device__ __host__ __forceinline__ int64_t ws_row(const WgArgs& a, int split, int tapi, int co) {
    return a.ws_layout ? (((int64_t)tapi * a.Cout + co) * a.nsplit + split) * a.Cin : (((int64_t)split * a.ntaps + tapi) * a.Cout + co) * a.Cin;
}

template <typename T> struct WC;
template <> struct WC<h16_t> { static constexpr int BK = 64, PB = 8, CB = 8; };
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
                if (a.in_sq && opnd[nb] == 1) {            // X operand squared on the fly (GDN: dgamma' = dn^T x^2)
#pragma unroll
                    for (int c = 0; c < PB; ++c) {
                        uint32_t* w4 = (uint32_t*)&t[c];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float lo = h2f_lo(w4[q]), hi = h2f_hi(w4[q]);
                            w4[q] = pack_h2(lo * lo, hi * hi);
                        }
                    }
                }
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
                h16x8 df[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) df[i] = *(const h16x8*)(ds + w_off<T>(wm * 64 + i * 32 + frow, ks * 2 + fh));
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[j] = *(const h16x8*)(xs + w_off<T>(wn * 64 + j * 32 + frow, ks * 2 + fh));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16_h16(df[i], xf[j], acc[i][j], 0, 0, 0);
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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wn * 64 + j * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (co < a.Cout && ci < a.Cin) a.out[ws_row(a, split, tapi, co) + ci] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------- bf16 fast path: LDS-DMA staging + transpose reads
// Same contraction as wgrad_kernel, but the [pixel][channel] tiles of dY and X go global -> LDS in their natural NHWC
// layout with global_load_lds_dwordx4 (no register staging, no in-register transposes) and the MFMA operands, which
// need 8 consecutive PIXELS of one channel per lane, come out of LDS through ds_read_b64_tr_b16: a 16-lane group reads
// a [4 pixels][16 channels] block and every lane receives one channel's 4 pixels.  The 16-byte slots of a 256-byte
// pixel row are XOR-ed with ((pixel & 3) << 2) on the DMA source side: a transposing read is served 32 lanes (two 16-lane groups = 4
// pixel rows x 64 bytes) per LDS cycle, and the four rows must land on four different 64-byte bank ranges.  (Rounds 1-3 shifted by 1:
// rows 0/1 and 2/3 of a half shared their banks -- SQ_LDS_BANK_CONFLICT = 50 % of the kernel's LDS cycles, profiles/r04_e_pmc_sq_train.json.)

typedef __attribute__((ext_vector_type(4))) short s16x4;
#ifndef WG_SWZ
#define WG_SWZ 2
#endif
#ifndef WGRAD_RING_DEFAULT
#define WGRAD_RING_DEFAULT 0
#endif

struct WgTrArgs {
    WgArgs w;
    float* zero_me; int zero_n;      // optional: block 0 clears this array (the bias gradient the next launch accumulates into)
    FastDiv dqw, dqh;
    int fastq;          // QW % 16 == 0 (and chunk % 64 == 0): the 16 pixels a wave stages per step share one image row
};

// BK pixels per stage, a ring of NST stages: NST - 1 stages are in flight while one is consumed.  <64, 2> (64 KB, two blocks per CU) is the
// form of rounds 2-4; <32, 5> (80 KB) keeps twice the bytes in flight per CU for the same two blocks (round 4: the kernel sits parked at
// its vmcnt / barrier half of its cycles with ONE 32 KB stage per block in flight).
// hw_bid / nb: this block's hardware id inside its job and the job's block count (the launch's own blockIdx / gridDim for the one-job form;
// its 8-aligned range of a batched launch, whose surplus ids the caller has already sent home)
template <int BK, int NST, bool SPREAD = false>
__device__ __forceinline__ void wgrad_tr_body(const WgTrArgs& A, const int hw_bid, const int nb) {
    const WgArgs& a = A.w;
    constexpr int TILE = BK * 256, STAGE = 2 * TILE;       // BK pixels x 128 channels x 2 B per operand
    constexpr int NI = BK / 16;                            // DMA instructions per wave, operand and stage (4 pixel rows each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (A.zero_me && hw_bid == 0)
        for (int i = tid; i < A.zero_n; i += NT) A.zero_me[i] = 0.f;       // the bias gradient's zero-fill rides on this launch (stream order: done before the next one)
    // logical block order: tap fastest, then channel tiles, K slice (pixel range) slowest, on XCD-contiguous ids -- the
    // 25 tap blocks of a pixel range read the same dY rows and overlapping X rows, so they should meet in one L2
    int bid = xcd_remap(hw_bid, nb);
    const int tapi = bid % a.ntaps; bid /= a.ntaps;
    const int cit = bid % a.ci_tiles; bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles; bid /= a.co_tiles;
    const int split = bid;
    const int tap = a.tap_id[0] + tapi;          // live taps are a raster-order prefix (checked by the launcher)
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int sh_y = ky - a.pad, sh_x = kx - a.pad;
    const int co0 = cot * TC, ci0 = cit * TC;
    const int64_t q_begin = split * a.chunk;
    const int64_t q_end = (q_begin + a.chunk < a.Q) ? q_begin + a.chunk : a.Q;

    // Buffer-addressed LDS-DMA (as in conv_igemm.hip): out-of-range offsets make the buffer unit write zeros, so padding
    // pixels, pixels beyond this block's slice and channels beyond the tensor need no zero page and no 64-bit pointers.
    // One operand is walked linearly (its pixel index IS q: dY for a conv, X for a transposed conv), the other carries the
    // tap shift and needs (b, qy, qx); those advance incrementally by 64 pixels per step -- a division only on a row wrap.
    constexpr uint32_t OOB = 0x80000000u;
    asm volatile("" ::"v"((__attribute__((address_space(3))) unsigned char*)smem) : "memory");
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)OOB, 0x00020000);
    const int lrow = lane >> 4, pslot = lane & 15;
    const int ls = pslot ^ ((lrow & 3) << WG_SWZ);                // row & 3 == lrow & 3 for all four DMA rows of a lane
    const bool d_ch = co0 + ls * 8 < a.Cout, x_ch = ci0 + ls * 8 < a.Cin;
    const uint32_t d_cb = (uint32_t)((a.y_co + co0 + ls * 8) * 2), x_cb = (uint32_t)((a.x_co + ci0 + ls * 8) * 2);
    const int SH = a.transposed ? a.Ho : a.H, SW = a.transposed ? a.Wo : a.W;        // extent of the shifted operand
    const uint32_t lin_ps = (uint32_t)((a.transposed ? a.x_ps : a.y_ps) * 2), sh_ps = (uint32_t)((a.transposed ? a.y_ps : a.x_ps) * 2);
    uint32_t qi[NI], lin_off[NI];
    int qx[NI], qy[NI], qb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const uint32_t q = (uint32_t)q_begin + (uint32_t)((wave * NI + i) * 4 + lrow);
        qi[i] = q;
        const uint32_t r1 = fdiv(q, A.dqw);
        qx[i] = (int)(q - r1 * (uint32_t)a.QW);
        const uint32_t b = fdiv(r1, A.dqh);
        qy[i] = (int)(r1 - b * (uint32_t)a.QH);
        qb[i] = (int)b;
        lin_off[i] = q * lin_ps;
    }
    // i0 .. i1: which of the stage's NI instruction pairs (the spread form issues one pair behind each k-step's MFMAs)
    auto issue_slow = [&](int buf, int i0, int i1) {
        unsigned char* dt = smem + buf * STAGE;
        unsigned char* xt = dt + TILE;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i < i0 || i >= i1) continue;
            const bool inq = qi[i] < (uint32_t)q_end;
            const int sy = qy[i] * a.stride + sh_y, sx = qx[i] * a.stride + sh_x;
            const bool sok = inq && (unsigned)sy < (unsigned)SH && (unsigned)sx < (unsigned)SW;
            const uint32_t s_off = (uint32_t)((qb[i] * SH + sy) * SW + sx) * sh_ps;
            uint32_t vd, vx;
            if (a.transposed) { vd = (sok && d_ch) ? s_off + d_cb : OOB; vx = (inq && x_ch) ? lin_off[i] + x_cb : OOB; }
            else { vd = (inq && d_ch) ? lin_off[i] + d_cb : OOB; vx = (sok && x_ch) ? s_off + x_cb : OOB; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (__attribute__((address_space(3))) void*)(dt + (wave * NI + i) * 1024), 16, (int)vd, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(xt + (wave * NI + i) * 1024), 16, (int)vx, 0, 0, 0);
            qi[i] += BK;
            lin_off[i] += BK * lin_ps;
            qx[i] += BK;
            if (qx[i] >= a.QW) {
                const uint32_t t = fdiv((uint32_t)qx[i], A.dqw);
                qx[i] -= (int)t * a.QW;
                qy[i] += (int)t;
                if (qy[i] >= a.QH) {
                    const uint32_t u = fdiv((uint32_t)qy[i], A.dqh);
                    qy[i] -= (int)u * a.QH;
                    qb[i] += (int)u;
                }
            }
        }
    };

    // Fast bookkeeping (QW % 16 == 0, every training layer above the 8x8 hyper maps): the 16 pixels of a wave's four DMA
    // instructions lie in one image row, so (b, qy, qx of the first pixel), the row validity and the row's byte offset
    // are wave-uniform SCALARS that go into the instruction's soffset; per lane only the x-range check of the shifted
    // operand is left (3 VALU per instruction).  The per-lane form above costs ~140 VALU per step -- more issue time than
    // the step's 16 MFMAs.
    const uint32_t lin_cb = a.transposed ? x_cb : d_cb, sft_cb = a.transposed ? d_cb : x_cb;
    const bool lin_ch = a.transposed ? x_ch : d_ch, sft_ch = a.transposed ? d_ch : x_ch;
    const uint32_t v_lin = lin_ch ? (uint32_t)lrow * lin_ps + lin_cb : OOB;
    const uint32_t v_sft = sft_ch ? (uint32_t)(lrow * a.stride) * sh_ps + sft_cb : OOB;
    const int lx = lrow * a.stride;
    const uint32_t neg_b = (uint32_t)a.pad * sh_ps;             // the shifted operand's resource starts pad pixels early: soffset >= 0
    const __amdgpu_buffer_rsrc_t lr = a.transposed ? xr : dr;
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)(a.transposed ? a.dy : a.x) - neg_b), 0, (int)OOB, 0x00020000);
    uint32_t sq_w = (uint32_t)q_begin + (uint32_t)(wave * NI * 4);
    int sqx, sqy, sqb;
    {
        const uint32_t r1 = fdiv(sq_w, A.dqw);
        sqx = (int)(sq_w - r1 * (uint32_t)a.QW);
        const uint32_t b = fdiv(r1, A.dqh);
        sqy = (int)(r1 - b * (uint32_t)a.QH);
        sqb = (int)b;
    }
    auto issue_fast = [&](int buf, int i0, int i1) {
        unsigned char* dt = smem + buf * STAGE;
        unsigned char* xt = dt + TILE;
        unsigned char* lt = a.transposed ? xt : dt;
        unsigned char* stt = a.transposed ? dt : xt;
        const bool inq = sq_w < (uint32_t)q_end;
        const int sy = sqy * a.stride + sh_y, sxw = sqx * a.stride + sh_x;
        const bool rowok = inq && (unsigned)sy < (unsigned)SH;
        const uint32_t so_l = sq_w * lin_ps;
        const uint32_t so_s = (uint32_t)(((sqb * SH + sy) * SW + sxw) * (int)sh_ps) + neg_b;
        const uint32_t vl = inq ? v_lin : OOB;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i < i0 || i >= i1) continue;
            const bool okx = rowok && (unsigned)(sxw + 4 * i * a.stride + lx) < (unsigned)SW;
            const uint32_t vs = okx ? v_sft : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lr, (__attribute__((address_space(3))) void*)(lt + (wave * NI + i) * 1024), 16, (int)vl,
                                                     (int)(so_l + (uint32_t)(4 * i) * lin_ps), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sr, (__attribute__((address_space(3))) void*)(stt + (wave * NI + i) * 1024), 16, (int)vs,
                                                     (int)(so_s + (uint32_t)(4 * i * a.stride) * sh_ps), 0, 0);
        }
        if (i1 < NI) return;              // the stage's bookkeeping advances with its last pair
        sq_w += BK;
        sqx += BK;
        if (sqx >= a.QW) {
            const uint32_t t = fdiv((uint32_t)sqx, A.dqw);
            sqx -= (int)t * a.QW;
            sqy += (int)t;
            if (sqy >= a.QH) {
                const uint32_t u = fdiv((uint32_t)sqy, A.dqh);
                sqy -= (int)u * a.QH;
                sqb += (int)u;
            }
        }
    };

    const int wm = wave & 1, wn = wave >> 1;
    const int frow = lane & 31, fh = lane >> 5;
    const int g = lane >> 4, t = lane & 15;
    // byte offset of this lane's 8-byte chunk inside a tile for channel-tile base cb, pixel base pb (multiples of 32 / 16)
    auto tr_off = [&](int cb, int pix) {
        const int ch = cb + (g & 1) * 16 + 4 * (t & 3);
        return pix * 256 + (((ch >> 3) ^ ((pix & 3) << WG_SWZ)) << 4) + (ch & 7) * 2;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int bslot = -1;                                   // block-uniform: which bias_part slot this block's column sums go to
    if (a.bias_part && cit == 0)
        for (int k = 0; k < a.nb_taps; ++k)
            if (tapi == a.b_tap[k]) bslot = k;
    f32x16 bacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;

    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int64_t nsteps = (q_end - q_begin + BK - 1) / BK;
    // the transform of the X operand (|x| for the abs-conv of encode_hyper, x^2 for the GDN gamma gradient) is a
    // compile-time variant of the loop: no branches between the transpose reads and the MFMAs
    auto main_loop = [&](auto abs_tag, auto sq_tag, auto fast_tag, auto bias_tag) {
        constexpr bool ABS = decltype(abs_tag)::value, SQ = decltype(sq_tag)::value, FASTQ = decltype(fast_tag)::value, BIAS = decltype(bias_tag)::value;
        const u32x4 ones_u = {H16_ONE_PAIR, H16_ONE_PAIR, H16_ONE_PAIR, H16_ONE_PAIR};
        const h16x8 ones = __builtin_bit_cast(h16x8, ones_u);
        auto issue = [&](int buf, int i0, int i1) {
            if constexpr (FASTQ) issue_fast(buf, i0, i1);
            else issue_slow(buf, i0, i1);
        };
        // stages 0 .. NST-2 go out first; every step then issues stage step + NST - 1 (past the end: out-of-range offsets, zeros into a
        // stage nobody reads -- the count of outstanding DMA instructions stays what the vmcnt below assumes)
        if (nsteps > 0) {
#pragma unroll
            for (int p = 0; p < NST - 1; ++p) issue(p, 0, NI);
        }
        int buf = 0;
        for (int64_t step = 0; step < nsteps; ++step) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 2 * NI) : "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char* dt = smem + buf * STAGE;
            const unsigned char* xt = dt + TILE;
            s16x4 dlo[2][2], dhi[2][2], xlo[2][2], xhi[2][2];
            auto ldf = [&](int set, int ks) {
                const int pix = ks * 16 + (g >> 1) * 8 + (t >> 2);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    dlo[set][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dt + tr_off(wm * 64 + i * 32, pix)));
                    dhi[set][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dt + tr_off(wm * 64 + i * 32, pix + 4)));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    xlo[set][j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xt + tr_off(wn * 64 + j * 32, pix)));
                    xhi[set][j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xt + tr_off(wn * 64 + j * 32, pix + 4)));
                }
            };
            const int nbuf = buf == 0 ? NST - 1 : buf - 1;      // the stage consumed one step ago
            if constexpr (!SPREAD) issue(nbuf, 0, NI);           // in front of the first fragment reads (behind them: 9.71 vs 9.675 ms per step, same box)
            __builtin_amdgcn_sched_barrier(0);
            ldf(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                if (ks + 1 < BK / 16) ldf((ks + 1) & 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                h16x8 df[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const s16x8 v = __builtin_shufflevector(dlo[ks & 1][i], dhi[ks & 1][i], 0, 1, 2, 3, 4, 5, 6, 7);
                    df[i] = __builtin_bit_cast(h16x8, v);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    s16x8 v = __builtin_shufflevector(xlo[ks & 1][j], xhi[ks & 1][j], 0, 1, 2, 3, 4, 5, 6, 7);
                    if (ABS) v = v & (short)0x7fff;
                    if (SQ) {
                        u32x4 u = __builtin_bit_cast(u32x4, v);
                        uint32_t* w4 = (uint32_t*)&u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float l2 = h2f_lo(w4[q]), h2 = h2f_hi(w4[q]);
                            w4[q] = pack_h2(l2 * l2, h2 * h2);
                        }
                        v = __builtin_bit_cast(s16x8, u);
                    }
                    xf[j] = __builtin_bit_cast(h16x8, v);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16_h16(df[i], xf[j], acc[i][j], 0, 0, 0);
                if constexpr (BIAS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) bacc[i] = mfma_32x32x16_h16(df[i], ones, bacc[i], 0, 0, 0);
                }
                if constexpr (SPREAD) issue(nbuf, ks, ks + 1);       // one instruction pair in the shadow of this k-step's MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
            buf = buf + 1 == NST ? 0 : buf + 1;
        }
    };
    if (bslot >= 0) {
        // one block in ntaps * ci_tiles (x stride^2 for a transposed conv): the squared-input (GDN gamma) form never carries a bias
        if (A.fastq) {
            if (a.in_abs) main_loop(std::true_type{}, std::false_type{}, std::true_type{}, std::true_type{});
            else main_loop(std::false_type{}, std::false_type{}, std::true_type{}, std::true_type{});
        } else {
            if (a.in_abs) main_loop(std::true_type{}, std::false_type{}, std::false_type{}, std::true_type{});
            else main_loop(std::false_type{}, std::false_type{}, std::false_type{}, std::true_type{});
        }
        // every column of bacc holds the same sums: column 0 (lanes 0 and 32) of the waves of ci half 0 writes them
        if (wn == 0 && frow == 0) {
            float* bp = a.bias_part + ((int64_t)split * a.nb_taps + bslot) * a.Cout;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (co < a.Cout) bp[co] = bacc[i][r];
                }
        }
    } else if (A.fastq) {
        if (a.in_sq) main_loop(std::false_type{}, std::true_type{}, std::true_type{}, std::false_type{});
        else if (a.in_abs) main_loop(std::true_type{}, std::false_type{}, std::true_type{}, std::false_type{});
        else main_loop(std::false_type{}, std::false_type{}, std::true_type{}, std::false_type{});
    } else {
        if (a.in_sq) main_loop(std::false_type{}, std::true_type{}, std::false_type{}, std::false_type{});
        else if (a.in_abs) main_loop(std::true_type{}, std::false_type{}, std::false_type{}, std::false_type{});
        else main_loop(std::false_type{}, std::false_type{}, std::false_type{}, std::false_type{});
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wn * 64 + j * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (co < a.Cout && ci < a.Cin) a.out[ws_row(a, split, tapi, co) + ci] = acc[i][j][r];
            }
        }
}
template <int BK, int NST, bool SPREAD = false>
__global__ __launch_bounds__(NT) void wgrad_tr_kernel(const WgTrArgs A) {
    wgrad_tr_body<BK, NST, SPREAD>(A, (int)blockIdx.x, (int)gridDim.x);
}

// Several layers' split-K launches in ONE grid (hesic_conv2d_wgrad_partial_batched; round 5).  One launch per layer costs each of the ~30
// 128-channel weight gradients of a training step its own ramp (every block waits for its first stage at the same moment), its own tail (every
// block writes its 64 KB partial at the same moment) and the idle slots of a 400- or 450-block grid on 512; a layer with 16 stages per
// block spends ~40 % of its launch there (10 GF in 28 us against 50 GF in 76 us for 64 stages per block).  Here the jobs' block ranges follow
// each other (8-aligned, so a job's XCD remap sees its own range), longest K slices first: a slot that finishes a block takes the next one,
// whatever layer it belongs to.  Results are bit-identical to the one-launch-per-layer form (same blocks, same order inside each block).
constexpr int WB_MAX = 14;                                  // jobs per launch: the argument block stays under the 4 KB of a kernel's arguments
struct WgTrBatch {
    int n;
    int start[WB_MAX + 1];                                  // multiples of 8
    int blocks[WB_MAX];                                     // real block count of the job (the rest of its range exits)
    WgTrArgs job[WB_MAX];
};
static_assert(sizeof(WgTrBatch) <= 4096, "kernel arguments: 4 KB");

template <int BK, int NST>
__global__ __launch_bounds__(NT) void wgrad_tr_batched_kernel(const WgTrBatch B) {
    int j = 0;
    while (j + 1 < B.n && (int)blockIdx.x >= B.start[j + 1]) ++j;       // uniform: scalar loop over <= 14 entries
    const int local = (int)blockIdx.x - B.start[j], nb8 = B.start[j + 1] - B.start[j];
    if (xcd_remap(local, nb8) >= B.blocks[j]) return;
    // the body reads its job through scalar loads of the argument block; the logical id is recomputed inside from the same (local, nb8)
    wgrad_tr_body<BK, NST, false>(B.job[j], local, nb8);
}


// ---------------------------------------------------------------- one kernel ROW of taps per block (round 5)
// wgrad_tr_kernel moves 32 KB of operands through L2 -> LDS for every 16 MFMAs of a wave: on the 128 -> 128 5x5 stride-2 layers of the
// 256^2 maps (conv2 / deconv3, 107 GFLOP each at B = 8) the 25 tap blocks of a pixel range re-read dY 25 times and X 6.25 times --
// 1.68 GB per launch out of L2 / the Infinity Cache, ~10 TB/s, the matrix pipe 26 % busy and half of the wave cycles parked at vmcnt
// (profiles/r04_f_pmc_sq_train.json).  Here a block owns the five taps (ky, 0..4) of one kernel row: for 64 consecutive q of ONE image row
// the linear operand (dY for a conv, X for a transposed conv) is staged once and serves all five taps, and the shifted operand's pixels
// q * 2 + kx - 2 of the five taps are 131 CONSECUTIVE pixels of one image row -- staged once as an even and an odd plane (tap kx reads
// plane kx & 1 from row (kx >> 1) on), so neighbouring taps share them too: 50 KB per 40 MFMAs of a wave instead of 32 KB per 16
// (0.31 GB per launch), 14 transposing LDS reads per 10 MFMAs instead of 16 per 8.  Eight waves (2 per SIMD): wave = 64 channels of the
// linear operand x 32 of the shifted one x 5 taps = 160 accumulator registers; a ring of three 50 KB stages, one block per CU.
// The split-K partials keep wgrad_tr_kernel's layout [split][tap][Cout][Cin] (same finishing passes); what grows is their volume:
// blocks x 5 taps x 64 KB (82 MB for 250 blocks), the price of the larger accumulator tile per CU.  The bias column sums ride on the
// VALU (the fragments a lane holds are 8 pixels of one channel) in the waves / blocks that see every dY pixel once.
struct WgRowArgs { WgArgs w; float* zero_me; int zero_n; FastDiv dqw, dqh; };

constexpr int ROW_LINB = 64 * 256, ROW_PLR = 68, ROW_SFTB = 2 * ROW_PLR * 256, ROW_STAGE = ROW_LINB + ROW_SFTB, ROW_NST = 3;
constexpr int ROW_LDS = ROW_NST * ROW_STAGE;

template <bool TR>
__global__ __launch_bounds__(512) void wgrad_row_kernel(const WgRowArgs A) {
    const WgArgs& a = A.w;
    constexpr int LINB = ROW_LINB, PLR = ROW_PLR, STAGE = ROW_STAGE, NST = ROW_NST;
    constexpr int NLIN = 16, NSFT = 2 * PLR / 4, NINS = NLIN + NSFT;       // DMA instructions (4 tile rows = 1 KB each) per stage: 16 + 34
    constexpr int PW = (NINS + 7) / 8;                                      // per wave: 7 (waves 0, 1) or 6
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (A.zero_me && blockIdx.x == 0)
        for (int i = tid; i < A.zero_n; i += 512) A.zero_me[i] = 0.f;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ky = bid % 5; bid /= 5;
    const int cit = bid % a.ci_tiles; bid /= a.ci_tiles;
    const int cot = bid % a.co_tiles; bid /= a.co_tiles;
    const int split = bid;
    const int co0 = cot * TC, ci0 = cit * TC;
    const int64_t q_begin = split * a.chunk;
    const int64_t q_end = (q_begin + a.chunk < a.Q) ? q_begin + a.chunk : a.Q;
    const int nsteps = (int)((q_end - q_begin) >> 6);          // chunk and Q are multiples of 64 (launcher)

    constexpr uint32_t OOB = 0x80000000u;
    asm volatile("" ::"v"((__attribute__((address_space(3))) unsigned char*)smem) : "memory");
    const int SH = TR ? a.Ho : a.H, SW = TR ? a.Wo : a.W;
    const uint32_t lin_ps = (uint32_t)((TR ? a.x_ps : a.y_ps) * 2), sh_ps = (uint32_t)((TR ? a.y_ps : a.x_ps) * 2);
    const uint32_t neg_b = 2u * sh_ps;                         // the shifted operand's resource starts pad = 2 pixels early: soffset >= 0
    const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc((void*)(TR ? a.x : a.dy), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)(TR ? a.dy : a.x) - neg_b), 0, (int)OOB, 0x00020000);
    const int lrow = lane >> 4, pslot = lane & 15;
    const int ls = pslot ^ ((lrow & 3) << 2);                  // 16-byte slot this lane FETCHES for LDS slot pslot (rows of an instruction are 4-aligned)
    const int lin_c = (TR ? ci0 : co0) + ls * 8, sft_c = (TR ? co0 : ci0) + ls * 8;
    const bool lin_ok = lin_c < (TR ? a.Cin : a.Cout), sft_ok = sft_c < (TR ? a.Cout : a.Cin);
    const uint32_t v_lin = lin_ok ? (uint32_t)lrow * lin_ps + (uint32_t)(((TR ? a.x_co : a.y_co) + lin_c) * 2) : OOB;
    const uint32_t sft_cb = (uint32_t)(((TR ? a.y_co : a.x_co) + sft_c) * 2);
    // instruction n = wave + 8 i of a stage: n < 16 -> rows 4n .. 4n+3 of the linear tile; else instruction n - 16 of the shifted tile
    // (rows 4 (n - 16) .. of plane 0 = even pixels for n - 16 < 17, of plane 1 = odd pixels from 17 on)
    int sxrel[PW];                                             // pixel (relative to the stage's first = q0 * 2 - 2) of this lane's row
    uint32_t v_sft[PW];
#pragma unroll
    for (int i = 2; i < PW; ++i) {
        const int mi = wave + 8 * i - NLIN, p = mi >= PLR / 4 ? 1 : 0;
        const int m = 4 * (mi - p * (PLR / 4)) + lrow;
        sxrel[i] = 2 * m + p;
        v_sft[i] = sft_ok ? (uint32_t)sxrel[i] * sh_ps + sft_cb : OOB;
    }
    uint32_t sq = (uint32_t)q_begin;                           // first q of the next stage to issue, and its (b, qy, qx)
    int sqx, sqy, sqb;
    {
        const uint32_t r1 = fdiv(sq, A.dqw);
        sqx = (int)(sq - r1 * (uint32_t)a.QW);
        const uint32_t b = fdiv(r1, A.dqh);
        sqy = (int)(r1 - b * (uint32_t)a.QH);
        sqb = (int)b;
    }
    auto issue = [&](int buf) {
        unsigned char* lt = smem + buf * STAGE;
        unsigned char* stt = lt + LINB;
        const bool inq = sq < (uint32_t)q_end;
        const int sy = sqy * 2 + ky - 2, sxb = sqx * 2 - 2;
        const bool rowok = inq && (unsigned)sy < (unsigned)SH;
        const uint32_t so_l = sq * lin_ps;
        const uint32_t so_s = (uint32_t)(((sqb * SH + sy) * SW + sqx * 2) * (int)sh_ps);
        const uint32_t vl = inq ? v_lin : OOB;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int n = wave + 8 * i;
            if (i < 2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(lr, (__attribute__((address_space(3))) void*)(lt + n * 1024), 16, (int)vl,
                                                         (int)(so_l + (uint32_t)(4 * n) * lin_ps), 0, 0);
            } else {
                if (i == PW - 1 && n >= NINS) continue;        // wave-uniform
                const bool okx = rowok && (unsigned)(sxb + sxrel[i]) < (unsigned)SW;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sr, (__attribute__((address_space(3))) void*)(stt + (n - NLIN) * 1024), 16,
                                                         (int)(okx ? v_sft[i] : OOB), (int)so_s, 0, 0);
            }
        }
        sq += 64;
        sqx += 64;
        if (sqx >= a.QW) {                                     // QW % 64 == 0: a stage never straddles a row
            sqx = 0;
            if (++sqy >= a.QH) { sqy = 0; ++sqb; }
        }
    };

    const int wl = wave & 1, wsd = wave >> 1;                  // 64-channel half of the linear operand, 32-channel quarter of the shifted one
    const int frow = lane & 31, fh = lane >> 5, g = lane >> 4, t = lane & 15;
    const int chs = (g & 1) * 16 + 4 * (t & 3), rsub = fh * 8 + (t >> 2);
    auto foff = [&](int cb, int rbase) {
        const int row = rbase + rsub, ch = cb + chs;
        return row * 256 + (((ch >> 3) ^ ((row & 3) << 2)) << 4) + (ch & 7) * 2;
    };
    int lo_off[2], so_off[5];
#pragma unroll
    for (int i = 0; i < 2; ++i) lo_off[i] = foff(wl * 64 + i * 32, 0);
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) so_off[kx] = LINB + foff(wsd * 32, (kx & 1) * PLR + (kx >> 1));

    f32x16 acc[5][2];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kx][i][r] = 0.f;
    // bias column sums: a conv's dY is the linear operand (every block sees all of it: the ky = 0 blocks of ci tile 0 sum it, in the two
    // waves of shifted-quarter 0); a transposed conv's dY is the shifted one: rows 2 qy + ky - 2 and pixels 2 qx + kx - 2 with ky, kx in
    // {2, 3} cover every dY pixel exactly once (slots (ky - 2) * 2 + (kx - 2), the order setup_bias_part lists them)
    const bool bias_blk = a.bias_part && cit == 0 && (TR ? (ky == 2 || ky == 3) : ky == 0);
    const bool bias_wave = bias_blk && (TR ? wl == 0 : wsd == 0);
    float bsum[2] = {0.f, 0.f};

    auto sum8 = [](const s16x4 lo, const s16x4 hi) {
        const u32x2 l = __builtin_bit_cast(u32x2, lo), h = __builtin_bit_cast(u32x2, hi);
        return ((h2f_lo(l.x) + h2f_hi(l.x)) + (h2f_lo(l.y) + h2f_hi(l.y))) + ((h2f_lo(h.x) + h2f_hi(h.x)) + (h2f_lo(h.y) + h2f_hi(h.y)));
    };
    auto main_loop = [&](auto bias_tag) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        if (nsteps > 0) {
#pragma unroll
            for (int p = 0; p < NST - 1; ++p) issue(p);
        }
        int buf = 0;
        for (int step = 0; step < nsteps; ++step) {
            if (wave < NINS - 8 * (PW - 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (PW - 1)) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(buf == 0 ? NST - 1 : buf - 1);                // the stage consumed one step ago
            const unsigned char* st = smem + buf * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned char* kb = st + ks * 4096;
                h16x8 lf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(kb + lo_off[i]));
                    const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(kb + lo_off[i] + 1024));
                    lf[i] = __builtin_bit_cast(h16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                    if constexpr (BIAS && !TR) bsum[i] += sum8(l0, l1);
                }
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const s16x4 s0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(kb + so_off[kx]));
                    const s16x4 s1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(kb + so_off[kx] + 1024));
                    const h16x8 sf = __builtin_bit_cast(h16x8, __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7));
                    if constexpr (BIAS && TR) {
                        if (kx == 2) bsum[0] += sum8(s0, s1);
                        if (kx == 3) bsum[1] += sum8(s0, s1);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if constexpr (TR) acc[kx][i] = mfma_32x32x16_h16(sf, lf[i], acc[kx][i], 0, 0, 0);
                        else acc[kx][i] = mfma_32x32x16_h16(lf[i], sf, acc[kx][i], 0, 0, 0);
                    }
                }
            }
            buf = buf + 1 == NST ? 0 : buf + 1;
        }
    };
    if (bias_wave) main_loop(std::true_type{});
    else main_loop(std::false_type{});

    if (bias_wave) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float s = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            if (fh == 0) {
                if constexpr (TR) {
                    const int co = co0 + wsd * 32 + frow;
                    if (co < a.Cout) a.bias_part[((int64_t)split * a.nb_taps + (ky - 2) * 2 + i) * a.Cout + co] = s;
                } else {
                    const int co = co0 + wl * 64 + i * 32 + frow;
                    if (co < a.Cout) a.bias_part[(int64_t)split * a.nb_taps * a.Cout + co] = s;
                }
            }
        }
    }
    // C[row = A's channel][col = B's channel]: col = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ci = TR ? ci0 + wl * 64 + i * 32 + frow : ci0 + wsd * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int co = TR ? co0 + wsd * 32 + rr : co0 + wl * 64 + i * 32 + rr;
                if (co < a.Cout && ci < a.Cin) a.out[ws_row(a, split, ky * 5 + kx, co) + ci] = acc[kx][i][r];
            }
        }
    }
}

// dw[tap_id[t]][..] = sum_s ws[s][t][..]; dead taps (masked conv) are zero-filled by the host memset
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int ntaps, int64_t per_tap,
                                    const WgArgs a) {
    const int64_t n = (int64_t)ntaps * per_tap;          // per_tap = Cout*Cin is a multiple of 4: 16-byte lanes
    for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        f32x4 s = *(const f32x4*)(ws + i);
        for (int k = 1; k < nsplit; ++k) s += *(const f32x4*)(ws + k * n + i);
        const int t = i / per_tap;
        *(f32x4*)(dw + (int64_t)a.tap_id[t] * per_tap + (i - t * per_tap)) = s;
    }
}

// Many K slices, few outputs (the 1x1 "convs" behind the GDN parameter gradients: up to 256 slices of one 128x128 tile):
// 16 float4 lanes x 16 slice groups per block, the groups meet in LDS in a fixed order -- 16x the parallelism of the
// one-thread-per-output loop above (which walked 256 slices serially in 16 blocks: 35 us for 16 MB).
__device__ __forceinline__ void reduce_wide_body(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int ntaps, int64_t per_tap,
                                                 const WgArgs& a, int bid, f32x4 (*red)[16]) {
    const int64_t n = (int64_t)ntaps * per_tap;
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t i = ((int64_t)bid * 16 + el) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (i < n) {
        int k = grp;
        for (; k + 16 < nsplit; k += 32) {
            s0 += *(const f32x4*)(ws + k * n + i);
            s1 += *(const f32x4*)(ws + (k + 16) * n + i);
        }
        if (k < nsplit) s0 += *(const f32x4*)(ws + k * n + i);
    }
    red[grp][el] = s0 + s1;
    __syncthreads();
    if (grp == 0 && i < n) {
        f32x4 s = red[0][el];
#pragma unroll
        for (int g = 1; g < 16; ++g) s += red[g][el];
        const int t = i / per_tap;
        *(f32x4*)(dw + (int64_t)a.tap_id[t] * per_tap + (i - t * per_tap)) = s;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int ntaps,
                                                                int64_t per_tap, const WgArgs a) {
    __shared__ f32x4 red[16][16];
    reduce_wide_body(ws, dw, nsplit, ntaps, per_tap, a, (int)blockIdx.x, red);
}

template <typename T>
__device__ __forceinline__ void colsum_body(const T* __restrict__ dy, float* __restrict__ db, int64_t P, int C, int ps, int co,
                                            int64_t rows_per_block, int bid, float* red) {
    // thread = one 16-byte chunk (8 bf16 / 4 fp32 channels) of a row; 256 / (C/CE) rows in flight per block iteration;
    // partial sums meet in LDS, one atomic per (block, channel)
    constexpr int CE = 16 / (int)sizeof(T);
    const int cpr = (C + CE - 1) / CE;                       // chunks per row
    const int rpi = 256 / cpr > 0 ? 256 / cpr : 1;           // rows per iteration
    const int chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr;
    const int64_t r0 = bid * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < P ? r0 + rows_per_block : P;
    float acc[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] = 0.f;
    const bool vec = (C % CE) == 0 && (ps % CE) == 0 && (co % CE) == 0;
    if (rsub < rpi) {
        auto add = [&](const u32x4 raw) {
            if constexpr (sizeof(T) == 2) {
                acc[0] += h2f_lo(raw.x); acc[1] += h2f_hi(raw.x);
                acc[2] += h2f_lo(raw.y); acc[3] += h2f_hi(raw.y);
                acc[4] += h2f_lo(raw.z); acc[5] += h2f_hi(raw.z);
                acc[6] += h2f_lo(raw.w); acc[7] += h2f_hi(raw.w);
            } else {
                acc[0] += __uint_as_float(raw.x); acc[1] += __uint_as_float(raw.y); acc[2] += __uint_as_float(raw.z); acc[3] += __uint_as_float(raw.w);
            }
        };
        int64_t r = r0 + rsub;
        if (vec) {
            const T* base = dy + co + chunk * CE;
            for (; r + 3 * rpi < r1; r += 4 * rpi) {           // four independent 16-byte loads in flight per lane
                const u32x4 v0 = *(const u32x4*)(base + r * ps), v1 = *(const u32x4*)(base + (r + rpi) * ps);
                const u32x4 v2 = *(const u32x4*)(base + (r + 2 * rpi) * ps), v3 = *(const u32x4*)(base + (r + 3 * rpi) * ps);
                add(v0); add(v1); add(v2); add(v3);
            }
            for (; r < r1; r += rpi) add(*(const u32x4*)(base + r * ps));
        } else {
            for (; r < r1; r += rpi) {
                const T* p = dy + r * ps + co + chunk * CE;
#pragma unroll
                for (int e = 0; e < CE; ++e)
                    if (chunk * CE + e < C) acc[e] += elem<T>::ld(p + e);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) red[threadIdx.x * CE + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float sum = 0.f;
        const int ch = c / CE, e = c % CE;
        for (int rs = 0; rs < rpi; ++rs) sum += red[(rs * cpr + ch) * CE + e];
        atomicAdd(db + c, sum);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ db, int64_t P, int C, int ps, int co,
                                                     int64_t rows_per_block) {
    __shared__ float red[256 * (16 / (int)sizeof(T))];
    colsum_body<T>(dy, db, P, C, ps, co, rows_per_block, (int)blockIdx.x, red);
}

// wgrad_reduce_kernel and colsum_kernel as one launch: blocks [0, n_red) sum the K slices, blocks [n_red, ...) the columns
// of dY (bias gradient).  The two are independent and each too small to fill the chip; a training step issues ~55 pairs.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_reduce_colsum_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int ntaps,
                                                                  int64_t per_tap, const WgArgs a, int n_red, const T* __restrict__ dy,
                                                                  float* __restrict__ db, int64_t P, int64_t rows_per_block) {
    __shared__ float red[256 * (16 / (int)sizeof(T))];
    if ((int)blockIdx.x >= n_red) {
        colsum_body<T>(dy, db, P, a.Cout, a.y_ps, a.y_co, rows_per_block, (int)blockIdx.x - n_red, red);
        return;
    }
    const int64_t n = (int64_t)ntaps * per_tap;
    for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)n_red * blockDim.x * 4) {
        f32x4 s = *(const f32x4*)(ws + i);
        for (int k = 1; k < nsplit; ++k) s += *(const f32x4*)(ws + k * n + i);
        const int t = i / per_tap;
        *(f32x4*)(dw + (int64_t)a.tap_id[t] * per_tap + (i - t * per_tap)) = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wgrad_reduce_wide_colsum_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int ntaps,
                                                                       int64_t per_tap, const WgArgs a, int n_red, const T* __restrict__ dy,
                                                                       float* __restrict__ db, int64_t P, int C, int ps, int co, int64_t rows_per_block) {
    __shared__ __attribute__((aligned(16))) float scratch[256 * (16 / (int)sizeof(T)) > 1024 ? 256 * (16 / (int)sizeof(T)) : 1024];
    if ((int)blockIdx.x >= n_red) {
        colsum_body<T>(dy, db, P, C, ps, co, rows_per_block, (int)blockIdx.x - n_red, scratch);
        return;
    }
    reduce_wide_body(ws, dw, nsplit, ntaps, per_tap, a, (int)blockIdx.x, (f32x4(*)[16])scratch);
}

// K-slice reduce straight into the PyTorch weight layout, optionally ACCUMULATING (dw += ...), + the bias column sums, one
// launch.  A training step used to run reduce (packed layout) -> unpack (PyTorch layout) -> AccumulateGrad copy / add per
// layer; with the gradients of all parameters living in one flat buffer (train.FlatGroup) this kernel adds the layer's
// gradient in place.  Block = a tile of 8 couts x 32 cins x a group of live taps: the slices are summed with coalesced
// 128-byte reads along cin, the tile turns round in LDS and leaves as runs along the destination's fastest index
// (conv: (Cout, Cin, KH, KW) -> TG taps of 32 cins; transposed conv: (Cin, Cout, KH, KW) -> TG taps of 8 couts).
struct FinishArgs {
    const float* ws; float* dw; int nsplit, ntaps, T_all, Cout, Cin, transposed, accumulate;
    int tiles_ci, tiles_co, tap_groups, taps_per_group, n_red;
    int8_t tap_id[25];
    const float* bias_part; int nb_parts, accumulate_bias;      // bias column sums left by wgrad_tr_kernel: [nb_parts][Cout], summed in order by ONE block
    int ws_layout;                                              // WgArgs::ws_layout of the launch that wrote ws
    int wide;                                                   // round 5: block = 8 couts x 128 cins x ONE tap, 512-byte row reads (finish_body)
};

template <typename T>
__device__ __forceinline__ void finish_body(const FinishArgs& f, const T* __restrict__ dy, float* __restrict__ db, int64_t P, int y_ps, int y_co,
                                            int64_t rows_per_block, int bid, float* scratch) {
    if (bid >= f.n_red) {
        if (f.bias_part) {
            // dbias (+)= sum of the per-slice column sums, fixed order (deterministic; the column-sum route below ends in atomics)
            for (int co = threadIdx.x; co < f.Cout; co += 256) {
                float s0 = 0.f, s1 = 0.f;
                int k = 0;
                for (; k + 1 < f.nb_parts; k += 2) { s0 += f.bias_part[(int64_t)k * f.Cout + co]; s1 += f.bias_part[(int64_t)(k + 1) * f.Cout + co]; }
                if (k < f.nb_parts) s0 += f.bias_part[(int64_t)k * f.Cout + co];
                db[co] = f.accumulate_bias ? db[co] + (s0 + s1) : (s0 + s1);
            }
            return;
        }
        colsum_body<T>(dy, db, P, f.Cout, y_ps, y_co, rows_per_block, bid - f.n_red, scratch);
        return;
    }
    if (f.wide) {
        // One tap, 8 couts, 128 cins per block: a wave instruction reads two whole 512-byte rows of a slice (the narrow form below reads 128-byte
        // quarters of eight rows: 1 GB of partials per training step at 2.9 TB/s).  The slices are summed in the narrow form's order (two
        // alternating accumulators over groups of eight) -- the same bits; the four sums of a lane go straight to the PyTorch layout.
        const int tci = bid % f.tiles_ci; bid /= f.tiles_ci;
        const int tco = bid % f.tiles_co;
        const int tl = bid / f.tiles_co;                        // live tap index
        const int co = tco * 8 + (threadIdx.x >> 5), ci = tci * 128 + (threadIdx.x & 31) * 4;
        if (co >= f.Cout || ci >= f.Cin) return;
        const int64_t per_tap = (int64_t)f.Cout * f.Cin;
        const int64_t n = f.ws_layout ? f.Cin : (int64_t)f.ntaps * per_tap;
        const float* src = f.ws + (f.ws_layout ? ((int64_t)tl * f.Cout + co) * f.nsplit * f.Cin : (int64_t)tl * per_tap + (int64_t)co * f.Cin) + ci;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        int k = 0;
        for (; k + 7 < f.nsplit; k += 8) {
            const f32x4 a0 = *(const f32x4*)(src + (int64_t)k * n), a1 = *(const f32x4*)(src + (int64_t)(k + 1) * n);
            const f32x4 a2 = *(const f32x4*)(src + (int64_t)(k + 2) * n), a3 = *(const f32x4*)(src + (int64_t)(k + 3) * n);
            const f32x4 a4 = *(const f32x4*)(src + (int64_t)(k + 4) * n), a5 = *(const f32x4*)(src + (int64_t)(k + 5) * n);
            const f32x4 a6 = *(const f32x4*)(src + (int64_t)(k + 6) * n), a7 = *(const f32x4*)(src + (int64_t)(k + 7) * n);
            s0 += a0; s1 += a1; s0 += a2; s1 += a3; s0 += a4; s1 += a5; s0 += a6; s1 += a7;
        }
        for (; k < f.nsplit; ++k) s0 += *(const f32x4*)(src + (int64_t)k * n);
        const f32x4 sv = s0 + s1;
        const int t = f.tap_id[tl];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ci + e >= f.Cin) break;
            const int64_t dst = f.transposed ? ((int64_t)(ci + e) * f.Cout + co) * f.T_all + t : ((int64_t)co * f.Cin + ci + e) * f.T_all + t;
            f.dw[dst] = f.accumulate ? f.dw[dst] + sv[e] : sv[e];
        }
        return;
    }
    const int tci = bid % f.tiles_ci; bid /= f.tiles_ci;
    const int tco = bid % f.tiles_co;
    const int tg = bid / f.tiles_co;
    const int t0 = tg * f.taps_per_group;
    const int tn = (t0 + f.taps_per_group <= f.ntaps ? f.taps_per_group : f.ntaps - t0);
    const int64_t per_tap = (int64_t)f.Cout * f.Cin;
    // slice stride and row stride (elements) of the workspace layout (WgArgs::ws_layout)
    const int64_t n = f.ws_layout ? f.Cin : (int64_t)f.ntaps * per_tap;
    const int64_t rowst = f.ws_layout ? (int64_t)f.nsplit * f.Cin : f.Cin;
    const int64_t tapst = f.ws_layout ? (int64_t)f.Cout * f.nsplit * f.Cin : per_tap;
    if ((f.Cin & 3) == 0) {
        // 16-byte lanes: thread = (tap lane tq of 4, cout cl of 8, 4 consecutive cins), the K slices of its value all in flight at once --
        // a quarter of the load instructions of the 4-byte form below for the same 128-byte row segments
        const int tq = threadIdx.x >> 6, cl4 = (threadIdx.x >> 3) & 7, i4 = threadIdx.x & 7;
        const int co4 = tco * 8 + cl4, ci4 = tci * 32 + i4 * 4;
        const bool in4 = co4 < f.Cout && ci4 < f.Cin;
        for (int tl = tq; tl < tn; tl += 4) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            if (in4) {
                const float* src = f.ws + (int64_t)(t0 + tl) * tapst + (int64_t)co4 * rowst + ci4;
                int k = 0;
                for (; k + 7 < f.nsplit; k += 8) {
                    const f32x4 a0 = *(const f32x4*)(src + (int64_t)k * n), a1 = *(const f32x4*)(src + (int64_t)(k + 1) * n);
                    const f32x4 a2 = *(const f32x4*)(src + (int64_t)(k + 2) * n), a3 = *(const f32x4*)(src + (int64_t)(k + 3) * n);
                    const f32x4 a4 = *(const f32x4*)(src + (int64_t)(k + 4) * n), a5 = *(const f32x4*)(src + (int64_t)(k + 5) * n);
                    const f32x4 a6 = *(const f32x4*)(src + (int64_t)(k + 6) * n), a7 = *(const f32x4*)(src + (int64_t)(k + 7) * n);
                    s0 += a0; s1 += a1; s0 += a2; s1 += a3; s0 += a4; s1 += a5; s0 += a6; s1 += a7;
                }
                for (; k < f.nsplit; ++k) s0 += *(const f32x4*)(src + (int64_t)k * n);
            }
            const f32x4 sv = s0 + s1;
            float* d = scratch + tl * 257 + cl4 * 32 + i4 * 4;
            d[0] = sv.x; d[1] = sv.y; d[2] = sv.z; d[3] = sv.w;
        }
    }
    const int cl = threadIdx.x >> 5, il = threadIdx.x & 31;
    const int co = tco * 8 + cl, ci = tci * 32 + il;
    const bool in = co < f.Cout && ci < f.Cin;
    for (int tl = 0; tl < ((f.Cin & 3) == 0 ? 0 : tn); ++tl) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (in) {
            const float* src = f.ws + (int64_t)(t0 + tl) * tapst + (int64_t)co * rowst + ci;
            int k = 0;
            for (; k + 7 < f.nsplit; k += 8) {           // eight loads in flight per lane; fixed order: deterministic
                const float a0 = src[(int64_t)k * n], a1 = src[(int64_t)(k + 1) * n], a2 = src[(int64_t)(k + 2) * n], a3 = src[(int64_t)(k + 3) * n];
                const float a4 = src[(int64_t)(k + 4) * n], a5 = src[(int64_t)(k + 5) * n], a6 = src[(int64_t)(k + 6) * n], a7 = src[(int64_t)(k + 7) * n];
                s0 += a0; s1 += a1; s2 += a2; s3 += a3; s0 += a4; s1 += a5; s2 += a6; s3 += a7;
            }
            for (; k + 3 < f.nsplit; k += 4) {
                s0 += src[(int64_t)k * n]; s1 += src[(int64_t)(k + 1) * n]; s2 += src[(int64_t)(k + 2) * n]; s3 += src[(int64_t)(k + 3) * n];
            }
            for (; k < f.nsplit; ++k) s0 += src[(int64_t)k * n];
        }
        scratch[tl * 257 + threadIdx.x] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    const int run = (f.transposed ? 8 : 32) * tn;          // values per outer index of the destination
    for (int o = threadIdx.x; o < 256 * tn; o += 256) {
        const int ol = o / run, rem = o - ol * run;
        const int inner = rem / tn, tl = rem - inner * tn;
        const int c_o = f.transposed ? tco * 8 + inner : tco * 8 + ol;
        const int c_i = f.transposed ? tci * 32 + ol : tci * 32 + inner;
        if (c_o >= f.Cout || c_i >= f.Cin) continue;
        const float v = f.transposed ? scratch[tl * 257 + inner * 32 + ol] : scratch[tl * 257 + ol * 32 + inner];
        const int t = f.tap_id[t0 + tl];
        const int64_t dst = f.transposed ? ((int64_t)c_i * f.Cout + c_o) * f.T_all + t : ((int64_t)c_o * f.Cin + c_i) * f.T_all + t;
        f.dw[dst] = f.accumulate ? f.dw[dst] + v : v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const FinishArgs f, const T* __restrict__ dy, float* __restrict__ db, int64_t P,
                                                           int y_ps, int y_co, int64_t rows_per_block) {
    __shared__ __attribute__((aligned(16))) float scratch[25 * 257 > 256 * (16 / (int)sizeof(T)) ? 25 * 257 : 256 * (16 / (int)sizeof(T))];
    finish_body<T>(f, dy, db, P, y_ps, y_co, rows_per_block, (int)blockIdx.x, scratch);
}

// The finishing passes of SEVERAL layers in one launch.  A training step ran 37 of them back to back, 16.8 us each on grids of
// ~500 - 800 blocks of a few hundred cycles: launch-to-launch gaps and half-empty tails rather than work.  The jobs (K-slice
// workspace, destination slot, bias source) travel by value in the kernel arguments; block -> job by a scan of <= 8 prefix counts.
constexpr int FIN_NB = 8;
struct FinishExtra { const void* dy; float* db; int64_t P, rpb; int y_ps, y_co; };
struct FinishBatch { int n; int start[FIN_NB + 1]; FinishArgs f[FIN_NB]; FinishExtra e[FIN_NB]; };

template <typename T>
__global__ __launch_bounds__(256) void wgrad_finish_batched_kernel(const FinishBatch fb) {
    __shared__ __attribute__((aligned(16))) float scratch[25 * 257 > 256 * (16 / (int)sizeof(T)) ? 25 * 257 : 256 * (16 / (int)sizeof(T))];
    int j = 0;
    while (j + 1 < fb.n && (int)blockIdx.x >= fb.start[j + 1]) ++j;
    const FinishExtra& e = fb.e[j];
    finish_body<T>(fb.f[j], (const T*)e.dy, e.db, e.P, e.y_ps, e.y_co, e.rpb, (int)blockIdx.x - fb.start[j], scratch);
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

// ---- narrow x wide weight gradient (g_a_conv1 3 -> 128 and g_s_conv4 128 -> 3, both 5x5 stride 2 pad 2).
// Both are  dW[wc][nc][tap] = sum_q WIDE[q][wc] * NARROW[2q + k - 2][nc]  with q over the grid of the 128-channel
// tensor (conv1: WIDE = dy, NARROW = x;  deconv4: WIDE = x, NARROW = dy) and the same weight index (wc*NC + nc)*25 + tap.
// thread = one wide channel (x 2 pixel halves); the narrow patch of a 8x16 q-tile sits in LDS and is read as
// wave-uniform (broadcast) rows; 75 accumulators live in registers across the block's tiles; one atomic per weight
// and block at the end.
template <int NC, typename WT>
__global__ __launch_bounds__(256) void sconv_wgrad_nw_kernel(const WT* __restrict__ wide, int64_t ws_b, int64_t ws_y, int64_t ws_x,
                                                             const void* __restrict__ narrow, int n_dtype, int64_t ns_b, int64_t ns_c,
                                                             int64_t ns_y, int64_t ns_x, float* __restrict__ dw, int B, int QH, int QW,
                                                             int NH, int NW) {
    constexpr int TH = 8, TW = 16, PH = 2 * TH + 3, PW = 2 * TW + 4;       // PW padded to keep rows 8-byte aligned
    __shared__ __attribute__((aligned(16))) float patch[NC * PH * PW];
    __shared__ float red[128][NC * 25 + 1];
    const int tid = threadIdx.x, wc = tid & 127, half = tid >> 7;
    float acc[NC * 25];
#pragma unroll
    for (int i = 0; i < NC * 25; ++i) acc[i] = 0.f;
    const int tiles_x = (QW + TW - 1) / TW, tiles_y = (QH + TH - 1) / TH;
    const int64_t ntiles = (int64_t)tiles_x * tiles_y * B;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / ((int64_t)tiles_x * tiles_y);
        __syncthreads();
        for (int i = tid; i < NC * PH * PW; i += 256) {
            const int px = i % PW, py = (i / PW) % PH, nc = i / (PW * PH);
            const int ny = 2 * ty * TH - 2 + py, nx = 2 * tx * TW - 2 + px;
            float v = 0.f;
            if ((unsigned)ny < (unsigned)NH && (unsigned)nx < (unsigned)NW)
                v = ld_any(narrow, b * ns_b + nc * ns_c + (int64_t)ny * ns_y + (int64_t)nx * ns_x, n_dtype);
            patch[i] = v;
        }
        __syncthreads();
        for (int r = half * (TH / 2); r < (half + 1) * (TH / 2); ++r) {
            const int qy = ty * TH + r;
            if (qy >= QH) break;
            // the row's 16 wide values first (independent loads in flight together), then 16 x 75 FMAs
            float wrow[TW];
#pragma unroll
            for (int c = 0; c < TW; ++c) {
                const int qx = tx * TW + c;
                wrow[c] = qx < QW ? elem<WT>::ld(wide + b * ws_b + (int64_t)qy * ws_y + (int64_t)qx * ws_x + wc) : 0.f;
            }
#pragma unroll 4
            for (int c = 0; c < TW; ++c) {
                const float wv = wrow[c];
#pragma unroll
                for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky) {
                        const float* pr = patch + (nc * PH + 2 * r + ky) * PW + 2 * c;     // 8-byte aligned, wave-uniform
                        const float2 p01 = *(const float2*)pr, p23 = *(const float2*)(pr + 2);
                        const float p4 = pr[4];
                        float* ac = acc + (nc * 5 + ky) * 5;
                        ac[0] += wv * p01.x; ac[1] += wv * p01.y; ac[2] += wv * p23.x; ac[3] += wv * p23.y; ac[4] += wv * p4;
                    }
            }
        }
    }
    __syncthreads();
    if (half == 1)
#pragma unroll
        for (int i = 0; i < NC * 25; ++i) red[wc][i] = acc[i];
    __syncthreads();
    if (half == 0)
#pragma unroll
        for (int i = 0; i < NC * 25; ++i) atomicAdd(dw + (int64_t)wc * NC * 25 + i, acc[i] + red[wc][i]);
}

// MFMA route for the same gradient: materialise the narrow side's im2col matrix P[q][nc*25 + tap] (bf16, 96 columns, the
// last 21 zero) and feed it with WIDE to the 1x1 weight-gradient kernel above: dW[wc][n] = sum_q WIDE[q][wc] P[q][n].
// The eight gathers of a chunk are issued together: offsets clamped to element 0 for padding / unused columns and the value replaced by
// zero afterwards, storage type chosen outside the loop (as nested `if`s around a per-element dtype switch every gather was a basic block of
// its own: eight dependent round trips per chunk, 64 us per launch for 100 MB).
template <typename T>
__device__ __forceinline__ void im2col_narrow_body(const T* __restrict__ narrow, int64_t ns_b, int64_t ns_c, int64_t ns_y, int64_t ns_x,
                                                   h16_t* __restrict__ P, int B, int QH, int QW, int NH, int NW, int NC) {
    const int64_t total = (int64_t)B * QH * QW * 12;          // 12 chunks of 8 columns per pixel
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ck = i % 12;
        const int64_t q = i / 12;
        const int qx = q % QW, qy = (q / QW) % QH, b = q / ((int64_t)QW * QH);
        float v[8];
        bool ok[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = ck * 8 + e;
            const int nc = n / 25, tap = n % 25, ky = tap / 5, kx = tap % 5;
            const int ny = 2 * qy - 2 + ky, nx = 2 * qx - 2 + kx;
            ok[e] = n < NC * 25 && (unsigned)ny < (unsigned)NH && (unsigned)nx < (unsigned)NW;
            v[e] = elem<T>::ld(narrow + (ok[e] ? b * ns_b + nc * ns_c + (int64_t)ny * ns_y + (int64_t)nx * ns_x : 0));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ok[e] ? v[e] : 0.f;
        *(u32x4*)(P + q * 96 + ck * 8) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
    }
}
__global__ void im2col_narrow_kernel(const void* __restrict__ narrow, int n_dtype, int64_t ns_b, int64_t ns_c, int64_t ns_y, int64_t ns_x,
                                     h16_t* __restrict__ P, int B, int QH, int QW, int NH, int NW, int NC) {
    if (n_dtype == HESIC_H16) im2col_narrow_body<h16_t>((const h16_t*)narrow, ns_b, ns_c, ns_y, ns_x, P, B, QH, QW, NH, NW, NC);
    else im2col_narrow_body<float>((const float*)narrow, ns_b, ns_c, ns_y, ns_x, P, B, QH, QW, NH, NW, NC);
}
// The same matrix with one THREAD per pixel row: a wave's lanes are 64 consecutive pixels of an image row, so each of the 75 gathers reads
// 64 values at stride 2 from ONE image row (4 cache lines per instruction; the chunk-per-thread form above spreads every instruction over
// ~15 rows x 5 pixels), and the row leaves as twelve 16-byte stores.  Same values, same layout.
template <typename T>
__global__ __launch_bounds__(256) void im2col_narrow_rows_kernel(const T* __restrict__ narrow, int64_t ns_b, int64_t ns_c, int64_t ns_y, int64_t ns_x,
                                                                 h16_t* __restrict__ P, int64_t Q, int QH, int QW, int NH, int NW, FastDiv dqw, FastDiv dqh) {
    // the 64 rows of a wave are 12 KB of CONSECUTIVE bytes of P: they go through LDS (13 slots per row: conflict-free 16-byte writes) and
    // leave lane-linear, 1 KB per store instruction (as twelve 16-byte stores per lane at a 192-byte stride the kernel took 51.7 us)
    __shared__ u32x4 stage[256 * 13];
    const int64_t q_raw = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t q = q_raw < Q ? q_raw : Q - 1;          // a surplus thread repeats the last row (not stored)
    const uint32_t r1 = fdiv((uint32_t)q, dqw);
    const int qx = (int)((uint32_t)q - r1 * (uint32_t)QW);
    const uint32_t b = fdiv(r1, dqh);
    const int qy = (int)(r1 - b * (uint32_t)QH);
    const T* base = narrow + (int64_t)b * ns_b + (int64_t)(2 * qy - 2) * ns_y + (int64_t)(2 * qx - 2) * ns_x;
    float v[76];
    v[75] = 0.f;
#pragma unroll
    for (int nc = 0; nc < 3; ++nc)
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            const bool rowok = (unsigned)(2 * qy - 2 + ky) < (unsigned)NH;
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const bool ok = rowok && (unsigned)(2 * qx - 2 + kx) < (unsigned)NW;
                const float t = elem<T>::ld(ok ? base + nc * ns_c + ky * ns_y + kx * ns_x : narrow);      // padding: element 0, replaced by zero
                v[nc * 25 + ky * 5 + kx] = ok ? t : 0.f;
            }
        }
#pragma unroll
    for (int ck = 0; ck < 12; ++ck) {
        u32x4 o = {0u, 0u, 0u, 0u};
        uint32_t* w4 = (uint32_t*)&o;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int n = ck * 8 + 2 * h;
            if (n < 75) w4[h] = pack_h2(v[n], v[n + 1]);
        }
        stage[threadIdx.x * 13 + ck] = o;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q0 = blockIdx.x * (int64_t)blockDim.x + wave * 64;      // first row of this wave
    u32x4* dst = (u32x4*)(P + q0 * 96);
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int c = it * 64 + lane, row = c / 12, ck = c - row * 12;
        if (q0 + row < Q) dst[c] = stage[(wave * 64 + row) * 13 + ck];
    }
}
// dw[wc*NCT + n] = sum_s part[s][wc][n]  (NCT = NC*25 real columns of the 96)
// dbias[co] = sum_s bpart[s][co] (the column sums wgrad_tr_kernel formed next to the 1x1 weight-gradient GEMM), fixed order: 8 slice groups x
// 128 couts per block, eight loads in flight per thread, the groups meet in LDS (one thread per cout walking up to 256 slices serially: 31 us)
__global__ __launch_bounds__(1024) void nw_bias_reduce_kernel(const float* __restrict__ bpart, float* __restrict__ db, int nsplit) {
    __shared__ float red[8][128];
    const int co = threadIdx.x & 127, grp = threadIdx.x >> 7;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = grp * 8; k0 < nsplit; k0 += 64) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nsplit) s[u] += bpart[(k0 + u) * 128 + co];
    }
    red[grp][co] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (grp == 0) {
        float t = red[0][co];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += red[g][co];
        db[co] = t;
    }
}
// The slices of the fused narrow <-> wide weight gradient summed in ONE launch (round 5): block = 32 consecutive patch columns of one wide
// channel (128-byte runs of a slice row) x 8 slice groups that meet in LDS, fixed order; blocks [384, 388) sum the bias column sums the same way.
// (nw_reduce_kernel below reads one value per lane from rows 49 KB apart -- 16 us for 12.6 MB -- and the bias took a second, one-block launch.)
__global__ __launch_bounds__(256) void nw_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit,
                                                        const float* __restrict__ bpart, float* __restrict__ db) {
    __shared__ float red[8][32];
    const int j = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int b = blockIdx.x;
    float s0 = 0.f, s1 = 0.f;
    if (b < 384) {
        const int wc = b / 3, n = (b - wc * 3) * 32 + j;
        const float* src = part + (int64_t)wc * 96 + n;
        int k = grp;
        for (; k + 8 < nsplit; k += 16) { s0 += src[(int64_t)k * 128 * 96]; s1 += src[(int64_t)(k + 8) * 128 * 96]; }
        if (k < nsplit) s0 += src[(int64_t)k * 128 * 96];
        red[grp][j] = s0 + s1;
        __syncthreads();
        if (grp == 0 && n < 75) {
            float t = red[0][j];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += red[g][j];
            dw[wc * 75 + n] = t;
        }
        return;
    }
    if (!bpart) return;
    const int co = (b - 384) * 32 + j;
    int k = grp;
    for (; k + 8 < nsplit; k += 16) { s0 += bpart[k * 128 + co]; s1 += bpart[(k + 8) * 128 + co]; }
    if (k < nsplit) s0 += bpart[k * 128 + co];
    red[grp][j] = s0 + s1;
    __syncthreads();
    if (grp == 0) {
        float t = red[0][j];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += red[g][j];
        db[co] = t;
    }
}
__global__ void nw_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit, int NCT) {
    // one wave per output value (the K split of this route is up to 256 deep: a serial sum per thread was 74 us)
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= 128 * NCT) return;
    const int wc = i / NCT, n = i % NCT;
    float s = 0.f;
    for (int k = lane; k < nsplit; k += 64) s += part[((int64_t)k * 128 + wc) * 96 + n];
    s = wave_sum(s);
    if (lane == 0) dw[i] = s;
}

// ---- narrow <-> wide, 5x5 stride 2 (g_a_conv1 3 -> 128, g_s_conv4 128 -> 3 transposed): one launch, no im2col matrix (round 5).
//   dW[wc][n = c * 25 + ky * 5 + kx] = sum_q WIDE[q][wc] * NARROW[c][2 qy + ky - 2][2 qx + kx - 2]
// Rounds 2-4 wrote the (Q x 96) 16-bit im2col matrix of the narrow image (84 MB at B = 8, 512^2), ran wgrad_tr_kernel on it as a one-tap
// GEMM (reading it and the 134 MB wide map), and summed the slices: 33 + 60 + 16 us per layer, five layers per step, for 0.63 GMAC each.
// Here a block walks 64-pixel row segments of the wide grid: the wide tile (64 px x 128 ch) arrives by LDS-DMA as in wgrad_row_kernel; the
// narrow WINDOW the segment's 75 patch columns are made of -- 3 channels x 5 rows x 131 consecutive pixels, fp32 -- arrives by 4-byte LDS-DMA
// as an even and an odd pixel plane per (channel, row) (tap kx = plane kx & 1 from element kx >> 1 on: unit-stride, conflict-free reads);
// the four waves turn it into the segment's (64 px x 96 col) 16-bit patch tile in LDS (24 columns per wave, lane = pixel), and both operands
// reach the matrix cores through transposing reads.  The launch is bound by the wide map's HBM read (134 MB) instead of three passes.
struct NwArgs {
    const void* wide; const float* narrow; float* part; float* bias_part;        // part [nsplit][128][96], bias_part [nsplit][128] or null
    int64_t ns_b, ns_c, ns_y, ns_x;                                              // narrow strides in elements
    int QH, QW, NH, NW;
    int64_t Q, chunk;
};
constexpr int NWF_WIDE = 64 * 256, NWF_WIN = 8192, NWF_STAGE = NWF_WIDE + NWF_WIN, NWF_NST = 3, NWF_PATCH = 64 * 256;
constexpr int NWF_LDS = NWF_NST * NWF_STAGE + NWF_PATCH;

template <bool BIAS>
__global__ __launch_bounds__(256) void wgrad_nw_fused_kernel(const NwArgs a) {
    constexpr int STAGE = NWF_STAGE, NST = NWF_NST, PLW = 68;                    // 68 elements per (channel, row, parity) plane of the window
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem + NST * STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t q_begin = split * a.chunk;
    const int64_t q_end = (q_begin + a.chunk < a.Q) ? q_begin + a.chunk : a.Q;
    const int nsteps = (int)((q_end - q_begin) >> 6);
    constexpr uint32_t OOB = 0x80000000u;
    asm volatile("" ::"v"((__attribute__((address_space(3))) unsigned char*)smem) : "memory");
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wide, 0, (int)OOB, 0x00020000);
    const int64_t neg = (2 * a.ns_y + 2 * a.ns_x) * 4;                            // the window starts two rows / two pixels before (2 qy, 2 qx)
    const __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)a.narrow - neg), 0, (int)OOB, 0x00020000);
    const int lrow = lane >> 4, pslot = lane & 15;
    const uint32_t v_w = (uint32_t)(lrow * 256 + ((pslot ^ ((lrow & 3) << 2)) << 4));
    // window element e = (wave * 8 + i) * 64 + lane of the stage -> plane row r = e / 68 = (c * 5 + ky) * 2 + parity, element m = e % 68
    int wky[8], wx[8];
    uint32_t v_n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = (wave * 8 + i) * 64 + lane;
        const int r = e / PLW, m = e - r * PLW;
        const int c = r / 10, rr = r - c * 10;
        wky[i] = rr >> 1;
        wx[i] = 2 * m + (rr & 1);
        v_n[i] = r < 30 ? (uint32_t)((c * a.ns_c + wky[i] * a.ns_y + wx[i] * a.ns_x) * 4) : OOB;
    }
    uint32_t sq = (uint32_t)q_begin;
    int sqx, sqy, sqb;
    {
        const uint32_t r1 = sq / (uint32_t)a.QW;
        sqx = (int)(sq - r1 * (uint32_t)a.QW);
        sqb = (int)(r1 / (uint32_t)a.QH);
        sqy = (int)(r1 - (uint32_t)sqb * (uint32_t)a.QH);
    }
    auto issue = [&](int buf) {
        unsigned char* wt = smem + buf * STAGE;
        unsigned char* nt = wt + NWF_WIDE;
        const bool inq = sq < (uint32_t)q_end;
        const uint32_t vw = inq ? v_w : OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = wave * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(wt + n * 1024), 16, (int)vw,
                                                     (int)((sq + (uint32_t)(4 * n)) * 256u), 0, 0);
        }
        const uint32_t so = (uint32_t)((sqb * a.ns_b + 2 * sqy * a.ns_y + 2 * sqx * a.ns_x) * 4);
        const int y0 = 2 * sqy - 2, x0 = 2 * sqx - 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = inq && (unsigned)(y0 + wky[i]) < (unsigned)a.NH && (unsigned)(x0 + wx[i]) < (unsigned)a.NW;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(nr, (__attribute__((address_space(3))) void*)(nt + (wave * 8 + i) * 256), 4,
                                                     (int)(ok ? v_n[i] : OOB), (int)so, 0, 0);
        }
        sq += 64;
        sqx += 64;
        if (sqx >= a.QW) {
            sqx = 0;
            if (++sqy >= a.QH) { sqy = 0; ++sqb; }
        }
    };
    const int frow = lane & 31, fh = lane >> 5, g = lane >> 4, t = lane & 15;
    const int chs = (g & 1) * 16 + 4 * (t & 3), rsub = fh * 8 + (t >> 2);
    auto foff = [&](int cb) {
        const int ch = cb + chs;
        return rsub * 256 + (((ch >> 3) ^ ((rsub & 3) << 2)) << 4) + (ch & 7) * 2;
    };
    const int a_off = foff(wave * 32);
    int b_off[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) b_off[n] = foff(n * 32);
    f32x16 acc[3];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float bsum = 0.f;
    if (nsteps > 0) {
#pragma unroll
        for (int p = 0; p < NST - 1; ++p) issue(p);
    }
    int buf = 0;
    for (int step = 0; step < nsteps; ++step) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 12) : "memory");
        __builtin_amdgcn_s_barrier();                       // the stage has landed for every wave; the patch tile of the last step is read out
        issue(buf == 0 ? NST - 1 : buf - 1);
        const unsigned char* st = smem + buf * STAGE;
        {
            // patch tile: lane = pixel, this wave's 24 columns (three 16-byte slots of the pixel's row); columns >= 75 are zero
            const float* win = (const float*)(st + NWF_WIDE);
            uint32_t w[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                float v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int col = wave * 24 + 2 * j + h;                    // wave-uniform
                    const int c = col / 25, r = col - c * 25, ky = r / 5, kx = r - ky * 5;
                    const float x = win[((c * 5 + ky) * 2 + (kx & 1)) * PLW + (kx >> 1) + lane];
                    v[h] = col < 75 ? x : 0.f;
                }
                w[j] = pack_h2(v[0], v[1]);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int slot = wave * 3 + k;
                *(u32x4*)(patch + lane * 256 + ((slot ^ ((lane & 3) << 2)) << 4)) = u32x4{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned char* kw = st + ks * 4096;
            const unsigned char* kp = patch + ks * 4096;
            const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(kw + a_off));
            const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(kw + a_off + 1024));
            const h16x8 af = __builtin_bit_cast(h16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
            if constexpr (BIAS) {
                const u32x2 lo = __builtin_bit_cast(u32x2, l0), hi = __builtin_bit_cast(u32x2, l1);
                bsum += ((h2f_lo(lo.x) + h2f_hi(lo.x)) + (h2f_lo(lo.y) + h2f_hi(lo.y))) + ((h2f_lo(hi.x) + h2f_hi(hi.x)) + (h2f_lo(hi.y) + h2f_hi(hi.y)));
            }
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const s16x4 s0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(kp + b_off[n]));
                const s16x4 s1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(kp + b_off[n] + 1024));
                const h16x8 bf = __builtin_bit_cast(h16x8, __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[n] = mfma_32x32x16_h16(af, bf, acc[n], 0, 0, 0);
            }
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
    }
    if constexpr (BIAS) {
        const float sb = bsum + __shfl_xor(bsum, 32, 64);
        if (fh == 0) a.bias_part[(int64_t)split * 128 + wave * 32 + frow] = sb;
    }
    float* out = a.part + (int64_t)split * 128 * 96;
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 96 + n * 32 + frow] = acc[n][r];
}

// ---- narrow x narrow, stride 1 (pre_conv 6 -> 3, after_conv 6 -> 3 transposed), 5x5 pad 2.
//   conv:        dW[co][ci][ky][kx] = sum_p dy[co][p] * x[ci][p + k - 2]
//   transposed:  dW[ci][co][ky][kx] = sum_i x[ci][i]  * dy[co][i + k - 2]
// i.e. in both cases  dW[a][s][k] = sum_p A[a][p] * S[s][p + k - 2]  with A the un-shifted tensor (dy | x), S the shifted one
// (x | dy).  A block stages a 16x64 tile of A and the haloed tile of S in LDS; thread = (a, s, ky) keeps 5 kx
// accumulators and slides along each row (one new S value + one A value per pixel).
// dw[a][s][ky][kx] = sum_p A[a][p] * S[s][p + (ky-2, kx-2)] for the 6 <-> 3 channel 5x5 stride-1 stages (A = dY, S = X for the
// conv; A = X, S = dY for the transposed op).  Lanes are PIXELS (64 columns of a tile row), the 450 sums live in registers:
// the NS*5 (s, ky) pairs are dealt to the 4 waves, a wave keeps NA*5 accumulators per pair (<= 120 per lane) and walks all
// 16 rows of the 16x64 LDS tile: per row and pair 5 shifted reads of S feed NA*5 FMAs.  Persistent blocks: the cross-lane
// reduction and the atomics happen once per block.
// DMA (round 5): both tensors fp32 with 32-bit offsets inside one image -- the tile rows go global -> LDS as 4-byte LDS-DMA, one instruction
// per (channel, row) of 64 pixels (+ one masked to four lanes for the halo columns 64 .. 67 of the shifted operand), addresses from scalar row
// arithmetic: no registers, no per-element div / mod, every row of the tile in flight at once.  The register-staged form (40 loads per thread
// and tile in rounds of 4 - 8, each element's (c, y, x) from two divisions) stays for the other storage types.
template <int NA, int NS_, bool DMA = false>
__global__ __launch_bounds__(256, 2) void sconv_wgrad_nn_kernel(const void* __restrict__ A, int a_dtype, int64_t as_b, int64_t as_c,
                                                             int64_t as_y, int64_t as_x, const void* __restrict__ S, int s_dtype,
                                                             int64_t ss_b, int64_t ss_c, int64_t ss_y, int64_t ss_x,
                                                             float* __restrict__ dw, float* __restrict__ part, int a_major, int B, int H,
                                                             int W) {
    constexpr int TH = 16, TW = 64, PH = TH + 4, PW = TW + 4;
    constexpr int NP = NS_ * 5, NPW = (NP + 3) / 4;              // (s, ky) pairs, pairs per wave
    constexpr int SB = NS_ > NA ? 4 : 8;                         // staging loads in flight per lane (the <3, 6> form has no registers for 8)
    __shared__ float at[NA * TH * TW];
    __shared__ float st[NS_ * PH * PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc[NPW][NA][5];
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[i][a][k] = 0.f;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int64_t ntiles = (int64_t)tiles_x * tiles_y * B;
    for (int64_t tile = xcd_remap(blockIdx.x, gridDim.x); tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / ((int64_t)tiles_x * tiles_y);
        __syncthreads();
        // staging: SB loads in flight per lane, then the LDS stores (as load / store pairs in a loop the compiler issued them one by
        // one: ~45 dependent memory round trips per tile, 150 us per launch for 75 MB); the storage type is chosen OUTSIDE the loops (a
        // per-element dtype branch keeps the loads in separate basic blocks, i.e. serial again)
        auto stage_a = [&](auto tag) {
            using T = decltype(tag);
            for (int i0 = 0; i0 < NA * TH * TW; i0 += 256 * SB) {
                float v[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int i = i0 + tid + 256 * u;
                    const int px = i % TW, py = (i / TW) % TH, c = i / (TW * TH);
                    const int y = ty * TH + py, x = tx * TW + px;
                    const bool ok = i < NA * TH * TW && y < H && x < W;
                    v[u] = elem<T>::ld((const T*)A + (ok ? b * as_b + c * as_c + (int64_t)y * as_y + (int64_t)x * as_x : 0));
                    v[u] = ok ? v[u] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < SB; ++u)
                    if (i0 + tid + 256 * u < NA * TH * TW) at[i0 + tid + 256 * u] = v[u];
            }
        };
        auto stage_s = [&](auto tag) {
            using T = decltype(tag);
            for (int i0 = 0; i0 < NS_ * PH * PW; i0 += 256 * SB) {
                float v[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int i = i0 + tid + 256 * u;
                    const int px = i % PW, py = (i / PW) % PH, c = i / (PW * PH);
                    const int y = ty * TH - 2 + py, x = tx * TW - 2 + px;
                    const bool ok = i < NS_ * PH * PW && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                    v[u] = elem<T>::ld((const T*)S + (ok ? b * ss_b + c * ss_c + (int64_t)y * ss_y + (int64_t)x * ss_x : 0));
                    v[u] = ok ? v[u] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < SB; ++u)
                    if (i0 + tid + 256 * u < NS_ * PH * PW) st[i0 + tid + 256 * u] = v[u];
            }
        };
        if constexpr (DMA) {
            constexpr uint32_t POISON = 0x80000000u;
            const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)A + b * as_b), 0, (int)POISON, 0x00020000);
            const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)S + b * ss_b), 0, (int)POISON, 0x00020000);
            const int xa = tx * TW + lane, xs0 = tx * TW - 2 + lane, xs1 = xs0 + 64;
            const bool xa_ok = xa < W, xs0_ok = (unsigned)xs0 < (unsigned)W, xs1_ok = (unsigned)xs1 < (unsigned)W;
            for (int idx = wave; idx < NA * TH; idx += 4) {                       // wave-uniform: scalar row arithmetic
                const int c = idx / TH, r = idx - c * TH, y = ty * TH + r;
                const uint32_t vo = (y < H && xa_ok) ? (uint32_t)((c * (int)as_c + y * (int)as_y + xa * (int)as_x) * 4) : POISON;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (__attribute__((address_space(3))) void*)(at + idx * TW), 4, (int)vo, 0, 0, 0);
            }
            for (int idx = wave; idx < NS_ * PH; idx += 4) {
                const int c = idx / PH, r = idx - c * PH, y = ty * TH - 2 + r;
                const bool yok = (unsigned)y < (unsigned)H;
                const int rb = (c * (int)ss_c + y * (int)ss_y) * 4;
                const uint32_t v0 = (yok && xs0_ok) ? (uint32_t)(rb + xs0 * (int)ss_x * 4) : POISON;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sr, (__attribute__((address_space(3))) void*)(st + idx * PW), 4, (int)v0, 0, 0, 0);
                if (lane < 4) {                                                   // halo columns 64 .. 67: only these four lanes write
                    const uint32_t v1 = (yok && xs1_ok) ? (uint32_t)(rb + xs1 * (int)ss_x * 4) : POISON;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(sr, (__attribute__((address_space(3))) void*)(st + idx * PW + 64), 4, (int)v1, 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (a_dtype == HESIC_H16) stage_a(h16_t{}); else stage_a(float{});
            if (s_dtype == HESIC_H16) stage_s(h16_t{}); else stage_s(float{});
        }
        __syncthreads();
        // Round 5: a lane owns FOUR consecutive pixels of a row (a wave = 4 rows x 64 columns per pass, four passes per tile): the eight S values
        // its four pixels share across the five kx shifts arrive as two 16-byte reads and feed 20 NA FMAs, where one pixel per lane read five
        // 4-byte values for 5 NA FMAs -- the loop was bound by LDS instructions (109 / 76 us per launch for 12 us of FMA issue).  The sums
        // are the same 450; only their association over pixels changes.
#pragma unroll 1
        for (int rg = 0; rg < TH / 4; ++rg) {
            const int r = rg * 4 + (lane >> 4), x0 = (lane & 15) * 4;
            f32x4 av[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) av[a] = *(const f32x4*)(at + (a * TH + r) * TW + x0);
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                const int p = wave + 4 * i;                       // wave-uniform
                if (p < NP) {
                    const int is = p / 5, ky = p - is * 5;
                    const float* sr = st + (is * PH + r + ky) * PW + x0;
                    const f32x4 s0 = *(const f32x4*)sr, s1 = *(const f32x4*)(sr + 4);
                    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            float t = acc[i][a][k];
                            t = fmaf(av[a].x, sv[k], t);
                            t = fmaf(av[a].y, sv[k + 1], t);
                            t = fmaf(av[a].z, sv[k + 2], t);
                            t = fmaf(av[a].w, sv[k + 3], t);
                            acc[i][a][k] = t;
                        }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave + 4 * i;
        if (p < NP) {
            const int is = p / 5, ky = p - is * 5;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                // weight layout: [a][s][ky][kx] (a_major) or [s][a][ky][kx]
                const int64_t base = a_major ? ((int64_t)(a * NS_ + is) * 5 + ky) * 5 : ((int64_t)(is * NA + a) * 5 + ky) * 5;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float v = wave_sum(acc[i][a][k]);
                    // 512 blocks x 450 same-address atomics cost more than the whole contraction (measured 0.27 of 0.55 ms):
                    // with a workspace every block leaves its partial row there and nn_partial_reduce_kernel adds them up
                    if (lane == 0) { if (part) part[(int64_t)blockIdx.x * (NA * NS_ * 25) + base + k] = v; else atomicAdd(dw + base + k, v); }
                }
            }
        }
    }
}

__global__ void nn_partial_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblocks, int n) {
    // one wave per output value: its lanes stride over the per-block partial rows
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    float s = 0.f;
    for (int b = lane; b < nblocks; b += 64) s += part[(int64_t)b * n + i];
    s = wave_sum(s);
    if (lane == 0) dw[i] = s;
}

__global__ __launch_bounds__(256) void sconv_dbias_kernel(const SWArgs a) {
    __shared__ float red[4];
    const int co = blockIdx.y;
    const int64_t n = (int64_t)a.B * a.Ho * a.Wo;
    float acc = 0.f;
    const int64_t plane = (int64_t)a.Ho * a.Wo;
    if (a.y_dtype == HESIC_F32 && a.ys_x == 1 && a.ys_y == a.Wo && (plane & 3) == 0 && ((a.ys_b | a.ys_c) & 3) == 0 && !((uintptr_t)a.dy & 15)) {
        // planar fp32 image gradient (the 3-channel outputs of g_s_conv4 / pre_conv / after_conv): 16-byte loads along the plane, no
        // per-element index arithmetic (the general loop below spent ~28 us on 25 MB: three 64-bit divisions per value)
        const int64_t q4 = plane >> 2;
        for (int b = 0; b < a.B; ++b) {
            const f32x4* src = (const f32x4*)((const float*)a.dy + (int64_t)b * a.ys_b + (int64_t)co * a.ys_c);
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
            for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < q4; i += (int64_t)gridDim.x * blockDim.x) s4 += src[i];
            acc += (s4.x + s4.y) + (s4.z + s4.w);
        }
    } else
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
// few channels (the 3-channel image-side GDNs): thread = pixel, all C*C + C sums in registers, wave reduction, then one atomic
// per wave and value -- the (i, j)-per-thread kernel above would run 9 lanes per block
template <int C>
__global__ __launch_bounds__(256) void gdn_bwd_param_small_kernel(const void* __restrict__ x, const float* __restrict__ dn,
                                                                  float* __restrict__ dgp, float* __restrict__ dbp, int64_t P, int dtype) {
    float g[C][C], bsum[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) g[i][j] = 0.f;
    }
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        float d[C], sq[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            d[c] = dn[p * C + c];
            const float xv = ld_any(x, p * C + c, dtype);
            sq[c] = xv * xv;
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            bsum[i] += d[i];
#pragma unroll
            for (int j = 0; j < C; ++j) g[i][j] = fmaf(d[i], sq[j], g[i][j]);
        }
    }
    // wave sums -> LDS -> one atomic per block and value (same-address atomics serialise in L2: keep them few)
    __shared__ float red[4][C * C + C];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const float b = wave_sum(bsum[i]);
        if ((threadIdx.x & 63) == 0) red[wv][C * C + i] = b;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float v = wave_sum(g[i][j]);
            if ((threadIdx.x & 63) == 0) red[wv][i * C + j] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < C * C + C) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x < C * C) atomicAdd(dgp + threadIdx.x, v);
        else atomicAdd(dbp + (threadIdx.x - C * C), v);
    }
}

// The three passes above in ONE for the 3-channel image-side GDNs (pre_gdn / after_gdn, newnet1.py:630,669, under autograd): thread = pixel,
// gamma' / beta' in registers, dn never leaves them -- dx and the C*C + C parameter sums come out of one read of x and gy (the passes
// took 42 + ~30 + 31 us per GDN on a 512^2 batch-8 image and wrote / re-read two fp32 workspaces).  Same formulas in the same order:
// dx is bit-identical; the parameter sums keep the per-thread accumulation and the one-atomic-per-block finish of the kernel above.
// PL: planar (B, C, HW) tensors -- blockIdx.y = image, P = HW, channel stride HW; otherwise NHWC with P = all pixels.
template <int C, typename T, bool PL = false>
__global__ __launch_bounds__(256) void gdn_bwd_small_fused_kernel(const T* __restrict__ x, const T* __restrict__ gy, const float* __restrict__ beta,
                                                                  const float* __restrict__ gamma, T* __restrict__ dx, float* __restrict__ dgp,
                                                                  float* __restrict__ dbp, int64_t P, int inverse, float beta_bound) {
    const int64_t cs = PL ? P : 1, ps = PL ? 1 : C;              // element strides of channel and pixel
    if (PL) { x += blockIdx.y * (int64_t)C * P; gy += blockIdx.y * (int64_t)C * P; dx += blockIdx.y * (int64_t)C * P; }
    float gm[C][C], bt[C], g[C][C], bsum[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        bt[i] = reparam(beta[i], beta_bound);
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) { gm[i][j] = reparam(gamma[i * C + j], kGammaBound); g[i][j] = 0.f; }
    }
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        float xv[C], gv[C], sq[C], dn[C], d0[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            xv[c] = elem<T>::ld(x + p * ps + c * cs);
            gv[c] = elem<T>::ld(gy + p * ps + c * cs);
            sq[c] = xv[c] * xv[c];
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            float norm = bt[i];
#pragma unroll
            for (int j = 0; j < C; ++j) norm += gm[i][j] * xv[j] * xv[j];
            if (inverse) {
                const float sqn = sqrtf(norm);
                dn[i] = 0.5f * gv[i] * xv[i] / sqn;
                d0[i] = gv[i] * sqn;
            } else {
                const float rs = rsqrtf(norm);
                dn[i] = -0.5f * gv[i] * xv[i] * rs * rs * rs;
                d0[i] = gv[i] * rs;
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float sj = 0.f;
#pragma unroll
            for (int k = 0; k < C; ++k) sj += gm[k][j] * dn[k];
            elem<T>::st(dx + p * ps + j * cs, d0[j] + 2.f * xv[j] * sj);
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            bsum[i] += dn[i];
#pragma unroll
            for (int j = 0; j < C; ++j) g[i][j] = fmaf(dn[i], sq[j], g[i][j]);
        }
    }
    __shared__ float red[4][C * C + C];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const float b = wave_sum(bsum[i]);
        if ((threadIdx.x & 63) == 0) red[wv][C * C + i] = b;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float v = wave_sum(g[i][j]);
            if ((threadIdx.x & 63) == 0) red[wv][i * C + j] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < C * C + C) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x < C * C) atomicAdd(dgp + threadIdx.x, v);
        else atomicAdd(dbp + (threadIdx.x - C * C), v);
    }
}

__global__ void gdn_bwd_chain_kernel(const float* __restrict__ beta, const float* __restrict__ gamma, const float* __restrict__ dgp,
                                     const float* __restrict__ dbp, float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                     float beta_bound, int accumulate) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < C * C) {
        const float th = gamma[e], g = dgp[e] * 2.f * fmaxf(th, kGammaBound);
        const float v = (th >= kGammaBound || g < 0.f) ? g : 0.f;
        dgamma[e] = accumulate ? dgamma[e] + v : v;
    }
    if (e < C) {
        const float th = beta[e], g = dbp[e] * 2.f * fmaxf(th, beta_bound);
        const float v = (th >= beta_bound || g < 0.f) ? g : 0.f;
        dbeta[e] = accumulate ? dbeta[e] + v : v;
    }
}

// The fused GDN backward's last step: sum the <= 256 block partials of (dgamma' | dbeta') in a fixed order (16 lanes of 4 floats x 16 slice
// groups per block, the groups meet in LDS: the shape of reduce_wide_body) and apply the reparametrisation chain on the way out --
// gdn_bwd_chain_kernel folded in, one launch per GDN less.
__global__ __launch_bounds__(256) void gdn_param_finish_kernel(const float* __restrict__ part, int nb, const float* __restrict__ beta,
                                                               const float* __restrict__ gamma, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float beta_bound, int accumulate) {
    constexpr int NP = 128 * 128 + 128;
    __shared__ f32x4 red[16][16];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = (blockIdx.x * 16 + el) * 4;          // NP is a multiple of 64: every lane is in range
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int k = grp;
    for (; k + 16 < nb; k += 32) {
        s0 += *(const f32x4*)(part + (int64_t)k * NP + i);
        s1 += *(const f32x4*)(part + (int64_t)(k + 16) * NP + i);
    }
    if (k < nb) s0 += *(const f32x4*)(part + (int64_t)k * NP + i);
    red[grp][el] = s0 + s1;
    __syncthreads();
    if (grp == 0) {
        f32x4 sum = red[0][el];
#pragma unroll
        for (int g = 1; g < 16; ++g) sum += red[g][el];
        const bool is_beta = i >= 128 * 128;
        const float* th_p = is_beta ? beta + (i - 128 * 128) : gamma + i;
        float* out = is_beta ? dbeta + (i - 128 * 128) : dgamma + i;
        const float bound = is_beta ? beta_bound : kGammaBound;
        const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {          // scalar accesses: parameters and gradient slots of a flat buffer are only 4-byte aligned
            const float th = th_p[e], g = sv[e] * 2.f * fmaxf(th, bound);
            const float v = (th >= bound || g < 0.f) ? g : 0.f;
            out[e] = accumulate ? out[e] + v : v;
        }
    }
}

// ---- GDN backward, bf16 storage, C = 128: one pass over (x, gy) on the matrix cores.
//   GEMM1  n = beta' + gamma' x^2          -> r = n^-1/2 | n^1/2,  t1 = g r,  dn = -1/2 g x r^3 | +1/2 g x / r
//   GEMM2  s_j = sum_i gamma'[i,j] dn_i     -> dx = t1 + 2 x s
// dn (bf16) also goes to a workspace; dgamma' = dn^T x^2 and dbeta' = colsum(dn) are then produced by the 1x1
// weight-gradient MFMA kernel (x squared on load).  After the tile load every wave owns its 32 pixel rows of both
// LDS tiles, so no block barrier is needed inside the tile loop.
// 256-byte rows, 16-byte slots XOR-ed with the row's low four bits, the two bit pairs swapped: any 16 consecutive rows put one logical slot
// on 16 different physical slots (the row-wise ds_read_b128 fragment reads), AND the rows 4k .. 4k+3 put the logical slots c .. c+3 on four
// different aligned groups of four (the transposing reads of the third GEMM: 4 pixel rows x 64 bytes per LDS cycle).  The plain
// slot ^ (row & 15) of rounds 2-3 served only the first: 52 % of the kernel's LDS cycles were bank conflicts.
#ifndef GB_SWZ_PLAIN
__device__ __forceinline__ int gb_off(int row, int slot) { return (row * 16 + (slot ^ (((row & 3) << 2) | ((row >> 2) & 3)))) * 16; }
#else
__device__ __forceinline__ int gb_off(int row, int slot) { return (row * 16 + (slot ^ (row & 15))) * 16; }
#endif

// PAR = 1 (the default path): the parameter gradients ride on the same pass.  dgamma'[i][j] = sum_p dn[p][i] x[p][j]^2 is a third GEMM
// with the PIXELS as K: both operands come out of the two LDS tiles through transposing reads
// (ds_read_b64_tr_b16: lane = one channel, 4 consecutive pixels per read), 8 k-steps x 4 MFMAs per wave and tile into a 128 x 128 fp32
// accumulator per block -- wave w owns columns 32 w .. 32 w + 31 (64 accumulation registers) and walks all 128 pixels of the tile, so the
// tile loop gains two block barriers; dbeta'[i] = sum_p dn[p][i] is summed from the same A fragments.  The block writes ONE
// (128 x 128 + 128) partial (`part`) at the end; dn never goes to HBM.  Before (round 2): dn written out (134 MB on a 256^2 map), read back with x by a 1-tap
// launch of wgrad_tr_kernel cut into 256 pixel slices, whose 256 partial tiles a third launch reduced next to the column sums of dn.
// The tile loop is ONE basic block: `inverse` is a template parameter, loads and stores are buffer-addressed (rows past P read zeros /
// are dropped by the range check, no branches), so the compiler can count the outstanding memory operations exactly -- the loop top
// waits for the next tile's loads only, not for the stores of the tile before (the branchy form waited vmcnt(0): a store round trip
// per tile) -- and can batch the LDS reads of the epilogues across channel groups.
template <bool PAR, bool INV>
__global__ __launch_bounds__(256) void gdn128_bwd_kernel(const h16_t* __restrict__ x, const h16_t* __restrict__ gy,
                                                         const float* __restrict__ beta, const float* __restrict__ gamma,
                                                         h16_t* __restrict__ dx, h16_t* __restrict__ dn_out, float* __restrict__ part,
                                                         int64_t P, float beta_bound) {
    constexpr bool inverse = INV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* gs = smem;                   // gamma'   [i][j]
    unsigned char* gt = smem + 32768;           // gamma'^T [j][i]
    unsigned char* xs = smem + 65536;           // x tile   [128 px][128 ch]  (becomes dx)
    unsigned char* ds = smem + 98304;           // gy tile  (becomes dn)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    // gamma' -> LDS in both orientations from ONE coalesced read: row-major 16-byte chunks as they come, the transpose as 2-byte scatters
    // (the strided gather gamma[(8 slot + e) * 128 + row] this replaces -- 32 dependent 4-byte loads 512 bytes apart per thread -- was most
    // of the ~10 us a launch spends outside its tiles)
    for (int c = tid; c < 128 * 16; c += 256) {
        const int row = c >> 4, slot = c & 15;
        const f32x4 a0 = *(const f32x4*)(gamma + row * 128 + slot * 8), a1 = *(const f32x4*)(gamma + row * 128 + slot * 8 + 4);
        const float v[8] = {reparam(a0.x, kGammaBound), reparam(a0.y, kGammaBound), reparam(a0.z, kGammaBound), reparam(a0.w, kGammaBound),
                            reparam(a1.x, kGammaBound), reparam(a1.y, kGammaBound), reparam(a1.z, kGammaBound), reparam(a1.w, kGammaBound)};
        const u32x4 pk = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
        *(u32x4*)(gs + gb_off(row, slot)) = pk;
        const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *(h16_t*)(gt + gb_off(slot * 8 + e, row >> 3) + (row & 7) * 2) = (h16_t)(e & 1 ? w4[e >> 1] >> 16 : w4[e >> 1] & 0xffffu);
    }
    float* bl = (float*)(smem + 131072);        // beta' (read back per tile: 64 registers of a lone wave's budget go to the third GEMM)
    if (tid < 128) bl[tid] = reparam(beta[tid], beta_bound);
    __syncthreads();
    const int64_t ntiles = (P + 127) / 128;
    const int r0 = wave * 32;                   // this wave's rows in both tiles
    // 128 KB of LDS = one block (one wave per SIMD) per CU: nothing else hides the HBM latency, so the rows of tile t+1
    // are requested into registers before tile t is computed (64 VGPRs; the budget of a lone wave is 512)
    u32x4 px[8], pg[8];
    const int nbytes = (int)(P * 256);          // P < 2^22 (host): 32-bit byte offsets
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)gy, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dxr = __builtin_amdgcn_make_buffer_rsrc((void*)dx, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dnr = __builtin_amdgcn_make_buffer_rsrc((void*)(PAR ? dx : dn_out), 0, nbytes, 0x00020000);
    const int lofs = (r0 + (lane >> 4)) * 256 + (lane & 15) * 16;      // this lane's first row / slot, bytes
    auto fetch = [&](int64_t t) {
        const int o = (int)t * 32768 + lofs;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            px[it] = __builtin_amdgcn_raw_buffer_load_b128(xr, o + it * 1024, 0, 0);
            pg[it] = __builtin_amdgcn_raw_buffer_load_b128(gr, o + it * 1024, 0, 0);
        }
    };
    fetch(blockIdx.x);
    // eight stores the range check drops: the loop is then entered with the same queue of memory operations its back edge carries
    // (16 loads, then 8 stores), and the wait at its top can be "all but the last 8" instead of "everything" on both paths
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 8; ++it) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, dxr, nbytes + it * 1024 + lofs, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 g3[4];                               // dgamma' partial of this block's pixels, columns 32 wave .. + 31: [i block]
    float cs = 0.f;                             // dbeta' partial: channel 32 wave + (lane & 31), this lane's pixel halves
    if constexpr (PAR) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) g3[j][r] = 0.f;
    }
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // wave-private 32 rows x 16 slots of x and gy: registers -> LDS, then the next tile's loads go out
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int c = it * 64 + lane, row = c >> 4, slot = c & 15;
            *(u32x4*)(xs + gb_off(r0 + row, slot)) = px[it];
            *(u32x4*)(ds + gb_off(r0 + row, slot)) = pg[it];
        }
        fetch(tile + gridDim.x);                 // past the last tile: every offset out of range, zeros
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const u32x4 raw = *(const u32x4*)(xs + gb_off(r0 + frow, ks * 2 + fh));
            const float f0 = h2f_lo(raw.x), f1 = h2f_hi(raw.x);
            const float f2 = h2f_lo(raw.y), f3 = h2f_hi(raw.y);
            const float f4 = h2f_lo(raw.z), f5 = h2f_hi(raw.z);
            const float f6 = h2f_lo(raw.w), f7 = h2f_hi(raw.w);
            const u32x4 sq = u32x4{pack_h2(f0 * f0, f1 * f1), pack_h2(f2 * f2, f3 * f3), pack_h2(f4 * f4, f5 * f5), pack_h2(f6 * f6, f7 * f7)};
            const h16x8 xf = __builtin_bit_cast(h16x8, sq);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const h16x8 gf = *(const h16x8*)(gs + gb_off(i * 32 + frow, ks * 2 + fh));
                acc[i] = mfma_32x32x16_h16(gf, xf, acc[i], 0, 0, 0);
            }
        }
        // lane: pixel r0+frow, channels i*32 + 8g + 4fh + e.  t1 stays in acc, dn replaces gy in LDS
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = i * 32 + 8 * g + 4 * fh;
                const int off = gb_off(r0 + frow, ch >> 3) + (ch & 7) * 2;
                const u32x2 xr = *(const u32x2*)(xs + off), gr = *(const u32x2*)(ds + off);
                const float xv[4] = {h2f_lo(xr.x), h2f_hi(xr.x), h2f_lo(xr.y), h2f_hi(xr.y)};
                const float gv[4] = {h2f_lo(gr.x), h2f_hi(gr.x), h2f_lo(gr.y), h2f_hi(gr.y)};
                float dn[4];
                const f32x4 bq = *(const f32x4*)(bl + ch);
                const float bvv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n = acc[i][4 * g + e] + bvv[e];
                    if (inverse) {
                        const float rs = __builtin_amdgcn_rsqf(n);    // one v_rsq instead of sqrt + division (outputs are bf16); n >= beta' > 0 is a normal number: no range scaling
                        dn[e] = 0.5f * gv[e] * xv[e] * rs;
                        acc[i][4 * g + e] = gv[e] * (n * rs);
                    } else {
                        const float rs = __builtin_amdgcn_rsqf(n);
                        dn[e] = -0.5f * gv[e] * xv[e] * rs * rs * rs;
                        acc[i][4 * g + e] = gv[e] * rs;
                    }
                }
                *(u32x2*)(ds + off) = u32x2{pack_h2(dn[0], dn[1]), pack_h2(dn[2], dn[3])};
            }
        // GEMM2: s[j][p] = sum_i gamma'^T[j][i] dn[p][i]
        f32x16 s2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s2[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const h16x8 df = *(const h16x8*)(ds + gb_off(r0 + frow, ks * 2 + fh));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const h16x8 gf = *(const h16x8*)(gt + gb_off(i * 32 + frow, ks * 2 + fh));
                s2[i] = mfma_32x32x16_h16(gf, df, s2[i], 0, 0, 0);
            }
        }
        if constexpr (PAR) {
            // GEMM3: A[i][p] = dn (ds tile), B[p][j] = x^2 (xs tile, still x here), K = the tile's 128 pixels; wave w owns the columns
            // j = 32 w .. 32 w + 31 of the result.  It reads every wave's rows: a barrier after dn is complete, another one before the
            // x tile turns into dx.  Rows past P are zero-filled on load (gy = 0 -> dn = 0): they add nothing.
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const int tg = lane >> 4, tt = lane & 15, wv = __builtin_amdgcn_readfirstlane(wave);
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int pix = ks * 16 + (tg >> 1) * 8 + (tt >> 2);
                auto tr = [&](const unsigned char* tile_, int cb, int px_) {
                    const int ch = cb + (tg & 1) * 16 + 4 * (tt & 3);
                    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile_ + gb_off(px_, ch >> 3) + (ch & 7) * 2));
                };
                // B: x^2 for the wave's own column block (squared once per block and element, not once per wave)
                const s16x8 vx = __builtin_shufflevector(tr(xs, wv * 32, pix), tr(xs, wv * 32, pix + 4), 0, 1, 2, 3, 4, 5, 6, 7);
                u32x4 ux = __builtin_bit_cast(u32x4, vx);
                uint32_t* w4 = (uint32_t*)&ux;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float l2 = h2f_lo(w4[q]), h2 = h2f_hi(w4[q]);
                    w4[q] = pack_h2(l2 * l2, h2 * h2);
                }
                const h16x8 xf = __builtin_bit_cast(h16x8, ux);
                {   // the column sums of dn: channel block `wave` by this wave (its own pair of reads: no wave-dependent branch in the loop)
                    const s16x8 vc = __builtin_shufflevector(tr(ds, wv * 32, pix), tr(ds, wv * 32, pix + 4), 0, 1, 2, 3, 4, 5, 6, 7);
                    const u32x4 ua = __builtin_bit_cast(u32x4, vc);
                    cs += ((h2f_lo(ua.x) + h2f_hi(ua.x)) + (h2f_lo(ua.y) + h2f_hi(ua.y))) +
                          ((h2f_lo(ua.z) + h2f_hi(ua.z)) + (h2f_lo(ua.w) + h2f_hi(ua.w)));
                }
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    const s16x8 va = __builtin_shufflevector(tr(ds, ib * 32, pix), tr(ds, ib * 32, pix + 4), 0, 1, 2, 3, 4, 5, 6, 7);
                    g3[ib] = mfma_32x32x16_h16(__builtin_bit_cast(h16x8, va), xf, g3[ib], 0, 0, 0);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = i * 32 + 8 * g + 4 * fh;
                const int off = gb_off(r0 + frow, ch >> 3) + (ch & 7) * 2;
                const u32x2 xr = *(const u32x2*)(xs + off);
                const float xv[4] = {h2f_lo(xr.x), h2f_hi(xr.x), h2f_lo(xr.y), h2f_hi(xr.y)};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[i][4 * g + e] + 2.f * xv[e] * s2[i][4 * g + e];
                *(u32x2*)(xs + off) = u32x2{pack_h2(o[0], o[1]), pack_h2(o[2], o[3])};
            }
        // wave-private copy-out of dx and dn (full 256-byte rows)
        const int so = (int)tile * 32768 + lofs;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int c = it * 64 + lane, row = c >> 4, slot = c & 15;
            __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4*)(xs + gb_off(r0 + row, slot)), dxr, so + it * 1024, 0, 0);
            if constexpr (!PAR) __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4*)(ds + gb_off(r0 + row, slot)), dnr, so + it * 1024, 0, 0);
        }
    }
    if constexpr (PAR) {
        float* out = part + (int64_t)blockIdx.x * (128 * 128 + 128);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 128 + wave * 32 + frow] = g3[ib][r];
        cs += __shfl_xor(cs, 32);
        if (fh == 0) out[128 * 128 + wave * 32 + frow] = cs;
    }
}

// (Round 5 built a two-pipelines-per-CU form of this pass, gdn128_bwd2_kernel: correct, 12 - 20 % slower -- the block barrier is the only cheap
// synchronisation and the dependences leave no balanced cut for two groups in lockstep.  Removed from the library in round 6; the kernel, its
// launch and the measurements are kept as profiles/experiments/r05_gdn128_bwd2_two_pipelines.patch.)

static WgTrArgs make_tr_args(const WgArgs& a, float* zero_me, int zero_n) {
    WgTrArgs A;
    A.w = a;
    A.zero_me = zero_me; A.zero_n = zero_n;
    A.dqw = make_fastdiv((uint32_t)a.QW);
    A.dqh = make_fastdiv((uint32_t)a.QH);
    constexpr bool slow = false;          // A/B switch for profiling
    A.fastq = (!slow && a.QW % 16 == 0 && a.chunk % 64 == 0) ? 1 : 0;
    return A;
}

void launch_wgrad_tr(const WgArgs& a, int64_t blocks, hipStream_t st, float* zero_me = nullptr, int zero_n = 0) {
    if (a.rowk) {
        WgRowArgs R;
        R.w = a; R.zero_me = zero_me; R.zero_n = zero_n;
        R.dqw = make_fastdiv((uint32_t)a.QW); R.dqh = make_fastdiv((uint32_t)a.QH);
        static bool rattr = false;
        if (!rattr) {
            (void)hipFuncSetAttribute((const void*)wgrad_row_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ROW_LDS);
            (void)hipFuncSetAttribute((const void*)wgrad_row_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ROW_LDS);
            rattr = true;
        }
        const dim3 g((unsigned)(5 * a.co_tiles * a.ci_tiles * a.nsplit)), b(512);
        if (a.transposed) hipLaunchKernelGGL(wgrad_row_kernel<true>, g, b, ROW_LDS, st, R);
        else hipLaunchKernelGGL(wgrad_row_kernel<false>, g, b, ROW_LDS, st, R);
        return;
    }
    const WgTrArgs A = make_tr_args(a, zero_me, zero_n);
    // the stage ring is <64, 2>.  Measured and dropped (training step, same box, ms; round 5): <64, 2> 9.89 / 9.91 | <32, 5> 11.30 | <32, 4> 11.32 | <64, 3>
    // (96 KB: one block per CU) 12.36 | <64, 2> with the stage's DMA instructions spread behind the k-steps' MFMAs 10.58 -- more bytes in flight, or
    // cheaper issue slots for the DMA instructions, are not what the loop is short of.
    const dim3 g((unsigned)blocks), b(NT);
    hipLaunchKernelGGL((wgrad_tr_kernel<64, 2>), g, b, 2 * 64 * 512, st, A);
}

int pick_splits(int64_t Q, int bk, int tiles, bool batched = false) {
    // aim at ~384 blocks (measured best on MI355X for the step as a whole: every extra slice is another fp32 partial tile
    // to write and reduce), at least 4 K-steps per block
    constexpr int target0 = 384;   // A/B switch
    // The layers with many pixels (first measured from Q >= 100000 on: 128 -> 128 5x5 on 256^2 inputs at B=8, 25 tap blocks per slice): 16 slices = 400 blocks leave 22 % of
    // the 512 block slots (256 CUs x 2) empty for the whole launch, 21 slices = 525 blocks run a second round for 13 of them; 20 slices = 500
    // blocks fill one round.  Training step, same box, alternating runs (ms): 384: 10.65 / 10.66 / 10.48 | 448: 10.70 (earlier box) | 475: 10.57 |
    // 500: 10.53 / 10.53 / 10.33 / 10.33 | 512: 11.01 (earlier box) | 1000: 10.64; 500 for EVERY layer: 10.48 (no gain).  0 = off (A/B).
    constexpr int big_target = 500;
    // from which pixel count on (same box, alternating runs, ms): 100000: 10.62 / 10.61 | 30000 (adds the 128 -> 128 layers on 128^2 inputs): 10.57 / 10.59 | 8000: 10.71
    constexpr int64_t big_q = 30000;      // A/B switch
    // batched route (hesic_conv2d_wgrad_nsplit(d, 1)): the grid is shared with the other queued layers, so a layer need not fill the 512 block
    // slots by itself -- fewer slices = fewer fp32 partial tiles to write and reduce.  Training step, same box, alternating runs (ms), small /
    // large-layer targets: 384 / 500: 9.187 / 9.189 | 256 / 500: 9.140 | 200 / 250: 9.077 / 9.074 | 128 / 250: 9.076 | 100 / 125: 9.308
    constexpr int bt_small = 200;
    constexpr int bt_big = 250;
    const int target = batched ? (Q >= big_q ? bt_big : bt_small) : ((big_target && Q >= big_q) ? big_target : target0);
    int64_t s = (target + tiles - 1) / tiles;
    const int64_t maxs = Q / (4 * bk) > 0 ? Q / (4 * bk) : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

// nsplit_override > 0: the caller names the K-slice count of a one-tap-per-block layer (the batched route, whose shared grids want fewer,
// longer slices than a launch that has to fill the machine by itself); the row kernel keeps its own count
int fill_args(const hesic_conv_desc* d, WgArgs& a, int nsplit_override = 0) {
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
    const int bk = d->dtype == HESIC_H16 ? WC<h16_t>::BK : WC<float>::BK;
    // wgrad_row_kernel: 5x5, stride 2, pad 2, every tap live, 64-pixel stages that stay inside one image row, and enough pixels that a
    // K slice per CU outweighs the 5-tap partial tiles every block leaves (HESIC_WGRAD_ROW=0: off, A/B; HESIC_WGRAD_ROW_MINQ)
    {
        const char* e = getenv("HESIC_WGRAD_ROW");
        const char* mq = getenv("HESIC_WGRAD_ROW_MINQ");
        const int64_t minq = mq ? atoll(mq) : 100000;
        const bool off32 = ((int64_t)a.B * a.H * a.W + 64) * a.x_ps * 2 < (1ll << 31) && ((int64_t)a.B * a.Ho * a.Wo + 64) * a.y_ps * 2 < (1ll << 31);
        a.rowk = (!e || atoi(e) != 0) && d->dtype == HESIC_H16 && d->KH == 5 && d->KW == 5 && d->stride == 2 && d->pad == 2 && n == 25 && !d->in_abs &&
                 a.QW % 64 == 0 && a.Q >= minq && a.Q < (1ll << 31) && off32;
    }
    if (a.rowk) {
        const int target = 256;                 // one block per CU (150 KB of LDS)
        const int64_t stages = a.Q / 64;
        int64_t s = target / (5 * a.co_tiles * a.ci_tiles);
        if (s < 1) s = 1;
        if (s > stages) s = stages;
        const int64_t per = (stages + s - 1) / s;
        a.chunk = per * 64;
        a.nsplit = (int)((stages + per - 1) / per);
        return 0;
    }
    a.nsplit = nsplit_override > 0 ? nsplit_override : pick_splits(a.Q, bk, n * a.co_tiles * a.ci_tiles);
    a.chunk = ((a.Q + a.nsplit - 1) / a.nsplit + bk - 1) / bk * bk;
    a.nsplit = (int)((a.Q + a.chunk - 1) / a.chunk);
    return 0;
}

}  // namespace

// ---- bias gradient through wgrad_tr_kernel (see WgArgs::bias_part): room for [nsplit][stride^2][Cout] floats behind the K-slice partials
static int64_t bias_part_bytes(const hesic_conv_desc* d, const WgArgs& a) { return (int64_t)a.nsplit * 4 * d->Cout * 4; }
// the launch conditions of wgrad_tr_kernel, shared by the callers that have to agree on who produced the bias sums
static bool wgrad_tr_path(const hesic_conv_desc* d, const WgArgs& a) {
    bool prefix = true;                               // live taps must be tap_id[0] + 0,1,2,... for the fast kernel
    for (int i = 0; i < a.ntaps; ++i) prefix = prefix && a.tap_id[i] == a.tap_id[0] + i;
    const bool off32 = ((int64_t)a.B * a.H * a.W + 64) * a.x_ps * 2 < (1ll << 31) && ((int64_t)a.B * a.Ho * a.Wo + 64) * a.y_ps * 2 < (1ll << 31);
    return d->dtype == HESIC_H16 && prefix && a.Q < (1ll << 31) && off32;
}
extern "C" int hesic_conv2d_wgrad_nsplit(const hesic_conv_desc* d, int batched) {
    if (!d || d->KH * d->KW > 25) return 0;
    WgArgs a;
    fill_args(d, a);
    constexpr int ring = WGRAD_RING_DEFAULT;
    if (a.rowk || !batched || ring != 0 || !wgrad_tr_path(d, a)) return a.nsplit;      // only the shared-grid kernel's layers take another count
    const int bk = d->dtype == HESIC_H16 ? WC<h16_t>::BK : WC<float>::BK;
    fill_args(d, a, pick_splits(a.Q, bk, a.ntaps * a.co_tiles * a.ci_tiles, true));
    return a.nsplit;
}

// point a.bias_part behind the weight partials in ws and list the taps whose blocks sum dY's columns; false: no such tap set
static bool setup_bias_part(const hesic_conv_desc* d, WgArgs& a, void* ws) {
    constexpr bool off = false;      // A/B switch: the separate column-sum blocks of rounds 1-3
    a.bias_part = nullptr; a.nb_taps = 0;
    if (off || !wgrad_tr_path(d, a) || a.ntaps < 1) return false;
    if (!d->transposed) { a.nb_taps = 1; a.b_tap[0] = 0; }
    else {
        // the tap set {pad + r : r < stride} covers dY only up to H * stride: other geometries (stride > 2, or an output larger than
        // H * stride, e.g. k = 4, p = 0, s = 2) take the column-sum blocks -- and b_tap holds stride^2 <= 4 entries
        if (d->tap_mask_lo || d->stride > 2 || d->stride < 1 || d->pad + d->stride > d->KH || d->pad + d->stride > d->KW ||
            d->Ho > d->H * d->stride || d->Wo > d->W * d->stride) return false;
        for (int ry = 0; ry < d->stride; ++ry)
            for (int rx = 0; rx < d->stride; ++rx) a.b_tap[a.nb_taps++] = (int8_t)((d->pad + ry) * d->KW + d->pad + rx);
        if (a.nb_taps > 4) { a.nb_taps = 0; return false; }
    }
    a.bias_part = (float*)ws + (int64_t)a.nsplit * a.ntaps * d->Cout * d->Cin;
    return true;
}

extern "C" int64_t hesic_conv2d_wgrad_ws_bytes(const hesic_conv_desc* d) {
    if (!d || d->KH * d->KW > 25) return 0;
    WgArgs a;
    fill_args(d, a);
    return (int64_t)a.nsplit * a.ntaps * d->Cout * d->Cin * 4 + bias_part_bytes(d, a);
}

extern "C" int64_t hesic_conv2d_wgrad_ws_bytes_n(const hesic_conv_desc* d, int nsplit) {
    if (!d || d->KH * d->KW > 25) return 0;
    WgArgs a;
    fill_args(d, a, nsplit);
    return (int64_t)a.nsplit * a.ntaps * d->Cout * d->Cin * 4 + bias_part_bytes(d, a);
}

extern "C" int hesic_conv2d_wgrad(const hesic_conv_desc* d, const void* x, const void* dy, float* dw_packed, float* dbias,
                                  void* ws, int64_t ws_bytes, void* stream) {
    HESIC_CHECK_ARG(d && x && dy && dw_packed, "conv2d_wgrad: null pointer");
    const int ce = d->dtype == HESIC_H16 ? 8 : 4;
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
    constexpr bool wg_legacy = false;
    bool prefix = true;                               // live taps must be tap_id[0] + 0,1,2,... for the fast kernel
    for (int i = 0; i < a.ntaps; ++i) prefix = prefix && a.tap_id[i] == a.tap_id[0] + i;
    const bool off32 = ((int64_t)a.B * a.H * a.W + 64) * a.x_ps * 2 < (1ll << 31) && ((int64_t)a.B * a.Ho * a.Wo + 64) * a.y_ps * 2 < (1ll << 31);
    bool db_zeroed = false;
    if (d->dtype == HESIC_H16 && !wg_legacy && prefix && a.Q < (1ll << 31) && off32) {
        launch_wgrad_tr(a, blocks, st, dbias, dbias ? d->Cout : 0);
        db_zeroed = dbias != nullptr;
    }
    else if (d->dtype == HESIC_H16) hipLaunchKernelGGL(wgrad_kernel<h16_t>, dim3((unsigned)blocks), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(wgrad_kernel<float>, dim3((unsigned)blocks), dim3(NT), 0, st, a);
    const int64_t per_tap = (int64_t)d->Cout * d->Cin;
    if (a.ntaps < d->KH * d->KW) zero_async(dw_packed, (int64_t)d->KH * d->KW * per_tap, st);
    constexpr bool split_launch = false;     // A/B switch for profiling
    if (dbias && !split_launch) {
        if (!db_zeroed) zero_async(dbias, d->Cout, st);
        const int64_t P = (int64_t)d->B * d->Ho * d->Wo;
        const int64_t rpb = P / 256 > 0 ? (P + 255) / 256 : 1;      // <= 256 column-sum blocks: each ends in C atomics
        const int n_col = (int)((P + rpb - 1) / rpb), n_red = grid_for(a.ntaps * per_tap / 4, 256);
        if (d->dtype == HESIC_H16)
            hipLaunchKernelGGL(wgrad_reduce_colsum_kernel<h16_t>, dim3((unsigned)(n_red + n_col)), dim3(256), 0, st, (const float*)ws, dw_packed,
                               a.nsplit, a.ntaps, per_tap, a, n_red, (const h16_t*)dy, dbias, P, rpb);
        else
            hipLaunchKernelGGL(wgrad_reduce_colsum_kernel<float>, dim3((unsigned)(n_red + n_col)), dim3(256), 0, st, (const float*)ws, dw_packed,
                               a.nsplit, a.ntaps, per_tap, a, n_red, (const float*)dy, dbias, P, rpb);
        HESIC_LAUNCH_RETURN("conv2d_wgrad");
    }
    if (a.nsplit >= 32 && a.ntaps * per_tap / 64 < 4096)
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((a.ntaps * per_tap / 4 + 15) / 16)), dim3(256), 0, st, (const float*)ws, dw_packed,
                           a.nsplit, a.ntaps, per_tap, a);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(a.ntaps * per_tap / 4, 256)), dim3(256), 0, st, (const float*)ws, dw_packed,
                           a.nsplit, a.ntaps, per_tap, a);
    if (dbias) {
        zero_async(dbias, d->Cout, st);
        const int64_t P = (int64_t)d->B * d->Ho * d->Wo;
        const int64_t rpb = P / 256 > 0 ? (P + 255) / 256 : 1;      // <= 256 blocks: every block ends in C atomics
        const unsigned g = (unsigned)((P + rpb - 1) / rpb);
        if (d->dtype == HESIC_H16)
            hipLaunchKernelGGL(colsum_kernel<h16_t>, dim3(g), dim3(256), 0, st, (const h16_t*)dy, dbias, P, d->Cout, d->y_pix_stride, d->y_c_off, rpb);
        else
            hipLaunchKernelGGL(colsum_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dy, dbias, P, d->Cout, d->y_pix_stride, d->y_c_off, rpb);
    }
    HESIC_LAUNCH_RETURN("conv2d_wgrad");
}

// finishing pass of one layer: the block layout of wgrad_finish_kernel; returns the number of bias column-sum blocks behind the n_red reduce blocks
static int make_finish(const hesic_conv_desc* d, const WgArgs& a, const void* ws, float* dw, int accumulate, bool with_bias, FinishArgs& f,
                       int64_t& P, int64_t& rpb, int accumulate_bias = 1) {
    const int T_all = d->KH * d->KW;
    memset(&f, 0, sizeof(f));
    f.ws = (const float*)ws; f.dw = dw; f.nsplit = a.nsplit; f.ntaps = a.ntaps; f.T_all = T_all; f.Cout = d->Cout; f.Cin = d->Cin;
    f.transposed = d->transposed; f.accumulate = (accumulate || a.ntaps < T_all) ? 1 : 0;
    f.ws_layout = a.ws_layout;
    memcpy(f.tap_id, a.tap_id, sizeof(f.tap_id));
    // A/B switch, OFF: HESIC_WGRAD_FINISH_WIDE=1 = blocks of 8 couts x 128 cins x one tap that read whole 512-byte rows and write their four
    // sums straight to the PyTorch layout.  Measured SLOWER on the training step (same box, alternating: 9.534 / 9.539 ms with it, 9.350 /
    // 9.352 without): the tap-strided 4-byte writes cost more than the wider reads save -- the 8 x 32 x taps tiles turn a tile round in LDS
    // and write runs along the destination's fastest index.
    constexpr bool wide_on = false;
    if (wide_on && (d->Cin & 3) == 0) {
        f.wide = 1;
        f.tiles_ci = (d->Cin + 127) / 128; f.tiles_co = (d->Cout + 7) / 8;
        f.taps_per_group = 1; f.tap_groups = a.ntaps;
        f.n_red = f.tiles_ci * f.tiles_co * a.ntaps;
        P = (int64_t)d->B * d->Ho * d->Wo;
        rpb = P / 256 > 0 ? (P + 255) / 256 : 1;
        if (with_bias && a.bias_part) {
            f.bias_part = a.bias_part; f.nb_parts = a.nsplit * a.nb_taps; f.accumulate_bias = accumulate_bias;
            return 1;
        }
        return with_bias ? (int)((P + rpb - 1) / rpb) : 0;
    }
    f.tiles_ci = (d->Cin + 31) / 32; f.tiles_co = (d->Cout + 7) / 8;
    const int tiles = f.tiles_ci * f.tiles_co;
    int groups = (512 + tiles - 1) / tiles;                      // enough blocks to cover the chip twice
    if (groups > a.ntaps) groups = a.ntaps;
    if (groups < 1) groups = 1;
    f.taps_per_group = (a.ntaps + groups - 1) / groups;
    f.tap_groups = (a.ntaps + f.taps_per_group - 1) / f.taps_per_group;
    f.n_red = tiles * f.tap_groups;
    P = (int64_t)d->B * d->Ho * d->Wo;
    rpb = P / 256 > 0 ? (P + 255) / 256 : 1;
    if (with_bias && a.bias_part) {                   // wgrad_tr_kernel left the column sums: one block adds them up
        f.bias_part = a.bias_part; f.nb_parts = a.nsplit * a.nb_taps; f.accumulate_bias = accumulate_bias;
        return 1;
    }
    return with_bias ? (int)((P + rpb - 1) / rpb) : 0;
}

static thread_local int g_wgrad_partial_only = 0;      // set by hesic_conv2d_wgrad_partial: stop after the split-K MFMA launch
// workspace layout of the direct / partial / finish_batched route (WgArgs::ws_layout).  A/B switch, OFF: HESIC_WGRAD_WS_LAYOUT=1 puts the
// slices of a (tap, cout) row next to each other.  Measured neutral on the training step (same box, alternating runs: 9.579 / 9.583 ms
// with it, 9.586 / 9.596 without): the finishing pass still reads 128-byte pieces (its 8 cout x 32 cin tiles) -- contiguity across
// slices alone does not raise its 2.9 TB/s; a 512-byte-wide tile would be the next step.
// (round 6, same box, two alternating runs each, ms per training step: this layout + the narrow finishing pass 8.887 / 8.897; slices of a (tap, cout) row
// adjacent 8.895 / 8.892; the 512-byte-wide finishing pass 9.037 / 9.049; both 9.027 / 9.042 -- the finishing pass's layout is not a lever)
static int direct_ws_layout() { return 0; }

extern "C" int hesic_conv2d_wgrad_direct(const hesic_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                         int accumulate, void* ws, int64_t ws_bytes, void* stream) {
    HESIC_CHECK_ARG(d && x && dy && (dw || g_wgrad_partial_only), "conv2d_wgrad_direct: null pointer");
    const int ce = d->dtype == HESIC_H16 ? 8 : 4;
    HESIC_CHECK_ARG(d->Cin % ce == 0 && d->Cout % ce == 0 && d->x_pix_stride % ce == 0 && d->y_pix_stride % ce == 0 &&
                        d->x_c_off % ce == 0 && d->y_c_off % ce == 0,
                    "conv2d_wgrad_direct: channels must be multiples of %d", ce);
    HESIC_CHECK_ARG(d->KH * d->KW <= 25, "conv2d_wgrad_direct: at most 25 taps");
    WgArgs a;
    fill_args(d, a);
    a.ws_layout = direct_ws_layout();
    const int64_t need = (int64_t)a.nsplit * a.ntaps * d->Cout * d->Cin * 4 + bias_part_bytes(d, a);
    HESIC_CHECK_ARG(ws && ws_bytes >= need, "conv2d_wgrad_direct: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)need);
    hipStream_t st = (hipStream_t)stream;
    a.x = x; a.dy = dy; a.out = (float*)ws;
    const int64_t blocks = (int64_t)a.ntaps * a.co_tiles * a.ci_tiles * a.nsplit;
    constexpr bool wlog = false;     // diagnostic: one line per weight-gradient launch (geometry, K slices)
    if (wlog)
        fprintf(stderr, "[hesic] wgrad %s B=%d %dx%d Cin=%d -> %dx%d Cout=%d k=%d s=%d taps=%d Q=%lld nsplit=%d chunk=%lld blocks=%lld row=%d\n",
                d->transposed ? "deconv" : "conv", d->B, d->H, d->W, d->Cin, d->Ho, d->Wo, d->Cout, d->KH, d->stride, a.ntaps, (long long)a.Q, a.nsplit,
                (long long)a.chunk, (long long)(a.rowk ? 5 * a.co_tiles * a.ci_tiles * a.nsplit : blocks), a.rowk);
    const bool bias_in_tr = setup_bias_part(d, a, ws);               // always (the partial-only call has no dbias, its finishing call does)
    float* zero_me = (dbias && !accumulate && !bias_in_tr) ? dbias : nullptr;       // accumulate: the caller's buffer already holds a value
    bool db_zeroed = false;
    if (wgrad_tr_path(d, a)) {
        launch_wgrad_tr(a, blocks, st, zero_me, zero_me ? d->Cout : 0);
        db_zeroed = true;
    } else if (d->dtype == HESIC_H16) hipLaunchKernelGGL(wgrad_kernel<h16_t>, dim3((unsigned)blocks), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(wgrad_kernel<float>, dim3((unsigned)blocks), dim3(NT), 0, st, a);
    if (g_wgrad_partial_only) HESIC_LAUNCH_RETURN("conv2d_wgrad_partial");
    if (zero_me && !db_zeroed) zero_async(dbias, d->Cout, st);
    const int T_all = d->KH * d->KW;
    if (a.ntaps < T_all && !accumulate) zero_async(dw, (int64_t)T_all * d->Cout * d->Cin, st);      // dead taps of a masked conv
    FinishArgs f;
    int64_t P, rpb;
    const int n_col = make_finish(d, a, ws, dw, accumulate, dbias != nullptr, f, P, rpb, accumulate);
    if (d->dtype == HESIC_H16)
        hipLaunchKernelGGL(wgrad_finish_kernel<h16_t>, dim3((unsigned)(f.n_red + n_col)), dim3(256), 0, st, f, (const h16_t*)dy, dbias, P,
                           d->y_pix_stride, d->y_c_off, rpb);
    else
        hipLaunchKernelGGL(wgrad_finish_kernel<float>, dim3((unsigned)(f.n_red + n_col)), dim3(256), 0, st, f, (const float*)dy, dbias, P,
                           d->y_pix_stride, d->y_c_off, rpb);
    HESIC_LAUNCH_RETURN("conv2d_wgrad_direct");
}

static bool nw_fast_case(const hesic_sconv_desc* d, bool& conv1) {
    const bool k5 = d->KH == 5 && d->KW == 5 && d->pad == 2 && d->stride == 2;
    conv1 = k5 && !d->transposed && d->Cin == 3 && d->Cout == 128 && d->ys_c == 1 && d->y_dtype == HESIC_H16 && d->H == 2 * d->Ho &&
            d->W == 2 * d->Wo && d->ys_x == 128 && d->ys_y == (int64_t)d->Wo * 128 && d->ys_b == (int64_t)d->Ho * d->Wo * 128;
    const bool dec4 = k5 && d->transposed && d->Cin == 128 && d->Cout == 3 && d->xs_c == 1 && d->x_dtype == HESIC_H16 && d->Ho == 2 * d->H &&
                      d->Wo == 2 * d->W && d->xs_x == 128 && d->xs_y == (int64_t)d->W * 128 && d->xs_b == (int64_t)d->H * d->W * 128;
    return conv1 || dec4;
}

static void nw_gemm_desc(int64_t Q, hesic_conv_desc& g) {
    memset(&g, 0, sizeof(g));
    g.B = 1; g.H = 1; g.W = (int32_t)Q; g.Cin = 96; g.Ho = 1; g.Wo = (int32_t)Q; g.Cout = 128; g.KH = g.KW = 1; g.stride = 1;
    g.dtype = HESIC_H16; g.x_pix_stride = 96; g.y_pix_stride = 128;
}

static bool nn_case(const hesic_sconv_desc* d) {
    return d->KH == 5 && d->KW == 5 && d->pad == 2 && d->stride == 1 && d->Cin == 6 && d->Cout == 3 && d->Ho == d->H && d->Wo == d->W;
}
constexpr int NN_BLOCKS = 512;

extern "C" int64_t hesic_sconv2d_wgrad_ws_bytes(const hesic_sconv_desc* d) {
    bool conv1;
    if (d && nn_case(d)) return (int64_t)NN_BLOCKS * 450 * 4;          // per-block partial rows of the 6 <-> 3 stages
    if (!d || !nw_fast_case(d, conv1)) return 0;
    const int64_t Q = conv1 ? (int64_t)d->B * d->Ho * d->Wo : (int64_t)d->B * d->H * d->W;
    if (Q >= (1ll << 22)) return 0;
    hesic_conv_desc g;
    nw_gemm_desc(Q, g);
    WgArgs a;
    fill_args(&g, a);
    return (Q * 96 * 2 + 255) / 256 * 256 + (int64_t)a.nsplit * 128 * 96 * 4 + (int64_t)a.nsplit * 128 * 4;      // im2col matrix, weight partials, bias partials
}

extern "C" int hesic_sconv2d_wgrad(const hesic_sconv_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                                   int64_t ws_bytes, void* stream) {
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
    constexpr bool legacy = false;
    const bool k5 = d->KH == 5 && d->KW == 5 && d->pad == 2;
    bool conv1 = false;
    const int64_t need = hesic_sconv2d_wgrad_ws_bytes(d);
    const bool mfma_route = !legacy && need > 0 && ws && ws_bytes >= need && nw_fast_case(d, conv1);
    if (!mfma_route) zero_async(dw, nw, st);          // the routes below accumulate with atomics; the MFMA route's finishing pass writes every element
    if (mfma_route) {
        // MFMA route: im2col of the 3-channel side, then the 1x1 weight-gradient kernel with WIDE as "dY"
        const int64_t Q = conv1 ? (int64_t)d->B * d->Ho * d->Wo : (int64_t)d->B * d->H * d->W;
        const int QH = conv1 ? d->Ho : d->H, QW = conv1 ? d->Wo : d->W, NH = conv1 ? d->H : d->Ho, NW = conv1 ? d->W : d->Wo;
        h16_t* P = (h16_t*)ws;
        float* part = (float*)((unsigned char*)ws + (Q * 96 * 2 + 255) / 256 * 256);
        bool fused_done = false;
        {
            // round 5: one fused launch (wgrad_nw_fused_kernel) when the narrow image is fp32 and a 64-pixel stage stays inside one row of the
            // wide grid; HESIC_NW_FUSED=0 is the A/B switch back to im2col + one-tap GEMM
            const char* e = getenv("HESIC_NW_FUSED");
            const int ndt0 = conv1 ? d->x_dtype : d->y_dtype;
            const int64_t nsb0 = conv1 ? d->xs_b : d->ys_b, nsc0 = conv1 ? d->xs_c : d->ys_c, nsy0 = conv1 ? d->xs_y : d->ys_y, nsx0 = conv1 ? d->xs_x : d->ys_x;
            const int64_t span = ((int64_t)(d->B - 1) * nsb0 + 2 * nsc0 + (int64_t)(NH + 2) * nsy0 + (int64_t)(NW + 140) * nsx0) * 4;
            if ((!e || atoi(e) != 0) && ndt0 == HESIC_F32 && QW % 64 == 0 && Q < (1ll << 23) && nsb0 >= 0 && nsc0 >= 0 && nsy0 >= 0 && nsx0 >= 0 &&
                span < (1ll << 31)) {
                hesic_conv_desc g0;
                nw_gemm_desc(Q, g0);
                WgArgs a0;
                fill_args(&g0, a0);                                   // the slice count the workspace was sized for
                NwArgs n;
                n.wide = conv1 ? dy : x; n.narrow = (const float*)(conv1 ? x : dy); n.part = part;
                n.ns_b = nsb0; n.ns_c = nsc0; n.ns_y = nsy0; n.ns_x = nsx0;
                n.QH = QH; n.QW = QW; n.NH = NH; n.NW = NW; n.Q = Q;
                const int64_t stages = Q / 64;
                int64_t ns = a0.nsplit < 256 ? a0.nsplit : 256;
                if (ns > stages) ns = stages;
                const int64_t per = (stages + ns - 1) / ns;
                n.chunk = per * 64;
                ns = (stages + per - 1) / per;
                float* bpart = part + ns * 128 * 96;
                const bool with_bias = conv1 && dbias;
                n.bias_part = with_bias ? bpart : nullptr;
                static bool nattr = false;
                if (!nattr) {
                    (void)hipFuncSetAttribute((const void*)wgrad_nw_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, NWF_LDS);
                    (void)hipFuncSetAttribute((const void*)wgrad_nw_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, NWF_LDS);
                    nattr = true;
                }
                if (with_bias) hipLaunchKernelGGL(wgrad_nw_fused_kernel<true>, dim3((unsigned)ns), dim3(256), NWF_LDS, st, n);
                else hipLaunchKernelGGL(wgrad_nw_fused_kernel<false>, dim3((unsigned)ns), dim3(256), NWF_LDS, st, n);
                hipLaunchKernelGGL(nw_finish_kernel, dim3(with_bias ? 388 : 384), dim3(256), 0, st, (const float*)part, dw, (int)ns,
                                   (const float*)(with_bias ? bpart : nullptr), dbias);
                if (with_bias) dbias = nullptr;
                fused_done = true;
            }
        }
        if (!fused_done) {
        constexpr bool chunk_form = false;      // A/B switch: the thread-per-chunk kernel of rounds 1-3
        const void* nimg = conv1 ? x : dy;
        const int ndt = conv1 ? d->x_dtype : d->y_dtype;
        const int64_t nsb = conv1 ? d->xs_b : d->ys_b, nsc = conv1 ? d->xs_c : d->ys_c, nsy = conv1 ? d->xs_y : d->ys_y, nsx = conv1 ? d->xs_x : d->ys_x;
        if (!chunk_form && Q < (1ll << 31)) {
            const FastDiv dqw = make_fastdiv((uint32_t)QW), dqh = make_fastdiv((uint32_t)QH);
            const dim3 grid((unsigned)((Q + 255) / 256));
            if (ndt == HESIC_H16)
                hipLaunchKernelGGL(im2col_narrow_rows_kernel<h16_t>, grid, dim3(256), 0, st, (const h16_t*)nimg, nsb, nsc, nsy, nsx, P, Q, QH, QW, NH, NW, dqw, dqh);
            else
                hipLaunchKernelGGL(im2col_narrow_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)nimg, nsb, nsc, nsy, nsx, P, Q, QH, QW, NH, NW, dqw, dqh);
        } else
            hipLaunchKernelGGL(im2col_narrow_kernel, dim3(grid_for(Q * 12, 256)), dim3(256), 0, st, nimg, ndt, nsb, nsc, nsy, nsx, P, d->B, QH, QW, NH, NW, 3);
        hesic_conv_desc g;
        nw_gemm_desc(Q, g);
        WgArgs a2;
        fill_args(&g, a2);
        a2.x = P; a2.dy = conv1 ? dy : x; a2.out = part;
        // conv1: the GEMM's "dY" operand IS dy (128 channels, 134 MB at B=8 512^2) -- its column sums (the bias gradient) come out of the same
        // launch (WgArgs::bias_part) instead of a second pass over it
        constexpr bool bias_colsum = false;      // A/B switch
        float* bpart = part + (int64_t)a2.nsplit * 128 * 96;
        if (conv1 && dbias && !bias_colsum) { a2.bias_part = bpart; a2.nb_taps = 1; a2.b_tap[0] = 0; }
        launch_wgrad_tr(a2, (int64_t)a2.ntaps * a2.co_tiles * a2.ci_tiles * a2.nsplit, st);
        hipLaunchKernelGGL(nw_reduce_kernel, dim3((128 * 75 + 3) / 4), dim3(256), 0, st, (const float*)part, dw, a2.nsplit, 75);
        if (a2.bias_part) {
            hipLaunchKernelGGL(nw_bias_reduce_kernel, dim3(1), dim3(1024), 0, st, (const float*)bpart, dbias, a2.nsplit);
            dbias = nullptr;                          // done: skip the column-sum pass below
        }
        }
    } else if (!legacy && k5 && d->stride == 2 && !d->transposed && d->Cin == 3 && d->Cout == 128 && d->ys_c == 1 && d->y_dtype == HESIC_H16 &&
        d->H == 2 * d->Ho && d->W == 2 * d->Wo) {
        // conv1: WIDE = dy (output grid), NARROW = x
        const int64_t tiles = (int64_t)((d->Wo + 15) / 16) * ((d->Ho + 7) / 8) * d->B;
        hipLaunchKernelGGL((sconv_wgrad_nw_kernel<3, h16_t>), dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(256), 0, st, (const h16_t*)dy,
                           d->ys_b, d->ys_y, d->ys_x, x, d->x_dtype, d->xs_b, d->xs_c, d->xs_y, d->xs_x, dw, d->B, d->Ho, d->Wo, d->H, d->W);
    } else if (!legacy && k5 && d->stride == 2 && d->transposed && d->Cin == 128 && d->Cout == 3 && d->xs_c == 1 && d->x_dtype == HESIC_H16 &&
               d->Ho == 2 * d->H && d->Wo == 2 * d->W) {
        // deconv4: WIDE = x (input grid), NARROW = dy
        const int64_t tiles = (int64_t)((d->W + 15) / 16) * ((d->H + 7) / 8) * d->B;
        hipLaunchKernelGGL((sconv_wgrad_nw_kernel<3, h16_t>), dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(256), 0, st, (const h16_t*)x,
                           d->xs_b, d->xs_y, d->xs_x, dy, d->y_dtype, d->ys_b, d->ys_c, d->ys_y, d->ys_x, dw, d->B, d->H, d->W, d->Ho, d->Wo);
    } else if (!legacy && k5 && d->stride == 1 && d->Cin == 6 && d->Cout == 3 && d->Ho == d->H && d->Wo == d->W) {
        const int64_t tiles = (int64_t)((d->W + 63) / 64) * ((d->H + 15) / 16) * d->B;
        const unsigned g = (unsigned)(tiles < NN_BLOCKS ? tiles : NN_BLOCKS);       // persistent blocks
        float* part = (ws && ws_bytes >= (int64_t)NN_BLOCKS * 450 * 4) ? (float*)ws : nullptr;
        // LDS-DMA staging: fp32 on both sides, non-negative strides, 32-bit byte offsets inside one image (HESIC_NN_DMA=0: register staging, A/B)
        constexpr bool dma_off = false;
        auto span = [&](int64_t sc, int64_t sy, int64_t sx, int C_) { return (C_ * sc + (int64_t)(d->H + 4) * sy + (int64_t)(d->W + 4) * sx) * 4; };
        const bool dma = !dma_off && d->x_dtype == HESIC_F32 && d->y_dtype == HESIC_F32 && d->xs_c >= 0 && d->xs_y >= 0 && d->xs_x >= 0 && d->ys_c >= 0 &&
                         d->ys_y >= 0 && d->ys_x >= 0 && span(d->xs_c, d->xs_y, d->xs_x, 6) < (1ll << 31) && span(d->ys_c, d->ys_y, d->ys_x, 3) < (1ll << 31);
        if (!d->transposed) {   // A = dy (co), S = x (ci); dW[co][ci][k]
            if (dma) hipLaunchKernelGGL((sconv_wgrad_nn_kernel<3, 6, true>), dim3(g), dim3(256), 0, st, dy, d->y_dtype, d->ys_b, d->ys_c, d->ys_y, d->ys_x, x,
                                        d->x_dtype, d->xs_b, d->xs_c, d->xs_y, d->xs_x, dw, part, 1, d->B, d->H, d->W);
            else hipLaunchKernelGGL((sconv_wgrad_nn_kernel<3, 6>), dim3(g), dim3(256), 0, st, dy, d->y_dtype, d->ys_b, d->ys_c, d->ys_y, d->ys_x, x,
                                    d->x_dtype, d->xs_b, d->xs_c, d->xs_y, d->xs_x, dw, part, 1, d->B, d->H, d->W);
        } else {                // A = x (ci), S = dy (co); dW[ci][co][k]
            if (dma) hipLaunchKernelGGL((sconv_wgrad_nn_kernel<6, 3, true>), dim3(g), dim3(256), 0, st, x, d->x_dtype, d->xs_b, d->xs_c, d->xs_y, d->xs_x, dy,
                                        d->y_dtype, d->ys_b, d->ys_c, d->ys_y, d->ys_x, dw, part, 1, d->B, d->H, d->W);
            else hipLaunchKernelGGL((sconv_wgrad_nn_kernel<6, 3>), dim3(g), dim3(256), 0, st, x, d->x_dtype, d->xs_b, d->xs_c, d->xs_y, d->xs_x, dy,
                                    d->y_dtype, d->ys_b, d->ys_c, d->ys_y, d->ys_x, dw, part, 1, d->B, d->H, d->W);
        }
        if (part) hipLaunchKernelGGL(nn_partial_reduce_kernel, dim3((450 + 3) / 4), dim3(256), 0, st, (const float*)part, dw, (int)g, 450);
    } else {
        const int64_t Q = (int64_t)d->B * (d->transposed ? d->H * d->W : d->Ho * d->Wo);
        const int gx = (int)((nw + 255) / 256);
        int gy = (int)(2048 / gx > 0 ? 2048 / gx : 1);
        if (gy > Q) gy = (int)Q;
        a.q_per_block = (Q + gy - 1) / gy;
        gy = (int)((Q + a.q_per_block - 1) / a.q_per_block);
        hipLaunchKernelGGL(sconv_wgrad_generic_kernel, dim3(gx, gy), dim3(256), 0, st, a);
    }
    if (dbias) {
        zero_async(dbias, d->Cout, st);
        const int64_t P = (int64_t)d->B * d->Ho * d->Wo;
        if (d->ys_c == 1 && d->Cout >= 32 && d->ys_x == d->Cout && d->ys_y == (int64_t)d->Wo * d->Cout &&
            d->ys_b == (int64_t)d->Ho * d->Wo * d->Cout) {       // dense NHWC: coalesced column sums
            const int64_t rpb = P / 256 > 0 ? (P + 255) / 256 : 1;      // <= 256 blocks: every block ends in C atomics
            const unsigned g = (unsigned)((P + rpb - 1) / rpb);
            if (d->y_dtype == HESIC_H16) hipLaunchKernelGGL(colsum_kernel<h16_t>, dim3(g), dim3(256), 0, st, (const h16_t*)dy, dbias, P, d->Cout, d->Cout, 0, rpb);
            else hipLaunchKernelGGL(colsum_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dy, dbias, P, d->Cout, d->Cout, 0, rpb);
        } else {
            hipLaunchKernelGGL(sconv_dbias_kernel, dim3(256, d->Cout), dim3(256), 0, st, a);
        }
    }
    HESIC_LAUNCH_RETURN("sconv2d_wgrad");
}

static int gdn_fast_desc(int64_t P, hesic_conv_desc& d) {
    // the parameter gradients of the fast path are a 1x1 "conv" weight gradient over P pixels: dY = dn, X = x
    memset(&d, 0, sizeof(d));
    d.B = 1; d.H = 1; d.W = (int32_t)P; d.Cin = 128; d.Ho = 1; d.Wo = (int32_t)P; d.Cout = 128; d.KH = d.KW = 1; d.stride = 1;
    d.dtype = HESIC_H16; d.x_pix_stride = 128; d.y_pix_stride = 128;
    return 0;
}

extern "C" int64_t hesic_gdn_backward_ws_bytes(int64_t P, int C) {
    int64_t generic = (2 * P * C + (int64_t)C * C + C) * 4;
    if (C == 128 && P < (1ll << 22)) {
        hesic_conv_desc d;
        gdn_fast_desc(P, d);
        WgArgs a;
        fill_args(&d, a);
        const int64_t fast = P * 128 * 2 + 256 + (int64_t)a.nsplit * 128 * 128 * 4 + (128 * 128 + 128) * 4;
        if (fast > generic) generic = fast;
        const int64_t tiles = (P + 127) / 128;
        const int64_t fused = ((tiles < 256 ? tiles : 256) + 1) * (128 * 128 + 128) * 4;       // one partial per block + the reduced gradient
        if (fused > generic) generic = fused;
    }
    return generic;
}

extern "C" int hesic_conv2d_wgrad(const hesic_conv_desc* d, const void* x, const void* dy, float* dw_packed, float* dbias,
                                  void* ws, int64_t ws_bytes, void* stream);

// ---- the parameter-gradient finishing passes of SEVERAL fused GDN backwards in one launch (round 5): a training step ran 15 of them, 6 us
// each, one behind every gdn128_bwd_kernel.  hesic_gdn_backward_partial leaves the block partials in the caller's workspace;
// hesic_gdn_param_finish_batched sums them (the order of gdn_param_finish_kernel: bit-identical) for up to GDN_FIN_NB layers per launch.
constexpr int GDN_FIN_NB = 16;
struct GdnFinJob { const float* part; const float* beta; const float* gamma; float* dgamma; float* dbeta; int nb; float bound; };
struct GdnFinBatch { int n, accumulate; GdnFinJob j[GDN_FIN_NB]; };
namespace {
__global__ __launch_bounds__(256) void gdn_param_finish_batched_kernel(const GdnFinBatch fb) {
    constexpr int NP = 128 * 128 + 128, BPJ = NP / 64;
    const int job = blockIdx.x / BPJ, blk = blockIdx.x - job * BPJ;
    const GdnFinJob& J = fb.j[job];
    __shared__ f32x4 red[16][16];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = (blk * 16 + el) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int k = grp;
    for (; k + 16 < J.nb; k += 32) {
        s0 += *(const f32x4*)(J.part + (int64_t)k * NP + i);
        s1 += *(const f32x4*)(J.part + (int64_t)(k + 16) * NP + i);
    }
    if (k < J.nb) s0 += *(const f32x4*)(J.part + (int64_t)k * NP + i);
    red[grp][el] = s0 + s1;
    __syncthreads();
    if (grp == 0) {
        f32x4 sum = red[0][el];
#pragma unroll
        for (int g = 1; g < 16; ++g) sum += red[g][el];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = i + e;
            if (idx < 128 * 128) {
                const float th = J.gamma[idx], g = sum[e] * 2.f * fmaxf(th, kGammaBound);
                const float v = (th >= kGammaBound || g < 0.f) ? g : 0.f;
                J.dgamma[idx] = fb.accumulate ? J.dgamma[idx] + v : v;
            } else {
                const int c = idx - 128 * 128;
                const float th = J.beta[c], g = sum[e] * 2.f * fmaxf(th, J.bound);
                const float v = (th >= J.bound || g < 0.f) ? g : 0.f;
                J.dbeta[c] = fb.accumulate ? J.dbeta[c] + v : v;
            }
        }
    }
}
}  // namespace

static thread_local int g_gdn_partial_only = 0;    // set by hesic_gdn_backward_partial: stop behind gdn128_bwd_kernel
extern "C" int hesic_gdn_backward_partial_ok(int64_t P, int C, int dtype);
static thread_local int g_gdn_accumulate = 0;      // set by hesic_gdn_backward_acc around its call

extern "C" int hesic_gdn_backward(const void* x, const void* dy, const float* beta, const float* gamma, void* dx, float* dbeta,
                                  float* dgamma, void* ws, int64_t P, int C, int inverse, float beta_min, int dtype, void* stream) {
    const int accumulate = g_gdn_accumulate;
    HESIC_CHECK_ARG(x && dy && beta && gamma && dx && (g_gdn_partial_only || (dbeta && dgamma)) && ws && P > 0 && C > 0, "gdn_backward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float bound = sqrtf(beta_min + kPedestal);
    constexpr bool legacy = false;
    HESIC_CHECK_ARG(!g_gdn_partial_only || hesic_gdn_backward_partial_ok(P, C, dtype), "gdn_backward_partial: only the fused 128-channel 16-bit form leaves block partials");
    if (!legacy && C == 128 && dtype == HESIC_H16 && P < (1ll << 22)) {
        hesic_conv_desc d;
        gdn_fast_desc(P, d);
        WgArgs a;
        fill_args(&d, a);
        unsigned char* base = (unsigned char*)ws;
        h16_t* dn = (h16_t*)base;
        int64_t off = (P * 128 * 2 + 255) / 256 * 256;
        void* wws = base + off;
        const int64_t wws_bytes = (int64_t)a.nsplit * 128 * 128 * 4;
        off += wws_bytes;
        float* dgp = (float*)(base + off);
        float* dbp = dgp + 128 * 128;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)gdn128_bwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 512);
            (void)hipFuncSetAttribute((const void*)gdn128_bwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 512);
            (void)hipFuncSetAttribute((const void*)gdn128_bwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 512);
            (void)hipFuncSetAttribute((const void*)gdn128_bwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 512);
            attr = true;
        }
        const int64_t tiles = (P + 127) / 128;
        const int nb = (int)(tiles < 256 ? tiles : 256);
        constexpr bool split_params = false;      // A/B switch: the three-launch form of round 2
        if (!split_params) {
            // one pass: dx + a (128 x 128 + 128) parameter-gradient partial per block, then the block partials summed in a fixed order
            constexpr int NP = 128 * 128 + 128;
            float* part = (float*)base;
            if (inverse) hipLaunchKernelGGL((gdn128_bwd_kernel<true, true>), dim3((unsigned)nb), dim3(256), 131072 + 512, st, (const h16_t*)x, (const h16_t*)dy, beta, gamma,
                                            (h16_t*)dx, (h16_t*)nullptr, part, P, bound);
            else hipLaunchKernelGGL((gdn128_bwd_kernel<true, false>), dim3((unsigned)nb), dim3(256), 131072 + 512, st, (const h16_t*)x, (const h16_t*)dy, beta, gamma,
                                    (h16_t*)dx, (h16_t*)nullptr, part, P, bound);
            {
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) { hesic_set_error("gdn_backward: %s", hipGetErrorString(e)); return (int)e; }
            }
            if (g_gdn_partial_only) HESIC_LAUNCH_RETURN("gdn_backward_partial");
            hipLaunchKernelGGL(gdn_param_finish_kernel, dim3((unsigned)(NP / 64)), dim3(256), 0, st, (const float*)part, nb, beta, gamma, dgamma, dbeta, bound, accumulate);
            HESIC_LAUNCH_RETURN("gdn_backward");
        }
        if (inverse) hipLaunchKernelGGL((gdn128_bwd_kernel<false, true>), dim3((unsigned)nb), dim3(256), 131072 + 512, st, (const h16_t*)x,
                                        (const h16_t*)dy, beta, gamma, (h16_t*)dx, dn, (float*)nullptr, P, bound);
        else hipLaunchKernelGGL((gdn128_bwd_kernel<false, false>), dim3((unsigned)nb), dim3(256), 131072 + 512, st, (const h16_t*)x,
                                (const h16_t*)dy, beta, gamma, (h16_t*)dx, dn, (float*)nullptr, P, bound);
        {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { hesic_set_error("gdn_backward: %s", hipGetErrorString(e)); return (int)e; }
        }
        // dgamma'[i][j] = sum_p dn[p][i] x[p][j]^2 ; dbeta'[i] = sum_p dn[p][i]
        // (same kernels as hesic_conv2d_wgrad, with the X operand squared on load)
        a.x = x; a.dy = dn; a.out = (float*)wws; a.in_sq = 1;
        const int64_t blocks = (int64_t)a.ntaps * a.co_tiles * a.ci_tiles * a.nsplit;
        constexpr bool wg_legacy = false;
        if (wg_legacy) hipLaunchKernelGGL(wgrad_kernel<h16_t>, dim3((unsigned)blocks), dim3(NT), 0, st, a);
        else launch_wgrad_tr(a, blocks, st, dbp, 128);
        if (wg_legacy) zero_async(dbp, 128, st);
        const int64_t rpb = P / 256 > 0 ? (P + 255) / 256 : 1;      // <= 256 column-sum blocks: every block ends in C atomics
        // K-slice reduce (slice-parallel form: this 1-tap problem is cut into up to 256 slices) + column sums of dn, one launch
        const int n_red = 128 * 128 / 64, n_col = (int)((P + rpb - 1) / rpb);
        hipLaunchKernelGGL(wgrad_reduce_wide_colsum_kernel<h16_t>, dim3((unsigned)(n_red + n_col)), dim3(256), 0, st, (const float*)wws, dgp, a.nsplit, 1,
                           (int64_t)128 * 128, a, n_red, (const h16_t*)dn, dbp, P, 128, 128, 0, rpb);
        hipLaunchKernelGGL(gdn_bwd_chain_kernel, dim3(64), dim3(256), 0, st, beta, gamma, dgp, dbp, dgamma, dbeta, C, bound, accumulate);
        HESIC_LAUNCH_RETURN("gdn_backward");
    }
    float* dn = (float*)ws;
    float* dx0 = dn + P * C;
    float* dgp = dx0 + P * C;
    float* dbp = dgp + (int64_t)C * C;
    zero_async(dgp, (int64_t)C * C + C, st);
    constexpr bool small_split = false;      // A/B switch: the three passes
    if (C == 3 && !small_split) {
        const dim3 g3(grid_for(P, 256 * 2, 1024));      // 128 blocks (the parameter pass's grid: few atomics) left half the CUs idle: 57.9 us
        if (dtype == HESIC_H16)
            hipLaunchKernelGGL((gdn_bwd_small_fused_kernel<3, h16_t>), g3, dim3(256), 0, st, (const h16_t*)x, (const h16_t*)dy, beta, gamma, (h16_t*)dx, dgp, dbp, P, inverse, bound);
        else
            hipLaunchKernelGGL((gdn_bwd_small_fused_kernel<3, float>), g3, dim3(256), 0, st, (const float*)x, (const float*)dy, beta, gamma, (float*)dx, dgp, dbp, P, inverse, bound);
        hipLaunchKernelGGL(gdn_bwd_chain_kernel, dim3(1), dim3(256), 0, st, beta, gamma, dgp, dbp, dgamma, dbeta, C, bound, accumulate);
        HESIC_LAUNCH_RETURN("gdn_backward");
    }
    hipLaunchKernelGGL(gdn_bwd_dn_kernel, dim3(grid_for(P * C, 256)), dim3(256), 0, st, x, dy, beta, gamma, dn, dx0, P, C, inverse, bound, dtype);
    hipLaunchKernelGGL(gdn_bwd_dx_kernel, dim3(grid_for(P * C, 256)), dim3(256), 0, st, x, gamma, dn, dx0, dx, P, C, dtype);
    const int gx = (C * C + 255) / 256;
    int gy = 2048 / gx > 0 ? 2048 / gx : 1;
    if (gy > P) gy = (int)P;
    const int64_t rpb = (P + gy - 1) / gy;
    gy = (int)((P + rpb - 1) / rpb);
    if (C == 3) hipLaunchKernelGGL(gdn_bwd_param_small_kernel<3>, dim3(grid_for(P, 256 * 8, 128)), dim3(256), 0, st, x, dn, dgp, dbp, P, dtype);
    else hipLaunchKernelGGL(gdn_bwd_param_kernel, dim3(gx, gy), dim3(256), 0, st, x, dn, dgp, dbp, P, C, dtype, rpb);
    hipLaunchKernelGGL(gdn_bwd_chain_kernel, dim3(gx), dim3(256), 0, st, beta, gamma, dgp, dbp, dgamma, dbeta, C, bound, accumulate);
    HESIC_LAUNCH_RETURN("gdn_backward");
}

extern "C" int hesic_gdn_backward_acc(const void* x, const void* dy, const float* beta, const float* gamma, void* dx, float* dbeta,
                                      float* dgamma, int accumulate, void* ws, int64_t P, int C, int inverse, float beta_min, int dtype,
                                      void* stream) {
    g_gdn_accumulate = accumulate ? 1 : 0;
    const int rc = hesic_gdn_backward(x, dy, beta, gamma, dx, dbeta, dgamma, ws, P, C, inverse, beta_min, dtype, stream);
    g_gdn_accumulate = 0;
    return rc;
}

extern "C" int hesic_gdn_backward_partial_ok(int64_t P, int C, int dtype) {
    constexpr bool legacy = false, split_params = false;
    return (!legacy && !split_params && C == 128 && dtype == HESIC_H16 && P > 0 && P < (1ll << 22)) ? 1 : 0;
}

extern "C" int hesic_gdn_backward_partial(const void* x, const void* dy, const float* beta, const float* gamma, void* dx, void* ws, int64_t P,
                                          int C, int inverse, float beta_min, int dtype, void* stream) {
    g_gdn_partial_only = 1;
    const int rc = hesic_gdn_backward(x, dy, beta, gamma, dx, nullptr, nullptr, ws, P, C, inverse, beta_min, dtype, stream);
    g_gdn_partial_only = 0;
    return rc;
}

extern "C" int hesic_gdn_param_finish_batched(int n, const void* const* ws, const int64_t* P, const float* const* beta, const float* const* gamma,
                                              float* const* dgamma, float* const* dbeta, const float* beta_min, int accumulate, void* stream) {
    HESIC_CHECK_ARG(n >= 0 && (n == 0 || (ws && P && beta && gamma && dgamma && dbeta && beta_min)), "gdn_param_finish_batched: null pointer");
    constexpr int NP = 128 * 128 + 128;
    for (int j0 = 0; j0 < n; j0 += GDN_FIN_NB) {
        GdnFinBatch fb;
        memset(&fb, 0, sizeof(fb));
        fb.n = n - j0 < GDN_FIN_NB ? n - j0 : GDN_FIN_NB;
        fb.accumulate = accumulate ? 1 : 0;
        for (int j = 0; j < fb.n; ++j) {
            const int q = j0 + j;
            HESIC_CHECK_ARG(ws[q] && beta[q] && gamma[q] && dgamma[q] && dbeta[q] && P[q] > 0, "gdn_param_finish_batched: job %d: bad arguments", q);
            for (int i = j0; i < q; ++i)
                HESIC_CHECK_ARG(dgamma[i] != dgamma[q], "gdn_param_finish_batched: jobs %d and %d add into the same gradient in one launch", i, q);
            const int64_t tiles = (P[q] + 127) / 128;
            fb.j[j].part = (const float*)ws[q]; fb.j[j].beta = beta[q]; fb.j[j].gamma = gamma[q]; fb.j[j].dgamma = dgamma[q]; fb.j[j].dbeta = dbeta[q];
            fb.j[j].nb = (int)(tiles < 256 ? tiles : 256);
            fb.j[j].bound = sqrtf(beta_min[q] + kPedestal);
        }
        hipLaunchKernelGGL(gdn_param_finish_batched_kernel, dim3((unsigned)(fb.n * (NP / 64))), dim3(256), 0, (hipStream_t)stream, fb);
    }
    HESIC_LAUNCH_RETURN("gdn_param_finish_batched");
}

extern "C" int hesic_gdn_backward_planar_acc(const void* x, const void* dy, const float* beta, const float* gamma, void* dx, float* dbeta,
                                             float* dgamma, int accumulate, void* ws, int B, int64_t HW, int C, int inverse, float beta_min,
                                             int dtype, void* stream) {
    HESIC_CHECK_ARG(x && dy && beta && gamma && dx && dbeta && dgamma && ws && B > 0 && HW > 0, "gdn_backward_planar: bad arguments");
    HESIC_CHECK_ARG(C == 3, "gdn_backward_planar: built for the 3-channel image-side GDNs");
    HESIC_CHECK_ARG(dtype == HESIC_H16 || dtype == HESIC_F32, "gdn_backward_planar: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const float bound = sqrtf(beta_min + kPedestal);
    float* dgp = (float*)ws;
    float* dbp = dgp + C * C;
    zero_async(dgp, C * C + C, st);
    int gx = grid_for(HW, 256 * 2, 1024);
    if ((int64_t)gx * B > 2048) gx = (int)(2048 / B > 0 ? 2048 / B : 1);
    const dim3 g3((unsigned)gx, (unsigned)B);
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL((gdn_bwd_small_fused_kernel<3, h16_t, true>), g3, dim3(256), 0, st, (const h16_t*)x, (const h16_t*)dy, beta, gamma, (h16_t*)dx, dgp, dbp, HW, inverse, bound);
    else
        hipLaunchKernelGGL((gdn_bwd_small_fused_kernel<3, float, true>), g3, dim3(256), 0, st, (const float*)x, (const float*)dy, beta, gamma, (float*)dx, dgp, dbp, HW, inverse, bound);
    hipLaunchKernelGGL(gdn_bwd_chain_kernel, dim3(1), dim3(256), 0, st, beta, gamma, dgp, dbp, dgamma, dbeta, C, bound, accumulate ? 1 : 0);
    HESIC_LAUNCH_RETURN("gdn_backward_planar");
}

extern "C" int hesic_conv2d_wgrad_partial(const hesic_conv_desc* d, const void* x, const void* dy, void* ws, int64_t ws_bytes, void* stream) {
    g_wgrad_partial_only = 1;
    const int rc = hesic_conv2d_wgrad_direct(d, x, dy, nullptr, nullptr, 1, ws, ws_bytes, stream);
    g_wgrad_partial_only = 0;
    return rc;
}

// The split-K launches of n layers (what hesic_conv2d_wgrad_partial does for one), the 128-channel-tile MFMA ones among them sharing grids of
// up to WB_MAX jobs (wgrad_tr_batched_kernel); layers on another kernel (wgrad_row_kernel, the VALU fallback) are launched one by one.  The
// partials land in each job's own workspace exactly as hesic_conv2d_wgrad_partial leaves them: hesic_conv2d_wgrad_finish_batched follows.
extern "C" int hesic_conv2d_wgrad_partial_batched(int n, const hesic_conv_desc* descs, const void* const* x, const void* const* dy, void* const* ws,
                                                  const int64_t* ws_bytes, const int32_t* nsplit, void* stream) {
    HESIC_CHECK_ARG(n >= 0 && (n == 0 || (descs && x && dy && ws && ws_bytes)), "conv2d_wgrad_partial_batched: null pointer");
    hipStream_t st = (hipStream_t)stream;
    constexpr int ring = WGRAD_RING_DEFAULT;
    struct Job { WgArgs a; int64_t blocks; };
    std::vector<Job> jobs;
    for (int i = 0; i < n; ++i) {
        const hesic_conv_desc* d = descs + i;
        HESIC_CHECK_ARG(x[i] && dy[i] && ws[i], "conv2d_wgrad_partial_batched: job %d: null pointer", i);
        WgArgs a;
        bool batched = d->KH * d->KW <= 25 && ring == 0;
        if (batched) {
            fill_args(d, a, nsplit ? nsplit[i] : 0);
            a.ws_layout = direct_ws_layout();
            batched = wgrad_tr_path(d, a) && !a.rowk;
        }
        if (!batched) {
            // another kernel takes this layer: it picks its own K-slice count, which a caller-named count has to agree with
            if (nsplit && nsplit[i] > 0) {
                WgArgs a0;
                fill_args(d, a0);
                HESIC_CHECK_ARG(a0.nsplit == nsplit[i], "conv2d_wgrad_partial_batched: job %d: %d K slices named, this layer's kernel takes %d", i, nsplit[i], a0.nsplit);
            }
            if (int rc = hesic_conv2d_wgrad_partial(d, x[i], dy[i], ws[i], ws_bytes[i], stream)) return rc;
            continue;
        }
        const int ce = 8;
        HESIC_CHECK_ARG(d->Cin % ce == 0 && d->Cout % ce == 0 && d->x_pix_stride % ce == 0 && d->y_pix_stride % ce == 0 && d->x_c_off % ce == 0 &&
                            d->y_c_off % ce == 0, "conv2d_wgrad_partial_batched: job %d: channels must be multiples of %d", i, ce);
        const int64_t need = (int64_t)a.nsplit * a.ntaps * d->Cout * d->Cin * 4 + bias_part_bytes(d, a);
        HESIC_CHECK_ARG(ws_bytes[i] >= need, "conv2d_wgrad_partial_batched: job %d: workspace too small (%lld < %lld)", i, (long long)ws_bytes[i], (long long)need);
        a.x = x[i]; a.dy = dy[i]; a.out = (float*)ws[i];
        setup_bias_part(d, a, ws[i]);
        jobs.push_back(Job{a, (int64_t)a.ntaps * a.co_tiles * a.ci_tiles * a.nsplit});
    }
    // longest K slices first: the launch ends on its shortest blocks
    std::stable_sort(jobs.begin(), jobs.end(), [](const Job& p, const Job& q) { return p.a.chunk > q.a.chunk; });
    for (size_t j0 = 0; j0 < jobs.size(); j0 += WB_MAX) {
        WgTrBatch B;
        memset(&B, 0, sizeof(B));
        B.n = (int)(jobs.size() - j0 < (size_t)WB_MAX ? jobs.size() - j0 : WB_MAX);
        for (int j = 0; j < B.n; ++j) {
            const Job& J = jobs[j0 + j];
            B.job[j] = make_tr_args(J.a, nullptr, 0);
            B.blocks[j] = (int)J.blocks;
            B.start[j + 1] = B.start[j] + (int)((J.blocks + 7) / 8 * 8);
        }
        if (B.n == 1) { launch_wgrad_tr(jobs[j0].a, jobs[j0].blocks, st); continue; }
        hipLaunchKernelGGL((wgrad_tr_batched_kernel<64, 2>), dim3((unsigned)B.start[B.n]), dim3(NT), 2 * 64 * 512, st, B);
    }
    HESIC_LAUNCH_RETURN("conv2d_wgrad_partial_batched");
}

extern "C" int hesic_conv2d_wgrad_finish_batched_n(int n, const hesic_conv_desc* descs, const void* const* ws, const void* const* dy, float* const* dw,
                                                   float* const* dbias, int accumulate, const int32_t* nsplit, void* stream);
extern "C" int hesic_conv2d_wgrad_finish_batched(int n, const hesic_conv_desc* descs, const void* const* ws, const void* const* dy, float* const* dw,
                                                 float* const* dbias, int accumulate, void* stream) {
    return hesic_conv2d_wgrad_finish_batched_n(n, descs, ws, dy, dw, dbias, accumulate, nullptr, stream);
}

extern "C" int hesic_conv2d_wgrad_finish_batched_n(int n, const hesic_conv_desc* descs, const void* const* ws, const void* const* dy, float* const* dw,
                                                   float* const* dbias, int accumulate, const int32_t* nsplit, void* stream) {
    HESIC_CHECK_ARG(n >= 0 && (n == 0 || (descs && ws && dy && dw && dbias)), "conv2d_wgrad_finish_batched: null pointer");
    hipStream_t st = (hipStream_t)stream;
    for (int j0 = 0; j0 < n; j0 += FIN_NB) {
        FinishBatch fb;
        memset(&fb, 0, sizeof(fb));
        fb.n = n - j0 < FIN_NB ? n - j0 : FIN_NB;
        for (int j = 0; j < fb.n; ++j) {
            const hesic_conv_desc* d = descs + j0 + j;
            HESIC_CHECK_ARG(ws[j0 + j] && dy[j0 + j] && dw[j0 + j], "conv2d_wgrad_finish_batched: job %d: null pointer", j0 + j);
            HESIC_CHECK_ARG(d->KH * d->KW <= 25 && d->dtype == descs[0].dtype, "conv2d_wgrad_finish_batched: job %d: at most 25 taps, one storage type per call", j0 + j);
            for (int i = 0; i < j0 + j; ++i)
                HESIC_CHECK_ARG(dw[i] != dw[j0 + j] || i < j0, "conv2d_wgrad_finish_batched: jobs %d and %d add into the same gradient in one launch", i, j0 + j);
            WgArgs a;
            fill_args(d, a, nsplit ? nsplit[j0 + j] : 0);
            a.ws_layout = direct_ws_layout();
            FinishExtra& e = fb.e[j];
            const bool bias_in_tr = setup_bias_part(d, a, (void*)ws[j0 + j]);      // the same decision hesic_conv2d_wgrad_partial took
            const int n_col = make_finish(d, a, ws[j0 + j], dw[j0 + j], accumulate, dbias[j0 + j] != nullptr, fb.f[j], e.P, e.rpb, accumulate);
            if (a.ntaps < d->KH * d->KW && !accumulate) zero_async(dw[j0 + j], (int64_t)d->KH * d->KW * d->Cout * d->Cin, st);
            if (dbias[j0 + j] && !accumulate && !bias_in_tr) zero_async(dbias[j0 + j], d->Cout, st);
            e.dy = dy[j0 + j]; e.db = dbias[j0 + j]; e.y_ps = d->y_pix_stride; e.y_co = d->y_c_off;
            fb.start[j + 1] = fb.start[j] + fb.f[j].n_red + n_col;
        }
        if (descs[0].dtype == HESIC_H16) hipLaunchKernelGGL(wgrad_finish_batched_kernel<h16_t>, dim3((unsigned)fb.start[fb.n]), dim3(256), 0, st, fb);
        else hipLaunchKernelGGL(wgrad_finish_batched_kernel<float>, dim3((unsigned)fb.start[fb.n]), dim3(256), 0, st, fb);
    }
    HESIC_LAUNCH_RETURN("conv2d_wgrad_finish_batched");
}
