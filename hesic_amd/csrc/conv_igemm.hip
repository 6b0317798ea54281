// Implicit-GEMM convolution / transposed convolution on the CDNA4 matrix cores.
//
// Replaces nn.Conv2d / nn.ConvTranspose2d of conv()/deconv() (compressai/models/utils.py:104-118) for
// the wide layers of the HESIC stacks (ywz/mywork/newnet1.py:420-692): Cin % 32 == 0, Cout % 8 == 0.
//
// GEMM view:  Y[cout, pixel] = sum_{tap, ci} Wp[tap][cout][ci] * X[pixel shifted by tap][ci]
//   M = Cout tile (BN rows of the weight panel)   -> MFMA "A" operand
//   N = 128 output pixels (an 8x16-ish 2-D patch) -> MFMA "B" operand
//   K = taps x Cin, walked in steps of 32 channels of one tap.
// With the weights on the A side every lane ends up holding 4 consecutive output channels of one
// pixel, so the epilogue can pack and stage a pixel-major tile in LDS and write full NHWC rows.
//
// A transposed stride-2 conv is run as its 4 output phases (sub-pixel decomposition): phase (py,px)
// is an ordinary stride-1 gather over 3x3 / 3x2 / 2x3 / 2x2 taps, so no zero-inserted input exists.
//
// Storage T is bf16 (v_mfma_f32_32x32x16_bf16) or fp32 (v_mfma_f32_32x32x2_f32, exact fp32 FMA chain,
// used for the tight-parity mode); accumulation is fp32 in both.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 128;      // pixels per block
constexpr int BK = 32;       // channels per K-step
constexpr int NTHREADS = 256;
constexpr int MAX_TAPS = 25;

struct IgemmArgs {
    const void* x;
    const void* w;
    const float* bias;
    void* y;
    int B, H, W, Cin, x_ps, x_co;
    int QH, QW, TH, TW, tw_shift, tiles_y, tiles_x;
    int in_step, out_step;
    int Ho, Wo, Cout, y_ps, y_co;
    int act, in_abs;
    int n_tiles, nphase;
    int KH, KW, stride, pad, transposed, ntaps_live;   // taps are derived arithmetically (no table loads in the K loop)
    const void* gdn_gamma;     // fused GDN epilogue: packed gamma' (hesic_gdn_pack_params); the fragment-order half is used here
    const float* gdn_beta;     // beta' fp32 [128]
    float acc_scale;           // hi/lo launches whose packed weights carry a power-of-two factor: the accumulators are multiplied by this (its inverse) once, behind the K loop
    void* y_pre;               // fused GDN, training: also store the conv output v = conv + bias (bf16, y's geometry) for GDN's backward
    FastDiv fd_nt, fd_tx, fd_ty, fd_b, fd_ph;                 // block-id decode without integer divisions
    int tap_parity;            // stride-2 conv: walk the taps parity class by parity class (see the K-loop cursor)
    int ksplit;                // > 1: the K loop (taps x channel chunks) is cut into ksplit slices, one block each, that
    float* ws;                 //      leave fp32 partial tiles in ws[slice][B][Ho][Wo][Cout] for splitk_reduce_kernel
    int x_group_step, tiles_per_group;   // grouped launch: cout tile nt reads input channels [x_co + (nt / tiles_per_group) * x_group_step, + Cin)
    int act2, act_split;       // couts >= act_split (a multiple of the cout tile) take activation act2 instead of act
    float* y32;                // bf16 fast path: also (or, with y == nullptr, only) store act(conv + bias) as fp32 straight from the
    int y32_ps, y32_co;        //      accumulators -- what feeds round() and the likelihoods must not pass through bf16 storage
    // bf16x3 ("hi/lo") operands, hesic_conv2d_forward_hilo (kernel template flag HL): x holds [hi(C) | lo(C)] per pixel (v = hi + lo
    // to 2^-17), the packed weights [w_hi(C) | w_lo(C)] per (tap, cout) (Cin = 2C here).  A stage brings BK/2 channels of all four
    // operand halves into LDS -- LDS row = [hi chunk | lo chunk] -- and the matrix cores form x_hi w_hi + x_lo w_hi + x_hi w_lo in
    // the fp32 accumulators: 3 MFMAs per staged byte pair instead of a 3x longer K loop (x_lo w_lo, 2^-18, is dropped).
    int y_hilo, y_abs;         // plain epilogue: y as [hi(Cout) | lo(Cout)] pairs, optionally of |v|
    const void* gdn_gamma_lo;  // hi/lo GDN epilogue (GDN = 3 | 4): fragment-order lo half of gamma' (hesic_gdn_pack_params_lo)
};

// Tap geometry of one launch phase, all wave-uniform scalars.
//   conv:        taps = the first ntaps_live kernel positions in raster order (a MaskedConv2d mask is such a prefix),
//                input offset d = k - pad, one phase.
//   transposed:  output phase (ry, rx) = (ph / s, ph % s) uses k = k0 + s*j with k0 = (r + pad) % s and reads the
//                input at q + (r + pad - k)/s;  output pixel = q*s + r.
struct Taps {
    int ry, rx, ky0, kx0, kst, nkx, ntaps;
};
__device__ __forceinline__ Taps make_taps(const IgemmArgs& a, int ph) {
    Taps t;
    if (a.transposed) {
        const int sh = a.stride >> 1, sm = a.stride - 1;         // stride is 1 or 2 (checked by the launcher): shifts, no division
        t.ry = ph >> sh; t.rx = ph & sm;
        t.ky0 = (t.ry + a.pad) & sm; t.kx0 = (t.rx + a.pad) & sm; t.kst = a.stride;
        const int nky = (a.KH - t.ky0 + sm) >> sh;
        t.nkx = (a.KW - t.kx0 + sm) >> sh;
        t.ntaps = nky * t.nkx;
    } else {
        t.ry = t.rx = t.ky0 = t.kx0 = 0; t.kst = 1; t.nkx = a.KW; t.ntaps = a.ntaps_live;
    }
    return t;
}
__device__ __forceinline__ void tap_at(const IgemmArgs& a, const Taps& t, int i, int& dy, int& dx, int& wt) {
    const int j = i / t.nkx, c = i - j * t.nkx;
    const int ky = t.ky0 + j * t.kst, kx = t.kx0 + c * t.kst;
    if (a.transposed) { dy = (t.ry + a.pad - ky) / a.stride; dx = (t.rx + a.pad - kx) / a.stride; }
    else { dy = ky - a.pad; dx = kx - a.pad; }
    wt = ky * a.KW + kx;
}

template <typename T> struct Cfg;
template <> struct Cfg<h16_t> {
    static constexpr int CE = 8;    // elements per 16-byte chunk
    static constexpr int CPR = 4;   // chunks per LDS row (BK*2/16)
    static constexpr int RPB = 4;   // LDS rows per 256-byte bank row
};
template <> struct Cfg<float> {
    static constexpr int CE = 4;
    static constexpr int CPR = 8;
    static constexpr int RPB = 2;
};

template <typename T>
__device__ __forceinline__ int lds_off(int row, int slot) {
    // byte offset of 16-byte slot `slot` of row `row`, XOR-swizzled so that the 16-lane groups of a
    // ds_read_b128 (rows r..r+3, r+12.., r+20..) land on distinct bank quads
    const int sw = slot ^ ((row / Cfg<T>::RPB) & (Cfg<T>::CPR - 1));
    return (row * Cfg<T>::CPR + sw) * 16;
}

__device__ __forceinline__ u32x4 abs_chunk(u32x4 v, h16_t) {
    return u32x4{v.x & 0x7fff7fffu, v.y & 0x7fff7fffu, v.z & 0x7fff7fffu, v.w & 0x7fff7fffu};
}
__device__ __forceinline__ u32x4 abs_chunk(u32x4 v, float) {
    return u32x4{v.x & 0x7fffffffu, v.y & 0x7fffffffu, v.z & 0x7fffffffu, v.w & 0x7fffffffu};
}

template <typename T, int BN>
__global__ __launch_bounds__(NTHREADS) void igemm_conv_kernel(const IgemmArgs a) {
    using C = Cfg<T>;
    constexpr int LA = BM * C::CPR / NTHREADS;              // x-tile 16B loads per thread per step
    constexpr int LB = (BN * C::CPR) / NTHREADS;            // w-tile loads per thread per step (>=1)
    static_assert(LB >= 1, "BN too small");
    constexpr int XT = BM * BK * (int)sizeof(T);            // bytes of one x stage
    constexpr int WT = BN * BK * (int)sizeof(T);
    constexpr int OROW = BN * (int)sizeof(T) + 16;          // padded epilogue row
    constexpr int STAGE = 2 * (XT + WT);
    constexpr int EPI = BM * OROW;
    constexpr int LDS_BYTES = STAGE > EPI ? STAGE : EPI;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- XCD-aware block remap: consecutive logical tiles share an L2
    int bid;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nt = bid % a.n_tiles;
    int rest = bid / a.n_tiles;
    const int tx = rest % a.tiles_x;
    rest /= a.tiles_x;
    const int ty = rest % a.tiles_y;
    rest /= a.tiles_y;
    const int b = rest % a.B;
    const int ph = rest / a.B;
    const int n0 = nt * BN;
    const Taps taps = make_taps(a, ph);
    const int kchunks = a.Cin / BK;
    const int nsteps = taps.ntaps * kchunks;

    const T* __restrict__ xg = (const T*)a.x;
    const T* __restrict__ wg = (const T*)a.w;

    // ---- per-thread staging coordinates (fixed over the K loop)
    const int slot = tid % C::CPR;
    const T* xrow[LA];
    int iy0[LA], ix0[LA];
    bool rowok[LA];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int row = tid / C::CPR + j * (NTHREADS / C::CPR);
        const int qy = ty * a.TH + (row >> a.tw_shift);
        const int qx = tx * a.TW + (row & (a.TW - 1));
        rowok[j] = (qy < a.QH) && (qx < a.QW);
        iy0[j] = qy * a.in_step;
        ix0[j] = qx * a.in_step;
        xrow[j] = xg + (((int64_t)b * a.H + iy0[j]) * a.W + ix0[j]) * a.x_ps + a.x_co + slot * C::CE;
    }
    const T* wrow[LB];
    bool wok[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int n = tid / C::CPR + j * (NTHREADS / C::CPR);
        wok[j] = (n0 + n) < a.Cout;
        wrow[j] = wg + (int64_t)(n0 + n) * a.Cin + slot * C::CE;
    }

    u32x4 xr[LA], wr[LB];
    auto load_step = [&](int step) {
        const int t = step / kchunks;
        const int c0 = (step - t * kchunks) * BK;
        int dy, dx, wt;
        tap_at(a, taps, t, dy, dx, wt);
        const int64_t xo = ((int64_t)dy * a.W + dx) * a.x_ps + c0;
        const int64_t wo = (int64_t)wt * a.Cout * a.Cin + c0;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const bool ok = rowok[j] && (unsigned)(iy0[j] + dy) < (unsigned)a.H && (unsigned)(ix0[j] + dx) < (unsigned)a.W;
            xr[j] = ok ? *(const u32x4*)(xrow[j] + xo) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) wr[j] = wok[j] ? *(const u32x4*)(wrow[j] + wo) : u32x4{0, 0, 0, 0};
    };
    auto store_step = [&](int buf) {
        unsigned char* xs = smem + buf * (XT + WT);
        unsigned char* ws = xs + XT;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int row = tid / C::CPR + j * (NTHREADS / C::CPR);
            u32x4 v = xr[j];
            if (a.in_abs) v = abs_chunk(v, T());
            *(u32x4*)(xs + lds_off<T>(row, slot)) = v;
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int row = tid / C::CPR + j * (NTHREADS / C::CPR);
            *(u32x4*)(ws + lds_off<T>(row, slot)) = wr[j];
        }
    };

    // ---- wave tiling: 2 (cout halves) x 2 (pixel halves)
    constexpr int MI = BN / 64;          // 32-cout tiles per wave
    constexpr int NI = 2;                // 32-pixel tiles per wave
    const int wm = wave & 1, wn = wave >> 1;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;

    load_step(0);
    store_step(0);
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) load_step(step + 1);
        const unsigned char* xs = smem + buf * (XT + WT);
        const unsigned char* ws = xs + XT;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h16x8 wf[MI], xf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    wf[i] = *(const h16x8*)(ws + lds_off<T>(wm * (BN / 2) + i * 32 + frow, ks * 2 + fh));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    xf[j] = *(const h16x8*)(xs + lds_off<T>(wn * 64 + j * 32 + frow, ks * 2 + fh));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = mfma_32x32x16_h16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // fp32: lane half h owns k = 16h..16h+15 of the step (any consistent k order is a valid GEMM)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x4 wf[MI], xf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    wf[i] = *(const f32x4*)(ws + lds_off<T>(wm * (BN / 2) + i * 32 + frow, fh * 4 + s));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    xf[j] = *(const f32x4*)(xs + lds_off<T>(wn * 64 + j * 32 + frow, fh * 4 + s));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i][e], xf[j][e], acc[i][j], 0, 0, 0);
            }
        }
        if (step + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias + activation, pack, stage pixel-major in LDS, write NHWC rows
    // C/D layout of the 32x32 MFMA: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (cout)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wm * (BN / 2) + i * 32 + 8 * g + 4 * fh;   // first of 4 consecutive couts
            float bv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (a.bias && (n0 + cl + e) < a.Cout) ? a.bias[n0 + cl + e] : 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int prow = wn * 64 + j * 32 + frow;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][j][4 * g + e] + bv[e], a.act);
                unsigned char* dst = smem + prow * OROW + cl * (int)sizeof(T);
                if constexpr (sizeof(T) == 2) {
                    *(u32x2*)dst = u32x2{pack_h2(v[0], v[1]), pack_h2(v[2], v[3])};
                } else {
                    *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
        }
    }
    __syncthreads();
    {
        constexpr int CPO = BN * (int)sizeof(T) / 16;           // 16B chunks per output row
        constexpr int TOT = BM * CPO;
        T* __restrict__ yg = (T*)a.y;
        const int oyo = taps.ry, oxo = taps.rx;
#pragma unroll
        for (int c = tid; c < TOT; c += NTHREADS) {
            const int prow = c / CPO, cc = c % CPO;
            const int qy = ty * a.TH + (prow >> a.tw_shift);
            const int qx = tx * a.TW + (prow & (a.TW - 1));
            const int ch = n0 + cc * C::CE;
            if (qy < a.QH && qx < a.QW && ch < a.Cout) {
                const int oy = qy * a.out_step + oyo, ox = qx * a.out_step + oxo;
                const int64_t off = (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.y_ps + a.y_co + ch;
                *(u32x4*)(yg + off) = *(const u32x4*)(smem + prow * OROW + cc * 16);
            }
        }
    }
}

// ------------------------------------------------------------------------- bf16 fast path (LDS-DMA staging)
// Same tiling and math as igemm_conv_kernel<h16_t, BN>, but the tiles go global -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write), BK is 64 when Cin allows it, and the loads of
// step s+1 are in flight while step s runs on the matrix cores.  The DMA writes LDS lane-linearly, so the
// XOR swizzle is applied on the SOURCE side: lane j of a wave instruction fetches the 16-byte chunk that
// belongs at LDS position j.  Padding taps / rows beyond the image or Cout are out-of-range buffer offsets (zero fill).

template <int LP>
__device__ __forceinline__ void wait_dma_groups(int k) {
    // leave the newest k stages (LP LDS-DMA instructions each, per wave) in flight, retire everything older
    switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LP) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LP) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LP) : "memory"); break;
    }
}

// GDN = 0: plain conv epilogue; 1 / 2: y = conv * rsqrt / sqrt(beta' + gamma' @ conv^2) fused (BN == 128 == Cout): the
// staged output tile is squared and sent through a second 128x128 MFMA contraction before it is written, so the
// activation never makes the HBM round trip between conv and (I)GDN (compressai/layers/gdn.py:55-70).
// two blocks per CU (2 waves per SIMD, <= 256 VGPRs) whenever the ring leaves LDS for two
constexpr int igemm_waves_per_eu(int bm, int bn, int bk, int ns, int nw) { return (ns * (bm + bn) * bk * 2 > 80 * 1024 ? 1 : 2) * nw / 4; }

// NW = waves per block: 4 (2x2 wave grid, 64x64 wave tiles).  NW = 8 (4x2 grid, 32 couts x 64 pixels per wave: twice the
// waves per SIMD, 1.5 fragment reads per MFMA) compiles and is correct but measured EQUAL on every layer (the loop is not
// limited by per-wave latency), so only NW = 4 is instantiated.
// (Rounds 2 - 5 carried a "wave-specialised" form -- NW extra LOADER waves per block issuing all LDS-DMA, the compute waves nothing but fragment
// reads + MFMAs + one barrier per stage.  Measured, conv 128 -> 128 5x5 s2 @256^2 B=8 + GDN, same box, back to back: 144.0 us against 143.8 us for
// this self-loading form at two blocks per CU; the ablations said why: without the DMA the self-loading form runs 84 us, without the MFMAs 86 us,
// without both 33 us -- the matrix phase (~58 us) and the LDS-fill phase (~60 us) ADD instead of overlapping, and the fill alone costs 39 us even
// from an L1-resident source: 1.64 GB through the ~64 B/clk/CU global -> LDS path, which a 128 x 128 tile needs at 62.5 B/clk/CU to feed the
// matrix cores at peak.  The bound is the tile's arithmetic intensity against that path, not issue scheduling.  Removed in round 6.)
template <int BMP, int BN, int BK, int NS, int GDN = 0, int NW = 4, int WS_UNUSED = 0, int HL = 0>
__global__ __launch_bounds__(NW * 64, igemm_waves_per_eu(BMP, BN, BK, NS, NW)) void igemm_glds_kernel(const IgemmArgs a) {
    constexpr int NTHREADS = NW * 64;          // COMPUTE threads (the epilogue's copy loops stride by this)
    using T = h16_t;
    constexpr int BM = BMP;                   // pixels per block: 128, 64 or 32 (small layers need more blocks)
    constexpr int CPR = BK * 2 / 16;          // 16-byte chunks per LDS row
    constexpr int RPB = 256 / (BK * 2);       // rows per 256-byte bank row
    constexpr int XT = BM * BK * 2, WT = BN * BK * 2, STAGE = XT + WT;
    constexpr int XI = BM * CPR / 64 / NW;    // x-tile DMA instructions per wave per step
    constexpr int WI = BN * CPR / 64 / NW;
    static_assert(XI >= 1 && WI >= 1, "tile too small");
    constexpr int WN = BM >= 256 ? NW / 2 : (BM >= 64 ? 2 : 1), WM = NW / WN;    // wave grid: WM cout slices x WN pixel slices
    constexpr int OROW = BN * 2 + 16;
    constexpr int EPI = BM * OROW;
    static_assert(NS >= 2 && NS <= 4 && (XI + WI) * 3 <= 63, "ring depth / vmcnt range");
    static_assert(WS_UNUSED == 0, "the loader-wave form was removed in round 6");
    static_assert(GDN == 0 || BN == 128, "fused GDN needs every channel of a pixel in the block");
    constexpr int BKH = HL ? BK / 2 : BK;     // channels a stage advances by (HL: a row holds BK/2 hi channels and their BK/2 lo partners)
    constexpr int YOFF = BM * 256;                            // fused GDN: squared tile at 0, output tile behind it
    constexpr int EPI_ALL = GDN ? 2 * YOFF : EPI;
    constexpr int LDS_BYTES = NS * STAGE > EPI_ALL ? NS * STAGE : EPI_ALL;
    // binary16 build, pair (I)GDN: the squares are scaled PER PIXEL (see the epilogue); the wave slices of a pixel exchange their maxima here
    constexpr bool DYN_SQ = HESIC_H16_IS_F16 && !HESIC_NO_DYN_SQ && GDN >= 3;
    // ... in the ring's spare room behind the two epilogue tiles when there is some (one more barrier, no more LDS: the <32,128,128> tile's ring
    // is exactly 80 KB = two blocks per CU, 512 bytes on top halved its occupancy, 62.8 -> 97.8 us), else in 1 - 2 KB of their own
    constexpr bool XMAX_IN_RING = DYN_SQ && (EPI_ALL + WM * BM * 4 <= NS * STAGE);
    constexpr int XMAX_BYTES = (DYN_SQ && !XMAX_IN_RING) ? WM * BM * 4 : 0;
    constexpr int XMAX_OFF = XMAX_IN_RING ? EPI_ALL : LDS_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + XMAX_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: LDS-DMA bases go to M0
    const int lw = wave;     // index among the waves that issue the DMA
    int bid;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // order: cout tile, output phase, tile x, tile y, image, K slice -- the phases of a transposed conv read the same input
    // tile, so they are neighbours in the launch order (same XCD, same time) and the tile leaves HBM once, not once per phase
    uint32_t rest = fdiv((uint32_t)bid, a.fd_nt);
    const int nt = bid - (int)rest * a.n_tiles;
    uint32_t q_ = fdiv(rest, a.fd_ph);
    const int ph = (int)(rest - q_ * (uint32_t)a.nphase);
    rest = q_; q_ = fdiv(rest, a.fd_tx);
    const int tx = (int)(rest - q_ * (uint32_t)a.tiles_x);
    rest = q_; q_ = fdiv(rest, a.fd_ty);
    const int ty = (int)(rest - q_ * (uint32_t)a.tiles_y);
    rest = q_; q_ = fdiv(rest, a.fd_b);
    const int b = (int)(rest - q_ * (uint32_t)a.B);
    const int kslice = (int)q_;
    const int n0 = nt * BN;
    const Taps taps = make_taps(a, ph);
    const int kchunks = a.Cin / BK;
    int step_lo = 0, step_hi = taps.ntaps * kchunks;
    if (a.ksplit > 1) { step_lo = kslice * step_hi / a.ksplit; step_hi = (kslice + 1) * step_hi / a.ksplit; }
    const int nsteps = step_hi - step_lo;
    const T* __restrict__ xg = (const T*)a.x;
    const T* __restrict__ wg = (const T*)a.w;

    constexpr int MI = BN / WM / 32, NI = BM / WN / 32;
    static_assert(MI >= 1 && NI >= 1, "wave tile too small");
    const int wm = wave % WM, wn = wave / WM;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    auto off = [&](int row, int slot) { return (row * CPR + (slot ^ ((row / RPB) & (CPR - 1)))) * 16; };

    // ---- buffer-addressed LDS-DMA: one resource per operand, a loop-invariant 32-bit VGPR offset per DMA row and a
    // wave-uniform SGPR offset for (tap, channel chunk).  Rows that fall into the padding (or beyond the q-grid /
    // Cout) carry an out-of-range offset: the buffer unit then writes zeros into LDS, so a stage costs no per-lane
    // address arithmetic at all; the per-row offsets are re-selected only when the tap changes.
    constexpr uint32_t OOB = 0x80000000u;
    // the buffer-load-to-LDS builtin is not modelled as a store to smem: let the ring escape through an empty asm so
    // that the "memory"-clobbering waits below count as writers of it
    asm volatile("" ::"v"((__attribute__((address_space(3))) unsigned char*)smem) : "memory");
    const int neg = (a.KH * a.W + a.KW) * a.x_ps;                  // elements; keeps every tap offset non-negative
    const int row0 = ty * a.TH * a.in_step;                        // offsets are relative to the tile's first input row
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(xg + ((int64_t)b * a.H + row0) * a.W * a.x_ps - neg), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (int)OOB, 0x00020000);
    const int prow = lane / CPR, pslot = lane % CPR;
    const int gco = a.x_group_step ? (nt / a.tiles_per_group) * a.x_group_step : 0;      // grouped launch: this cout tile's input slice
    const int act_eff = (a.act_split && n0 >= a.act_split) ? a.act2 : a.act;
    uint32_t xoff[XI], xv[XI], wv[WI];
    int iy0[XI], ix0[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = (lw * XI + i) * (64 / CPR) + prow;
        const int ls = pslot ^ ((row / RPB) & (CPR - 1));
        const int qy = ty * a.TH + (row >> a.tw_shift), qx = tx * a.TW + (row & (a.TW - 1));
        const bool ok = qy < a.QH && qx < a.QW;
        iy0[i] = ok ? qy * a.in_step : (int)0xc0000000;            // far outside: every tap of a dead row reads zeros
        ix0[i] = qx * a.in_step;
        const int lc = HL ? (ls % (CPR / 2)) * 8 + (ls / (CPR / 2)) * (a.Cin / 2) : ls * 8;     // HL: slots >= CPR/2 come from the lo half, C channels further
        xoff[i] = (uint32_t)((((qy * a.in_step - row0) * a.W + qx * a.in_step) * a.x_ps + a.x_co + gco + lc) * 2);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = (lw * WI + i) * (64 / CPR) + prow;
        const int ls = pslot ^ ((row / RPB) & (CPR - 1));
        const int lc = HL ? (ls % (CPR / 2)) * 8 + (ls / (CPR / 2)) * (a.Cin / 2) : ls * 8;
        wv[i] = (n0 + row) < a.Cout ? (uint32_t)(((n0 + row) * a.Cin + lc) * 2) : OOB;
    }
    // tap cursor (all SGPR): input displacement moves by +1 (conv) / -1 (transposed phase) per tap index
    const int dstep = a.transposed ? -1 : 1;
    const int dy_base = a.transposed ? (taps.ry + a.pad - taps.ky0) >> (a.stride >> 1) : -a.pad;      // exact: multiple of the stride
    const int dx_base = a.transposed ? (taps.rx + a.pad - taps.kx0) >> (a.stride >> 1) : -a.pad;
    const uint32_t wtap = (uint32_t)(a.Cout * a.Cin * 2);
    // Tap order.  A stride-2 conv reads input pixel (2q + k): taps of equal (ky & 1, kx & 1) touch the SAME quarter of the
    // input pixels (shifted by whole output pixels), the other three parities touch disjoint quarters.  Walking the taps
    // in raster order therefore cycles the whole input footprint of the co-resident blocks (~9 MB per XCD on the big
    // layers) through the 4 MB L2 several times (measured: 511 MB fetched for 134 MB of input); walking them parity class
    // by parity class -- (0,0): 9 taps, (0,1): 6, (1,0): 6, (1,1): 4 -- keeps one 2 MB quarter resident at a time.
    const bool par = a.tap_parity != 0;
    int kst_e = par ? 2 : taps.kst, ky0_e = taps.ky0, kx0_e = taps.kx0, cls = 0;
    int nkx_e = par ? (a.KW + 1) >> 1 : taps.nkx, nky_e = par ? (a.KH + 1) >> 1 : 0x7fffffff;
    int cur_j = 0, cur_c = 0, cur_chunk = 0;
    if (step_lo) { const int tap_lo = step_lo / kchunks; cur_j = tap_lo / taps.nkx; cur_c = tap_lo % taps.nkx; cur_chunk = step_lo % kchunks; }
    int dy = dy_base + dstep * cur_j, dx = dx_base + dstep * cur_c;
    int dx_row = dx_base;                                 // dx at the start of a tap row of the current class
    uint32_t s_x = 0, s_w = 0;
    auto set_tap = [&]() {
        s_x = (uint32_t)(((dy * a.W + dx) * a.x_ps + neg) * 2);
        s_w = (uint32_t)((ky0_e + cur_j * kst_e) * a.KW + kx0_e + cur_c * kst_e) * wtap;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const bool ok = (unsigned)(iy0[i] + dy) < (unsigned)a.H && (unsigned)(ix0[i] + dx) < (unsigned)a.W;
            xv[i] = ok ? xoff[i] : OOB;
        }
    };
    set_tap();
    auto next_tap = [&]() {
        dx += dstep * (par ? 2 : 1);
        if (++cur_c == nkx_e) {
            cur_c = 0; dx = dx_row; ++cur_j; dy += dstep * (par ? 2 : 1);
            if (cur_j == nky_e) {                         // parity walk only: next class
                ++cls;
                ky0_e = cls >> 1; kx0_e = cls & 1;
                nkx_e = (a.KW - kx0_e + 1) >> 1; nky_e = (a.KH - ky0_e + 1) >> 1;
                cur_j = 0; dy = ky0_e - a.pad; dx_row = kx0_e - a.pad; dx = dx_row;
            }
        }
        set_tap();
    };
    auto issue = [&](int buf) {
        const uint32_t sx = s_x + (uint32_t)(cur_chunk * BKH * 2), sw = s_w + (uint32_t)(cur_chunk * BKH * 2);
        unsigned char* xs = smem + buf * STAGE;
        unsigned char* ws = xs + XT;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(xs + (lw * XI + i) * 1024),
                                                     16, (int)xv[i], (int)sx, 0, 0);
#pragma unroll
        for (int i = 0; i < WI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(ws + (lw * WI + i) * 1024),
                                                     16, (int)wv[i], (int)sw, 0, 0);
        if (++cur_chunk == kchunks) {
            cur_chunk = 0;
            next_tap();
        }
    };

#ifndef IGEMM_HL_MFMA16
#define IGEMM_HL_MFMA16 1
#endif
    constexpr bool HL16 = IGEMM_HL_MFMA16 && HL != 0 && BK >= 64;      // pair launches on the 16x16x32 instruction (every pair layer of the models: Cin_k % 64 == 0)
    // Main loop.  Fragment reads are software-pipelined one k-substep ahead of the MFMAs that consume them (two
    // register sets), and the DMA issue for stage s+NS-1 sits between the first fragment read of stage s and its
    // MFMAs, so neither the LDS latency nor the DMA bookkeeping is exposed in front of the matrix pipe.
    auto main_loop = [&](auto abs_tag) {
        constexpr bool ABS = decltype(abs_tag)::value;
        constexpr int KS = BK / 16;
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nsteps) issue(s);
        int buf = 0, nxt = NS - 1;
        for (int step = 0; step < nsteps; ++step) {
            const int rem = nsteps - 1 - step;
            if constexpr (NS == 2) wait_dma_groups<XI + WI>(0);
            else wait_dma_groups<XI + WI>(rem < NS - 2 ? rem : NS - 2);
            __builtin_amdgcn_s_barrier();
            const unsigned char* xs = smem + buf * STAGE;
            const unsigned char* ws = xs + XT;
            buf = (buf + 1 == NS) ? 0 : buf + 1;
            if constexpr (HL16) {
                // hi/lo operands on v_mfma_f32_16x16x32 (round 6): one instruction spans 32 channels, so a 64-wide stage (32 hi channels and their lo
                // partners) is ONE k-step.  Same fragment bytes and accumulator registers as the 32x32x16 form below (16-row blocks: lane = (row lane & 15,
                // k-quarter lane >> 4), slot ks2 * 4 + quarter of the hi / lo half), but half the accumulator traffic per flop -- at the board's power cap,
                // where these launches run, the register-only loop of this instruction holds 1.95 - 2.0 PFLOP/s against 1.73.  The accumulators live in
                // `acc`'s registers as 16 x 16 tiles (tile (a, b) = registers ((a & 1) * 2 + (b & 1)) * 4 .. + 3 of acc[a >> 1][b >> 1]) and are turned
                // into the 32 x 32 layout the epilogues read ONCE behind the K loop (hl16_to_32 below).  Summation order per output value: stages as
                // before, inside a 32-channel step hi hi, hi lo, lo hi -- the same in every tile variant.
                constexpr int KS32 = BK / 64, LO = CPR / 2, MA = 2 * MI, NB = 2 * NI;
                const int c16 = lane & 15, q16 = lane >> 4;
                h16x8 wh[MA], wl[MA], xh[NB], xl[NB];
                auto ld16 = [&](int ks2) {
#pragma unroll
                    for (int i = 0; i < MA; ++i) {
                        const int r = wm * (BN / WM) + i * 16 + c16;
                        wh[i] = *(const h16x8*)(ws + off(r, ks2 * 4 + q16));
                        if constexpr (HL == 1) wl[i] = *(const h16x8*)(ws + off(r, LO + ks2 * 4 + q16));
                    }
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int r = wn * (BM / WN) + j * 16 + c16;
                        xh[j] = *(const h16x8*)(xs + off(r, ks2 * 4 + q16));
                        xl[j] = *(const h16x8*)(xs + off(r, LO + ks2 * 4 + q16));
                    }
                };
                ld16(0);
                __builtin_amdgcn_sched_barrier(0);
                if (step + NS - 1 < nsteps) issue(nxt);
                nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks2 = 0; ks2 < KS32; ++ks2) {
                    if (ks2) { ld16(ks2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                    for (int i = 0; i < MA; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            const int r0 = ((i & 1) * 2 + (j & 1)) * 4;
                            f32x16& A_ = acc[i >> 1][j >> 1];
                            f32x4 c = {A_[r0], A_[r0 + 1], A_[r0 + 2], A_[r0 + 3]};
                            c = mfma_16x16x32_h16(wh[i], xh[j], c, 0, 0, 0);
                            c = mfma_16x16x32_h16(wh[i], xl[j], c, 0, 0, 0);
                            if constexpr (HL == 1) c = mfma_16x16x32_h16(wl[i], xh[j], c, 0, 0, 0);
                            A_[r0] = c[0]; A_[r0 + 1] = c[1]; A_[r0 + 2] = c[2]; A_[r0 + 3] = c[3];
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
            if constexpr (HL) {
                // hi/lo operands: per 16-deep k-substep four fragment groups (w_hi, w_lo, x_hi, x_lo) feed 3 * MI * NI MFMAs.
                // HL == 2 (round 4, the "x3c2" analysis mode's g_a_conv3 / conv4): the weights are SINGLE values rounded with error feedback
                // over the taps (the w_lo half of the packed rows is zero and is neither read nor multiplied): two products per pair
                constexpr int KSH = BK / 32, LO = CPR / 2;
                h16x8 wh[2][MI], wl[2][MI], xh[2][NI], xl[2][NI];
                auto ldh = [&](int set, int ks) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int r = wm * (BN / WM) + i * 32 + frow;
                        wh[set][i] = *(const h16x8*)(ws + off(r, ks * 2 + fh));
                        if constexpr (HL == 1) wl[set][i] = *(const h16x8*)(ws + off(r, LO + ks * 2 + fh));
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int r = wn * (BM / WN) + j * 32 + frow;
                        xh[set][j] = *(const h16x8*)(xs + off(r, ks * 2 + fh));
                        xl[set][j] = *(const h16x8*)(xs + off(r, LO + ks * 2 + fh));
                    }
                };
                ldh(0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (step + NS - 1 < nsteps) issue(nxt);
                nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) {
                    if (ks + 1 < KSH) ldh((ks + 1) & 1, ks + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            acc[i][j] = mfma_32x32x16_h16(wh[ks & 1][i], xh[ks & 1][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = mfma_32x32x16_h16(wh[ks & 1][i], xl[ks & 1][j], acc[i][j], 0, 0, 0);
                            if constexpr (HL == 1) acc[i][j] = mfma_32x32x16_h16(wl[ks & 1][i], xh[ks & 1][j], acc[i][j], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
            h16x8 wf[2][MI], xf[2][NI];
            auto ldf = [&](int set, int ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i) wf[set][i] = *(const h16x8*)(ws + off(wm * (BN / WM) + i * 32 + frow, ks * 2 + fh));
#pragma unroll
                for (int j = 0; j < NI; ++j) xf[set][j] = *(const h16x8*)(xs + off(wn * (BM / WN) + j * 32 + frow, ks * 2 + fh));
            };
            ldf(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (step + NS - 1 < nsteps) issue(nxt);
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) ldf((ks + 1) & 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (ABS) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        u32x4 v = __builtin_bit_cast(u32x4, xf[ks & 1][j]);
                        v = u32x4{v.x & 0x7fff7fffu, v.y & 0x7fff7fffu, v.z & 0x7fff7fffu, v.w & 0x7fff7fffu};
                        xf[ks & 1][j] = __builtin_bit_cast(h16x8, v);
                    }
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = mfma_32x32x16_h16(wf[ks & 1][i], xf[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (a.in_abs) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    if constexpr (HL16) {
        // hl16_to_32: the wave's (BN / WM) x (BM / WN) fp32 tile through its own slice of the (now dead) ring -- written from the 16 x 16 tiles (lane =
        // pixel lane & 15, couts 4 (lane >> 4) .. + 3 of a 16-cout block), read back in the 32 x 32 layout (lane = pixel lane & 31, couts
        // 8 g + 4 (lane >> 5) .. + 3 of a 32-cout block); 16-byte slot s of a pixel row at s ^ (pixel & (slots - 1)).  Once per tile: ~2 x 4 MI NI
        // LDS instructions per lane against thousands of MFMAs.
        constexpr int CW = BN / WM, PW = BM / WN, SL = CW / 4;             // couts, pixels, 16-byte slots per pixel row of the wave's tile
        static_assert(NW * CW * PW * 4 <= LDS_BYTES, "the accumulator turn-round needs the wave tiles' fp32 size in LDS");
        unsigned char* tb = smem + wave * (CW * PW * 4);
        const int c16 = lane & 15, q16 = lane >> 4;
        asm volatile("s_barrier" ::: "memory");                            // every wave is done reading the ring
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
            for (int j = 0; j < 2 * NI; ++j) {
                const int r0 = ((i & 1) * 2 + (j & 1)) * 4, px = j * 16 + c16, sl = i * 4 + q16;
                const f32x16& A_ = acc[i >> 1][j >> 1];
                *(f32x4*)(tb + (px * SL + (sl ^ (px & (SL - 1)))) * 16) = f32x4{A_[r0], A_[r0 + 1], A_[r0 + 2], A_[r0 + 3]};
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // wave-private: the LDS executes a wave's accesses in order; the compiler must not
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int px = j * 32 + frow, sl = i * 8 + 2 * g + fh;
                    const f32x4 v = *(const f32x4*)(tb + (px * SL + (sl ^ (px & (SL - 1)))) * 16);
                    acc[i][j][4 * g] = v[0]; acc[i][j][4 * g + 1] = v[1]; acc[i][j][4 * g + 2] = v[2]; acc[i][j][4 * g + 3] = v[3];
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // ... and the epilogues' own barriers stand before their first LDS write
    }
    if constexpr (HL != 0) {
        // pair weights packed as (w * 2^s)_hi | (w * 2^s)_lo (binary16: the lo half of an unscaled 0.02 is a subnormal half, the pair then
        // carries 2^-20 instead of 2^-22): the sums come out times 2^s, exactly; one multiply per value here, in front of every epilogue
        // (a K slice's partial tile is scaled the same way: the reduce adds scaled partials)
        if (a.acc_scale != 1.f) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= a.acc_scale;
        }
    }
    if constexpr (GDN == 0) {
        if (a.ksplit > 1) {
            // K slice: raw fp32 partial sums straight from the accumulators (16 bytes per lane and channel quad)
            float* __restrict__ wsp = a.ws + (int64_t)kslice * a.B * a.Ho * a.Wo * a.Cout;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int pr = wn * (BM / WN) + j * 32 + frow;
                const int qy = ty * a.TH + (pr >> a.tw_shift), qx = tx * a.TW + (pr & (a.TW - 1));
                if (qy < a.QH && qx < a.QW) {
                    const int oy = qy * a.out_step + taps.ry, ox = qx * a.out_step + taps.rx;
                    float* dst = wsp + (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + n0;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
                            if (n0 + cl < a.Cout)
                                *(f32x4*)(dst + cl) = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        }
                }
            }
            return;
        }
        if (a.y32) {
            // fp32 copy of the output (latents y / z and the sigma, mu maps): 16 bytes per lane and channel quad, the two lane
            // halves of a pixel side by side (32-byte runs; the whole 4*BN-byte row leaves the wave within 8 instructions)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int pr = wn * (BM / WN) + j * 32 + frow;
                const int qy = ty * a.TH + (pr >> a.tw_shift), qx = tx * a.TW + (pr & (a.TW - 1));
                if (qy < a.QH && qx < a.QW) {
                    const int oy = qy * a.out_step + taps.ry, ox = qx * a.out_step + taps.rx;
                    float* dst = a.y32 + (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.y32_ps + a.y32_co + n0;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
                            if (n0 + cl < a.Cout) {
                                const f32x4 bq = a.bias ? *(const f32x4*)(a.bias + n0 + cl) : f32x4{0.f, 0.f, 0.f, 0.f};
                                *(f32x4*)(dst + cl) = f32x4{apply_act(acc[i][j][4 * g] + bq[0], act_eff), apply_act(acc[i][j][4 * g + 1] + bq[1], act_eff),
                                                            apply_act(acc[i][j][4 * g + 2] + bq[2], act_eff), apply_act(acc[i][j][4 * g + 3] + bq[3], act_eff)};
                            }
                        }
                }
            }
            if (!a.y) return;
        }
        // plain epilogue; with a.y_hilo the output is written twice through the same staging tile: channels [y_co, +Cout) take
        // hi = bf16(v), channels [y_co + Cout, +Cout) lo = bf16(v - hi), v = act(conv + bias) (|v| with a.y_abs: the hyper-analysis reads |y|)
        T* __restrict__ yg = (T*)a.y;
        const int oyo = taps.ry, oxo = taps.rx;
        const int nhalf = a.y_hilo ? 2 : 1;
        for (int half = 0; half < nhalf; ++half) {
            __syncthreads();     // every wave is done with the ring (or with the previous half's rows) before the tile is rewritten
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
                    float bv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = (a.bias && (n0 + cl + e) < a.Cout) ? a.bias[n0 + cl + e] : 0.f;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int pr = wn * (BM / WN) + j * 32 + frow;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = apply_act(acc[i][j][4 * g + e] + bv[e], act_eff);
                            if (a.y_abs) v[e] = fabsf(v[e]);
                        }
                        uint32_t h0, l0, h1, l1;
                        split_h2(v[0], v[1], h0, l0);
                        split_h2(v[2], v[3], h1, l1);
                        const u32x2 o = half ? u32x2{l0, l1} : u32x2{h0, h1};
                        *(u32x2*)(smem + pr * OROW + cl * 2) = o;
                    }
                }
            }
            __syncthreads();
            constexpr int CPO = BN * 2 / 16;
            constexpr int TOT = BM * CPO;
#pragma unroll
            for (int c = tid; c < TOT; c += NTHREADS) {
                const int pr = c / CPO, cc = c % CPO;
                const int qy = ty * a.TH + (pr >> a.tw_shift), qx = tx * a.TW + (pr & (a.TW - 1));
                const int ch = n0 + cc * 8;
                if (qy < a.QH && qx < a.QW && ch < a.Cout) {
                    const int oy = qy * a.out_step + oyo, ox = qx * a.out_step + oxo;
                    const int64_t o = (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.y_ps + a.y_co + half * a.Cout + ch;
                    *(u32x4*)(yg + o) = *(const u32x4*)(smem + pr * OROW + cc * 16);
                }
            }
        }
    } else if constexpr (GDN >= 3) {
        // ---- fused (I)GDN epilogue on hi/lo operands (analysis path of the bf16x3 mode): same data flow as the epilogue below,
        // but nothing passes through ONE bf16: the squares go to LDS as a hi tile and a lo tile, the contraction is
        // gamma_hi sq_hi + gamma_hi sq_lo + gamma_lo sq_hi on top of beta' (fp32 accumulators), the product v * rsqrt(nrm) stays
        // fp32 and leaves as [hi(128) | lo(128)] per pixel -- 2^-17 relative instead of 2^-9 at every layer boundary.
        constexpr bool INV = GDN == 4;
        // Binary16 build (DYN_SQ): a square v^2 * 2^-6 of |v| < 0.0625 is a SUBNORMAL half (absolute step 6e-8, its lo partner carries nothing):
        // with beta' = 1e-2 and activations of 0.1 the norm comes out 4e-6 relative wrong, with beta' = 1e-4 up to 1e-3
        // (profiles/scripts/gdn_pair_precision.py) -- worse than bfloat16 pairs (1.4e-6), and far from the 2^-22 the pair carries elsewhere.
        // So a pixel's squares are formed from u = v * 2^k with k chosen from the pixel's own largest |v| over all 128 channels (|u| in
        // [2^6, 2^7): u^2 <= 2^14 fits, the largest square keeps all 22 bits, a channel 2^-13 below it is still normal); the contraction then
        // holds 64 * 4^k * sum(gamma' v^2) (gamma' is packed times 64), beta' is added AFTER it in fp32: n = beta' + nrm * 2^(-6 - 2k).
        // Powers of two throughout: no rounding of its own.  bfloat16 build: its range needs none of this (k = -3, as before).
        f32x16 nrm[MI][NI];
        [[maybe_unused]] float sq_c[NI], sq_inv[NI];
        // beta' of this lane's channels is used behind the contraction: small tiles (registers to spare, one tile per block: the L2 round trip
        // would sit in front of the output) request it here, the 128- / 256-pixel tiles (252+ registers) re-read it there
        constexpr bool EARLY_BE = DYN_SQ && BM <= 64;
        [[maybe_unused]] f32x4 beq[EARLY_BE ? MI : 1][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
                const f32x4 t = a.bias ? *(const f32x4*)(a.bias + cl) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 be = {0.f, 0.f, 0.f, 0.f};
                if constexpr (!DYN_SQ) be = *(const f32x4*)(a.gdn_beta + cl);
                else if constexpr (EARLY_BE) beq[i][g] = *(const f32x4*)(a.gdn_beta + cl);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < NI; ++j) { acc[i][j][4 * g + e] += t[e]; nrm[i][j][4 * g + e] = be[e]; }
            }
        if constexpr (DYN_SQ) {
            float* xmax = (float*)(smem + XMAX_OFF);                      // [WM][BM]
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[i][j][r]));
                m = fmaxf(m, __shfl_xor(m, 32));                          // the other 4-channel halves of the same pixel
                sq_c[j] = m;
                // outside the ring it can be written now (slower waves may still be reading the ring); inside it, behind the barrier below
                if (!XMAX_IN_RING && WM > 1 && fh == 0) xmax[wm * BM + wn * (BM / WN) + j * 32 + frow] = m;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        h16x8 gq[MI][8];
        {
            const h16x8* gfr = (const h16x8*)a.gdn_gamma + 128 * 16;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) gq[i][ks] = gfr[((wm * MI + i) * 8 + ks) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        auto split2 = [](float p, float q, uint32_t& hi, uint32_t& lo) { split_h2(p, q, hi, lo); };
        if constexpr (DYN_SQ && !XMAX_IN_RING) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... and the pixel maxima are visible
        else asm volatile("s_barrier" ::: "memory");                   // every wave is done reading the ring
        if constexpr (XMAX_IN_RING && WM > 1) {
            float* xmax = (float*)(smem + XMAX_OFF);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (fh == 0) xmax[wm * BM + wn * (BM / WN) + j * 32 + frow] = sq_c[j];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if constexpr (DYN_SQ) {
            const float* xmax = (const float*)(smem + XMAX_OFF);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                float m = sq_c[j];
                if (WM > 1) {
#pragma unroll
                    for (int w = 0; w < WM; ++w) m = fmaxf(m, xmax[w * BM + wn * (BM / WN) + j * 32 + frow]);
                }
                int ex = 0;
                (void)frexpf(m, &ex);                                      // m in [2^(ex-1), 2^ex); 0 -> ex = 0
                int k = 7 - ex;
                k = k > 40 ? 40 : (k < -24 ? -24 : k);                     // inf / NaN / denormal maxima: stay finite
                sq_c[j] = ldexpf(1.f, k);
                sq_inv[j] = ldexpf(1.f, -6 - 2 * k);
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int pr = wn * (BM / WN) + j * 32 + frow;
                    float v0 = acc[i][j][4 * g], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                    uint32_t h0, l0, h1, l1;
                    if constexpr (DYN_SQ) {
                        v0 *= sq_c[j]; v1 *= sq_c[j]; v2 *= sq_c[j]; v3 *= sq_c[j];
                        split2(v0 * v0, v1 * v1, h0, l0);
                        split2(v2 * v2, v3 * v3, h1, l1);
                    } else {
                    split2(v0 * v0 * H16_SQ_SCALE, v1 * v1 * H16_SQ_SCALE, h0, l0);
                    split2(v2 * v2 * H16_SQ_SCALE, v3 * v3 * H16_SQ_SCALE, h1, l1);
                    }
                    const u32x2 h = u32x2{h0, h1}, l = u32x2{l0, l1};
                    const int o = pr * 256 + (((cl >> 3) ^ (pr & 15)) << 4) + (cl & 7) * 2;
                    *(u32x2*)(smem + o) = h;
                    *(u32x2*)(smem + YOFF + o) = l;
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // squares visible to every wave
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            h16x8 qh[NI], ql[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int pr = wn * (BM / WN) + j * 32 + frow;
                qh[j] = *(const h16x8*)(smem + pr * 256 + (((ks * 2 + fh) ^ (pr & 15)) << 4));
                ql[j] = *(const h16x8*)(smem + YOFF + pr * 256 + (((ks * 2 + fh) ^ (pr & 15)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    nrm[i][j] = mfma_32x32x16_h16(gq[i][ks], qh[j], nrm[i][j], 0, 0, 0);
                    nrm[i][j] = mfma_32x32x16_h16(gq[i][ks], ql[j], nrm[i][j], 0, 0, 0);
                }
        }
        // gamma'_lo * sq_hi: only the strict pair form (HL == 1).  The single-operand (HL == 0: g_a_conv2 of "x3c2") and single-weight
        // (HL == 2) launches stop at gamma'_hi * (sq_hi + sq_lo): their conv already carries 2^-12 per operand, a single gamma' adds < 3 %
        // to the flip count (CPU study)
        if constexpr (HL == 1) {
            {
                const h16x8* gfr = (const h16x8*)a.gdn_gamma_lo;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) gq[i][ks] = gfr[((wm * MI + i) * 8 + ks) * 64 + lane];
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                h16x8 qh[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int pr = wn * (BM / WN) + j * 32 + frow;
                    qh[j] = *(const h16x8*)(smem + pr * 256 + (((ks * 2 + fh) ^ (pr & 15)) << 4));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) nrm[i][j] = mfma_32x32x16_h16(gq[i][ks], qh[j], nrm[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has consumed both square tiles
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int pr = wn * (BM / WN) + j * 32 + frow;
                    float v[4];
                    [[maybe_unused]] f32x4 be = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (EARLY_BE) be = beq[i][g];
                    else if constexpr (DYN_SQ) be = *(const f32x4*)(a.gdn_beta + cl);      // L1 / L2 hits: 512 bytes per layer
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float n = DYN_SQ ? fmaf(nrm[i][j][4 * g + e], sq_inv[j], be[e]) : nrm[i][j][4 * g + e];
                        v[e] = acc[i][j][4 * g + e] * (INV ? sqrtf(n) : rsqrtf(n));
                    }
                    uint32_t h0, l0, h1, l1;
                    split2(v[0], v[1], h0, l0);
                    split2(v[2], v[3], h1, l1);
                    const u32x2 h = u32x2{h0, h1}, l = u32x2{l0, l1};
                    const int o = pr * 256 + (((cl >> 3) ^ (pr & 15)) << 4) + (cl & 7) * 2;
                    *(u32x2*)(smem + o) = h;
                    *(u32x2*)(smem + YOFF + o) = l;
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        constexpr int TOT = BM * 16;
        T* __restrict__ yg = (T*)a.y;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int c = tid; c < TOT; c += NTHREADS) {
                const int pr = c >> 4, cc = c & 15;
                const int qy = ty * a.TH + (pr >> a.tw_shift), qx = tx * a.TW + (pr & (a.TW - 1));
                if (qy < a.QH && qx < a.QW) {
                    const int oy = qy * a.out_step + taps.ry, ox = qx * a.out_step + taps.rx;
                    const int64_t o = (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.y_ps + a.y_co + half * 128 + cc * 8;
                    *(u32x4*)(yg + o) = *(const u32x4*)(smem + half * YOFF + pr * 256 + ((cc ^ (pr & 15)) << 4));
                }
            }
        }
    } else {
        // ---- fused (I)GDN epilogue.  Every lane already owns 4 consecutive channels of its pixels, so
        //   1. bias, beta' and the gamma' MFMA fragments (fragment order in global memory: one coalesced 16-byte load per
        //      lane and fragment, no LDS copy) are requested first and stay in flight across the barriers below;
        //   2. v = conv + bias is squared in fp32 registers and only the bf16 squares go to LDS (XOR-swizzled 256-byte pixel
        //      rows: the ring is reused, exactly BM*256 bytes) -- the one cross-wave exchange the channel contraction needs;
        //   3. nrm = beta' + gamma' @ v^2 runs on the matrix cores with the accumulators preloaded with beta';
        //   4. y = v * rsqrt(nrm) (GDN) or v * sqrt(nrm) (IGDN) is staged behind the squared tile, so writing it needs no
        //      barrier against lanes still reading squares.
        // Barriers are raw s_barrier + lgkmcnt waits: a __syncthreads() would drain the gamma' loads.
        float bv[MI][4][4];
        f32x16 nrm[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
                const f32x4 t = a.bias ? *(const f32x4*)(a.bias + cl) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 be = *(const f32x4*)(a.gdn_beta + cl);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bv[i][g][e] = t[e];
#pragma unroll
                    for (int j = 0; j < NI; ++j) nrm[i][j][4 * g + e] = be[e];
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        h16x8 gq[MI][8];
        {
            const h16x8* gfr = (const h16x8*)a.gdn_gamma + 128 * 16;       // second half of the packed buffer: fragment order
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) gq[i][ks] = gfr[((wm * MI + i) * 8 + ks) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x2 sq[MI][NI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][4 * g + e] + bv[i][g][e];
                        acc[i][j][4 * g + e] = v[e];                   // keep the fp32 conv output for the final product
                    }
                    sq[i][j][g] = u32x2{pack_sq2(v[0], v[1]), pack_sq2(v[2], v[3])};
                }
        asm volatile("s_barrier" ::: "memory");                        // every wave is done reading the ring
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int pr = wn * (BM / WN) + j * 32 + frow;
                    *(u32x2*)(smem + pr * 256 + (((cl >> 3) ^ (pr & 15)) << 4) + (cl & 7) * 2) = sq[i][j][g];
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // squares visible to every wave
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            h16x8 qf[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int pr = wn * (BM / WN) + j * 32 + frow;
                qf[j] = *(const h16x8*)(smem + pr * 256 + (((ks * 2 + fh) ^ (pr & 15)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) nrm[i][j] = mfma_32x32x16_h16(gq[i][ks], qf[j], nrm[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int pr = wn * (BM / WN) + j * 32 + frow;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float n = nrm[i][j][4 * g + e];
                        v[e] = acc[i][j][4 * g + e] * (GDN == 2 ? __builtin_amdgcn_sqrtf(n) : rsqrtf(n));   // n >= beta' > 0: raw v_sqrt_f32 (1 ulp; the output is bf16)
                    }
                    *(u32x2*)(smem + YOFF + pr * 256 + (((cl >> 3) ^ (pr & 15)) << 4) + (cl & 7) * 2) =
                        u32x2{pack_h2(v[0], v[1]), pack_h2(v[2], v[3])};
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        constexpr int TOT = BM * 16;
        const int oyo = taps.ry, oxo = taps.rx;
        auto store_tile = [&](T* __restrict__ dstp, int lds_off) {
#pragma unroll
            for (int c = tid; c < TOT; c += NTHREADS) {
                const int pr = c >> 4, cc = c & 15;
                const int qy = ty * a.TH + (pr >> a.tw_shift), qx = tx * a.TW + (pr & (a.TW - 1));
                if (qy < a.QH && qx < a.QW) {
                    const int oy = qy * a.out_step + oyo, ox = qx * a.out_step + oxo;
                    const int64_t o = (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.y_ps + a.y_co + cc * 8;
                    *(u32x4*)(dstp + o) = *(const u32x4*)(smem + lds_off + pr * 256 + ((cc ^ (pr & 15)) << 4));
                }
            }
        };
        if (a.y_pre) {
            // training: the squared tile is dead (every wave is past its GDN MFMAs), so v goes out through its place
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int pr = wn * (BM / WN) + j * 32 + frow;
                        *(u32x2*)(smem + pr * 256 + (((cl >> 3) ^ (pr & 15)) << 4) + (cl & 7) * 2) =
                            u32x2{pack_h2(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_h2(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                    }
                }
        }
        store_tile((T*)a.y, YOFF);
        if (a.y_pre) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            store_tile((T*)a.y_pre, 0);
        }
    }
}

// ------------------------------------------------------------------------- transposed stride-2 conv: the four output phases in ONE block
// deconv() (compressai/models/utils.py:112-118) on the big maps of g_s (ywz/mywork/newnet1.py:603-624) and the data gradient of the
// stride-2 analysis convs.  igemm_glds_kernel runs a transposed layer as 4 x tiles blocks of 18 / 12 / 12 / 8 K stages (9 / 6 / 6 / 4 taps
// x 2 channel chunks): per block ~6 us of fixed cost (launch, tile decode, pipeline fill, epilogue, the store drain in front of the exit)
// on ~15 us of K loop.  Here a block keeps its 128 q-pixels x 128 couts and walks the phases in turn as ONE 50-stage pipeline:
//   * the LDS-DMA cursor runs ahead of the matrix cores ACROSS the phase boundary: stage 0 of phase p+1 is requested in the last stage
//     of phase p and lands under p's epilogue;
//   * the epilogue lives in the ring buffer the last stage of the phase was read from (32 KB = 128 pixels x 256 B; the (I)GDN form
//     reuses it for squares, then output), the other buffer belongs to the DMA in flight -- 64 KB per block, two blocks per CU as before;
//   * output rows leave through buffer stores (out-of-range pixels: dropped by the address check, always 8 stores per lane) and the
//     first stage of the next phase waits with a COUNTED vmcnt: the stores drain under the next K loop, not in front of it;
//   * the per-row DMA offsets do not depend on the phase (same q-tile), only the tap displacement does.
// Summation order per output value is the one of igemm_glds_kernel (taps in raster order, channel chunk innermost): bit-identical.
// Ablation builds (profiles/scripts/tr4_ablation.sh; never in the shipped libraries): -DTR4_ABL=<bits>  1: no LDS-DMA instructions, 2: no MFMAs in
// the K loop (the fragment reads stay), 4: no fragment reads (MFMAs on whatever the registers hold), 8: no epilogue (accumulators kept alive, nothing
// squared / contracted / stored).  Compile-time, so the K loop stays one basic block in every variant.
#ifndef TR4_ABL
#define TR4_ABL 0
#endif
#ifndef TR4_HALO
#define TR4_HALO 1
#endif
// HALO = 1 (round 6; Cin = 128, 5 x 5, pad 2, q-tile 8 x 16): the X operand is not staged per (tap, channel chunk) -- 25 taps x 2 chunks x 16 KB
// per block, 9 distinct shifts of ONE 10 x 18 input patch -- but ONCE per block, as that patch with all 128 channels (180 pixels x 256 B = 45 KB,
// through registers), and the fragment reads of a tap address it at the tap's shift; the ring then carries weight tiles only (2 x 16 KB),
// so a block moves 0.85 MB through the global -> LDS path instead of 1.6 MB.  LDS per block 45 + 32 KB (+ beta'): still two blocks per CU, and the
// epilogue tile (32 KB) takes BOTH ring buffers -- the next phase's first weight tile is requested behind the epilogue's last LDS read instead
// of under it.  16-byte slot s of halo pixel (hy, hx) sits at position s ^ (hx & 15): a b128 fragment read's 16-lane groups ({0-3, 12-15} of one
// tile row + {4-11} of the next) then cover the 64 banks exactly once at every shift.  Same taps, same order, same chunks: bit-identical.
template <int GDN, int HALO = 0>
__global__ __launch_bounds__(256, 2) void igemm_tr4_kernel(const IgemmArgs a) {
    using T = h16_t;
    constexpr int BM = 128, BN = 128, BK = 64, NW = 4, NT = NW * 64;
    constexpr int CPR = BK * 2 / 16, RPB = 256 / (BK * 2);
    constexpr int XT = HALO ? 0 : BM * BK * 2, WT = BN * BK * 2, STAGE = XT + WT;
    constexpr int XI = HALO ? 0 : BM * CPR / 64 / NW, WI = BN * CPR / 64 / NW;
    constexpr int HW_ = 18, HPIX = 10 * HW_, XH_OFF = 2 * WT, XH_BYTES = HALO ? HPIX * 256 : 0;      // halo: 10 rows x 18 pixels x 128 channels
    constexpr int WM = 2, MI = BN / WM / 32, NI = BM / 2 / 32;
    constexpr int NST = BM * 16 / NT;                         // 16-byte stores per lane and output tile
    static_assert((HALO ? 2 * STAGE : STAGE) == BM * 256 && NST == 8, "the epilogue tile takes exactly one stage buffer (HALO: both weight buffers)");
    constexpr int BETA_OFF = 2 * STAGE + XH_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[BETA_OFF + (GDN ? 512 : 0)];   // ring + (halo) + beta' (fp32 [128])

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (GDN && tid < 128) *(float*)(smem + BETA_OFF + tid * 4) = a.gdn_beta[tid];     // read behind the K loop's barriers
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    uint32_t rest = fdiv((uint32_t)bid, a.fd_nt);
    const int nt = bid - (int)rest * a.n_tiles;
    uint32_t q_ = fdiv(rest, a.fd_tx);
    const int tx = (int)(rest - q_ * (uint32_t)a.tiles_x);
    rest = q_; q_ = fdiv(rest, a.fd_ty);
    const int ty = (int)(rest - q_ * (uint32_t)a.tiles_y);
    const int b = (int)q_;
    const int n0 = nt * BN;
    const int kchunks = a.Cin / BK;
    const T* __restrict__ xg = (const T*)a.x;

    const int wm = wave % WM, wn = wave / WM;
    const int frow = lane & 31, fh = lane >> 5;
    auto off = [&](int row, int slot) { return (row * CPR + (slot ^ ((row / RPB) & (CPR - 1)))) * 16; };

    // ---- producer: buffer-addressed LDS-DMA, as in igemm_glds_kernel (q-grid step 1: in_step == 1)
    constexpr uint32_t OOB = 0x80000000u;
    asm volatile("" ::"v"((__attribute__((address_space(3))) unsigned char*)smem) : "memory");
    const int neg = (a.KH * a.W + a.KW) * a.x_ps;
    const int row0 = ty * a.TH;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(xg + ((int64_t)b * a.H + row0) * a.W * a.x_ps - neg), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)OOB, 0x00020000);
    const int prow = lane / CPR, pslot = lane % CPR;
    uint32_t xoff[XI ? XI : 1], xv[XI ? XI : 1], wv[WI];
    int iy0[XI ? XI : 1], ix0[XI ? XI : 1];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = (wave * XI + i) * (64 / CPR) + prow;
        const int ls = pslot ^ ((row / RPB) & (CPR - 1));
        const int qy = ty * a.TH + (row >> a.tw_shift), qx = tx * a.TW + (row & (a.TW - 1));
        const bool ok = qy < a.QH && qx < a.QW;
        iy0[i] = ok ? qy : (int)0xc0000000;
        ix0[i] = qx;
        xoff[i] = (uint32_t)((((qy - row0) * a.W + qx) * a.x_ps + a.x_co + ls * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = (wave * WI + i) * (64 / CPR) + prow;
        const int ls = pslot ^ ((row / RPB) & (CPR - 1));
        wv[i] = (uint32_t)(((n0 + row) * a.Cin + ls * 8) * 2);
    }
    if constexpr (HALO != 0) {
        // the input patch of the tile, once, through registers (16 bytes per lane and step: slot e & 15 of halo pixel e >> 4): 12 loads per lane
        // in flight together, then 12 LDS writes.  (As 45 LDS-DMA instructions per block -- destinations up to 155 KB into the CU's LDS for the second
        // co-resident block, which no other kernel here has -- it is as correct and as fast: 152 - 162 vs 159 us, same box, alternating.)
        constexpr int HSTEPS = (HPIX * 16 + NT - 1) / NT;
        u32x4 hv[HSTEPS];
#pragma unroll
        for (int i = 0; i < HSTEPS; ++i) {
            const int e = i * NT + tid, hp = e >> 4, hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = row0 - 1 + hy, ix = tx * a.TW - 1 + hx;
            const int ls = (e & 15) ^ (hx & 15);
            const bool ok = hp < HPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const uint32_t vo = ok ? (uint32_t)(((((hy - 1) * a.W + ix) * a.x_ps + a.x_co + ls * 8) + neg) * 2) : OOB;
            hv[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)vo, 0, 0);          // out of range: zeros
        }
#pragma unroll
        for (int i = 0; i < HSTEPS; ++i) {
            const int e = i * NT + tid;
            if (e < HPIX * 16) *(u32x4*)(smem + XH_OFF + e * 16) = hv[i];
        }
    }
    // consumer side of the halo: this lane's pixels of the tile (q-tile 8 x 16: pixel p = (p >> 4, p & 15)) as halo addresses at shift (0, 0)
    [[maybe_unused]] int hbase[BM / 2 / 32], hsw[BM / 2 / 32];
    if constexpr (HALO != 0) {
#pragma unroll
        for (int j = 0; j < BM / 2 / 32; ++j) {
            const int pq = (wave / 2) * (BM / 2) + j * 32 + (lane & 31);
            hbase[j] = XH_OFF + (((pq >> 4) + 1) * HW_ + (pq & 15) + 1) * 256;
            hsw[j] = (pq & 15) + 1;
        }
    }
    // phase (ry, rx) = (ph >> 1, ph & 1): taps k = k0 + 2 j with k0 = (r + pad) & 1, input displacement (r + pad - k) / 2 = d0 - j
    auto phase_geo = [&](int ph, int& ky0, int& kx0, int& nky, int& nkx, int& dyb, int& dxb) {
        const int ry = ph >> 1, rx = ph & 1;
        ky0 = (ry + a.pad) & 1; kx0 = (rx + a.pad) & 1;
        nky = (a.KH - ky0 + 1) >> 1; nkx = (a.KW - kx0 + 1) >> 1;
        dyb = (ry + a.pad - ky0) >> 1; dxb = (rx + a.pad - kx0) >> 1;
    };
    const uint32_t wtap = (uint32_t)(a.Cout * a.Cin * 2);
    int p_ph = 0, p_ky0, p_kx0, p_nky, p_nkx, p_dyb, p_dxb;
    phase_geo(0, p_ky0, p_kx0, p_nky, p_nkx, p_dyb, p_dxb);
    int cur_j = 0, cur_c = 0, cur_chunk = 0, dy = p_dyb, dx = p_dxb;
    uint32_t s_x = 0, s_w = 0;
    auto set_tap = [&]() {
        s_x = (uint32_t)(((dy * a.W + dx) * a.x_ps + neg) * 2);
        s_w = (uint32_t)((p_ky0 + cur_j * 2) * a.KW + p_kx0 + cur_c * 2) * wtap;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const bool ok = (unsigned)(iy0[i] + dy) < (unsigned)a.H && (unsigned)(ix0[i] + dx) < (unsigned)a.W;
            xv[i] = ok ? xoff[i] : OOB;
        }
    };
    set_tap();
    auto issue = [&](int buf) {
        const uint32_t sx = s_x + (uint32_t)(cur_chunk * BK * 2), sw = s_w + (uint32_t)(cur_chunk * BK * 2);
        unsigned char* xs = smem + buf * STAGE;
        unsigned char* ws = xs + XT;
        (void)sx;
        if constexpr (!(TR4_ABL & 1)) {
#pragma unroll
            for (int i = 0; i < XI; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(xs + (wave * XI + i) * 1024), 16, (int)xv[i], (int)sx, 0, 0);
#pragma unroll
            for (int i = 0; i < WI; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(ws + (wave * WI + i) * 1024), 16, (int)wv[i], (int)sw, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < XI; ++i) asm volatile("" ::"v"(xv[i]), "s"(sx), "v"(xs));
#pragma unroll
            for (int i = 0; i < WI; ++i) asm volatile("" ::"v"(wv[i]), "s"(sw), "v"(ws));
        }
        if (++cur_chunk == kchunks) {
            cur_chunk = 0;
            dx -= 1;
            if (++cur_c == p_nkx) {
                cur_c = 0; dx = p_dxb; ++cur_j; dy -= 1;
                if (cur_j == p_nky) {                                      // the cursor moves on to the next output phase
                    p_ph = (p_ph + 1) & 3;
                    phase_geo(p_ph, p_ky0, p_kx0, p_nky, p_nkx, p_dyb, p_dxb);
                    cur_j = 0; dy = p_dyb; dx = p_dxb;
                }
            }
            set_tap();
        }
    };

    // ---- output side: one buffer resource per image, the phase enters as a scalar offset
    const int Wo = a.Wo;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)a.y + (int64_t)b * a.Ho * Wo * a.y_ps), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t ypr = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)(a.y_pre ? a.y_pre : a.y) + (int64_t)b * a.Ho * Wo * a.y_ps), 0, (int)OOB, 0x00020000);
    // gamma' fragments (fragment order: one 16-byte piece per lane and (cout slice, k-step)): ONE 32-bit lane offset + scalar offsets --
    // as 64-bit global pointers the 16 addresses were hoisted out of the phase loop into 32 registers (and spilled)
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const h16x8*)(GDN ? a.gdn_gamma : a.w) + (GDN ? 128 * 16 : 0)), 0, 128 * 128 * 2, 0x00020000);
    auto store_tile = [&](const __amdgpu_buffer_rsrc_t& rs, const unsigned char* eb, int ry, int rx) {
        int t = tid;
        asm volatile("" : "+v"(t));                                      // keep the address arithmetic inside the epilogue (registers)
        const uint32_t so = (uint32_t)((ry * Wo + rx) * a.y_ps * 2);
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int pr = (t >> 4) + k * (NT / 16), cc = t & 15;
            const int qy = ty * a.TH + (pr >> a.tw_shift), qx = tx * a.TW + (pr & (a.TW - 1));
            // phase offset in the VGPR offset, soffset = 0: see the store-hazard note in store_tile_then_issue below
            const uint32_t vo = (qy < a.QH && qx < a.QW) ? (uint32_t)(((2 * qy * Wo + 2 * qx) * a.y_ps + a.y_co + n0 + cc * 8) * 2) + so : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4*)(eb + pr * 256 + ((cc ^ (pr & 15)) << 4)), rs, (int)vo, 0, 0);
        }
    };

    // HALO: the phase's last store -- tile rows into registers, barrier (every wave is done with the tile = with both weight buffers), the next
    // phase's first weight tile requested, then the stores: the request is older than the NST stores, the next phase waits with vmcnt(NST)
    f32x16 acc[MI][NI];
    int gbuf = 0;
    auto store_tile_then_issue = [&](const __amdgpu_buffer_rsrc_t& rs, const unsigned char* eb, int ry, int rx, bool more) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const uint32_t so = (uint32_t)((ry * Wo + rx) * a.y_ps * 2);
        u32x4 rows[NST];
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int pr = (t >> 4) + k * (NT / 16), cc = t & 15;
            rows[k] = *(const u32x4*)(eb + pr * 256 + ((cc ^ (pr & 15)) << 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (more) issue(gbuf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int pr = (t >> 4) + k * (NT / 16), cc = t & 15;
            const int qy = ty * a.TH + (pr >> a.tw_shift), qx = tx * a.TW + (pr & (a.TW - 1));
            // the phase offset goes INTO the VGPR offset (soffset = 0): with a register soffset the compiler's hazard recognizer lets a VALU
            // write into a 16-byte store's data registers follow the store directly (here: the next store's address, computed into a
            // register of the previous store's data), and under memory-pipe back-pressure -- two blocks per CU -- gfx950 then stores the
            // NEW contents: 4 - 17 % wrong pixels, different on every launch, never in the last store of a lane (see also sconv.hip)
            const uint32_t vo = (qy < a.QH && qx < a.QW) ? (uint32_t)(((2 * qy * Wo + 2 * qx) * a.y_ps + a.y_co + n0 + cc * 8) * 2) + so : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(rows[k], rs, (int)vo, 0, 0);
        }
    };
    issue(0);
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
        int c_ky0, c_kx0, c_nky, c_nkx, c_dyb, c_dxb;
        phase_geo(ph, c_ky0, c_kx0, c_nky, c_nkx, c_dyb, c_dxb);
        const int nsteps = c_nky * c_nkx * kchunks;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        auto stage = [&](auto last_tag, int step) {
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr int KS = BK / 16;
            __builtin_amdgcn_s_barrier();
            const unsigned char* xs = smem + gbuf * STAGE;
            const unsigned char* ws = xs + XT;
            gbuf ^= 1;
            h16x8 wf[2][MI], xf[2][NI];
            [[maybe_unused]] int hx_[NI];
            [[maybe_unused]] int hchunk = 0;
            if constexpr (HALO != 0) {
                // tap of this stage (two channel chunks per tap): its shift moves the lane's halo address and the slot rotation
                const int tap = step >> 1, tj = (tap >= c_nkx) + (tap >= 2 * c_nkx), tc = tap - tj * c_nkx;
                const int dy_ = c_dyb - tj, dx_ = c_dxb - tc;
                hchunk = (step & 1) * 128;
#pragma unroll
                for (int j = 0; j < NI; ++j) hx_[j] = (hbase[j] + (dy_ * HW_ + dx_) * 256) | (((fh ^ (hsw[j] + dx_)) & 15) << 4);
            }
            auto ldf = [&](int set, int ks) {
                if constexpr (HALO != 0) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) wf[set][i] = *(const h16x8*)(ws + off(wm * (BN / WM) + i * 32 + frow, ks * 2 + fh));
#pragma unroll
                    for (int j = 0; j < NI; ++j) xf[set][j] = *(const h16x8*)(smem + (hx_[j] ^ (hchunk + ks * 32)));
                    return;
                }
                if constexpr (TR4_ABL & 4) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) asm volatile("" : "=v"(wf[set][i]) : "v"(ws));
#pragma unroll
                    for (int j = 0; j < NI; ++j) asm volatile("" : "=v"(xf[set][j]) : "v"(xs));
                    return;
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) wf[set][i] = *(const h16x8*)(ws + off(wm * (BN / WM) + i * 32 + frow, ks * 2 + fh));
#pragma unroll
                for (int j = 0; j < NI; ++j) xf[set][j] = *(const h16x8*)(xs + off(wn * (BM / 2) + j * 32 + frow, ks * 2 + fh));
            };
            ldf(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!LAST || (ph < 3 && !HALO)) issue(gbuf);           // LAST: stage 0 of the next phase, into the buffer the epilogue does not use
                                                                   // (HALO: the epilogue takes both weight buffers -- requested behind it)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) ldf((ks + 1) & 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TR4_ABL & 2) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(wf[ks & 1][i]));
#pragma unroll
                    for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(xf[ks & 1][j]));
                } else {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = mfma_32x32x16_h16(wf[ks & 1][i], xf[ks & 1][j], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // first stage of a later phase: everything but the previous epilogue's stores (the newest vector-memory operations of this wave)
        // has to be back; the lgkmcnt part covers that epilogue's last LDS reads before the barrier hands its buffer to the DMA
        if (ph == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (lgkmcnt: HALO's patch, written by ds_write)
        // HALO: the weight tile was requested just in front of the phase's last NST stores: everything older than those stores has to be back
        // (counted, like the unfused form's wait below; vmcnt(0) measured the same -- the stores are as old as the request)
        else if constexpr (HALO != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        else if (GDN && a.y_pre) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST) : "memory");
        for (int step = 0; step < nsteps - 1; ++step) {
            stage(std::false_type{}, step);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // (I)GDN: the gamma' fragments of the epilogue are requested in front of the phase's last stage -- an L2 round trip that would
        // otherwise sit between the two barriers of the epilogue (and, the counter being in-order, drag the next phase's first DMA with it)
        // (first cout half only: all of it costs 16 spilled registers; the second half is requested at the top of the epilogue and is
        // not needed before the first half's contraction has been issued)
        h16x8 gq[GDN ? MI : 1][8];
        if constexpr (GDN != 0) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                gq[0][ks] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(gr, lane * 16, ((wm * MI + 0) * 8 + ks) * 1024, 0));
        }
        stage(std::true_type{}, nsteps - 1);

        // ---- epilogue in the buffer of the stage just computed (HALO: in both weight buffers)
        unsigned char* eb = HALO ? smem : smem + (gbuf ^ 1) * STAGE;
        const int ry = ph >> 1, rx = ph & 1;
        if constexpr ((TR4_ABL & 8) != 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
            if constexpr (GDN != 0) asm volatile("" ::"v"(gq[0][0]), "v"(gq[0][7]));
#pragma unroll
            for (int k = 0; k < NST; ++k) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, yr, (int)OOB, 0, 0);      // keeps the counted vmcnt of the next phase right
            if (GDN && a.y_pre)
#pragma unroll
                for (int k = 0; k < NST; ++k) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, yr, (int)OOB, 0, 0);
            (void)eb; (void)ry; (void)rx;
        } else if constexpr (GDN == 0) {
            const int act_eff = a.act;
            int fr_ = frow, fh_ = fh;
            asm volatile("" : "+v"(fr_), "+v"(fh_));
            asm volatile("s_barrier" ::: "memory");                        // every wave is done reading the ring buffer
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh_;
                    const f32x4 bq = a.bias ? *(const f32x4*)(a.bias + n0 + cl) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int pr = wn * (BM / 2) + j * 32 + fr_;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][j][4 * g + e] + bq[e], act_eff);
                        *(u32x2*)(eb + pr * 256 + (((cl >> 3) ^ (pr & 15)) << 4) + (cl & 7) * 2) = u32x2{pack_h2(v[0], v[1]), pack_h2(v[2], v[3])};
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (HALO != 0) store_tile_then_issue(yr, eb, ry, rx, ph < 3);
            else store_tile(yr, eb, ry, rx);
        } else {
            // fused (I)GDN (compressai/layers/gdn.py:55-70), the data flow of igemm_glds_kernel's epilogue in ONE 32 KB tile: squares in,
            // contraction on the matrix cores one cout half at a time (32 norm registers live instead of 64: the DMA cursor's state
            // stays in registers through the epilogue), the product in place in the accumulators, barrier, output over the squares
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh;
                    const f32x4 t = a.bias ? *(const f32x4*)(a.bias + cl) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[i][j][4 * g + e] += t[e];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 1; i < MI; ++i)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    gq[i][ks] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(gr, lane * 16, ((wm * MI + i) * 8 + ks) * 1024, 0));
            __builtin_amdgcn_sched_barrier(0);
            auto put_tile = [&](auto sq_tag) {
                constexpr bool SQ = decltype(sq_tag)::value;
                int fr_ = frow, fh_ = fh;
                asm volatile("" : "+v"(fr_), "+v"(fh_));               // the 16 swizzled addresses are rebuilt here, not carried through the K loop
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh_;
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const int pr = wn * (BM / 2) + j * 32 + fr_;
                            const float v0 = acc[i][j][4 * g], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                            *(u32x2*)(eb + pr * 256 + (((cl >> 3) ^ (pr & 15)) << 4) + (cl & 7) * 2) =
                                SQ ? u32x2{pack_sq2(v0, v1), pack_sq2(v2, v3)} : u32x2{pack_h2(v0, v1), pack_h2(v2, v3)};
                        }
                    }
            };
            asm volatile("s_barrier" ::: "memory");                        // every wave is done reading the ring buffer
            if (a.y_pre) {
                // training: v = conv + bias for GDN's backward leaves first (the accumulators turn into the output below)
                put_tile(std::false_type{});
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                store_tile(ypr, eb, ry, rx);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            put_tile(std::true_type{});
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // squares visible to every wave
            int fr2 = frow, fh2 = fh;
            asm volatile("" : "+v"(fr2), "+v"(fh2));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                f32x16 nrm[NI];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = wm * (BN / WM) + i * 32 + 8 * g + 4 * fh2;
                    const f32x4 be = *(const f32x4*)(smem + BETA_OFF + cl * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NI; ++j) nrm[j][4 * g + e] = be[e];
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    h16x8 qf[NI];
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int pr = wn * (BM / 2) + j * 32 + fr2;
                        qf[j] = *(const h16x8*)(eb + pr * 256 + (((ks * 2 + fh2) ^ (pr & 15)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) nrm[j] = mfma_32x32x16_h16(gq[i][ks], qf[j], nrm[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= (GDN == 2 ? __builtin_amdgcn_sqrtf(nrm[j][r]) : rsqrtf(nrm[j][r]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has its square fragments: the tile may be overwritten
            put_tile(std::false_type{});
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (HALO != 0) store_tile_then_issue(yr, eb, ry, rx, ph < 3);
            else store_tile(yr, eb, ry, rx);
        }
    }
}

// y = act(sum of the K-slice partials + bias) as bf16 (y) and / or fp32 (y32): one thread per pixel and 8 channels
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int nslice, int64_t npix, int Cout,
                                                            const float* __restrict__ bias, int act, h16_t* __restrict__ y,
                                                            int y_ps, int y_co, float* __restrict__ y32, int y32_ps, int y32_co,
                                                            int y_hilo, int y_abs) {
    const int cg = Cout >> 3;
    const int64_t total = npix * cg, slice = npix * Cout;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / cg;
        const int c = (int)(i - p * cg) * 8;
        f32x4 lo = bias ? *(const f32x4*)(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 hi = bias ? *(const f32x4*)(bias + c + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float* src = ws + p * Cout + c;
        for (int s = 0; s < nslice; ++s) {
            lo += *(const f32x4*)(src + s * slice);
            hi += *(const f32x4*)(src + s * slice + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = apply_act(lo[e], act); hi[e] = apply_act(hi[e], act); }
        if (y32) {
            *(f32x4*)(y32 + p * y32_ps + y32_co + c) = lo;
            *(f32x4*)(y32 + p * y32_ps + y32_co + c + 4) = hi;
        }
        if (y) {
            if (y_abs) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = fabsf(lo[e]); hi[e] = fabsf(hi[e]); }
            }
            const u32x4 o = u32x4{pack_h2(lo[0], lo[1]), pack_h2(lo[2], lo[3]), pack_h2(hi[0], hi[1]), pack_h2(hi[2], hi[3])};
            *(u32x4*)(y + p * y_ps + y_co + c) = o;
            if (y_hilo)       // second half of a hi/lo pair map: bf16(v - bf16(v)) at channel offset Cout
                *(u32x4*)(y + p * y_ps + y_co + Cout + c) =
                    u32x4{pack_h2(lo[0] - h2f_lo(o.x), lo[1] - h2f_hi(o.x)),
                          pack_h2(lo[2] - h2f_lo(o.y), lo[3] - h2f_hi(o.y)),
                          pack_h2(hi[0] - h2f_lo(o.z), hi[1] - h2f_hi(o.z)),
                          pack_h2(hi[2] - h2f_lo(o.w), hi[3] - h2f_hi(o.w))};
        }
    }
}

// ------------------------------------------------------------------------- weight packing
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, const float* __restrict__ mask, T* __restrict__ wp,
                                   int Cout, int Cin, int KH, int KW, int transposed, int flip) {
    const int64_t n = (int64_t)KH * KW * Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = i % Cin;
        int64_t r = i / Cin;
        const int co = r % Cout;
        const int tap = r / Cout;
        int ky = tap / KW, kx = tap % KW;
        if (flip) { ky = KH - 1 - ky; kx = KW - 1 - kx; }
        const int64_t src = transposed ? (((int64_t)ci * Cout + co) * KH + ky) * KW + kx
                                       : (((int64_t)co * Cin + ci) * KH + ky) * KW + kx;
        float v = w[src];
        if (mask) v *= mask[src];
        elem<T>::st(wp + i, v);
    }
}

// 16-bit packing with ERROR FEEDBACK over the taps of each (cout, cin) pair (inference, Conv2d).  Rounding every weight on its own leaves
// 25 independent errors per pair; here the rounding error of a tap is carried into the next one of a serpentine walk over the KH x KW
// window (consecutive taps = neighbouring input pixels), so the errors of a pair sum to less than ONE half-ulp.  What that buys: a feature
// map is spatially smooth over a 5 x 5 window, so sum_t dW_t x_t ~ x * sum_t dW_t -- the weight-rounding part of a one-product layer's
// error nearly vanishes (CPU study profiles/scripts/precision_study.py, g_a_conv2 on single fp16 operands, 512^2: latent flips 6.3e-4 ->
// 3.9e-4 against 3.7e-4 with EXACT weights; on white-noise inputs it would cost sqrt(2)).  Same storage, same kernels, chosen at pack time.
__global__ void pack_weight_shaped_kernel(const float* __restrict__ w, h16_t* __restrict__ wp, int Cout, int Cin, int KH, int KW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (co, ci), ci fastest: coalesced 2-byte stores per tap
    if (i >= Cout * Cin) return;
    const int ci = i % Cin, co = i / Cin;
    const float* src = w + (int64_t)i * KH * KW;
    float e = 0.f;
    for (int ky = 0; ky < KH; ++ky)
        for (int j = 0; j < KW; ++j) {
            const int kx = (ky & 1) ? KW - 1 - j : j;
            const float tgt = src[ky * KW + kx] + e;
            const h16_t q = f2h(tgt);
            e = tgt - h2f(q);
            wp[((int64_t)(ky * KW + kx) * Cout + co) * Cin + ci] = q;
        }
}

// The same for a ConvTranspose2d weight (Cin, Cout, KH, KW) of stride s (round 5: the synthesis stacks): an output pixel of phase (py, px)
// sums only the taps with (ky % s, kx % s) = phase over neighbouring INPUT pixels, so the error is fed back inside each phase's tap
// class (9 / 6 / 6 / 4 taps for 5 x 5, stride 2), serpentine inside the class.  Measured at a trained 31 dB operating point
// (profiles/scripts/synthesis_precision.py): plain rounding of the four synthesis layers' weights moves the PSNR by +-(2 - 9)e-4 dB per
// layer -- all of the default mode's residual deviation, against a 1e-3 bar.
__global__ void pack_weight_shaped_tr_kernel(const float* __restrict__ w, h16_t* __restrict__ wp, int Cout, int Cin, int KH, int KW, int s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (co, ci), ci fastest: coalesced 2-byte stores per tap
    if (i >= Cout * Cin) return;
    const int ci = i % Cin, co = i / Cin;
    const float* src = w + ((int64_t)ci * Cout + co) * KH * KW;
    for (int cy = 0; cy < s; ++cy)
        for (int cx = 0; cx < s; ++cx) {
            const int nx = (KW - cx + s - 1) / s;
            float e = 0.f;
            int row = 0;
            for (int ky = cy; ky < KH; ky += s, ++row)
                for (int j = 0; j < nx; ++j) {
                    const int kx = cx + s * ((row & 1) ? nx - 1 - j : j);
                    const float tgt = src[ky * KW + kx] + e;
                    const h16_t q = f2h(tgt);
                    e = tgt - h2f(q);
                    wp[((int64_t)(ky * KW + kx) * Cout + co) * Cin + ci] = q;
                }
        }
}

// All weight repacks of a training step in ONE launch (an eager step issued 68 of them, ~7 us each).  Block -> job by a
// search over the jobs' first-block table; a block moves a tile of 8 couts x 32 cins x all taps through LDS so that both
// sides are wide: the source is read in runs of 32*taps (conv) or 8*taps (transposed conv) consecutive floats, the
// destination written in 64-byte runs of 32 cins.  (Element-wise gathering read the fp32 source with a 100-byte stride:
// 250 us for the 18 M elements of a HESIC step.)
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const hesic_pack_job* __restrict__ jobs, int n_jobs) {
    __shared__ float tile[256 * MAX_TAPS];
    int lo = 0, hi = n_jobs - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {                       // last job whose first block is <= bid (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= bid) lo = mid; else hi = mid - 1;
    }
    const hesic_pack_job j = jobs[lo];
    const int khw = j.KH * j.KW;
    const int tiles_ci = (j.Cin + 31) / 32;
    const int lb = bid - j.block0;
    const int co0 = (lb / tiles_ci) * 8, ci0 = (lb % tiles_ci) * 32;
    const int run = (j.transposed ? 8 : 32) * khw;              // consecutive source floats per outer index
    const int n_outer = j.transposed ? 32 : 8;
    // (outer index, position inside its run of consecutive source floats): two nested loops, no division per element -- the flat form
    // `e / run`, `rem / khw` with run-time divisors spent more issue cycles on the index split than on the move (172 us per step)
    // Loads first, LDS stores after, sixteen loads in flight per lane: written as load-store pairs in a loop with run-time bounds the
    // compiler issued them one by one, ~25 dependent HBM round trips per block (172 us per step for 108 MB).
    auto row = [&](int ol, int64_t& base, int64_t& lim) {
        const int co_o = j.transposed ? co0 : co0 + ol, ci_o = j.transposed ? ci0 + ol : ci0;
        const bool outer_ok = j.transposed ? ci_o < j.Cin : co_o < j.Cout;
        base = j.transposed ? ((int64_t)ci_o * j.Cout + co0) * khw : ((int64_t)co_o * j.Cin + ci0) * khw;
        lim = outer_ok ? (j.transposed ? (int64_t)j.Cout - co0 : (int64_t)j.Cin - ci0) * khw : 0;       // floats of this run that exist
    };
    if (j.transposed) {                     // 32 runs of 8 * khw <= 200 floats: one load per lane and run
        for (int o8 = 0; o8 < 32; o8 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                int64_t base, lim;
                row(o8 + u, base, lim);
                const int rem = threadIdx.x;
                v[u] = 0.f;
                if (rem < run && rem < lim) {
                    v[u] = j.w[base + rem];
                    if (j.mask) v[u] *= j.mask[base + rem];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if ((int)threadIdx.x < run) tile[(o8 + u) * run + threadIdx.x] = v[u];
        }
    } else {                                // 8 runs of 32 * khw <= 800 floats: four loads per lane and run, four runs at a time
        for (int o2 = 0; o2 < 8; o2 += 4) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t base, lim;
                row(o2 + u, base, lim);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rem = threadIdx.x + 256 * k;
                    v[u][k] = 0.f;
                    if (rem < run && rem < lim) {
                        v[u][k] = j.w[base + rem];
                        if (j.mask) v[u][k] *= j.mask[base + rem];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((int)threadIdx.x + 256 * k < run) tile[(o2 + u) * run + threadIdx.x + 256 * k] = v[u][k];
        }
    }
    __syncthreads();
    if (j.dtype == HESIC_H16 && (j.Cin & 7) == 0) {
        // bf16 destination: 8 cins = one 16-byte store per lane (2-byte stores moved 128 bytes per wave instruction: 175 us per step)
        for (int o = threadIdx.x; o < 32 * khw; o += 256) {
            const int t = o >> 5, cl = (o >> 2) & 7, il = (o & 3) * 8;    // tap, cout in tile, first of 8 cins in tile
            const int co = co0 + cl, ci = ci0 + il;
            if (co >= j.Cout || ci >= j.Cin) continue;
            const int ts = j.flip ? khw - 1 - t : t;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = j.transposed ? tile[((il + e) * 8 + cl) * khw + ts] : tile[(cl * 32 + il + e) * khw + ts];
            *(u32x4*)((h16_t*)j.w_packed + ((int64_t)t * j.Cout + co) * j.Cin + ci) =
                u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
        }
        return;
    }
    for (int o = threadIdx.x; o < 256 * khw; o += 256) {
        const int t = o >> 8, cl = (o >> 5) & 7, il = o & 31;    // tap, cout in tile, cin in tile (fastest)
        const int co = co0 + cl, ci = ci0 + il;
        if (co >= j.Cout || ci >= j.Cin) continue;
        const int ts = j.flip ? khw - 1 - t : t;
        const float v = j.transposed ? tile[(il * 8 + cl) * khw + ts] : tile[(cl * 32 + il) * khw + ts];
        const int64_t dst = ((int64_t)t * j.Cout + co) * j.Cin + ci;
        if (j.dtype == HESIC_H16) ((h16_t*)j.w_packed)[dst] = f2h(v);
        else ((float*)j.w_packed)[dst] = v;
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, const float* __restrict__ mask, float* __restrict__ dw,
                                    int Cout, int Cin, int KH, int KW, int transposed) {
    const int64_t n = (int64_t)KH * KW * Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the PyTorch layout so the writes are coalesced
        int64_t r = i;
        const int kx = r % KW; r /= KW;
        const int ky = r % KH; r /= KH;
        int co, ci;
        if (transposed) { co = r % Cout; ci = r / Cout; } else { ci = r % Cin; co = r / Cin; }
        float v = dwp[((int64_t)(ky * KW + kx) * Cout + co) * Cin + ci];
        if (mask) v *= mask[i];
        dw[i] = v;
    }
}

int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

// gamma' = max(gamma, 2^-18)^2 - 2^-36 as bf16, twice: [0, 128*128) in the LDS image order of the image-side fused kernel
// (sconv_n2w_gdn_kernel), [128*128, 2*128*128) in MFMA A-fragment order for igemm_glds_kernel -- fragment (rb, ks) holds,
// for lane l, gamma'[rb*32 + (l & 31)][ks*16 + (l >> 5)*8 + 0..7].  beta' likewise (fp32).
__device__ __forceinline__ void gdn_pack_slot(int i, const float* __restrict__ beta, const float* __restrict__ gamma, float beta_bound,
                                              h16_t* __restrict__ gp, float* __restrict__ bp) {
    if (i >= 128 * 16) return;
    const int row = i >> 4, slot = i & 15;
    const float ped = 1.0f / 68719476736.0f, gb = 1.0f / 262144.0f;
    h16_t* dst = gp + (row * 16 + (slot ^ (row & 15))) * 8;
    h16_t* frag = gp + 128 * 128 + ((((row >> 5) * 8 + (slot >> 1)) * 64) + (slot & 1) * 32 + (row & 31)) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = fmaxf(gamma[row * 128 + slot * 8 + e], gb);
        dst[e] = frag[e] = f2h((t * t - ped) * H16_SQ_UNSCALE);
    }
    if (i < 128) {
        const float t = fmaxf(beta[i], beta_bound);
        bp[i] = t * t - ped;
    }
}

__global__ void gdn_pack_kernel(const float* __restrict__ beta, const float* __restrict__ gamma, float beta_bound,
                                h16_t* __restrict__ gp, float* __restrict__ bp) {
    gdn_pack_slot(blockIdx.x * blockDim.x + threadIdx.x, beta, gamma, beta_bound, gp, bp);      // one 16-byte slot (8 values) per thread: 128 rows x 16 slots
}

// every GDN of a model in one launch (the training step's repack behind the optimiser update): 8 blocks per job
__global__ void gdn_pack_batched_kernel(const hesic_gdn_pack_job* __restrict__ jobs) {
    const hesic_gdn_pack_job j = jobs[blockIdx.x >> 3];
    gdn_pack_slot((blockIdx.x & 7) * blockDim.x + threadIdx.x, j.beta, j.gamma, sqrtf(j.beta_min + 1.0f / 68719476736.0f), (h16_t*)j.gamma_packed, j.beta_packed);
}

// lo half of gamma' for the hi/lo GDN epilogue, MFMA A-fragment order (as the second half of gdn_pack_kernel's buffer):
// gamma'_lo = bf16(gamma' - float(bf16(gamma'))), so gamma'_hi + gamma'_lo = gamma' to 2^-17
__global__ void gdn_pack_lo_kernel(const float* __restrict__ gamma, h16_t* __restrict__ glo) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 128 * 16) return;
    const int row = i >> 4, slot = i & 15;
    const float ped = 1.0f / 68719476736.0f, gb = 1.0f / 262144.0f;
    h16_t* frag = glo + ((((row >> 5) * 8 + (slot >> 1)) * 64) + (slot & 1) * 32 + (row & 31)) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = fmaxf(gamma[row * 128 + slot * 8 + e], gb);
        const float g = (t * t - ped) * H16_SQ_UNSCALE;
        frag[e] = f2h(g - h2f(f2h(g)));
    }
}

}  // namespace

// ------------------------------------------------------------------------------- C ABI
extern "C" int hesic_pack_conv_weight(const float* w, const float* mask, void* wp, int Cout, int Cin, int KH, int KW,
                                      int transposed, int flip, int dtype, void* stream) {
    HESIC_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "pack_conv_weight: bad arguments");
    const int64_t n = (int64_t)KH * KW * Cout * Cin;
    const int g = grid_for(n, 256);
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(pack_weight_kernel<h16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, w, mask, (h16_t*)wp,
                           Cout, Cin, KH, KW, transposed, flip);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, w, mask, (float*)wp,
                           Cout, Cin, KH, KW, transposed, flip);
    HESIC_LAUNCH_RETURN("pack_conv_weight");
}

extern "C" int hesic_pack_conv_weight_shaped(const float* w, void* wp, int Cout, int Cin, int KH, int KW, void* stream) {
    HESIC_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && (int64_t)Cout * Cin < (1ll << 31), "pack_conv_weight_shaped: bad arguments");
    hipLaunchKernelGGL(pack_weight_shaped_kernel, dim3((unsigned)((Cout * Cin + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (h16_t*)wp, Cout, Cin, KH, KW);
    HESIC_LAUNCH_RETURN("pack_conv_weight_shaped");
}

extern "C" int hesic_pack_conv_weight_shaped_tr(const float* w, void* wp, int Cout, int Cin, int KH, int KW, int stride, void* stream) {
    HESIC_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && stride >= 1 && (int64_t)Cout * Cin < (1ll << 31), "pack_conv_weight_shaped_tr: bad arguments");
    hipLaunchKernelGGL(pack_weight_shaped_tr_kernel, dim3((unsigned)((Cout * Cin + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (h16_t*)wp, Cout, Cin, KH, KW, stride);
    HESIC_LAUNCH_RETURN("pack_conv_weight_shaped_tr");
}

extern "C" int hesic_pack_conv_weights_batched(const hesic_pack_job* jobs_device, int n_jobs, int total_blocks, void* stream) {
    HESIC_CHECK_ARG(jobs_device && n_jobs > 0 && total_blocks > 0, "pack_conv_weights_batched: bad arguments");
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, n_jobs);
    HESIC_LAUNCH_RETURN("pack_conv_weights_batched");
}

extern "C" int hesic_unpack_conv_wgrad(const float* dwp, const float* mask, float* dw, int Cout, int Cin, int KH, int KW,
                                       int transposed, void* stream) {
    HESIC_CHECK_ARG(dwp && dw, "unpack_conv_wgrad: null pointer");
    const int64_t n = (int64_t)KH * KW * Cout * Cin;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dwp, mask, dw, Cout,
                       Cin, KH, KW, transposed);
    HESIC_LAUNCH_RETURN("unpack_conv_wgrad");
}

static std::atomic<int> g_phase4_mode{getenv("HESIC_IGEMM_PHASE4") ? atoi(getenv("HESIC_IGEMM_PHASE4")) : 1};
extern "C" int hesic_conv2d_set_phase_fusion(int mode) {
    if (mode < 0 || mode > 2) { hesic_set_error("conv2d_set_phase_fusion: mode must be 0 (off), 1 (auto) or 2 (always)"); return -1; }
    return g_phase4_mode.exchange(mode);
}
static thread_local int* g_plan_out = nullptr;
static thread_local const void* g_gdn_gamma = nullptr;   // set by hesic_conv2d_gdn_forward around its call to the launcher
static thread_local const float* g_gdn_beta = nullptr;
static thread_local int g_gdn_mode = 0;
static thread_local void* g_y_pre = nullptr;              // set by hesic_conv2d_gdn_forward_train
static thread_local float* g_ws = nullptr;               // set by hesic_conv2d_forward_ws: split-K workspace
static thread_local size_t g_ws_bytes = 0;
static thread_local size_t* g_ws_need = nullptr;         // set by hesic_conv2d_ws_bytes: only report the workspace size
static thread_local int g_groups = 1, g_x_group_step = 0, g_act2 = 0, g_act_split = 0;   // set by hesic_conv2d_forward_grouped
static thread_local float* g_y32 = nullptr;              // set by hesic_conv2d_forward_f32out: fp32 copy of the output
static thread_local int g_y32_ps = 0, g_y32_co = 0;
static thread_local int g_hilo = 0;                      // set by hesic_conv2d_forward_hilo: bf16x3 operands
static thread_local const void* g_gdn_gamma_lo = nullptr;
static thread_local int g_y_hilo = 0, g_y_abs = 0;
static thread_local float g_hilo_acc_scale = 1.f;      // hesic_conv2d_hilo_set_acc_scale: consumed by the next hi/lo launch of this thread
static thread_local float g_hilo_acc_scale_active = 1.f;      // ... the value that launch runs with (set and cleared by conv2d_forward_hilo_n)

extern "C" int hesic_gdn_pack_params(const float* beta, const float* gamma, float beta_min, void* gamma_packed, float* beta_packed,
                                     int C, void* stream) {
    HESIC_CHECK_ARG(beta && gamma && gamma_packed && beta_packed, "gdn_pack_params: null pointer");
    HESIC_CHECK_ARG(C == 128, "gdn_pack_params: the fused conv+GDN epilogue is built for C == 128 (got %d)", C);
    hipLaunchKernelGGL(gdn_pack_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, beta, gamma, sqrtf(beta_min + 1.0f / 68719476736.0f),
                       (h16_t*)gamma_packed, beta_packed);
    HESIC_LAUNCH_RETURN("gdn_pack_params");
}   // when set, hesic_conv2d_forward only reports its tile choice

extern "C" int hesic_gdn_pack_params_batched(const hesic_gdn_pack_job* jobs_device, int n_jobs, void* stream) {
    HESIC_CHECK_ARG(jobs_device && n_jobs > 0 && n_jobs < (1 << 20), "gdn_pack_params_batched: bad arguments");
    hipLaunchKernelGGL(gdn_pack_batched_kernel, dim3(8 * n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
    HESIC_LAUNCH_RETURN("gdn_pack_params_batched");
}

extern "C" int hesic_gdn_pack_params_lo(const float* gamma, void* gamma_lo_packed, int C, void* stream) {
    HESIC_CHECK_ARG(gamma && gamma_lo_packed, "gdn_pack_params_lo: null pointer");
    HESIC_CHECK_ARG(C == 128, "gdn_pack_params_lo: the fused conv+GDN epilogue is built for C == 128 (got %d)", C);
    hipLaunchKernelGGL(gdn_pack_lo_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, gamma, (h16_t*)gamma_lo_packed);
    HESIC_LAUNCH_RETURN("gdn_pack_params_lo");
}

extern "C" int hesic_conv2d_forward(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                    void* y, void* stream);

extern "C" int hesic_conv2d_variant(const hesic_conv_desc* d, int* bm_bn_bk_glds) {
    HESIC_CHECK_ARG(d && bm_bn_bk_glds, "conv2d_variant: null pointer");
    g_plan_out = bm_bn_bk_glds;
    const int rc = hesic_conv2d_forward(d, (const void*)16, (const void*)16, nullptr, (void*)16, nullptr);
    g_plan_out = nullptr;
    return rc;
}

extern "C" int hesic_conv2d_gdn_forward(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                        const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* stream) {
    HESIC_CHECK_ARG(d && gamma_packed && beta_packed, "conv2d_gdn_forward: null pointer");
    HESIC_CHECK_ARG(d->dtype == HESIC_H16 && d->Cout == 128 && d->act == HESIC_ACT_NONE && d->Cin % 32 == 0,
                    "conv2d_gdn_forward: needs bf16 storage, Cout == 128, Cin %% 32 == 0 and no activation");
    g_gdn_gamma = gamma_packed; g_gdn_beta = beta_packed; g_gdn_mode = inverse ? 2 : 1;
    const int rc = hesic_conv2d_forward(d, x, w_packed, bias, y, stream);
    g_gdn_gamma = nullptr; g_gdn_beta = nullptr; g_gdn_mode = 0;
    return rc;
}

extern "C" int hesic_conv2d_gdn_forward_train(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                              const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* y_pre,
                                              void* stream) {
    HESIC_CHECK_ARG(y_pre, "conv2d_gdn_forward_train: null pointer");
    g_y_pre = y_pre;
    const int rc = hesic_conv2d_gdn_forward(d, x, w_packed, bias, gamma_packed, beta_packed, inverse, y, stream);
    g_y_pre = nullptr;
    return rc;
}

extern "C" int hesic_conv2d_forward(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                    void* y, void* stream) {
    HESIC_CHECK_ARG(d && x && w_packed && (y || g_y32), "conv2d_forward: null pointer");
    HESIC_CHECK_ARG(d->Cin % BK == 0, "conv2d_forward: Cin=%d must be a multiple of %d (use hesic_sconv2d_forward)", d->Cin, BK);
    HESIC_CHECK_ARG(!g_hilo || ((g_gdn_mode == 0 || g_gdn_mode >= 3) && g_groups == 1 && !g_act_split), "conv2d_forward_hilo: plain or hi/lo GDN epilogue, no groups");
    const int hilo = g_hilo;                       // bf16x3 operands: x = [hi | lo] (2 Cin channels), weights [w_hi | w_lo] (2 Cin per tap and cout)
    HESIC_CHECK_ARG(!hilo || d->Cin % 32 == 0, "conv2d_forward_hilo: Cin must be a multiple of 32");
    const int cin_k = hilo ? 2 * d->Cin : d->Cin;  // channels per tap of the packed weights; a stage of BK covers BK/2 logical channels then,
                                                   // so every "Cin % BK" tile-selection rule below applies to cin_k
    const int ce = d->dtype == HESIC_H16 ? 8 : 4;
    HESIC_CHECK_ARG(d->Cout % ce == 0 && d->y_c_off % ce == 0 && d->y_pix_stride % ce == 0 && d->x_c_off % ce == 0 &&
                        d->x_pix_stride % ce == 0,
                    "conv2d_forward: channel counts/offsets must be multiples of %d", ce);
    HESIC_CHECK_ARG(d->KH * d->KW <= MAX_TAPS, "conv2d_forward: at most %d taps", MAX_TAPS);
    HESIC_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv2d_forward: stride must be 1 or 2");
    HESIC_CHECK_ARG(d->dtype == HESIC_H16 || d->dtype == HESIC_F32, "conv2d_forward: bad dtype");
    HESIC_CHECK_ARG(d->x_c_off + (hilo ? 2 : 1) * d->Cin <= d->x_pix_stride && d->y_c_off + ((g_gdn_mode >= 3 || (hilo && g_y_hilo)) ? 2 : 1) * d->Cout <= d->y_pix_stride,
                    "conv2d_forward: channel slice out of range");

    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y;
    a.gdn_gamma = g_gdn_gamma; a.gdn_beta = g_gdn_beta; a.y_pre = g_y_pre; a.gdn_gamma_lo = g_gdn_gamma_lo; a.y_hilo = g_y_hilo; a.y_abs = g_y_abs;
    a.acc_scale = hilo ? g_hilo_acc_scale_active : 1.f;
    a.y32 = g_y32; a.y32_ps = g_y32_ps; a.y32_co = g_y32_co;
    const int gdn = g_gdn_mode;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = cin_k; a.x_ps = d->x_pix_stride; a.x_co = d->x_c_off;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.y_ps = d->y_pix_stride; a.y_co = d->y_c_off;
    a.act = d->act; a.in_abs = d->in_abs;
    const int s = d->stride, p = d->pad;
    a.KH = d->KH; a.KW = d->KW; a.stride = s; a.pad = p; a.transposed = d->transposed;
    if (!d->transposed) {
        HESIC_CHECK_ARG(d->Ho == (d->H + 2 * p - d->KH) / s + 1 && d->Wo == (d->W + 2 * p - d->KW) / s + 1,
                        "conv2d_forward: output size does not match");
        a.QH = d->Ho; a.QW = d->Wo; a.in_step = s; a.out_step = 1; a.nphase = 1;
        int live = d->KH * d->KW;
        if (d->tap_mask_lo) {
            // the kernels walk a raster-order PREFIX of the taps (what MaskedConv2d masks 'A'/'B' are)
            live = 0;
            while (live < d->KH * d->KW && ((d->tap_mask_lo >> live) & 1)) ++live;
            HESIC_CHECK_ARG(live > 0 && (d->tap_mask_lo >> live) == 0, "conv2d_forward: tap mask must be a raster-order prefix");
        }
        a.ntaps_live = live;
    } else {
        HESIC_CHECK_ARG(d->Ho == d->H * s && d->Wo == d->W * s, "conv2d_forward: transposed output must be H*stride");
        HESIC_CHECK_ARG(!d->tap_mask_lo, "conv2d_forward: tap mask unsupported for transposed conv");
        HESIC_CHECK_ARG(d->KH >= s && d->KW >= s, "conv2d_forward: transposed kernel smaller than the stride");
        // output o = q*s + r gets input i = q + (r + p - k)/s for taps k == (r+p) mod s  (see make_taps)
        a.QH = d->H; a.QW = d->W; a.in_step = 1; a.out_step = s; a.nphase = s * s;
        a.ntaps_live = 0;
    }
    // cout tile: 128 (twice the FLOP per staged pixel byte) unless it would waste more than 1/8 of the MFMA work
    const int pad128 = ((d->Cout + 127) / 128) * 128, pad64 = ((d->Cout + 63) / 64) * 64;
    const int BN = (pad128 - pad64) * 8 > pad128 ? 64 : 128;
    a.n_tiles = (d->Cout + BN - 1) / BN;
    const bool fast = d->dtype == HESIC_H16;
    HESIC_CHECK_ARG(!g_y32 || (fast && !gdn), "conv2d_forward_f32out: bf16 storage without the fused GDN epilogue only");
    // pixel tile: 128, shrunk to 64 / 32 (fast path only) until the grid has ~1.5 blocks per CU
    int bm = 128;
    auto count_blocks = [&](int m) {
        int tw = 16;
        while (tw > 1 && tw / 2 >= a.QW) tw /= 2;
        if (tw > m) tw = m;
        const int th = m / tw;
        return (int64_t)a.n_tiles * ((a.QW + tw - 1) / tw) * ((a.QH + th - 1) / th) * a.B * a.nphase;
    };
    // hesic_conv2d_set_phase_fusion(2): eligible transposed layers take the 128-pixel tile whatever the grid (the fused kernel's only tile)
    const bool tr4_shape = d->dtype == HESIC_H16 && d->transposed && s == 2 && BN == 128 && cin_k % 64 == 0 && !hilo && !g_y32 && g_groups == 1 &&
                           !g_act_split && !a.in_abs && d->Cout % 128 == 0 && gdn <= 2;
    const bool tr4_forced = tr4_shape && g_phase4_mode.load(std::memory_order_relaxed) >= 2;
    if (fast && !tr4_forced) {
        if (count_blocks(128) < 384) bm = 64;
        if (bm == 64 && count_blocks(64) < 384 && BN == 128 && cin_k % 64 == 0) bm = 32;
    }
    // Split-K for the low-resolution layers (hyper path: 8x8 .. 32x32 maps): with so few pixels a full-K block per tile
    // leaves most CUs idle and makes every block stream the whole weight tensor.  Given a workspace, the K loop is cut
    // into up to 8 slices (>= 4 stages each) on the largest pixel tile the map fills, partial tiles go to the workspace
    // in fp32 and splitk_reduce_kernel applies bias / activation / bf16 rounding.
    int ksplit = 1;
    if (fast && !gdn && (g_ws || g_ws_need)) {
        const int qpix = a.QH * a.QW;
        int bm_s = qpix >= 128 ? 128 : (qpix >= 64 ? 64 : 32);
        if (bm_s == 32 && !(BN == 128 && cin_k % 64 == 0)) bm_s = 64;
        // Launches whose output feeds round() or a likelihood (the pair layers and every fp32-latent launch) decide the split PER IMAGE: the
        // slice count fixes the summation order, and a pair's latents must not depend on what else is in the batch (round 4: pair 7 of 8
        // and the pair alone differed in the last bit of y -- 2 slices against 8 -- which flips a latent per ~3 pairs)
        const int per_img = (hilo || g_y32) ? a.B : 1;
        const int64_t nb = count_blocks(bm_s) / per_img;
        const int bk_ = cin_k % 64 == 0 ? 64 : 32;
        const int min_taps = d->transposed ? (d->KH / s) * (d->KW / s) : a.ntaps_live;
        const int min_steps = min_taps * (cin_k / bk_);
        // one block per CU is the target; a very long K loop (>= 200 stages: the 960-channel data gradients) is cut further, to two
        // co-resident blocks per CU (183 us unsplit -> 108 us at 4 slices -> measured below at 8)
        const int target = min_steps >= 200 ? 512 : 256;
        int S = (int)((target + nb - 1) / nb);
        if (S > 8) S = 8;
        if (S > min_steps / 4) S = min_steps / 4;
        // measured on MI355X (B=8): it pays when even 32-pixel tiles leave half the CUs idle, or when K is very long;
        // the short K loops of transposed phases and mid-sized maps lose more to the reduce pass than they gain
        const bool one_phase = !d->transposed || s == 1;      // a stride-1 transposed conv (the data gradient of a stride-1 conv) is one phase with the full K loop
        const bool starved = one_phase && count_blocks(32) / per_img < 128;
        // long K on a small map (>= 64 stages: the 192 -> 128 5x5 layer of encode_hyper at 32x32, 75 stages): four K slices on 128-pixel
        // tiles instead of 256 blocks of 32 pixels, 46.6 -> 27.5 + 5 us (round 2; 100 was the round-1 threshold)
        const bool long_k = one_phase && min_steps >= 64;
        if (nb < 256 && S >= 2 && (starved || long_k)) { ksplit = S; bm = bm_s; }
    }
    // 256-pixel tile, 8 waves of 64 x 64 (2 cout x 4 pixel slices), one block per CU, for the pair conv + GDN launch (the forward's largest:
    // g_a_conv2 + GDN on pairs, twice per forward): the weight tile is shared by twice the pixels.  Same box, graph replay, alternating (round 5):
    // 128-pixel tile 345.8 / 344.0 us, this one 332.6 / 331.8.  The K walk of an output is the same in every tile: bit-identical results.
    // Measured and dropped (kept out of the library since round 6): the same tile for single-operand layers (146 vs 153 us conv, 223 vs 216 us
    // transposed: no gain, one 8-wave block per CU marches through its barriers in lockstep), 4 waves of 64 couts x 128 pixels (359 us: a lone wave
    // per SIMD loses its latency cover), a ring of three 48 KB stages (332.2 / 337.9 us), loader waves next to the compute waves (144.0 vs 143.8 us).
    if (fast && hilo == 1 && gdn == 3 && bm == 128 && BN == 128 && cin_k % 64 == 0 && ksplit == 1 && count_blocks(256) >= 384) bm = 256;
    if (g_groups > 1 || g_act_split) {
        HESIC_CHECK_ARG(fast && !gdn, "conv2d_forward_grouped: bf16 storage, no fused GDN");
        HESIC_CHECK_ARG(d->Cout % g_groups == 0 && (d->Cout / g_groups) % BN == 0 && g_act_split % BN == 0,
                        "conv2d_forward_grouped: couts per group and the activation split must be multiples of the cout tile (%d)", BN);
        HESIC_CHECK_ARG(d->x_c_off + (g_groups - 1) * g_x_group_step + d->Cin <= d->x_pix_stride, "conv2d_forward_grouped: input slice out of range");
        ksplit = 1; bm = (bm == 256) ? 128 : bm;          // one block per output tile: the K-slice reduce knows neither groups nor two activations
        a.x_group_step = g_groups > 1 ? g_x_group_step : 0;
        a.tiles_per_group = (d->Cout / g_groups) / BN;
        a.act2 = g_act2; a.act_split = g_act_split;
    }
    const size_t ws_need = ksplit > 1 ? (size_t)ksplit * d->B * d->Ho * d->Wo * d->Cout * sizeof(float) : 0;
    if (g_ws_need) { *g_ws_need = ws_need; return 0; }
    HESIC_CHECK_ARG(ws_need <= g_ws_bytes, "conv2d_forward_ws: workspace too small (%zu < %zu bytes)", g_ws_bytes, ws_need);
    a.ksplit = ksplit; a.ws = g_ws;
    // 2-D pixel patch: as square as the q-grid allows
    int TW = 16;
    while (TW > 1 && TW / 2 >= a.QW) TW /= 2;
    if (bm == 128 && a.QW >= 32 && a.QH < 8) TW = 32;
    if (TW > bm) TW = bm;
    int TH = bm / TW;
    a.TW = TW; a.TH = TH; a.tw_shift = ilog2(TW);
    a.tiles_x = (a.QW + TW - 1) / TW; a.tiles_y = (a.QH + TH - 1) / TH;
    const int64_t nblocks = (int64_t)a.n_tiles * a.tiles_x * a.tiles_y * a.B * a.nphase * ksplit;
    HESIC_CHECK_ARG(nblocks > 0 && nblocks < (1ll << 31), "conv2d_forward: bad grid");
    // The four output phases of a transposed stride-2 layer in one block (igemm_tr4_kernel).  Fused blocks are 4x as long, so a ragged
    // last round of the 512 block slots (256 CUs x 2) costs 4x as much: measured on MI355X (128 -> 128, +IGDN / plain, us unfused ->
    // fused): 1024 blocks 180.7 -> 161.5 / 177 -> 148; 512: 104 -> 93.6 / 101 -> 84.9; 576: 113.9 -> 117.4 / 115.3 -> 108.5; 256: 59.6 ->
    // 51.0 and 48.5 -> 54.0 / ties.  Auto mode uses it from 384 fused blocks on when its rounds are >= 70 % full.
    // hesic_conv2d_set_phase_fusion / HESIC_IGEMM_PHASE4 = 0: never, 1: auto, 2: whenever the shape is eligible.
    bool use_tr4 = false;
    {
        const int mode = g_phase4_mode.load(std::memory_order_relaxed);
        const int64_t nb4 = nblocks / 4, rounds = (nb4 + 511) / 512;
        const bool fills = mode >= 2 || (nb4 >= 384 && nb4 * 10 >= rounds * 512 * 7);
        use_tr4 = fast && mode && fills && bm == 128 && tr4_shape && ksplit == 1 && (int64_t)d->Ho * d->Wo * a.y_ps * 2 < (1ll << 31);
    }
    if (g_plan_out) {
        if (use_tr4) { g_plan_out[0] = 128; g_plan_out[1] = 128; g_plan_out[2] = 64; g_plan_out[3] = 2; return 0; }
        g_plan_out[0] = bm; g_plan_out[1] = BN; g_plan_out[2] = fast ? (((bm == 32) || (bm == 64 && BN == 64)) && cin_k % 128 == 0 ? 128 : (cin_k % 64 == 0 ? 64 : 32)) : BK; g_plan_out[3] = fast ? 1 : 0;
        return 0;
    }
    a.tap_parity = (!d->transposed && s == 2 && d->KH >= 2 && d->KW >= 2 && a.ntaps_live == d->KH * d->KW && ksplit == 1) ? 1 : 0;
    // (a class -> channel chunk -> tap walk of the parity classes cut the pair launch's fetch traffic 800 -> 655 MB at equal time, 343 vs 346 us, but
    // the summation order then depends on the K step of the tile variant, and with it the last bit of y on the batch size: removed in round 6)
    a.fd_nt = make_fastdiv((uint32_t)a.n_tiles); a.fd_tx = make_fastdiv((uint32_t)a.tiles_x); a.fd_ty = make_fastdiv((uint32_t)a.tiles_y);
    a.fd_b = make_fastdiv((uint32_t)a.B); a.fd_ph = make_fastdiv((uint32_t)a.nphase);
    const dim3 grid((unsigned)nblocks), block(NTHREADS);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_GLDS(M_, N_, K_, S_)                                                                          \
    do {                                                                                                    \
        if (hilo == 2) {                                                                                    \
            if (N_ == 128 && gdn == 3) hipLaunchKernelGGL((igemm_glds_kernel<M_, 128, K_, S_, 3, 4, 0, 2>), grid, block, 0, st, a);       \
            else if (N_ == 128 && gdn == 4) hipLaunchKernelGGL((igemm_glds_kernel<M_, 128, K_, S_, 4, 4, 0, 2>), grid, block, 0, st, a);  \
            else hipLaunchKernelGGL((igemm_glds_kernel<M_, N_, K_, S_, 0, 4, 0, 2>), grid, block, 0, st, a);                              \
        }                                                                                                   \
        else if (hilo) {                                                                                    \
            if (N_ == 128 && gdn == 3) hipLaunchKernelGGL((igemm_glds_kernel<M_, 128, K_, S_, 3, 4, 0, 1>), grid, block, 0, st, a);       \
            else if (N_ == 128 && gdn == 4) hipLaunchKernelGGL((igemm_glds_kernel<M_, 128, K_, S_, 4, 4, 0, 1>), grid, block, 0, st, a);  \
            else hipLaunchKernelGGL((igemm_glds_kernel<M_, N_, K_, S_, 0, 4, 0, 1>), grid, block, 0, st, a);                              \
        }                                                                                                   \
        else if (N_ == 128 && gdn == 3) hipLaunchKernelGGL((igemm_glds_kernel<M_, 128, K_, S_, 3>), grid, block, 0, st, a);  \
        else if (N_ == 128 && gdn == 4) hipLaunchKernelGGL((igemm_glds_kernel<M_, 128, K_, S_, 4>), grid, block, 0, st, a);  \
        else if (N_ == 128 && gdn == 1) hipLaunchKernelGGL((igemm_glds_kernel<M_, 128, K_, S_, 1>), grid, block, 0, st, a);  \
        else if (N_ == 128 && gdn == 2) hipLaunchKernelGGL((igemm_glds_kernel<M_, 128, K_, S_, 2>), grid, block, 0, st, a);  \
        else hipLaunchKernelGGL((igemm_glds_kernel<M_, N_, K_, S_, 0>), grid, block, 0, st, a);                              \
    } while (0)
#define LAUNCH_GLDS_NS(M_, N_, K_)                                   \
    do {                                                             \
        if (deep) LAUNCH_GLDS(M_, N_, K_, 4);                        \
        else LAUNCH_GLDS(M_, N_, K_, 2);                             \
    } while (0)
    if (fast) {
        // ring depth: a 2-deep ring relies on 2-3 co-resident blocks per CU to hide the L2 latency; layers whose grid
        // is too small for that (low resolutions) get a 4-deep ring instead, as long as every block of the grid still
        // fits in LDS at once (160 KB per CU)
        // the buffer-addressed DMA keeps 32-bit offsets relative to the tile's first input row and the packed weights
        HESIC_CHECK_ARG(((int64_t)(TH * a.in_step + 2 * d->KH) * a.W + 2 * d->KW) * a.x_ps * 2 < (1ll << 31) &&
                            (int64_t)d->KH * d->KW * d->Cout * cin_k * 2 < (1ll << 31),
                        "conv2d_forward: image rows / weights too large for 32-bit tile offsets");
        const int bk = cin_k % 64 == 0 ? 64 : 32;
        const int stage = (bm + BN) * bk * 2;
        const int64_t per_cu = (nblocks + 255) / 256;
        const bool deep = bm < 128 ? per_cu * 4 * stage <= 160 * 1024 : (bk == 32 && per_cu * 4 * stage <= 160 * 1024);
        if (bm == 256) {
            // pair conv + GDN on 256 pixels x 128 couts: 8 waves of 64 x 64, one block per CU (see the tile choice above)
            hipLaunchKernelGGL((igemm_glds_kernel<256, 128, 64, 2, 3, 8, 0, 1>), grid, dim3(512), 0, st, a);
        } else if (use_tr4) {
            // the four output phases of a tile in one block (igemm_tr4_kernel): a quarter of the blocks, one 50-stage pipeline each
            const dim3 grid4((unsigned)(nblocks / 4));
            // the input patch staged once per block (HALO): 128 input channels, 5 x 5 taps at pad 2 (shifts -1 .. 1), the 8 x 16 q-tile
            const bool halo = TR4_HALO && cin_k == 128 && d->KH == 5 && d->KW == 5 && d->pad == 2 && a.TW == 16 && a.TH == 8;
            if (halo) {
                if (gdn == 1) hipLaunchKernelGGL((igemm_tr4_kernel<1, 1>), grid4, block, 0, st, a);
                else if (gdn == 2) hipLaunchKernelGGL((igemm_tr4_kernel<2, 1>), grid4, block, 0, st, a);
                else hipLaunchKernelGGL((igemm_tr4_kernel<0, 1>), grid4, block, 0, st, a);
            }
            else if (gdn == 1) hipLaunchKernelGGL((igemm_tr4_kernel<1>), grid4, block, 0, st, a);
            else if (gdn == 2) hipLaunchKernelGGL((igemm_tr4_kernel<2>), grid4, block, 0, st, a);
            else hipLaunchKernelGGL((igemm_tr4_kernel<0>), grid4, block, 0, st, a);
        } else if (bm == 128) {
            if (bk == 64) { if (BN == 128) LAUNCH_GLDS(128, 128, 64, 2); else LAUNCH_GLDS_NS(128, 64, 64); }
            else { if (BN == 128) LAUNCH_GLDS_NS(128, 128, 32); else LAUNCH_GLDS_NS(128, 64, 32); }
        } else if (bm == 64) {
            // same lever as for the 32-pixel tile: BK = 128 where two 2-stage blocks still fit a CU (64 x 64 tiles: 64 KB; 128 ->
            // 192 5x5 s2 @64x64 B=8: 28.4 -> 22.3 us) or where the grid is one block per CU anyway (64 x 128 tiles, 96 KB)
            if (bk == 64 && BN == 64 && cin_k % 128 == 0) LAUNCH_GLDS(64, 64, 128, 2);
            else if (bk == 64) { if (BN == 128) LAUNCH_GLDS_NS(64, 128, 64); else LAUNCH_GLDS_NS(64, 64, 64); }
            else { if (BN == 128) LAUNCH_GLDS_NS(64, 128, 32); else LAUNCH_GLDS_NS(64, 64, 32); }
        } else {
            // 32-pixel tiles are bound by per-stage bookkeeping and barrier waits (PMC: ~95 scalar/vector instructions per 4
            // MFMAs): BK = 128 halves the stage count (8 MFMAs per wave and stage, 2-deep ring of 40 KB stages) --
            // 128 -> 128 5x5 s1 @32x32 B=8: 23.4 -> 18.8 us; a 3-deep ring was slower (20.7).
            if (cin_k % 128 == 0) LAUNCH_GLDS(32, 128, 128, 2);
            else LAUNCH_GLDS_NS(32, 128, 64);
        }
    } else {
        if (BN == 128) hipLaunchKernelGGL((igemm_conv_kernel<float, 128>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((igemm_conv_kernel<float, 64>), grid, block, 0, st, a);
    }
    if (ksplit > 1) {
        const int64_t npix = (int64_t)d->B * d->Ho * d->Wo;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(npix * (d->Cout / 8), 256)), dim3(256), 0, st, (const float*)g_ws, ksplit,
                           npix, d->Cout, bias, d->act, (h16_t*)y, d->y_pix_stride, d->y_c_off, g_y32, g_y32_ps, g_y32_co, g_y_hilo, g_y_abs);
    }
    HESIC_LAUNCH_RETURN("conv2d_forward");
}

extern "C" size_t hesic_conv2d_ws_bytes(const hesic_conv_desc* d) {
    size_t need = 0;
    if (!d) return 0;
    g_ws_need = &need;
    const int rc = hesic_conv2d_forward(d, (const void*)16, (const void*)16, nullptr, (void*)16, nullptr);
    g_ws_need = nullptr;
    return rc == 0 ? need : 0;
}

/* The same query for hesic_conv2d_forward_f32out: a launch with an fp32 latent output decides its K split per image (a pair's latents
 * must not depend on the batch), so its scratch can differ from the plain launch's. */
extern "C" size_t hesic_conv2d_f32out_ws_bytes(const hesic_conv_desc* d) {
    size_t need = 0;
    if (!d) return 0;
    g_ws_need = &need; g_y32 = (float*)16; g_y32_ps = d->Cout; g_y32_co = 0;
    const int rc = hesic_conv2d_forward(d, (const void*)16, (const void*)16, nullptr, (void*)16, nullptr);
    g_ws_need = nullptr; g_y32 = nullptr; g_y32_ps = g_y32_co = 0;
    return rc == 0 ? need : 0;
}

extern "C" int hesic_conv2d_forward_ws(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                       void* y, void* ws, size_t ws_bytes, void* stream) {
    g_ws = (float*)ws; g_ws_bytes = ws ? ws_bytes : 0;
    const int rc = hesic_conv2d_forward(d, x, w_packed, bias, y, stream);
    g_ws = nullptr; g_ws_bytes = 0;
    return rc;
}

extern "C" int hesic_conv2d_forward_f32out(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                           void* y, float* y_f32, int y32_pix_stride, int y32_c_off, void* ws, size_t ws_bytes,
                                           void* stream) {
    HESIC_CHECK_ARG(d && y_f32, "conv2d_forward_f32out: null pointer");
    HESIC_CHECK_ARG(d->dtype == HESIC_H16, "conv2d_forward_f32out: the fp32 copy exists for bf16 storage (fp32 storage is fp32 already)");
    HESIC_CHECK_ARG(y32_c_off % 4 == 0 && y32_pix_stride % 4 == 0 && y32_c_off + d->Cout <= y32_pix_stride,
                    "conv2d_forward_f32out: fp32 channel slice must be 16-byte aligned and in range");
    g_y32 = y_f32; g_y32_ps = y32_pix_stride; g_y32_co = y32_c_off;
    g_ws = (float*)ws; g_ws_bytes = ws ? ws_bytes : 0;
    const int rc = hesic_conv2d_forward(d, x, w_packed, bias, y, stream);
    g_ws = nullptr; g_ws_bytes = 0;
    g_y32 = nullptr; g_y32_ps = g_y32_co = 0;
    return rc;
}

extern "C" int hesic_conv2d_forward_grouped(const hesic_conv_desc* d, int groups, int x_group_step, int act2, int act_split, const void* x,
                                            const void* w_packed, const float* bias, void* y, float* y_f32, int y32_pix_stride,
                                            int y32_c_off, void* stream) {
    HESIC_CHECK_ARG(d && groups >= 1 && x_group_step >= 0 && act_split >= 0 && act_split <= d->Cout, "conv2d_forward_grouped: bad arguments");
    HESIC_CHECK_ARG(d->dtype == HESIC_H16, "conv2d_forward_grouped: bf16 storage");
    HESIC_CHECK_ARG(!y_f32 || (y32_c_off % 4 == 0 && y32_pix_stride % 4 == 0 && y32_c_off + d->Cout <= y32_pix_stride),
                    "conv2d_forward_grouped: fp32 channel slice must be 16-byte aligned and in range");
    g_groups = groups; g_x_group_step = x_group_step; g_act2 = act2; g_act_split = act_split;
    g_y32 = y_f32; g_y32_ps = y32_pix_stride; g_y32_co = y32_c_off;
    const int rc = hesic_conv2d_forward(d, x, w_packed, bias, y, stream);
    g_groups = 1; g_x_group_step = g_act2 = g_act_split = 0;
    g_y32 = nullptr; g_y32_ps = g_y32_co = 0;
    return rc;
}

// One weight of a grouped launch into its cout slice [co_off, co_off + Cout) of a packed buffer [KH*KW][Cout_total][Cin].
namespace {
__global__ void pack_weight_slice_kernel(const float* __restrict__ w, h16_t* __restrict__ wp, int Cout, int Cin, int KH, int KW, int transposed,
                                         int Cout_total, int co_off) {
    const int64_t n = (int64_t)KH * KW * Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = i % Cin;
        int64_t r = i / Cin;
        const int co = r % Cout, tap = r / Cout;
        const int ky = tap / KW, kx = tap % KW;
        const int64_t src = transposed ? (((int64_t)ci * Cout + co) * KH + ky) * KW + kx : (((int64_t)co * Cin + ci) * KH + ky) * KW + kx;
        wp[((int64_t)tap * Cout_total + co_off + co) * Cin + ci] = f2h(w[src]);
    }
}
}  // namespace

extern "C" int hesic_pack_conv_weight_slice(const float* w, void* wp, int Cout, int Cin, int KH, int KW, int transposed, int Cout_total,
                                            int co_off, void* stream) {
    HESIC_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && co_off >= 0 && co_off + Cout <= Cout_total, "pack_conv_weight_slice: bad arguments");
    const int64_t n = (int64_t)KH * KW * Cout * Cin;
    hipLaunchKernelGGL(pack_weight_slice_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (h16_t*)wp, Cout, Cin, KH, KW,
                       transposed, Cout_total, co_off);
    HESIC_LAUNCH_RETURN("pack_conv_weight_slice");
}

/* bf16x3 implicit GEMM (kernel template flag HL): d->Cin / d->Cout are the LOGICAL channel counts. */
static int conv2d_forward_hilo_n(int products, const hesic_conv_desc* d, const void* x_hilo, const void* w_packed_hilo, const float* bias,
                                         const void* gamma_packed, const void* gamma_lo_packed, const float* beta_packed, int inverse,
                                         void* y_hilo, int y_abs, float* y_f32, int y32_pix_stride, int y32_c_off, void* ws, size_t ws_bytes, void* stream) {
    // the accumulator scale named for THIS launch (hesic_conv2d_hilo_set_acc_scale) is taken and cleared first: a call refused by one of the
    // checks below must not leave it behind for an unrelated later launch
    const float acc_scale_now = g_hilo_acc_scale;
    g_hilo_acc_scale = 1.f;
    HESIC_CHECK_ARG(d && x_hilo && w_packed_hilo && (y_hilo || y_f32), "conv2d_forward_hilo: null pointer");
    HESIC_CHECK_ARG(d->dtype == HESIC_H16 && !d->in_abs, "conv2d_forward_hilo: bf16 storage, no |x| on load (use y_abs on the producer)");
    const bool gdn = gamma_packed != nullptr;
    if (gdn) {
        HESIC_CHECK_ARG(gamma_lo_packed && beta_packed && y_hilo && !y_f32 && !y_abs, "conv2d_forward_hilo: the fused GDN form writes y_hilo only and needs both gamma' halves");
        HESIC_CHECK_ARG(d->Cout == 128 && d->act == HESIC_ACT_NONE, "conv2d_forward_hilo: fused GDN needs Cout == 128 and no activation");
    }
    HESIC_CHECK_ARG(!y_f32 || (y32_c_off % 4 == 0 && y32_pix_stride % 4 == 0 && y32_c_off + d->Cout <= y32_pix_stride),
                    "conv2d_forward_hilo: fp32 channel slice must be 16-byte aligned and in range");
    g_hilo = products == 2 ? 2 : 1;
    g_hilo_acc_scale_active = acc_scale_now;
    g_gdn_gamma = gamma_packed; g_gdn_gamma_lo = gamma_lo_packed; g_gdn_beta = beta_packed; g_gdn_mode = gdn ? (inverse ? 4 : 3) : 0;
    g_y32 = y_f32; g_y32_ps = y32_pix_stride; g_y32_co = y32_c_off;
    g_y_hilo = (!gdn && y_hilo) ? 1 : 0; g_y_abs = y_abs ? 1 : 0;
    g_ws = (float*)ws; g_ws_bytes = ws ? ws_bytes : 0;
    const int rc = hesic_conv2d_forward(d, x_hilo, w_packed_hilo, bias, y_hilo, stream);
    g_ws = nullptr; g_ws_bytes = 0;
    g_hilo = 0; g_y_hilo = g_y_abs = 0;
    g_gdn_gamma = nullptr; g_gdn_gamma_lo = nullptr; g_gdn_beta = nullptr; g_gdn_mode = 0;
    g_y32 = nullptr; g_y32_ps = g_y32_co = 0;
    g_hilo_acc_scale_active = 1.f;                        // one launch only
    return rc;
}

extern "C" int hesic_conv2d_hilo_set_acc_scale(float scale) {
    HESIC_CHECK_ARG(scale > 0.f && scale == scale && scale < 3.0e38f, "conv2d_hilo_set_acc_scale: a positive finite factor (a power of two) expected");
    g_hilo_acc_scale = scale;
    return 0;
}

extern "C" int hesic_conv2d_forward_hilo(const hesic_conv_desc* d, const void* x_hilo, const void* w_packed_hilo, const float* bias,
                                         const void* gamma_packed, const void* gamma_lo_packed, const float* beta_packed, int inverse,
                                         void* y_hilo, int y_abs, float* y_f32, int y32_pix_stride, int y32_c_off, void* ws, size_t ws_bytes, void* stream) {
    return conv2d_forward_hilo_n(3, d, x_hilo, w_packed_hilo, bias, gamma_packed, gamma_lo_packed, beta_packed, inverse, y_hilo, y_abs, y_f32, y32_pix_stride,
                                 y32_c_off, ws, ws_bytes, stream);
}

/* Pairs x SINGLE weights: w_packed_hilo has the same [tap][Cout][2 Cin] layout, its first Cin values per row are the weights (rounded with
 * error feedback over the taps: hesic_pack_conv_weight_shaped), the second Cin are not read.  Two products per pair; the fused GDN stops
 * at gamma'_hi (gamma_lo_packed is not read either). */
extern "C" int hesic_conv2d_forward_hilo_w1(const hesic_conv_desc* d, const void* x_hilo, const void* w_packed_hilo, const float* bias,
                                            const void* gamma_packed, const void* gamma_lo_packed, const float* beta_packed, int inverse,
                                            void* y_hilo, int y_abs, float* y_f32, int y32_pix_stride, int y32_c_off, void* ws, size_t ws_bytes, void* stream) {
    return conv2d_forward_hilo_n(2, d, x_hilo, w_packed_hilo, bias, gamma_packed, gamma_lo_packed, beta_packed, inverse, y_hilo, y_abs, y_f32, y32_pix_stride,
                                 y32_c_off, ws, ws_bytes, stream);
}

/* Single 16-bit operands (one product per MAC) with the hi/lo (I)GDN epilogue: v = conv + bias stays in the fp32 accumulators, the
 * squares and gamma' go through the contraction as pairs, y leaves as [hi(128) | lo(128)] per pixel for a hi/lo consumer. */
extern "C" int hesic_conv2d_gdn_forward_hilo_out(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                                 const void* gamma_packed, const void* gamma_lo_packed, const float* beta_packed, int inverse,
                                                 void* y_hilo, void* stream) {
    HESIC_CHECK_ARG(d && x && w_packed && gamma_packed && gamma_lo_packed && beta_packed && y_hilo, "conv2d_gdn_forward_hilo_out: null pointer");
    HESIC_CHECK_ARG(d->dtype == HESIC_H16 && !d->in_abs && d->Cout == 128 && d->act == HESIC_ACT_NONE && d->Cin % 32 == 0 && !d->transposed,
                    "conv2d_gdn_forward_hilo_out: 16-bit storage, Conv2d, Cout == 128, Cin %% 32 == 0, no activation");
    g_gdn_gamma = gamma_packed; g_gdn_gamma_lo = gamma_lo_packed; g_gdn_beta = beta_packed; g_gdn_mode = inverse ? 4 : 3;
    const int rc = hesic_conv2d_forward(d, x, w_packed, bias, y_hilo, stream);
    g_gdn_gamma = nullptr; g_gdn_gamma_lo = nullptr; g_gdn_beta = nullptr; g_gdn_mode = 0;
    return rc;
}

extern "C" size_t hesic_conv2d_hilo_ws_bytes(const hesic_conv_desc* d) {
    size_t need = 0;
    if (!d) return 0;
    g_ws_need = &need; g_hilo = 1;
    const int rc = hesic_conv2d_forward(d, (const void*)16, (const void*)16, nullptr, (void*)16, nullptr);
    g_ws_need = nullptr; g_hilo = 0;
    return rc == 0 ? need : 0;
}
