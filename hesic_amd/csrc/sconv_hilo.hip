// g_a_conv1 + g_a_gdn1 (conv(3, N) 5x5 stride 2 -> GDN, newnet1.py:583-584 / :633-634) on hi/lo bf16 operand pairs: the first
// layer of the bf16x3 analysis path (include/hesic_hip.h, "hi/lo" section).  Same data flow as sconv_n2w_gdn_fast_kernel
// (csrc/sconv.hip) -- the image rows of a 32-pixel tile are gathered straight into MFMA B fragments, the weights are the A
// operand from an LDS image, the GDN contraction takes its squares from the wave's own accumulators (K-permuted gamma' image) --
// but every operand is a PAIR (hi = bf16(v), lo = bf16(v - hi)) and every product three MFMAs (hi hi + hi lo + lo hi; lo lo,
// 2^-18, is dropped): x from the fp32 image, w and gamma' from fp32 parameters, the squares and the output from the fp32
// accumulators.  The output leaves as [hi(128) | lo(128)] per pixel for the hi/lo implicit-GEMM layers behind it.
//
// Four 32 KB weight images (w_hi, w_lo, gamma'_hi, gamma'_lo) + 4 KB of row staging per wave (the tile's 32 pixels x 64 channels of one half
// at a time) = the CU's whole 160 KB of LDS: one block of eight waves per CU, two per SIMD, so one wave's VALU phases (operand
// splitting, squares, rsqrt, output staging: about as many issue cycles per tile as its 192 MFMAs) run under the other's MFMAs.
#include "common.h"

#ifdef N2W_DBG
__device__ long long g_n2w_dbg[2][16];
#define N2W_T(k) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && (wave == 0 || wave == 4) && it_dbg == 3 && lane == 0) g_n2w_dbg[wave >> 2][k] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define N2W_T(k) do {} while (0)
#endif

namespace {

struct HArgs {
    const float* x; const unsigned char* img; const float* bias; const float* beta; h16_t* y;
    int B, H, W, Ho, Wo;
    int64_t xs_b, xs_c, xs_y, ys_b, ys_y, ys_x;
    FastDiv fd_tx, fd_ty;
};

__device__ __forceinline__ void split2(float p, float q, uint32_t& hi, uint32_t& lo) { split_h2(p, q, hi, lo); }
// image samples: no saturation step (the binary16 build takes |x| <= 65504 -- an image)
__device__ __forceinline__ void split2_img(float p, float q, uint32_t& hi, uint32_t& lo) {
    hi = pack_h2_raw(p, q);
    lo = pack_h2_raw(sub_h2_lo(p, hi), sub_h2_hi(q, hi));
}
// fp32 -> 16-bit terms whose sum is the value to ~2^-22 (binary16: two terms, bfloat16: three)
constexpr int H16_TERMS = HESIC_H16_IS_F16 ? 2 : 3;
__device__ __forceinline__ void terms3(float v, uint32_t& t01, uint32_t& t2) {
    const uint32_t a = pack_h2_raw(h16_clamp(v), 0.f) & 0xffffu;
    const float r1 = h16_clamp(v) - h2f_lo(a);
    const uint32_t b = pack_h2_raw(r1, 0.f) & 0xffffu;
    t01 = a | (b << 16);
    t2 = H16_TERMS == 3 ? (pack_h2_raw(r1 - h2f_lo(b), 0.f) & 0xffffu) : 0u;
}

// OUT1 = 1: the output leaves as ONE 16-bit value per channel (128 channels per pixel; half the stores) -- for a consumer that multiplies
// single operands (the "x3c2" analysis mode: g_a_conv2, 70 % of g_a's MACs, at one product per MAC).  Its value is then rounded to 2^-12
// anyway, so the kernel's own arithmetic only needs ~2^-15: the w_lo and gamma'_lo products are dropped (TWO MFMAs per operand pair, 128
// instead of 192 per tile) -- x and the squares stay pairs, w is rounded with error feedback over the taps (the image is smooth: the
// weight-rounding error of the sum cancels, see pack_weight_shaped_kernel), gamma' single.  CPU study (precision_study.py schemes, 512^2,
// g_a_conv2 single): 3.97e-4 flipped latents with the full pair arithmetic here, 4.09e-4 with this form.
// (A barrier rotation that put one wave of a SIMD into its matrix phase while the other stored -- PING -- ran 75.5 vs 81.8 us back to back but 0.5 % LESS on
// the 8-pair step, twice; one wave per SIMD was only 1.3x slower than two.  Both forms left the library in round 6: profiles/experiments/r05_n2w_ping_rotation.patch.)
template <int INV, int OUT1, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW / 4) void n2w_gdn_hilo_kernel(const HArgs a) {
    constexpr int KS = 5, R = 15, NT = NW * 64;
    constexpr bool EARLY_REQ = OUT1 != 0 && NW == 4;
    constexpr uint32_t POISON = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned char* wl_hi = smem;                 // conv weights [128 co][16 slots ^ (co & 15)] of 8 bf16: slot r = ci*5 + ky, values kx 0..4
    const unsigned char* wl_lo = smem + 32768;
    const unsigned char* gl_hi = smem + 65536;         // gamma' [128 out][16 slots ^ (row & 15)], K order permuted to the accumulator layout
    const unsigned char* gl_lo = smem + 98304;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    unsigned char* os = smem + 131072 + wave * 4096;   // 32 pixel rows of 128 bytes (64 channels of one half), 16-byte chunks XOR (pixel & 7)
    // A-fragment address of (32-row block i, k-step ks) inside an image: row = i*32 + frow, so (row & 15) == (frow & 15) and the lane part
    // is ONE offset per k-step, the block / image part an immediate -- 8 address registers instead of one per (i, ks, image)
    auto fa = [&](int ks) { return frow * 256 + ((((ks * 2 + fh) ^ (frow & 15))) << 4); };

    const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + 1) / 2;       // wave tile = 2 output rows x 16 columns
    const int ntiles = tiles_x * tiles_y * a.B;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tstride = (int)gridDim.x * NW;
    u32x2 raw[8][3];
    int tb = 0, tty = 0, ttx = 0;
    auto request = [&](int tile) {                    // the 24 row-pair loads of a tile (see sconv_n2w_gdn_fast_kernel)
        const uint32_t q = fdiv((uint32_t)tile, a.fd_tx);
        ttx = tile - (int)q * tiles_x;
        tb = (int)fdiv(q, a.fd_ty);
        tty = (int)q - tb * tiles_y;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)tb * a.xs_b - 2), 0, (int)POISON, 0x00020000);
        int fhl = fh;                                 // opaque copy: the per-k-step row constants below are two VALU ops each -- recomputed per
        asm volatile("" : "+v"(fhl));                 // tile instead of hoisted out of the tile loop into ~30 long-lived registers (which spilled)
        const int oy = tty * 2 + (frow >> 4), ox = ttx * 16 + (frow & 15);
        const bool pok = oy < a.Ho && ox < a.Wo;
        const int iy0 = oy * 2 - 2, ix0 = ox * 2 - 2;
        const uint32_t base = (uint32_t)((iy0 * (int)a.xs_y + ix0) * 4 + 8);
        const bool ok0 = pok && ox > 0, ok2 = pok && ix0 + 4 < a.W;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int ra = 2 * ks, rb = 2 * ks + 1;
            const int kya = ra % KS, kyb = rb < R ? rb % KS : 0x40000000, cia = ra / KS, cib = rb / KS;
            const uint32_t offa = (uint32_t)((cia * (int)a.xs_c + kya * (int)a.xs_y) * 4), offb = (uint32_t)((cib * (int)a.xs_c + (rb % KS) * (int)a.xs_y) * 4);
            const bool okr = (unsigned)(iy0 + (fhl ? kyb : kya)) < (unsigned)a.H;
            const uint32_t v = base + (fhl ? offb : offa);
            const uint32_t v0 = (okr && ok0) ? v : POISON, v1 = (okr && pok) ? v : POISON, v2 = (okr && ok2) ? v : POISON;
            raw[ks][0] = __builtin_amdgcn_raw_buffer_load_b64(xr, (int)v0, 0, 0);
            raw[ks][1] = __builtin_amdgcn_raw_buffer_load_b64(xr, (int)v1, 8, 0);
            raw[ks][2] = __builtin_amdgcn_raw_buffer_load_b64(xr, (int)v2, 16, 0);
        }
    };
    int tile = lb * NW + wave;
    if (tile < ntiles) request(tile);                 // in flight while the block copies its weight images

    {
        const u32x4* src = (const u32x4*)a.img;       // 8192 slots of 16 bytes, laid out by n2w_hilo_pack_kernel
#pragma unroll
        for (int r = 0; r < 8192 / (8 * NT); ++r) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[tid + (r * 8 + j) * NT];
#pragma unroll
            for (int j = 0; j < 8; ++j) *(u32x4*)(smem + (tid + (r * 8 + j) * NT) * 16) = v[j];
        }
    }
    __syncthreads();

    const uint32_t st_lane = (uint32_t)(((lane >> 3) * (int)a.ys_x + (lane & 7) * 8) * 2);       // store lane = (pixel of 8, 16-byte chunk of a 128-byte line)
    const __amdgpu_buffer_rsrc_t bias_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? 512 : 0, 0x00020000);      // no bias: every load returns zeros
    const __amdgpu_buffer_rsrc_t beta_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.beta, 0, 512, 0x00020000);
    // OUT1: bias and beta' enter through the matrix cores -- one k-step whose A fragment holds the value as 16-bit terms in k = 0..2 of its
    // row (lanes of the upper k half: zeros) against a B fragment of ones: 2 x 4 MFMAs per tile start the two accumulator sets instead of 2 x 16
    // buffer loads whose latency sat in front of each MFMA phase; the terms live in registers for the whole launch
    uint32_t cb01[4], cb2[4], ce01[4], ce2[4];
    u32x4 ones_b = u32x4{0u, 0u, 0u, 0u};
    if constexpr (OUT1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bias_rs, (i * 32 + frow) * 4, 0, 0));
            const float ev = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(beta_rs, (i * 32 + frow) * 4, 0, 0));
            terms3(bv * H16_SQ_ROOT, cb01[i], cb2[i]);      // the accumulators hold conv * sqrt(H16_SQ_SCALE), the norms beta' * H16_SQ_SCALE + ...:
            terms3(ev * H16_SQ_SCALE, ce01[i], ce2[i]);     // the output v * rsqrt(norm) is unchanged, the squares need no scaling multiply
            if (fh) { cb01[i] = cb2[i] = ce01[i] = ce2[i] = 0u; }
        }
        ones_b = u32x4{H16_ONE_PAIR, H16_ONE_PAIR & 0xffffu, 0u, 0u};
    }
    [[maybe_unused]] int it_dbg = -1;
    for (; tile < ntiles; tile += tstride) {
        ++it_dbg;
        N2W_T(0);
        const int b = tb, ty = tty, tx = ttx;        // of the tile whose rows are in `raw`
        u32x4 xh[8], xl[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x2 p0 = __builtin_bit_cast(f32x2, raw[ks][0]), p1 = __builtin_bit_cast(f32x2, raw[ks][1]), p2 = __builtin_bit_cast(f32x2, raw[ks][2]);
            uint32_t h0, l0, h1, l1, h2, l2;
            split2_img(p0.x, p0.y, h0, l0);
            split2_img(p1.x, p1.y, h1, l1);
            split2_img(p2.x, 0.f, h2, l2);
            xh[ks] = u32x4{h0, h1, h2, 0u};
            xl[ks] = u32x4{l0, l1, l2, 0u};
        }
        // bias / beta' are re-read per tile (buffer loads, L1 hits) through an offset the optimiser cannot see through: hoisted out of
        // the tile loop their 2 x 64 values per lane would sit in registers next to the accumulators and spill.  (A laundered
        // POINTER turned them into flat loads, which also count on lgkmcnt and stalled every LDS wait behind a global round trip.)
        uint32_t pofs = (uint32_t)(fh * 16);
        asm volatile("" : "+v"(pofs));
        N2W_T(1);
        N2W_T(2);
        f32x16 acc[4];
        if constexpr (OUT1) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = mfma_32x32x16_h16(__builtin_bit_cast(h16x8, u32x4{cb01[i], cb2[i], 0u, 0u}), __builtin_bit_cast(h16x8, ones_b), z, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bias_rs, (int)(pofs + (uint32_t)((i * 32 + 8 * g) * 4)), 0, 0));
                    acc[i][4 * g] = bv.x; acc[i][4 * g + 1] = bv.y; acc[i][4 * g + 2] = bv.z; acc[i][4 * g + 3] = bv.w;
                }
        }
        // conv: the w_hi fragments are double-buffered one k-step ahead; the w_lo fragments of a k-step are requested at its start and
        // used last (behind 8 MFMAs = their LDS latency)
        {
            h16x8 wh[2][4], wl[4];
            auto ldh = [&](int set, int ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wh[set][i] = *(const h16x8*)(wl_hi + fa(ks) + i * 8192);
                }
            };
            ldh(0, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if constexpr (!OUT1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        wl[i] = *(const h16x8*)(wl_lo + fa(ks) + i * 8192);
                    }
                }
                if (ks + 1 < 8) ldh((ks + 1) & 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                const h16x8 fxh = __builtin_bit_cast(h16x8, xh[ks]), fxl = __builtin_bit_cast(h16x8, xl[ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma_32x32x16_h16(wh[ks & 1][i], fxh, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma_32x32x16_h16(wh[ks & 1][i], fxl, acc[i], 0, 0, 0);
                if constexpr (!OUT1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = mfma_32x32x16_h16(wl[i], fxh, acc[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        N2W_T(3);
        if constexpr (EARLY_REQ) {
            // the next tile's rows are requested in front of the GDN contraction: its 68 MFMAs, the rsqrt pass and the staging passes cover
            // the round trip (requested behind them, ~half of it was exposed: one wave per SIMD measured 14 k cycles per tile for 8 k of work)
            __builtin_amdgcn_sched_barrier(0);
            request(tile + tstride < ntiles ? tile + tstride : tile);
            __builtin_amdgcn_sched_barrier(0);
        }
        // binary16 build, pair output: the squares of this pixel are formed from v * 2^k, k from the pixel's largest |v| over its 128 channels
        // (this lane's 64 and lane ^ 32's), beta' is added behind the contraction -- see the hi/lo (I)GDN epilogue of igemm_glds_kernel
        // (conv_igemm.hip) for the why: fixed-scale squares of small activations are subnormal halves
        constexpr bool DYN_SQ = HESIC_H16_IS_F16 && !HESIC_NO_DYN_SQ && !OUT1;
        [[maybe_unused]] float sq_c = 1.f, sq_inv = 1.f;
        if constexpr (DYN_SQ) {
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(acc[i][r])), fabsf(acc[i][r + 1]));     // v_max3_f32 with |.| modifiers
            m = fmaxf(m, __shfl_xor(m, 32));
            int ex = 0;
            (void)frexpf(m, &ex);
            int k = 7 - ex;
            k = k > 40 ? 40 : (k < -24 ? -24 : k);
            sq_c = ldexpf(1.f, k);
            sq_inv = ldexpf(1.f, -6 - 2 * k);                   // gamma' is packed times 64 (H16_SQ_UNSCALE)
        }
        f32x16 nrm[4];
        if constexpr (DYN_SQ) {
            // started by the first k-step's MFMAs (zero C operand)
        } else if constexpr (OUT1) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                nrm[i] = mfma_32x32x16_h16(__builtin_bit_cast(h16x8, u32x4{ce01[i], ce2[i], 0u, 0u}), __builtin_bit_cast(h16x8, ones_b), z, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 be = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(beta_rs, (int)(pofs + (uint32_t)((i * 32 + 8 * g) * 4)), 0, 0));
                    nrm[i][4 * g] = be.x; nrm[i][4 * g + 1] = be.y; nrm[i][4 * g + 2] = be.z; nrm[i][4 * g + 3] = be.w;
                }
        }
        // GDN contraction: the squares of k-step ks + 1 (VALU) and its gamma'_hi fragments (LDS) are prepared before the 12 MFMAs of
        // k-step ks are issued; gamma'_lo is requested at the start of its k-step and used last
        {
            h16x8 gh[2][4], gl[4], fq[2][2];
            auto prep = [&](int set, int ks) {
                const int si = ks >> 1, so = (ks & 1) * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    gh[set][i] = *(const h16x8*)(gl_hi + fa(ks) + i * 8192);
                }
                uint32_t qh[4], ql[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sq_scale = OUT1 ? 1.f : H16_SQ_SCALE;
                    float s0, s1;
                    if constexpr (DYN_SQ) {
                        const float u0 = acc[si][so + 2 * e] * sq_c, u1 = acc[si][so + 2 * e + 1] * sq_c;
                        s0 = u0 * u0; s1 = u1 * u1;
                    } else {
                        s0 = acc[si][so + 2 * e] * acc[si][so + 2 * e] * sq_scale; s1 = acc[si][so + 2 * e + 1] * acc[si][so + 2 * e + 1] * sq_scale;
                    }
                    split2(s0, s1, qh[e], ql[e]);
                }
                fq[set][0] = __builtin_bit_cast(h16x8, u32x4{qh[0], qh[1], qh[2], qh[3]});
                fq[set][1] = __builtin_bit_cast(h16x8, u32x4{ql[0], ql[1], ql[2], ql[3]});
            };
            prep(0, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if constexpr (!OUT1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        gl[i] = *(const h16x8*)(gl_lo + fa(ks) + i * 8192);
                    }
                }
                if (ks + 1 < 8) prep((ks + 1) & 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (DYN_SQ && ks == 0) {      // the accumulators start at zero: an inline-constant C operand, no 64 moves per tile
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        nrm[i] = mfma_32x32x16_h16(gh[ks & 1][i], fq[ks & 1][0], z, 0, 0, 0);
                    } else nrm[i] = mfma_32x32x16_h16(gh[ks & 1][i], fq[ks & 1][0], nrm[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) nrm[i] = mfma_32x32x16_h16(gh[ks & 1][i], fq[ks & 1][1], nrm[i], 0, 0, 0);
                if constexpr (!OUT1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) nrm[i] = mfma_32x32x16_h16(gl[i], fq[ks & 1][0], nrm[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        N2W_T(4);
        N2W_T(5);
        // y = v * rsqrt(nrm) (GDN) / v * sqrt(nrm) (IGDN) in fp32, kept in acc
        if constexpr (DYN_SQ) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 be = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(beta_rs, (int)(pofs + (uint32_t)((i * 32 + 8 * g) * 4)), 0, 0));
                    nrm[i][4 * g] = fmaf(nrm[i][4 * g], sq_inv, be.x); nrm[i][4 * g + 1] = fmaf(nrm[i][4 * g + 1], sq_inv, be.y);
                    nrm[i][4 * g + 2] = fmaf(nrm[i][4 * g + 2], sq_inv, be.z); nrm[i][4 * g + 3] = fmaf(nrm[i][4 * g + 3], sq_inv, be.w);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] *= INV ? __builtin_amdgcn_sqrtf(nrm[i][r]) : __builtin_amdgcn_rsqf(nrm[i][r]);
        N2W_T(6);
        if constexpr (!EARLY_REQ) {
            // the next tile's rows are requested here: `raw` (48 registers) is then not live across the MFMA phases, and the loads have
            // the staging passes -- and the other wave of the SIMD -- to come back
            __builtin_amdgcn_sched_barrier(0);
            request(tile + tstride < ntiles ? tile + tstride : tile);      // unconditional: a conditional update would keep the OLD `raw` live across
                                                                           // both MFMA phases (48 registers; the allocator spilled them)
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (int64_t)b * a.ys_b), 0, (int)POISON, 0x00020000);
        // four passes through the 4 KB of wave-private staging: (hi | lo) x (channels 0..63 | 64..127), all 32 pixels of the tile, 128
        // bytes (one cache line) per pixel and pass.  Every lane takes part in every pass, the hi passes cost one pack per channel
        // pair, only the lo passes form the residual.
#pragma unroll
        for (int pass = 0; pass < (OUT1 ? 2 : 4); ++pass) {
            const int half = pass >> 1, cg = pass & 1;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = cg * 2 + ii, cl = ii * 32 + 8 * g + 4 * fh;              // channel inside this pass's 64
                    uint32_t h0 = pack_h2(acc[i][4 * g], acc[i][4 * g + 1]), h1 = pack_h2(acc[i][4 * g + 2], acc[i][4 * g + 3]);
                    if (half) {
                        h0 = pack_h2(acc[i][4 * g] - h2f_lo(h0), acc[i][4 * g + 1] - h2f_hi(h0));
                        h1 = pack_h2(acc[i][4 * g + 2] - h2f_lo(h1), acc[i][4 * g + 3] - h2f_hi(h1));
                    }
                    *(u32x2*)(os + frow * 128 + (((cl >> 3) ^ (frow & 7)) << 4) + (cl & 7) * 2) = u32x2{h0, h1};
                }
            // the staging rows are written as 8-byte and read as 16-byte vectors: the compiler must not reorder them on type-based
            // alias grounds (it did: every other pixel came out with the previous pass's data) -- compiler barriers; the LDS itself
            // executes a wave's accesses in order
            asm volatile("" ::: "memory");
            N2W_T(8 + pass * 3);
            u32x4 rowv[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pr = it * 8 + (lane >> 3);
                rowv[it] = *(const u32x4*)(os + pr * 128 + (((lane & 7) ^ (pr & 7)) << 4));
            }
            asm volatile("" ::: "memory");
            N2W_T(9 + pass * 3);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int y2 = ty * 2 + (it >> 1), x2 = tx * 16 + (it & 1) * 8;
                const int so = (y2 * (int)a.ys_y + x2 * (int)a.ys_x + half * 128 + cg * 64) * 2;                       // scalar
                const bool ok = y2 < a.Ho && x2 + (lane >> 3) < a.Wo;
                // tile offset in the VGPR offset, soffset = 0: see the store-hazard note in sconv_n2w_gdn_fast_kernel
                __builtin_amdgcn_raw_buffer_store_b128(rowv[it], yr, (int)(ok ? st_lane + (uint32_t)so : POISON), 0, 0);
            }
            N2W_T(10 + pass * 3);
        }
        N2W_T(7);
    }
}

// LDS images of the kernel above, built once per weight update: [w_hi | w_lo | gamma'_hi | gamma'_lo], 32 KB each.
//   conv:   row co, slot r = ci*5 + ky (r < 15) at position r ^ (co & 15), 8 values = kx 0..4, 0, 0, 0
//   gamma': row i (output channel), slot q = 2 ks + h at position q ^ (i & 15), 8 values = gamma'[i][16 ks + 4 h + 0..3],
//           gamma'[i][16 ks + 8 + 4 h + 0..3] -- the order in which a lane of the GDN contraction holds its squares
// ``shaped``: the hi half of a conv weight is its 16-bit rounding WITH error feedback over the 25 taps of its (cout, cin) pair (serpentine
// walk, as pack_weight_shaped_kernel) -- for the two-product form of the kernel (OUT1), which multiplies w_hi only.
__global__ void n2w_hilo_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma, unsigned char* __restrict__ img, int shaped,
                                     float wscale = 1.f) {
    // the single-output form (shaped != 0) works on conv values scaled by sqrt(H16_SQ_SCALE) (a power of two: exact), so that their squares
    // are the scaled squares without a multiply per value: conv weights x that factor, gamma' plain (scale and unscale cancel), bias and
    // beta' scaled inside the kernel
    // wscale (a power of two; pair form only): the conv weights go in as (w * wscale)_hi | (w * wscale)_lo so that the lo half of a small weight
    // is a normal half; the caller passes bias * wscale and beta' * wscale^2 -- GDN's output v / sqrt(beta' + sum gamma' v^2) does not change
    const float wsc = shaped ? H16_SQ_ROOT : wscale, gsc = shaped ? 1.f : H16_SQ_UNSCALE;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4096) return;
    float v[8];
    int row, pos;
    if (idx < 2048) {
        row = idx >> 4;
        const int r = idx & 15, ci = r / 5, ky = r % 5;
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) v[kx] = (r < 15 && kx < 5) ? w[((row * 3 + ci) * 5 + ky) * 5 + kx] * wsc : 0.f;
        if (shaped && r < 15) {
            // replay the walk of this (cout, cin) pair up to row ky: the error carried into it, then this row's five values
            const float* src = w + (row * 3 + ci) * 25;
            float e = 0.f;
            for (int yy = 0; yy <= ky; ++yy)
                for (int j = 0; j < 5; ++j) {
                    const int kx = (yy & 1) ? 4 - j : j;
                    const float tgt = src[yy * 5 + kx] * wsc + e;
                    const float q = h2f(f2h(tgt));
                    e = tgt - q;
                    if (yy == ky) v[kx] = q;
                }
        }
        pos = r ^ (row & 15);
    } else {
        const int j = idx - 2048, q = j & 15, ks = q >> 1, h = q & 1;
        row = j >> 4;
        const float ped = 1.0f / 68719476736.0f, gb = 1.0f / 262144.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * ks + 4 * h + (e & 3) + (e >> 2) * 8;
            const float t = fmaxf(gamma[row * 128 + c], gb);
            v[e] = (t * t - ped) * gsc;
        }
        pos = q ^ (row & 15);
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2(v[2 * e], v[2 * e + 1], hi[e], lo[e]);
    unsigned char* dst = img + (idx < 2048 ? 0 : 65536) + (row * 16 + pos) * 16;
    *(u32x4*)dst = u32x4{hi[0], hi[1], hi[2], hi[3]};
    *(u32x4*)(dst + 32768) = u32x4{lo[0], lo[1], lo[2], lo[3]};
}

}  // namespace

#ifdef N2W_DBG
extern "C" int hesic_debug_n2w_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_n2w_dbg), sizeof(long long) * 32); }
#endif

extern "C" int hesic_sconv_pack_weight_image_hilo(const float* w, const float* gamma, void* image, void* stream) {
    HESIC_CHECK_ARG(w && gamma && image, "sconv_pack_weight_image_hilo: null pointer");
    hipLaunchKernelGGL(n2w_hilo_pack_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, gamma, (unsigned char*)image, 0);
    HESIC_LAUNCH_RETURN("sconv_pack_weight_image_hilo");
}

extern "C" int hesic_sconv_pack_weight_image_hilo_scaled(const float* w, const float* gamma, float wscale, void* image, void* stream) {
    HESIC_CHECK_ARG(w && gamma && image && wscale > 0.f && wscale < 3.0e38f, "sconv_pack_weight_image_hilo_scaled: null pointer or bad scale");
    hipLaunchKernelGGL(n2w_hilo_pack_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, gamma, (unsigned char*)image, 0, wscale);
    HESIC_LAUNCH_RETURN("sconv_pack_weight_image_hilo_scaled");
}

extern "C" int hesic_sconv_pack_weight_image_hilo_out1(const float* w, const float* gamma, void* image, void* stream) {
    HESIC_CHECK_ARG(w && gamma && image, "sconv_pack_weight_image_hilo_out1: null pointer");
    hipLaunchKernelGGL(n2w_hilo_pack_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, gamma, (unsigned char*)image, 1);
    HESIC_LAUNCH_RETURN("sconv_pack_weight_image_hilo_out1");
}

static int n2w_gdn_hilo_launch(const hesic_sconv_desc* d, const float* x, const void* image_hilo, const float* bias,
                               const float* beta_packed, int inverse, void* y_hilo, int out1, void* stream) {
    HESIC_CHECK_ARG(d && x && image_hilo && beta_packed && y_hilo, "sconv2d_gdn_forward_hilo: null pointer");
    HESIC_CHECK_ARG(!d->transposed && d->Cin == 3 && d->Cout == 128 && d->KH == 5 && d->KW == 5 && d->stride == 2 && d->pad == 2 &&
                        d->x_dtype == HESIC_F32 && d->y_dtype == HESIC_H16 && d->act == HESIC_ACT_NONE && d->ys_c == 1 && d->ys_x >= (out1 ? 128 : 256) &&
                        (d->ys_x % 8) == 0 && (d->ys_y % 8) == 0 && (d->ys_b % 8) == 0,
                    "sconv2d_gdn_forward_hilo: built for the 3 -> 128 5x5 stride-2 stage, fp32 image in, [hi | lo] bf16 NHWC out (pixel stride >= 256)");
    HESIC_CHECK_ARG(d->Ho == (d->H + 4 - 5) / 2 + 1 && d->Wo == (d->W + 4 - 5) / 2 + 1, "sconv2d_gdn_forward_hilo: output size does not match");
    const int64_t tiles = (int64_t)((d->Wo + 15) / 16) * ((d->Ho + 1) / 2) * d->B;
    HESIC_CHECK_ARG(d->xs_x == 1 && d->W % 2 == 0 && tiles < (1ll << 30) && d->xs_c >= 0 && d->xs_y >= 0 &&
                        (2 * d->xs_c + (int64_t)(d->H + 4) * d->xs_y + d->W) * 4 < (1ll << 31) &&
                        ((int64_t)d->Ho * d->ys_y + (int64_t)d->Wo * d->ys_x) * 2 < (1ll << 31),
                    "sconv2d_gdn_forward_hilo: needs fp32 planes with unit pixel stride, an even width and 32-bit offsets inside one image");
    HArgs a;
    a.x = x; a.img = (const unsigned char*)image_hilo; a.bias = bias; a.beta = beta_packed; a.y = (h16_t*)y_hilo;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
    a.xs_b = d->xs_b; a.xs_c = d->xs_c; a.xs_y = d->xs_y; a.ys_b = d->ys_b; a.ys_y = d->ys_y; a.ys_x = d->ys_x;
    a.fd_tx = make_fastdiv((uint32_t)((d->Wo + 15) / 16)); a.fd_ty = make_fastdiv((uint32_t)((d->Ho + 1) / 2));
    const size_t lds = 160 * 1024;
    const unsigned grid = (unsigned)((tiles + 7) / 8 < 256 ? (tiles + 7) / 8 : 256);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)n2w_gdn_hilo_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)n2w_gdn_hilo_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)n2w_gdn_hilo_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)n2w_gdn_hilo_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const dim3 g(grid), blk(512);
    hipStream_t st = (hipStream_t)stream;
    if (out1) {
        if (inverse) hipLaunchKernelGGL((n2w_gdn_hilo_kernel<1, 1>), g, blk, lds, st, a);
        else hipLaunchKernelGGL((n2w_gdn_hilo_kernel<0, 1>), g, blk, lds, st, a);
    } else {
        if (inverse) hipLaunchKernelGGL((n2w_gdn_hilo_kernel<1, 0>), g, blk, lds, st, a);
        else hipLaunchKernelGGL((n2w_gdn_hilo_kernel<0, 0>), g, blk, lds, st, a);
    }
    HESIC_LAUNCH_RETURN("sconv2d_gdn_forward_hilo");
}

extern "C" int hesic_sconv2d_gdn_forward_hilo(const hesic_sconv_desc* d, const float* x, const void* image_hilo, const float* bias,
                                              const float* beta_packed, int inverse, void* y_hilo, void* stream) {
    return n2w_gdn_hilo_launch(d, x, image_hilo, bias, beta_packed, inverse, y_hilo, 0, stream);
}

extern "C" int hesic_sconv2d_gdn_forward_hilo_out1(const hesic_sconv_desc* d, const float* x, const void* image_hilo, const float* bias,
                                                   const float* beta_packed, int inverse, void* y, void* stream) {
    return n2w_gdn_hilo_launch(d, x, image_hilo, bias, beta_packed, inverse, y, 1, stream);
}
