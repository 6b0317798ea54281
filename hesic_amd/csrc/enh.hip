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
// Round 6: the ROWS of the weight fragments are permuted (fragment row 8 j + 4 hh + e holds cout 16 hh + 4 j + e), so a lane's sixteen
// accumulators are the couts 16 h .. 16 h + 15 of ONE pixel: 32 contiguous bytes of the NHWC row.  Residuals are read and outputs stored as two
// 16-byte pieces per lane (a wave instruction covers 2 KB of consecutive pixels), the intermediate halo is written as two 16-byte chunks per
// lane -- no fp32 turn-round through LDS any more (rounds 1 - 5: 4 ds_write_b128 + 4 ds_read_b128 and ~80 VALU instructions per 32 pixels;
// 18 KB of LDS per block).  Same sums per value: bit-identical.
// HBM-bound by construction: 64 B in + 64 B out per pixel (134 + 134 MB per layer at B=8 512x512).
#include "common.h"

namespace {

struct C32Args {
    const h16_t* x; const float* w; const float* bias; const void* res1; const void* res2; void* y;
    int B, H, W, Cout, act, tiles_x, tiles_y;
    FastDiv fd_tx, fd_ty;
};

constexpr int TH = 16, TW = 32, HW_ = TW + 2, HH = TH + 2, HPIX = HH * HW_;     // halo: 18 x 34 pixels of 64 bytes


// Next-tile halo pieces through inline-asm buffer loads the compiler cannot see: tracked loads pending at the head of the row-group loop
// (which has stores in it) made it flush them there -- `s_waitcnt vmcnt(0)` right behind the request, the whole round trip exposed once per
// tile.  The pieces are waited for by hand at the top of the next tile (c32_wait_pieces: vmcnt(0) -- the younger stores of partial tiles
// are predicated, their number is not a constant).
__device__ __forceinline__ u32x4 c32_rsrc(const void* p) {
    const uint64_t a_ = (uint64_t)p;
    return u32x4{(uint32_t)a_, (uint32_t)(a_ >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
}
__device__ __forceinline__ void c32_load_piece(u32x4& dst, int off, const u32x4& rs) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(off), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void c32_wait_pieces(u32x4 (&pc)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < N; ++u) asm volatile("" : "+v"(pc[u]));       // the values are defined from here on
}

// fragment row -> output channel (see the header): row 8 j + 4 hh + e holds cout 16 hh + 4 j + e
__device__ __forceinline__ int c32_cout_of_row(int row) { return 16 * ((row >> 2) & 1) + 4 * (row >> 3) + (row & 3); }

// act(v) with the activation known at compile time where the caller knows it (LK: LeakyReLU as max(v, 0.01 v) -- two instructions instead of the
// multiply / compare / two selects of the run-time form; same value for every finite v)
template <bool LK>
__device__ __forceinline__ float c32_act(float v, int act) {
    if constexpr (LK) return fmaxf(v, 0.01f * v);
    else return apply_act(v, act);
}

// act(acc + bias) + res1 + res2 of a lane's sixteen couts -> two 16-byte stores.  NR = how many residual tensors take part (a missing one is not
// added as + 0: one instruction per value).
template <int NR, bool LK>
__device__ __forceinline__ void c32_store_row(h16_t* dst, const f32x16& acc, const float (&bv)[16], int act, const u32x4 (&r1v)[2], const u32x4 (&r2v)[2]) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c32_act<LK>(acc[8 * it + e] + bv[8 * it + e], act);
        const uint32_t ra[4] = {r1v[it].x, r1v[it].y, r1v[it].z, r1v[it].w}, rb[4] = {r2v[it].x, r2v[it].y, r2v[it].z, r2v[it].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (NR >= 2) {
                v[2 * e] += h2f_lo(ra[e]) + h2f_lo(rb[e]);
                v[2 * e + 1] += h2f_hi(ra[e]) + h2f_hi(rb[e]);
            } else if constexpr (NR == 1) {
                v[2 * e] += h2f_lo(ra[e]);
                v[2 * e + 1] += h2f_hi(ra[e]);
            }
        }
        *(u32x4*)(dst + 8 * it) = u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
    }
}

// NRES (MODE 0): how many residual tensors the launch carries, at compile time -- with run-time null checks the row loop of a plain layer
// still ended every row group in `s_waitcnt vmcnt(0)` (for loads it never issues), i.e. waited out the next tile's halo prefetch and
// the previous row group's stores.
template <int MODE, int NRES = 2>      // 0: 32 couts, bf16 NHWC out (+ bf16 NHWC residuals); 1: <= 4 couts, fp32 planar out (+ fp32 planar residual)
__global__ __launch_bounds__(256, 2) void c32_conv3x3_kernel(const C32Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char halo[HPIX * 64];
    constexpr uint32_t POISON = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    // weights -> 18 A fragments: lane (fragment row p = cout C32_COUT_OF_ROW(p), half h) holds channels k*16 + h*8 + [0,8) of tap t.  The (Cout,32,3,3) fp32
    // tensor comes in through LDS with coalesced loads (row pitch 289 floats: the per-lane gathers below hit 32 different
    // banks); gathering it straight from global memory cost every block ~9000 scattered cache-line requests.
    h16x8 wf[9][2];
    {
        float* wst = (float*)halo;                               // 32 x 289 floats = 37 KB <= the halo buffer
        const int nw = a.Cout * 288;
        for (int i = tid; i < 32 * 288; i += 256) {
            const int co = i / 288, r = i - co * 288;
            wst[co * 289 + r] = i < nw ? a.w[i] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = wst[c32_cout_of_row(p) * 289 + (k * 16 + h * 8 + e) * 9 + t];
                wf[t][k] = __builtin_bit_cast(h16x8, u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])});
            }
        __syncthreads();
    }
    float bv[16];          // accumulator r of this lane = cout 16 h + r
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = 16 * h + r;
        bv[r] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;
    }
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    // halo -> registers: 16-byte piece i = (pixel i >> 2, slot i & 3) holds channel chunk slot ^ ((pixel >> 2) & 3); pixels
    // outside the image are poisoned offsets (zeros = the conv's zero padding).  The next tile's pieces are requested as soon
    // as this tile's are in LDS, so their HBM latency runs under the MFMA phase.
    u32x4 pc[10];
    // MODE 1, NRES 0: at most three couts, the image this conv refines requested with the halo (below); NRES 1: four couts, in-loop loads
    constexpr bool PRE = MODE == 1 && NRES == 0;
    [[maybe_unused]] float rpn[PRE ? TH / 4 : 1][3];
    int cb = 0, cty = 0, ctx = 0;
    auto request = [&](int tile) {
        const uint32_t q = fdiv((uint32_t)tile, a.fd_tx);
        ctx = tile - (int)q * a.tiles_x;
        cb = (int)fdiv(q, a.fd_ty);
        cty = (int)q - cb * a.tiles_y;
        const int y0 = cty * TH - 1, x0 = ctx * TW - 1;
        const u32x4 xr = c32_rsrc(a.x + (int64_t)cb * a.H * a.W * 32);
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int i = tid + 256 * u;
            const int hp = i >> 2, slot = i & 3;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = hp < HPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int chunk = slot ^ ((hp >> 2) & 3);
            c32_load_piece(pc[u], ok ? (int)(((iy * a.W + ix) * 32 + chunk * 8) * 2) : (int)POISON, xr);
        }
        if constexpr (PRE) {
            // the fp32 image this conv refines (res1, planar): this lane's values for its four row groups of the tile, requested WITH the halo.
            // As ordinary loads inside the row loop they were the newest vector-memory operations at every row group's end, and the wait
            // for them -- the compiler cannot see the halo requests above, so it emitted vmcnt(0) -- waited out the next tile's whole halo
            // once per row group: 80 us for a launch whose bytes take 35.
            const u32x4 rr = c32_rsrc((const float*)a.res1 + (int64_t)cb * a.Cout * a.H * a.W);
#pragma unroll
            for (int rq = 0; rq < TH / 4; ++rq)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int y = cty * TH + wave + 4 * rq, x = ctx * TW + p;
                    const bool ok = a.res1 && h == 0 && c < a.Cout && y < a.H && x < a.W;
                    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(rpn[rq][c]) : "v"(ok ? ((c * a.H + y) * a.W + x) * 4 : (int)POISON), "s"(rr) : "memory");
                }
        }
    };
    int tile = blockIdx.x;
    request(tile < ntiles ? tile : 0);
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = cb, ty = cty, tx = ctx;                     // of the tile whose halo is in `pc`
        c32_wait_pieces(pc);
        [[maybe_unused]] float rpc[PRE ? TH / 4 : 1][3];
        if constexpr (PRE) {
#pragma unroll
            for (int rq = 0; rq < TH / 4; ++rq)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    asm volatile("" : "+v"(rpn[rq][c]));          // defined from here on (c32_wait_pieces waited for every load)
                    rpc[rq][c] = rpn[rq][c];
                }
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int i = tid + 256 * u;
            if (i < HPIX * 4) *(u32x4*)(halo + i * 16) = pc[u];
        }
        __syncthreads();
        request(tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile);      // unconditional (asm outputs must be defined on every path)
#pragma unroll 1
        for (int rq = 0; rq < TH / 4; ++rq) {
            const int yl = wave + 4 * rq;
            const int y = ty * TH + yl, x = tx * TW + p;
            const bool live = y < a.H && x < a.W;
            // a lane's 32 output bytes (couts 16 h .. 16 h + 15 of pixel p): residuals are requested here, before the MFMAs
            const int64_t o = (((int64_t)b * a.H + y) * a.W + x) * 32 + 16 * h;
            u32x4 r1v[2], r2v[2];
            float rp[4];
            if constexpr (MODE == 0) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    r1v[it] = (NRES >= 1 && live) ? *(const u32x4*)((const h16_t*)a.res1 + o + 8 * it) : u32x4{0u, 0u, 0u, 0u};
                    r2v[it] = (NRES >= 2 && live) ? *(const u32x4*)((const h16_t*)a.res2 + o + 8 * it) : u32x4{0u, 0u, 0u, 0u};
                }
            } else if constexpr (PRE) {
                // (the row loop is `unroll 1`: the row group's values are selected, not indexed)
#pragma unroll
                for (int c = 0; c < 3; ++c) rp[c] = rq == 0 ? rpc[0][c] : rq == 1 ? rpc[1][c] : rq == 2 ? rpc[2][c] : rpc[3][c];
                rp[3] = 0.f;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    rp[c] = (a.res1 && live && h == 0 && c < a.Cout) ? ((const float*)a.res1)[(((int64_t)b * a.Cout + c) * a.H + y) * a.W + x] : 0.f;
            }
            f32x16 acc, acc1;                                      // two chains: consecutive MFMAs never wait on each other
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hp = (yl + t / 3) * HW_ + p + t % 3;
                const unsigned char* row = halo + hp * 64;
                const int sw = (hp >> 2) & 3;
                const h16x8 xf0 = *(const h16x8*)(row + ((h ^ sw) << 4)), xf1 = *(const h16x8*)(row + (((2 + h) ^ sw) << 4));
                acc = mfma_32x32x16_h16(wf[t][0], xf0, acc, 0, 0, 0);
                acc1 = mfma_32x32x16_h16(wf[t][1], xf1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
            // D[fragment row][pixel]: lane holds pixel p, accumulator r = cout 16 h + r (permuted weight rows)
            if constexpr (MODE == 0) {
                if (live) c32_store_row<NRES, false>((h16_t*)a.y + o, acc, bv, a.act, r1v, r2v);
            } else {
                if (live && h == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < a.Cout)
                            ((float*)a.y)[(((int64_t)b * a.Cout + c) * a.H + y) * a.W + x] = apply_act(acc[c] + bv[c], a.act) + rp[c];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- the 6 -> 32 input conv, straight from the two images
// Enhancement.conv1 on torch.cat((x, x_another_warp), 1) (newnet1.py:300-301).  Rounds 3 - 5: pack_images_c32 wrote the two planar fp32 images as
// channels 0..5 of a zero-padded 32-channel map (50 us, 134 MB out) and c32_conv3x3_kernel read it back with a zero-padded weight (72 us, 134 MB in,
// 18 MFMAs per 32 pixels of which 9 multiplied zeros).  Here the halo is built from the six image planes directly (fp32 -> 16-bit, 32 bytes per pixel:
// chunk 0 = the six channels + two zeros, chunk 1 = zeros; chunk c in slot c ^ ((pixel >> 3) & 1)), nine K = 16 MFMAs per group: 50 MB in, 134 MB out.
// Same products and the same order of the non-zero ones: bit-identical to the two-launch route.
struct C6Args {
    const float* xa; const float* xb; const float* w; const float* bias; void* y;
    int B, H, W, act, tiles_x, tiles_y;
    FastDiv fd_tx, fd_ty;
};

__global__ __launch_bounds__(256, 2) void c32_conv6_kernel(const C6Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char halo[HPIX * 32];
    __shared__ float wst[32 * 55];
    constexpr uint32_t POISON = 0x80000000u;
    constexpr int NPX = (HPIX + 255) / 256;              // halo pixels per thread: 3
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    h16x8 wf[9];
    {
        for (int i = tid; i < 32 * 54; i += 256) wst[(i / 54) * 55 + i % 54] = a.w[i];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (h == 0 && e < 6) ? wst[c32_cout_of_row(p) * 55 + e * 9 + t] : 0.f;
            wf[t] = __builtin_bit_cast(h16x8, u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])});
        }
    }
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = a.bias ? a.bias[16 * h + r] : 0.f;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int64_t plane = (int64_t)a.H * a.W;
    float pv[NPX][6];
    int cb = 0, cty = 0, ctx = 0;
    auto request = [&](int tile) {
        const uint32_t q = fdiv((uint32_t)tile, a.fd_tx);
        ctx = tile - (int)q * a.tiles_x;
        cb = (int)fdiv(q, a.fd_ty);
        cty = (int)q - cb * a.tiles_y;
        const int y0 = cty * TH - 1, x0 = ctx * TW - 1;
        const u32x4 ra = c32_rsrc(a.xa + (int64_t)cb * 3 * plane), rb = c32_rsrc(a.xb + (int64_t)cb * 3 * plane);
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const int hp = tid + 256 * u;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = hp < HPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int o = (iy * a.W + ix) * 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(pv[u][c]) : "v"(ok ? o + (int)(c * plane * 4) : (int)POISON), "s"(ra) : "memory");
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(pv[u][3 + c]) : "v"(ok ? o + (int)(c * plane * 4) : (int)POISON), "s"(rb) : "memory");
            }
        }
    };
    int tile = blockIdx.x;
    request(tile < ntiles ? tile : 0);
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = cb, ty = cty, tx = ctx;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
#pragma unroll
            for (int c = 0; c < 6; ++c) asm volatile("" : "+v"(pv[u][c]));
            const int hp = tid + 256 * u;
            if (hp < HPIX) {
                const int sw = (hp >> 3) & 1;
                *(u32x4*)(halo + hp * 32 + (sw << 4)) = u32x4{pack_h2(pv[u][0], pv[u][1]), pack_h2(pv[u][2], pv[u][3]), pack_h2(pv[u][4], pv[u][5]), 0u};
                *(u32x4*)(halo + hp * 32 + ((1 ^ sw) << 4)) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();
        request(tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile);
#pragma unroll 1
        for (int rq = 0; rq < TH / 4; ++rq) {
            const int yl = wave + 4 * rq;
            const int y = ty * TH + yl, x = tx * TW + p;
            const bool live = y < a.H && x < a.W;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hp = (yl + t / 3) * HW_ + p + t % 3;
                const h16x8 xf = *(const h16x8*)(halo + hp * 32 + ((h ^ ((hp >> 3) & 1)) << 4));
                acc = mfma_32x32x16_h16(wf[t], xf, acc, 0, 0, 0);
            }
            const u32x4 none[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            if (live) c32_store_row<0, false>((h16_t*)a.y + (((int64_t)b * a.H + y) * a.W + x) * 32 + 16 * h, acc, bv, a.act, none, none);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- a whole ResidualBlock per launch
// leaky(conv2(leaky(conv1(x)))) + x (+ the Enhancement_Block's outer skip), compressai/layers/layers.py:125-147: two launches of the kernel above move
// 64 B in + 64 B out per pixel TWICE; here the intermediate map never leaves the CU.  One block of EIGHT waves per CU, two roles, software-pipelined over
// the block's tiles: producer waves run conv1 of tile s into an intermediate region in LDS (bias, LeakyReLU, 16-bit, ZERO outside the image: conv2 pads
// the intermediate map, not conv1's extrapolation), consumer waves conv2 of tile s - 1 from the other region + identity + outer skip.
// (Rounds 3 - 5 ran this on 16 x 32 tiles in 32-pixel groups with the 32 x 32 x 16 MFMA and an fp32 turn-round of the output through LDS: 129 / 144 us
// per launch without / with outer skip at B=8 512^2.  Ablations of that kernel -- profiles/r06_en_ablation.txt: full 157 us on random data, no MFMAs 107,
// no fragment reads 115, neither 92, no identity loads / stores 121, skeleton 40, no producer 95, no consumer 111: phases and roles ADD -- and its counters
// (r06_a_pmc_sq_en.json: MFMA pipe 30 % busy, 14 VALU per MFMA, 27 % of LDS cycles conflicts) led to the form below; the old kernel is kept as
// profiles/experiments/r05_c32_resblock_tile16x32.patch.)
struct RBArgs {
    const h16_t* x; const float* w1; const float* b1; const float* w2; const float* b2; const void* res2; void* y;
    int B, H, W, act, tiles_x, tiles_y;
    FastDiv fd_tx, fd_ty;
};

// Ablation builds (profiles/scripts/en_ablation.sh; never in the shipped libraries): -DRB_ABL=<bits>  1: no MFMAs (fragment reads stay), 2: no fragment
// reads, 4: the consumers' identity / outer-skip loads and the output stores dropped, 8: no producer work at all, 16: no consumer work at all.
#ifndef RB_ABL
#define RB_ABL 0
#endif

// ---------------------------------------------------------------------------------------------- ResidualBlock, row-rolling form (round 6)
// What bounded the kernel above (profiles/r06_en_ablation.txt, r06_a_pmc_sq_en.json): every 32-pixel group re-reads its nine shifted fragments from
// LDS -- 648 ds_read_b128 per 16 x 32 tile, ~5.2 k LDS cycles per CU and tile, as many as the matrix pipe needs per SIMD --, the epilogues cost as many
// VALU cycles as the MFMAs matrix cycles, and the phases of the two lock-stepped roles add up.  Here a wave owns a COLUMN of 32 pixels and walks down
// its rows:
//   * the six fragments of an input row (three tap columns x two 16-pixel halves) are read ONCE and serve the three output rows that touch it
//     (6 reads per 36 MFMAs instead of 36), four accumulator sets rotate: output row r accumulates while input rows r .. r + 2 pass, its epilogue
//     stands one input row later, in the same scheduling region as MFMAs that do not depend on it;
//   * v_mfma_f32_16x16x32 (K = 32 = all input channels of a tap) with the weight rows permuted so that a lane ends up with EIGHT CONSECUTIVE couts of
//     one pixel: fragment row 4 g + e of cout half c2 holds cout 8 g + 4 c2 + e, lane (pixel px, g) of the D tile then owns couts 8 g .. 8 g + 7 =
//     one 16-byte chunk of the NHWC row, four lanes a whole 64-byte pixel, a wave instruction 1 KB of consecutive pixels -- identity / outer-skip
//     loads, output stores and the intermediate region's LDS writes need no turn-round (the 32 x 32 x 16 form leaves a lane 4 + 4 + 4 + 4 couts;
//     with permuted rows 2 x 16 bytes at a 64-byte stride: measured 73 us of a 159 us launch in loads / stores);
//   * the next tile's input halo lands by LDS-DMA in a second buffer (no register holds it; rounds 3 - 5 parked it in 24 registers per lane);
//   * tile 14 x 30 output pixels: the intermediate region is 16 x 32 (a row = exactly one 32-pixel column group, four producer waves x four rows),
//     the input halo 18 x 34.  A pixel's 64 bytes sit at (row * width + col) * 64, channel chunk c in slot c ^ ((col >> 1) & 3): conflict-free for
//     the fragment reads of every tap column (brute-forced over the b128 lane groups) and for the 16-byte writes.
// Summation order per output value: taps in raster order, one MFMA per tap.  Not the order of c32_conv3x3_kernel (two K = 16 halves per tap): the
// ResidualBlock launch agrees with two launches of it to the last bit of the 16-bit result in all but ~1.5e-3 of the values (tested).
#ifdef R3_STAMP          /* cycle stamps of block 0 (profiles/scripts/en_stamps.py; never in the shipped libraries) */
__device__ unsigned long long* g_r3_stamp = nullptr;
#define R3_ST(slot) do { if (blockIdx.x == 0 && lane == 0 && s >= 2 && s < 6) g_r3_stamp[((s - 2) * 8 + wave) * 8 + (slot)] = clock64(); } while (0)
#else
#define R3_ST(slot) do {} while (0)
#endif
constexpr int R3_TH = 14, R3_TW = 30;
constexpr int R3_IH = R3_TH + 4, R3_IW = R3_TW + 4, R3_MH = R3_TH + 2, R3_MW = R3_TW + 2;
constexpr int R3_IPIX = R3_IH * R3_IW, R3_MPIX = R3_MH * R3_MW;              // 612, 512 pixels
constexpr int R3_NDMA = (R3_IPIX * 4 + 63) / 64;                             // 1 KB LDS-DMA instructions per input halo: 39 (the last one partial)
constexpr int R3_IBYTES = R3_NDMA * 1024, R3_MBYTES = (R3_MPIX + 2) * 64;    // (the consumers' dead columns 30, 31 read two pixels past a row)
constexpr int R3_LDS_DATA = 2 * R3_IBYTES + 2 * R3_MBYTES;                   // two input halos, two intermediate regions: 142 KB
constexpr int R3_LDS = R3_LDS_DATA + R3_NDMA * 64 * 2;                       // + the DMA pieces' halo positions (5 KB)
static_assert(R3_IBYTES >= 32 * 289 * 4, "the weight staging buffer lives in the input halo");

typedef f32x4 c32_acc_t[2][2];          // [16-pixel half q][cout half c2]: couts 8 g + 4 c2 + e of pixel 16 q + px

// R output rows of a 32-pixel column from R + 2 input rows at `src` (LDS, PITCH bytes per row).  Output row r accumulates in iterations r .. r + 2;
// fin(r, acc) stands in iteration r + 3 (the last one behind the loop), pre(r) -- its residual requests -- one iteration ahead of that.
template <int R, int PITCH, typename PRE, typename FIN>
__device__ __forceinline__ void c32_roll(const unsigned char* src, const int (&foff)[3][2], const h16x8 (&wf)[9][2], const f32x4 (&bias)[2], PRE&& pre, FIN&& fin) {
    c32_acc_t acc[R];
    h16x8 f[3][2];          // ONE set: the fragments of tap column kx are re-requested for the next input row right behind their last MFMA
    auto rd = [&](int i, int kx) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#if RB_ABL & 2
            f[kx][q] = wf[(i + kx) % 9][q];
#else
            f[kx][q] = *(const h16x8*)(src + i * PITCH + foff[kx][q]);
#endif
        }
    };
    rd(0, 0); rd(0, 1); rd(0, 2);
#pragma unroll
    for (int i = 0; i < R + 2; ++i) {
        if (i >= 3) { pre(i - 3, 0); fin(i - 3, acc[i - 3], 0); }
        if (i == R + 1) pre(R - 1, 1);          // the last row's requests one input row early: its epilogue stands behind the loop, with no MFMAs to cover a wait
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int ky = i - r;
                if (ky < 0 || ky > 2) continue;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // the first MFMA of a row starts from the bias (couts 8 g + 4 c2 + e): no add in the epilogue
#if RB_ABL & 1
                        asm volatile("" ::"v"(wf[ky * 3 + kx][c2]), "v"(f[kx][q]));
                        acc[r][q][c2] = (ky == 0 && kx == 0) ? bias[c2] : acc[r][q][c2];
#else
                        acc[r][q][c2] = mfma_16x16x32_h16(wf[ky * 3 + kx][c2], f[kx][q], (ky == 0 && kx == 0) ? bias[c2] : acc[r][q][c2], 0, 0, 0);
#endif
                    }
            }
            if (i + 1 < R + 2) rd(i + 1, kx);
        }
        __builtin_amdgcn_iglp_opt(0);            // (measured: 103.6 / 131.6 us per launch without / with outer skip against 108.5 / 135.1 without the hint)
        __builtin_amdgcn_sched_barrier(0);       // one input row per scheduling region (without it: 109.6 / 161.8 us)
    }
    fin(R - 1, acc[R - 1], 1);
}

// act(acc) + res1 + res2 of a lane's eight couts -> 16 bytes (the bias is already in the accumulators)
template <int NR, bool LK>
__device__ __forceinline__ u32x4 c32_pack8(const f32x4& a0, const f32x4& a1, int act, const u32x4& r1, const u32x4& r2) {
    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c32_act<LK>(v[e], act);
    const uint32_t ra[4] = {r1.x, r1.y, r1.z, r1.w}, rb[4] = {r2.x, r2.y, r2.z, r2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (NR >= 2) {
            v[2 * e] += h2f_lo(ra[e]) + h2f_lo(rb[e]);
            v[2 * e + 1] += h2f_hi(ra[e]) + h2f_hi(rb[e]);
        } else if constexpr (NR == 1) {
            v[2 * e] += h2f_lo(ra[e]);
            v[2 * e + 1] += h2f_hi(ra[e]);
        }
    }
    return u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
}

template <bool RES2, bool LK>
__global__ __launch_bounds__(512) void c32_resblock_r3_kernel(const RBArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char r3_smem[];
    unsigned char* hin0 = r3_smem;                             // two input halos 18 x 34 (39 KB each: 39 DMA instructions of 1 KB)
    unsigned char* hmid0 = r3_smem + 2 * R3_IBYTES;            // two intermediate regions 16 x 32 (+ 2 pixels)
    // the buffer-load-to-LDS builtin is not modelled as a store to LDS: let the buffers escape through an empty asm so that the "memory"-clobbering
    // waits below count as writers of them
    asm volatile("" ::"v"((__attribute__((address_space(3))) unsigned char*)r3_smem) : "memory");
    constexpr uint32_t POISON = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 0: producer (conv1), 1: consumer (conv2).  The consumers are waves 0 .. 3: the two waves of a SIMD share its VALU issue by age, and the
    // consumer -- epilogue with the residual adds, loads, stores -- is the stage's longer role (stamped: 9.8 k vs 6.5 k cycles with the roles the other way round)
    const int role = 1 - (wave >> 2), w4 = wave & 3;
    const int px = lane & 15, g = lane >> 4;                   // fragment / tile column, 8-channel group
    h16x8 wf[9][2];
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        float* wst = (float*)hin0;                               // 32 x 289 floats = 37 KB <= an input halo buffer
        const float* w = which ? a.w2 : a.w1;
        for (int i = tid; i < 32 * 288; i += 512) {
            const int co = i / 288, r = i - co * 288;
            wst[co * 289 + r] = w[i];
        }
        __syncthreads();
        if (role == which) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int co = 8 * (px >> 2) + 4 * c2 + (px & 3);          // fragment row px of cout half c2
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = wst[co * 289 + (8 * g + e) * 9 + t];
                    wf[t][c2] = __builtin_bit_cast(h16x8, u32x4{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])});
                }
        }
        __syncthreads();
    }
    f32x4 bias[2];        // a lane's couts 8 g + 4 c2 + e: the initial value of its accumulators
    {
        const float* bp = role ? a.b2 : a.b1;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) bias[c2] = bp ? *(const f32x4*)(bp + 8 * g + 4 * c2) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // per-lane constants: fragment reads (column 16 q + px + kx of a row, channel chunk g) and the producers' 16-byte writes (column 16 q + px)
    int foff[3][2], woff[2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = 16 * q + px + kx;
            foff[kx][q] = col * 64 + ((g ^ ((col >> 1) & 3)) << 4);
        }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int col = 16 * q + px;
        woff[q] = col * 64 + ((g ^ ((col >> 1) & 3)) << 4);
    }

    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int n_my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    int cb = 0, cty = 0, ctx = 0;
    // the input halo of `tile` into buffer `buf` by LDS-DMA: 16-byte piece i = (pixel i >> 2, slot i & 3) holds channel chunk slot ^ ((column >> 1) & 3);
    // a wave instruction lands 64 consecutive pieces (1 KB), PRODUCER wave w issues instructions w, w + 4, ...; pixels outside the image carry an out-of-range
    // offset (the buffer unit writes zeros = the conv's zero padding).  No register holds the tile while it is in flight.
    // halo position (row << 8 | column) of every 16-byte DMA piece, once per block, in LDS behind the buffers (5 KB): the division by the halo width
    // is not repeated per stage and no register carries the pieces' constants through the row loops
    constexpr int NISS = 4;          // the four PRODUCER waves issue the DMA in one burst at the stage top (measured: the pieces spread two per input row
                                     // over the producers' row loop 121 / 136 us against 114 / 137 on the same box; all eight waves issuing and awaiting it 113 / 147 us per launch
    const int jw = w4;               // without / with outer skip against 105 / 132 -- the consumers then wait for their stores at every stage top)
    constexpr int NPCS = (R3_NDMA + NISS - 1) / NISS;
    unsigned short* dpos = (unsigned short*)(r3_smem + R3_LDS_DATA);
    for (int i = tid; i < R3_NDMA * 64; i += 512) {
        const int hp = i >> 2;
        const int hy = hp / R3_IW, hx = hp - hy * R3_IW;
        dpos[i] = (unsigned short)(hp < R3_IPIX ? (hy << 8) | hx : 0x7f00);          // (a piece behind the halo: its row is never inside an image)
    }
    __syncthreads();
    auto request = [&](int tile, int buf) {
        const uint32_t q = fdiv((uint32_t)tile, a.fd_tx);
        ctx = tile - (int)q * a.tiles_x;
        cb = (int)fdiv(q, a.fd_ty);
        cty = (int)q - cb * a.tiles_y;
        const int y0 = cty * R3_TH - 2, x0 = ctx * R3_TW - 2;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)cb * a.H * a.W * 32), 0, (int)POISON, 0x00020000);
        unsigned char* dst = hin0 + buf * R3_IBYTES;
        const int base = (y0 * a.W + x0) * 64;
#pragma unroll
        for (int u = 0; u < NPCS; ++u) {
            const int j = jw + NISS * u;                         // wave-uniform
            if (j < R3_NDMA) {
                const int dp = dpos[j * 64 + lane], hy = dp >> 8, hx = dp & 255;
                const bool ok = (unsigned)(y0 + hy) < (unsigned)a.H && (unsigned)(x0 + hx) < (unsigned)a.W;
                const int rel = (hy * a.W + hx) * 64 + (((lane & 3) ^ ((hx >> 1) & 3)) << 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16,
                                                         ok ? base + rel : (int)POISON, 0, 0, 0);
            }
        }
    };
    // (every wave decodes the tile; only the producers move data: they have no other vector-memory traffic, so their `s_waitcnt vmcnt(0)` in front of
    // the stage barrier waits for the halo alone -- the consumers' stores drain under the next stage instead of in front of its barrier)
    auto decode = [&](int tile) {
        const uint32_t q = fdiv((uint32_t)tile, a.fd_tx);
        ctx = tile - (int)q * a.tiles_x;
        cb = (int)fdiv(q, a.fd_ty);
        cty = (int)q - cb * a.tiles_y;
    };
    if (role == 0) request((int)blockIdx.x, 0); else decode((int)blockIdx.x);
    int pb = 0, pty = 0, ptx = 0;          // tile of the previous stage (the consumer's)
#pragma unroll 1
    for (int s = 0; s <= n_my; ++s) {
        const int b = cb, ty = cty, tx = ctx;                     // tile s, whose halo lands in buffer s & 1
        const unsigned char* hin = hin0 + (s & 1) * R3_IBYTES;
        R3_ST(0);
        if (role == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of it
        R3_ST(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        R3_ST(2);
        if (s + 1 < n_my) {                                       // the other buffer: its readers finished a stage ago
            if (role == 0) request((int)blockIdx.x + (s + 1) * (int)gridDim.x, (s + 1) & 1);
            else decode((int)blockIdx.x + (s + 1) * (int)gridDim.x);
        }
        R3_ST(3);
        if (role == 0) {
            if (s < n_my && !(RB_ABL & 8)) {
                // intermediate rows 4 w4 .. 4 w4 + 3 (image rows ty * 14 - 1 + ...), columns = image tx * 30 - 1 + col; ZERO outside the image:
                // conv2 pads the intermediate map, not conv1's extrapolation
                unsigned char* hmid = hmid0 + (s & 1) * R3_MBYTES + (4 * w4) * (R3_MW * 64);
                const int ix0 = tx * R3_TW - 1 + px, iy0 = ty * R3_TH - 1 + 4 * w4;
                c32_roll<4, R3_IW * 64>(hin + (4 * w4) * (R3_IW * 64), foff, wf, bias, [](int, int) {}, [&](int r, const c32_acc_t& acc, int) {
                    const bool yin = (unsigned)(iy0 + r) < (unsigned)a.H;
                    const u32x4 none = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        u32x4 o = c32_pack8<0, LK>(acc[q][0], acc[q][1], a.act, none, none);
                        if (!(yin && (unsigned)(ix0 + 16 * q) < (unsigned)a.W)) o = none;
                        *(u32x4*)(hmid + r * (R3_MW * 64) + woff[q]) = o;
                    }
                });
            }
        } else if (s >= 1 && !(RB_ABL & 16)) {
            // output rows: waves 0, 1 take four (0 - 3, 4 - 7), waves 2, 3 three (8 - 10, 11 - 13)
            const int r0 = w4 < 2 ? 4 * w4 : 8 + 3 * (w4 - 2);
            const unsigned char* hmid = hmid0 + ((s - 1) & 1) * R3_MBYTES + r0 * (R3_MW * 64);
            const int x0 = ptx * R3_TW + px, y0 = pty * R3_TH + r0;
            const bool xok[2] = {x0 < a.W, px + 16 < R3_TW && x0 + 16 < a.W};
            // identity, outer skip and output through buffer descriptors of the image: a dead lane (outside the tile / the image) carries an
            // out-of-range offset -- loads return zeros, stores are dropped -- so the whole row loop stays ONE basic block per input row (exec-masked
            // loads / stores put a branch around each and the scheduler could not move the epilogue's VALU work under the MFMAs)
            const int64_t img = (int64_t)pb * a.H * a.W * 32;
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + img), 0, (int)POISON, 0x00020000);
            const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)((const h16_t*)(RES2 ? a.res2 : (const void*)a.x) + img), 0, (int)POISON, 0x00020000);
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)((h16_t*)a.y + img), 0, (int)POISON, 0x00020000);
            const int ob = ((y0 * a.W + x0) * 32 + 8 * g) * 2;               // byte offset of (row 0, half 0) inside the image
            u32x4 r1v[2][2], r2v[2][2];          // [set][16-pixel half]: set 1 = the last row's (requested early)
            auto pre = [&](int r, int set) {            // identity and outer skip of output row r: requested at the top of the input row whose MFMAs cover the wait
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bool lv = xok[q] && y0 + r < a.H;
                    const int o = lv ? ob + (r * a.W + 16 * q) * 64 : (int)POISON;
#if RB_ABL & 4
                    r1v[set][q] = r2v[set][q] = u32x4{(uint32_t)o, 0u, 0u, 0u};
#else
                    r1v[set][q] = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 0);          // the identity (L2: the producers read it a stage ago)
                    if constexpr (RES2) r2v[set][q] = __builtin_amdgcn_raw_buffer_load_b128(sr, o, 0, 0);
                    else r2v[set][q] = u32x4{0u, 0u, 0u, 0u};
#endif
                }
            };
            auto fin = [&](int r, const c32_acc_t& acc, int set) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bool lv = xok[q] && y0 + r < a.H;
                    const u32x4 o = c32_pack8<RES2 ? 2 : 1, LK>(acc[q][0], acc[q][1], a.act, r1v[set][q], r2v[set][q]);
#if RB_ABL & 4
                    __builtin_amdgcn_raw_buffer_store_b128(o, yr, (lv && o.x == 0x12345678u) ? ob + (r * a.W + 16 * q) * 64 : (int)POISON, 0, 0);
#else
                    __builtin_amdgcn_raw_buffer_store_b128(o, yr, lv ? ob + (r * a.W + 16 * q) * 64 : (int)POISON, 0, 0);
#endif
                }
            };
            if (w4 < 2) c32_roll<4, R3_MW * 64>(hmid, foff, wf, bias, pre, fin);
            else c32_roll<3, R3_MW * 64>(hmid, foff, wf, bias, pre, fin);
        }
        // end of stage: the producers' LDS writes are out (lgkmcnt), nobody waits for global stores here (__syncthreads would: vmcnt(0))
        R3_ST(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        R3_ST(5);
        pb = b; pty = ty; ptx = tx;
    }
}

// cat(xa, xb) (two fp32 planar 3-channel images, newnet1.py:300) as channels 0..5 of a zero-padded 32-channel NHWC bf16 map:
// the input of the 6 -> 32 conv in the layout of the kernel above.  One thread per pixel: six coalesced plane reads, one
// 64-byte row out.
__global__ __launch_bounds__(256) void pack_images_c32_kernel(const float* __restrict__ xa, const float* __restrict__ xb, h16_t* __restrict__ out,
                                                              int B, int64_t HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, q = i - b * HW;
        const float* pa = xa + b * 3 * HW + q;
        const float* pb = xb + b * 3 * HW + q;
        const u32x4 v0 = {pack_h2(pa[0], pa[HW]), pack_h2(pa[2 * HW], pb[0]), pack_h2(pb[HW], pb[2 * HW]), 0u};
        const u32x4 z = {0u, 0u, 0u, 0u};
        u32x4* o = (u32x4*)(out + i * 32);
        o[0] = v0; o[1] = z; o[2] = z; o[3] = z;
    }
}

}  // namespace

extern "C" int hesic_pack_images_c32(const float* xa, const float* xb, void* out, int B, int H, int W, void* stream) {
    HESIC_CHECK_ARG(xa && xb && out && B > 0 && H > 0 && W > 0, "pack_images_c32: bad arguments");
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(pack_images_c32_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, xa, xb, (h16_t*)out, B, (int64_t)H * W);
    HESIC_LAUNCH_RETURN("pack_images_c32");
}

extern "C" int hesic_conv3x3_c32_forward_img6(const float* xa, const float* xb, const float* w, const float* bias, int act, void* y, int B, int H, int W,
                                              void* stream) {
    HESIC_CHECK_ARG(xa && xb && w && y && B > 0 && H > 0 && W > 0, "conv3x3_c32_forward_img6: bad arguments");
    HESIC_CHECK_ARG((int64_t)H * W * 64 < (1ll << 31) && (int64_t)3 * H * W * 4 < (1ll << 31), "conv3x3_c32_forward_img6: image too large for 32-bit offsets");
    C6Args a;
    a.xa = xa; a.xb = xb; a.w = w; a.bias = bias; a.y = y; a.B = B; a.H = H; a.W = W; a.act = act;
    a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
    a.fd_tx = make_fastdiv((uint32_t)a.tiles_x); a.fd_ty = make_fastdiv((uint32_t)a.tiles_y);
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * B;
    HESIC_CHECK_ARG(ntiles < (1ll << 31), "conv3x3_c32_forward_img6: too many tiles");
    hipLaunchKernelGGL(c32_conv6_kernel, dim3((unsigned)(ntiles < 512 ? ntiles : 512)), dim3(256), 0, (hipStream_t)stream, a);
    HESIC_LAUNCH_RETURN("conv3x3_c32_forward_img6");
}

extern "C" int hesic_conv3x3_c32_forward(const void* x, const float* w, const float* bias, int Cout, int act, const void* res1,
                                         const void* res2, void* y, int B, int H, int W, void* stream) {
    HESIC_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0, "conv3x3_c32_forward: bad arguments");
    HESIC_CHECK_ARG(Cout == 32 || (Cout >= 1 && Cout <= 4), "conv3x3_c32_forward: Cout must be 32 (bf16 NHWC out) or <= 4 (fp32 planar out)");
    HESIC_CHECK_ARG(Cout == 32 || !res2, "conv3x3_c32_forward: the planar form takes one residual");
    HESIC_CHECK_ARG((int64_t)H * W * 64 < (1ll << 31), "conv3x3_c32_forward: image too large for 32-bit offsets");
    C32Args a;
    a.x = (const h16_t*)x; a.w = w; a.bias = bias; a.res1 = res1; a.res2 = res2; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
    a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
    a.fd_tx = make_fastdiv((uint32_t)a.tiles_x); a.fd_ty = make_fastdiv((uint32_t)a.tiles_y);
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * B;
    HESIC_CHECK_ARG(ntiles < (1ll << 31), "conv3x3_c32_forward: too many tiles");
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);          // persistent: two blocks per CU, weights packed once per block
    if (Cout == 32) {
        if (!a.res1 && a.res2) { a.res1 = a.res2; a.res2 = nullptr; }        // one residual: it is res1
        if (a.res2) hipLaunchKernelGGL((c32_conv3x3_kernel<0, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
        else if (a.res1) hipLaunchKernelGGL((c32_conv3x3_kernel<0, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((c32_conv3x3_kernel<0, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    } else if (a.Cout <= 3) hipLaunchKernelGGL((c32_conv3x3_kernel<1, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((c32_conv3x3_kernel<1, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    HESIC_LAUNCH_RETURN("conv3x3_c32_forward");
}

#ifdef R3_STAMP
extern "C" int hesic_en_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_r3_stamp), &p, sizeof(p)); }
#endif
extern "C" int hesic_resblock_c32_forward(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, int act,
                                          const void* res2, void* y, int B, int H, int W, void* stream) {
    HESIC_CHECK_ARG(x && w1 && w2 && y && B > 0 && H > 0 && W > 0, "resblock_c32_forward: bad arguments");
    HESIC_CHECK_ARG(x != y, "resblock_c32_forward: in-place is not supported (neighbouring tiles read the input halo)");
    HESIC_CHECK_ARG((int64_t)H * W * 64 < (1ll << 31), "resblock_c32_forward: image too large for 32-bit offsets");
    RBArgs a;
    a.x = (const h16_t*)x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.res2 = res2; a.y = y;
    a.B = B; a.H = H; a.W = W; a.act = act;
    a.tiles_x = (W + R3_TW - 1) / R3_TW; a.tiles_y = (H + R3_TH - 1) / R3_TH;
    a.fd_tx = make_fastdiv((uint32_t)a.tiles_x); a.fd_ty = make_fastdiv((uint32_t)a.tiles_y);
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * B;
    HESIC_CHECK_ARG(ntiles < (1ll << 31), "resblock_c32_forward: too many tiles");
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);          // persistent: one block per CU, both weight sets packed once per block
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)c32_resblock_r3_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_LDS);
        (void)hipFuncSetAttribute((const void*)c32_resblock_r3_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_LDS);
        (void)hipFuncSetAttribute((const void*)c32_resblock_r3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_LDS);
        (void)hipFuncSetAttribute((const void*)c32_resblock_r3_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_LDS);
        attr = true;
    }
    const dim3 g(grid), bk(512);
    hipStream_t st = (hipStream_t)stream;
    if (act == HESIC_ACT_LEAKY) {          // the ResidualBlock's activation (layers.py:131): known at compile time
        if (res2) hipLaunchKernelGGL((c32_resblock_r3_kernel<true, true>), g, bk, R3_LDS, st, a);
        else hipLaunchKernelGGL((c32_resblock_r3_kernel<false, true>), g, bk, R3_LDS, st, a);
    } else {
        if (res2) hipLaunchKernelGGL((c32_resblock_r3_kernel<true, false>), g, bk, R3_LDS, st, a);
        else hipLaunchKernelGGL((c32_resblock_r3_kernel<false, false>), g, bk, R3_LDS, st, a);
    }
    HESIC_LAUNCH_RETURN("resblock_c32_forward");
}

// ---------------------------------------------------------------------------------------------- weight gradient, 32 channels
// dW[co][ci][ky][kx] = sum over pixels of g[p][co] * x[p + (ky-1, kx-1)][ci] for the 3x3 convs above (stage 2 trains the enhancement
// net, newnet1.py:272-311): 9 x 32 x 32 numbers out of 2 M pixels per layer.  The implicit-GEMM weight-gradient kernel serves this
// shape with a 128 x 128 channel tile and one block per tap (16x the MFMA work, 9x the reads): 828 us per layer at B=8 512x512,
// 36 layers per step.  Here a wave owns a strip of 64 pixels of one image row: the strip of g (4 KB) and the three halo rows of x
// (3 x 66 pixels) go to wave-private LDS with LDS-DMA (NHWC rows of 32 bf16 channels are contiguous: lane-linear copies), double
// buffered; per 16 pixels one transposing fragment read of g (ds_read_tr16_b64: a lane wants 8 consecutive PIXELS of one channel)
// feeds ten MFMAs -- nine taps against the shifted x fragments and one against a fragment of ones for the bias gradient.  Every
// wave keeps its 10 accumulators over all its strips, a block leaves one [10][32][32] fp32 partial; a finishing kernel sums the
// partials into the PyTorch layout.  HBM-bound: g and x are read once (134 + 134 MB per layer).
namespace {

struct C32WgArgs {
    const h16_t* x; const h16_t* g; float* part;
    int B, H, W, strips_x, nstrips;
    FastDiv fd_sx, fd_h;
};

constexpr int WG_STRIP = 64, WG_ROWB = (WG_STRIP + 2) * 64 + 64;      // bytes of a staged x halo row (66 pixels, padded to 5 x 1 KB DMA pieces = 5120? no: see below)
constexpr int WG_XROW = 5 * 1024;                                     // a halo row is fetched as 5 lane-linear 1 KB pieces (80 pixels, 66 used)
constexpr int WG_BUF = 4096 + 3 * WG_XROW;                            // one stage: g strip + three x rows = 19456 bytes

__global__ __launch_bounds__(256, 1) void c32_wgrad_kernel(const C32WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    constexpr uint32_t OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* mine = wsm + wave * (2 * WG_BUF);
    asm volatile("" ::"v"((__attribute__((address_space(3))) unsigned char*)wsm) : "memory");
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)OOB, 0x00020000);
    const int gw = (int)(blockIdx.x * 4 + wave), nw = (int)gridDim.x * 4;

    auto issue = [&](int strip, int buf) {
        const uint32_t q = fdiv((uint32_t)strip, a.fd_sx);
        const int sx = strip - (int)q * a.strips_x;
        const int b = (int)fdiv(q, a.fd_h), y = (int)q - b * a.H;
        const int x0 = sx * WG_STRIP;
        unsigned char* base = mine + buf * WG_BUF;
        // g strip: pixels [x0, x0 + 64) of row y, 4 pieces of 16 pixels; lane = (pixel of 16, 16-byte chunk of its 64 bytes)
        const uint32_t rowoff = (uint32_t)(((b * a.H + y) * a.W) * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = x0 + i * 16 + (lane >> 2);
            const uint32_t v = px < a.W ? rowoff + (uint32_t)(px * 64 + (lane & 3) * 16) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (__attribute__((address_space(3))) void*)(base + i * 1024), 16, (int)v, 0, 0, 0);
        }
        // x halo rows y-1, y, y+1: pixels [x0 - 1, x0 + 79) as 5 pieces (66 used); outside the image -> zeros
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            const bool rok = (unsigned)yy < (unsigned)a.H;
            const uint32_t ro = (uint32_t)(((b * a.H + yy) * a.W) * 64);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int px = x0 - 1 + i * 16 + (lane >> 2);
                const bool ok = rok && (unsigned)px < (unsigned)a.W && (i < 4 || (lane >> 2) < 2);
                const uint32_t v = ok ? ro + (uint32_t)(px * 64 + (lane & 3) * 16) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(base + 4096 + r * WG_XROW + i * 1024), 16, (int)v, 0, 0, 0);
            }
        }
    };

    f32x16 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const h16x8 ones = __builtin_bit_cast(h16x8, u32x4{H16_ONE_PAIR, H16_ONE_PAIR, H16_ONE_PAIR, H16_ONE_PAIR});
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int gq = lane >> 4, t16 = lane & 15;
    // transposing read: a 16-lane group reads [4 pixels][16 channels]; lane -> (pixel t16 >> 2, channels 4 (t16 & 3) ..), and receives
    // 4 pixels of ONE channel.  Fragment lane (channel = lane & 31, k half = lane >> 5): channels (gq & 1) * 16 + .., pixels (gq >> 1) * 8 + ..
    const int choff = ((gq & 1) * 16 + 4 * (t16 & 3)) * 2, pixl = (gq >> 1) * 8 + (t16 >> 2);

    int strip = gw, buf = 0;
    if (strip < a.nstrips) issue(strip, 0);
    for (; strip < a.nstrips; strip += nw) {
        if (strip + nw < a.nstrips) {
            issue(strip + nw, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(19)" ::: "memory");        // the 19 pieces of the next strip may stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char* gt = mine + buf * WG_BUF;
        const unsigned char* xt = gt + 4096;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int pix = ks * 16 + pixl;
            const s16x4 dlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(gt + pix * 64 + choff));
            const s16x4 dhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(gt + (pix + 4) * 64 + choff));
            const h16x8 df = __builtin_bit_cast(h16x8, __builtin_shufflevector(dlo, dhi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r = t / 3, dx = t % 3;                       // halo row, pixel shift (halo pixel 0 = image pixel x0 - 1)
                const unsigned char* xp = xt + r * WG_XROW + (pix + dx) * 64 + choff;
                const s16x4 xlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)xp);
                const s16x4 xhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xp + 4 * 64));
                const h16x8 xf = __builtin_bit_cast(h16x8, __builtin_shufflevector(xlo, xhi, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[t] = mfma_32x32x16_h16(df, xf, acc[t], 0, 0, 0);
            }
            acc[9] = mfma_32x32x16_h16(df, ones, acc[9], 0, 0, 0);
        }
        buf ^= 1;
        asm volatile("" ::: "memory");
    }
    // the block's four partials meet in LDS (the staging buffers are free now), wave 0 writes ONE [10][32 co][32 ci] partial per block:
    // D row = co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column = ci = lane & 31
    __syncthreads();
    float* red = (float*)wsm;
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave - 1) * 10240 + (t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.part + (int64_t)blockIdx.x * (10 * 1024);
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), o = (t * 16 + r) * 64 + lane;
                out[t * 1024 + co * 32 + (lane & 31)] = (acc[t][r] + red[o]) + (red[10240 + o] + red[20480 + o]);
            }
    }
}

// dw[co][ci][ky][kx] (+)= sum of the partials' [tap][co][ci]; db[co] (+)= sum of [9][co][0]
__global__ __launch_bounds__(256) void c32_wgrad_finish_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw,
                                                               float* __restrict__ db, int co_n, int ci_n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // over [10][32][32]
    if (i >= 10 * 1024) return;
    const int t = i >> 10, co = (i >> 5) & 31, ci = i & 31;
    if (t == 9 ? (ci != 0 || !db || co >= co_n) : (co >= co_n || ci >= ci_n)) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 4 <= nparts; p += 4) {
        s0 += part[(int64_t)p * 10240 + i]; s1 += part[(int64_t)(p + 1) * 10240 + i];
        s2 += part[(int64_t)(p + 2) * 10240 + i]; s3 += part[(int64_t)(p + 3) * 10240 + i];
    }
    for (; p < nparts; ++p) s0 += part[(int64_t)p * 10240 + i];
    const float s = (s0 + s1) + (s2 + s3);
    if (t == 9) { db[co] = accumulate ? db[co] + s : s; return; }
    float* d = dw + ((int64_t)co * ci_n + ci) * 9 + t;
    *d = accumulate ? *d + s : s;
}

}  // namespace

extern "C" int64_t hesic_conv3x3_c32_wgrad_ws_bytes(void) { return (int64_t)256 * 10 * 1024 * sizeof(float); }

extern "C" int hesic_conv3x3_c32_wgrad(const void* x, const void* g, float* dw, float* dbias, int Cout, int Cin, int accumulate, void* ws,
                                       int64_t ws_bytes, int B, int H, int W, void* stream) {
    HESIC_CHECK_ARG(x && g && dw && ws && B > 0 && H > 0 && W > 0, "conv3x3_c32_wgrad: bad arguments");
    HESIC_CHECK_ARG(Cout >= 1 && Cout <= 32 && Cin >= 1 && Cin <= 32, "conv3x3_c32_wgrad: 1 <= Cout, Cin <= 32 (x and g are 32-channel maps; narrower convs use their first channels)");
    HESIC_CHECK_ARG((int64_t)B * H * W * 64 < (1ll << 31), "conv3x3_c32_wgrad: tensors too large for 32-bit offsets");
    HESIC_CHECK_ARG(ws_bytes >= hesic_conv3x3_c32_wgrad_ws_bytes(), "conv3x3_c32_wgrad: workspace too small");
    C32WgArgs a;
    a.x = (const h16_t*)x; a.g = (const h16_t*)g; a.part = (float*)ws;
    a.B = B; a.H = H; a.W = W; a.strips_x = (W + WG_STRIP - 1) / WG_STRIP;
    const int64_t ns = (int64_t)a.strips_x * H * B;
    HESIC_CHECK_ARG(ns < (1ll << 31), "conv3x3_c32_wgrad: too many strips");
    a.nstrips = (int)ns;
    a.fd_sx = make_fastdiv((uint32_t)a.strips_x); a.fd_h = make_fastdiv((uint32_t)H);
    const int grid = (int)((ns + 3) / 4 < 256 ? (ns + 3) / 4 : 256);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)c32_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(c32_wgrad_kernel, dim3(grid), dim3(256), 4 * 2 * WG_BUF, (hipStream_t)stream, a);
    hipLaunchKernelGGL(c32_wgrad_finish_kernel, dim3(40), dim3(256), 0, (hipStream_t)stream, (const float*)ws, grid, dw, dbias, Cout, Cin, accumulate);
    HESIC_LAUNCH_RETURN("conv3x3_c32_wgrad");
}
