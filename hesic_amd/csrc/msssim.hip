// MS-SSIM on the device: the second quality metric the reference reports next to PSNR (ywz/mywork/test3real.py:107-109,
// newtrain6_real.py:90-91: pytorch_msssim.ms_ssim(x_hat, x, data_range=1, size_average=False)).  pytorch_msssim is third party and
// absent; the published algorithm (Wang, Simoncelli, Bovik 2003, as that package implements it) is restated in
// oracle/hesic_oracle.py::ms_ssim and here:
//   per scale, per image and channel: an 11-tap normalised Gaussian (sigma 1.5) applied separably WITHOUT padding to x, y, x^2, y^2, xy;
//   cs = (2 s_xy + C2) / (s_x^2 + s_y^2 + C2), ssim = (2 mu_x mu_y + C1) / (mu_x^2 + mu_y^2 + C1) * cs, summed over the valid positions;
//   between scales a 2 x 2 average pool (zero padding of odd sides, divisor 4).
// HBM-bound by construction (two fp32 images read once per scale: 12.6 MB at 512 x 512 x 3 x 2 views x 8 pairs); a block owns a 32 x 32
// tile of valid positions: the 42 x 42 input patches of x and y go to LDS once, the horizontal pass leaves five 42 x 32 maps in LDS, the
// vertical pass + the two ratios run per output pixel, the block's partial sums leave through ONE pair of fp64 atomics.
#include "common.h"

namespace {

constexpr int WIN = 11, HALO = WIN - 1, TS = 32, PS = TS + HALO;      // tile of valid outputs, patch side

struct SsimArgs {
    const float* x; const float* y;
    int64_t xs[4], ys[4];          // element strides (b, c, row, col) of the two images
    int B, C, H, W, Ho, Wo, tiles_x, tiles_y;
    float win[WIN];
    float C1, C2;
    double* sums;                  // [B * C][2]: sum of ssim, sum of cs over the valid positions
};

__global__ __launch_bounds__(256) void ssim_scale_kernel(const SsimArgs a) {
    __shared__ float px[PS][PS + 1], py[PS][PS + 1];
    __shared__ float hq[5][PS][TS + 1];
    __shared__ double red[4][2];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int c = t % a.C, b = t / a.C;
    const int oy0 = ty * TS, ox0 = tx * TS;
    const float* xb = a.x + b * a.xs[0] + c * a.xs[1];
    const float* yb = a.y + b * a.ys[0] + c * a.ys[1];
    for (int i = tid; i < PS * PS; i += 256) {
        const int r = i / PS, q = i % PS, iy = oy0 + r, ix = ox0 + q;
        const bool ok = iy < a.H && ix < a.W;
        px[r][q] = ok ? xb[iy * a.xs[2] + ix * a.xs[3]] : 0.f;
        py[r][q] = ok ? yb[iy * a.ys[2] + ix * a.ys[3]] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < PS * TS; i += 256) {            // horizontal pass: rows of the patch x the tile's 32 columns
        const int r = i / TS, q = i % TS;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float w = a.win[k], u = px[r][q + k], v = py[r][q + k];
            sx += w * u; sy += w * v; sxx += w * (u * u); syy += w * (v * v); sxy += w * (u * v);
        }
        hq[0][r][q] = sx; hq[1][r][q] = sy; hq[2][r][q] = sxx; hq[3][r][q] = syy; hq[4][r][q] = sxy;
    }
    __syncthreads();
    double acc_s = 0.0, acc_c = 0.0;
    for (int i = tid; i < TS * TS; i += 256) {
        const int r = i / TS, q = i % TS;
        if (oy0 + r >= a.Ho || ox0 + q >= a.Wo) continue;
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float w = a.win[k];
#pragma unroll
            for (int j = 0; j < 5; ++j) m[j] += w * hq[j][r + k][q];
        }
        const float mxx = m[0] * m[0], myy = m[1] * m[1], mxy = m[0] * m[1];
        const float vx = m[2] - mxx, vy = m[3] - myy, cxy = m[4] - mxy;
        const float cs = (2.f * cxy + a.C2) / (vx + vy + a.C2);
        const float ss = (2.f * mxy + a.C1) / (mxx + myy + a.C1) * cs;
        acc_s += (double)ss; acc_c += (double)cs;
    }
    acc_s = wave_sum_d(acc_s); acc_c = wave_sum_d(acc_c);
    if ((tid & 63) == 0) { red[tid >> 6][0] = acc_s; red[tid >> 6][1] = acc_c; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(a.sums + (int64_t)(b * a.C + c) * 2, red[0][0] + red[1][0] + red[2][0] + red[3][0]);
        atomicAdd(a.sums + (int64_t)(b * a.C + c) * 2 + 1, red[0][1] + red[1][1] + red[2][1] + red[3][1]);
    }
}

// F.avg_pool2d(x, 2, padding = (H % 2, W % 2)) of pytorch_msssim: zero padding on both sides of an odd side, divisor always 4;
// output (H + 2 ph - 2) / 2 + 1.  Planar fp32 out (contiguous); one thread per output value.
__global__ void avgpool2_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, float* __restrict__ y,
                                int B, int C, int H, int W, int Ho, int Wo, int ph, int pw) {
    const int64_t n = (int64_t)B * C * Ho * Wo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = i % Wo;
        int64_t r = i / Wo;
        const int oy = r % Ho; r /= Ho;
        const int c = r % C, b = r / C;
        const float* p = x + b * sb + c * sc;
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int iy = 2 * oy + dy - ph, ix = 2 * ox + dx - pw;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) s += p[iy * sy + ix * sx];
            }
        y[i] = 0.25f * s;
    }
}

}  // namespace

extern "C" int hesic_ssim_scale(const float* x, const int64_t x_strides[4], const float* y, const int64_t y_strides[4], int B, int C, int H,
                                int W, float data_range, double* sums, void* stream) {
    HESIC_CHECK_ARG(x && y && x_strides && y_strides && sums && B > 0 && C > 0, "ssim_scale: bad arguments");
    HESIC_CHECK_ARG(H >= WIN && W >= WIN, "ssim_scale: image side %d x %d smaller than the %d-tap window", H, W, WIN);
    SsimArgs a;
    a.x = x; a.y = y;
    for (int i = 0; i < 4; ++i) { a.xs[i] = x_strides[i]; a.ys[i] = y_strides[i]; }
    a.B = B; a.C = C; a.H = H; a.W = W; a.Ho = H - HALO; a.Wo = W - HALO;
    a.tiles_x = (a.Wo + TS - 1) / TS; a.tiles_y = (a.Ho + TS - 1) / TS;
    float g[WIN], sum = 0.f;                                       // pytorch_msssim._fspecial_gauss_1d in fp32
    for (int i = 0; i < WIN; ++i) { const float co = (float)(i - WIN / 2); g[i] = expf(-(co * co) / (2.f * 1.5f * 1.5f)); sum += g[i]; }
    for (int i = 0; i < WIN; ++i) a.win[i] = g[i] / sum;
    a.C1 = (0.01f * data_range) * (0.01f * data_range); a.C2 = (0.03f * data_range) * (0.03f * data_range);
    a.sums = sums;
    const int64_t blocks = (int64_t)B * C * a.tiles_x * a.tiles_y;
    HESIC_CHECK_ARG(blocks < (1ll << 31), "ssim_scale: bad grid");
    hipLaunchKernelGGL(ssim_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    HESIC_LAUNCH_RETURN("ssim_scale");
}

extern "C" int hesic_avgpool2_pad(const float* x, const int64_t x_strides[4], float* y, int B, int C, int H, int W, void* stream) {
    HESIC_CHECK_ARG(x && x_strides && y && B > 0 && C > 0 && H > 1 && W > 1, "avgpool2_pad: bad arguments");
    const int ph = H % 2, pw = W % 2, Ho = (H + 2 * ph - 2) / 2 + 1, Wo = (W + 2 * pw - 2) / 2 + 1;
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for((int64_t)B * C * Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream, x, x_strides[0],
                       x_strides[1], x_strides[2], x_strides[3], y, B, C, H, W, Ho, Wo, ph, pw);
    HESIC_LAUNCH_RETURN("avgpool2_pad");
}
