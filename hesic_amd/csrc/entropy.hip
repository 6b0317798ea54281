// Fused quantise + likelihood kernels of the entropy models (forward and backward):
//   EntropyBottleneck.forward          compressai/entropy_models/entropy_models.py:384-411 (+ :350-382)
//   GaussianConditional.forward        :546-554 (+ :528-544)
//   GaussianMixtureConditional.forward :661-702   (the HESIC addition)
// The reference runs ~30 (EB) / ~60 (GMM, K=5) tiny ATen kernels; here each latent is read once and
// y_hat / likelihood / symbols are written once.  All arithmetic is fp32; erfc/exp/tanh use the full
// precision device functions (no fast-math) because the likelihood floors at 1e-9.
#include "common.h"

namespace {

// --------------------------------------------------------------------------- EntropyBottleneck
// packed per-channel parameters (raw, i.e. before softplus / tanh), stride HESIC_EB_PARAM_STRIDE:
//   [0,3)   M0 (3x1)   [3,12) M1 (3x3)  [12,21) M2  [21,30) M3  [30,33) M4 (1x3)
//   [33,36) b0  [36,39) b1  [39,42) b2  [42,45) b3  [45] b4
//   [46,49) f0  [49,52) f1  [52,55) f2  [55,58) f3      [58] median
constexpr int EB_M0 = 0, EB_M1 = 3, EB_M4 = 30, EB_B0 = 33, EB_B4 = 45, EB_F0 = 46, EB_MED = 58, EB_NP = 58;

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float signf(float x) { return (x > 0.f) - (x < 0.f); }

struct EBParams {
    float sp[33];    // softplus(matrices)
    float b[13];
    float tf[12];    // tanh(factors)
};

// slot 59 of a parameter row marks a table whose softplus / tanh have already been applied (hesic_eb_prepare_params:
// the inference cache does the 45 transcendentals per channel once instead of once per thread and launch)
constexpr int EB_READY = 59;
// slot 60: the likelihood lower bound of the module (EntropyModel(likelihood_bound=...), entropy_models.py:60-66); 0 = no bound
constexpr int EB_BOUND = 60;

__device__ __forceinline__ void eb_load(const float* p, EBParams& q) {
    if (p[EB_READY] != 0.f) {
#pragma unroll
        for (int i = 0; i < 33; ++i) q.sp[i] = p[i];
#pragma unroll
        for (int i = 0; i < 13; ++i) q.b[i] = p[EB_B0 + i];
#pragma unroll
        for (int i = 0; i < 12; ++i) q.tf[i] = p[EB_F0 + i];
        return;
    }
#pragma unroll
    for (int i = 0; i < 33; ++i) q.sp[i] = softplusf(p[i]);
#pragma unroll
    for (int i = 0; i < 13; ++i) q.b[i] = p[EB_B0 + i];
#pragma unroll
    for (int i = 0; i < 12; ++i) q.tf[i] = tanhf(p[EB_F0 + i]);
}

// The parameter rows of a block's EB_CL channels in LDS, transformed ONCE per block (round 5).  Rounds 1-4 had every thread fetch its
// channel's 60 values with 60 strided loads (64 cache lines per wave instruction) and run the 45 softplus / tanh itself -- for ONE pixel per
// thread on the 8 x 8 hyper-latents of a training step: eb_fwd 26 us, eb_bwd 67 us for 65 536 elements.  Rows are padded to 65 floats
// (a wave reads 16 different rows: 16 different banks).  prep: softplus / tanh applied (slot layout of the raw row); sg: sigmoid of the raw
// matrix entries = d softplus / d raw, what the backward multiplies with (null: forward only).
constexpr int EB_CL = 16, EB_PL = 16, EB_ROW = HESIC_EB_PARAM_STRIDE + 1;
__device__ __forceinline__ void eb_stage(const float* __restrict__ params, int c0, int C, float (*prep)[EB_ROW], float (*sg)[EB_ROW]) {
    // EB_CL * 64 floats = 256 threads x one float4, coalesced
    const int e = threadIdx.x * 4, ch = e >> 6, slot = e & 63;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c0 + ch < C) v = *(const f32x4*)(params + (int64_t)(c0 + ch) * HESIC_EB_PARAM_STRIDE + slot);
    const float ready = (c0 + ch < C) ? params[(int64_t)(c0 + ch) * HESIC_EB_PARAM_STRIDE + EB_READY] : 1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sl = slot + k;
        const float r = v[k];
        float o = r;
        if (ready == 0.f) {
            if (sl < 33) o = softplusf(r);
            else if (sl >= EB_F0 && sl < EB_F0 + 12) o = tanhf(r);
        }
        prep[ch][sl] = o;
        if (sg) sg[ch][sl] = sl < 33 ? sigmoidf(r) : 0.f;
    }
}
__device__ __forceinline__ void eb_load_prepared(const float* p, EBParams& q) {
#pragma unroll
    for (int i = 0; i < 33; ++i) q.sp[i] = p[i];
#pragma unroll
    for (int i = 0; i < 13; ++i) q.b[i] = p[EB_B0 + i];
#pragma unroll
    for (int i = 0; i < 12; ++i) q.tf[i] = p[EB_F0 + i];
}

// forward of the 1-3-3-3-3-1 cumulative; keeps pre-activations when `pre` != nullptr (backward)
__device__ __forceinline__ float eb_logits(const EBParams& q, float v, float (*pre)[3], float (*hin)[3]) {
    float h[3], t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        t[r] = q.sp[EB_M0 + r] * v + q.b[r];
        if (pre) pre[0][r] = t[r];
        h[r] = t[r] + q.tf[r] * tanhf(t[r]);
    }
#pragma unroll
    for (int l = 1; l < 4; ++l) {
        if (hin) { hin[l][0] = h[0]; hin[l][1] = h[1]; hin[l][2] = h[2]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* m = q.sp + EB_M1 + (l - 1) * 9 + r * 3;
            t[r] = m[0] * h[0] + m[1] * h[1] + m[2] * h[2] + q.b[3 * l + r];
            if (pre) pre[l][r] = t[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) h[r] = t[r] + q.tf[3 * l + r] * tanhf(t[r]);
    }
    if (hin) { hin[4][0] = h[0]; hin[4][1] = h[1]; hin[4][2] = h[2]; }
    return q.sp[EB_M4] * h[0] + q.sp[EB_M4 + 1] * h[1] + q.sp[EB_M4 + 2] * h[2] + q.b[12];
}

// block = EB_CL channels x EB_PL pixel lanes; the block's parameter rows are staged (and transformed) once in LDS (eb_stage)
template <typename T, typename TO = T>
__global__ __launch_bounds__(EB_CL * EB_PL) void eb_fwd_kernel(const T* __restrict__ z, const float* __restrict__ params, const T* __restrict__ noise,
                              TO* __restrict__ zhat, float* __restrict__ lik, int32_t* __restrict__ sym, int64_t P, int C) {
    __shared__ float prep[EB_CL][EB_ROW];
    const int cl = threadIdx.x & (EB_CL - 1), pl = threadIdx.x / EB_CL;
    const int c0 = blockIdx.y * EB_CL, c = c0 + cl;
    eb_stage(params, c0, C, prep, nullptr);
    __syncthreads();
    if (c >= C) return;
    EBParams q;
    eb_load_prepared(prep[cl], q);
    const float med = prep[cl][EB_MED], bound = prep[cl][EB_BOUND];
    for (int64_t p = (int64_t)blockIdx.x * EB_PL + pl; p < P; p += (int64_t)gridDim.x * EB_PL) {
        const int64_t i = p * C + c;
        const float zv = elem<T>::ld(z + i);
        float v;
        if (noise) {
            v = zv + elem<T>::ld(noise + i);
        } else {
            const float r = rintf(zv - med);
            if (sym) sym[i] = (int32_t)r;
            v = r + med;
        }
        const float lo = eb_logits(q, v - 0.5f, nullptr, nullptr), up = eb_logits(q, v + 0.5f, nullptr, nullptr);
        const float s = -signf(lo + up);
        const float l = fabsf(sigmoidf(s * up) - sigmoidf(s * lo));
        elem<TO>::st(zhat + i, v);
        lik[i] = fmaxf(l, bound);
    }
}

__global__ void eb_prepare_kernel(const float* __restrict__ raw, float* __restrict__ out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* p = raw + (int64_t)c * HESIC_EB_PARAM_STRIDE;
    float* o = out + (int64_t)c * HESIC_EB_PARAM_STRIDE;
    for (int i = 0; i < 33; ++i) o[i] = softplusf(p[i]);
    for (int i = 0; i < 13; ++i) o[EB_B0 + i] = p[EB_B0 + i];
    for (int i = 0; i < 12; ++i) o[EB_F0 + i] = tanhf(p[EB_F0 + i]);
    o[EB_MED] = p[EB_MED];
    o[EB_READY] = 1.f;
    o[EB_BOUND] = p[EB_BOUND];
    for (int i = EB_BOUND + 1; i < HESIC_EB_PARAM_STRIDE; ++i) o[i] = 0.f;
}

// backward through one logits evaluation: accumulates d(params) into gp[58] and returns d/dv
__device__ __forceinline__ float eb_logits_bwd(const EBParams& q, const float* sg, float v, float gout, float* gp) {
    float pre[4][3], hin[5][3];
    eb_logits(q, v, pre, hin);
    // layer 4: out = sp4 . h3 + b4
    float dh[3];
    gp[EB_B4] += gout;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gp[EB_M4 + j] += gout * hin[4][j] * sg[EB_M4 + j];
        dh[j] = gout * q.sp[EB_M4 + j];
    }
#pragma unroll
    for (int l = 3; l >= 1; --l) {
        float dpre[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float th = tanhf(pre[l][r]);
            const float tf = q.tf[3 * l + r];
            gp[EB_F0 + 3 * l + r] += dh[r] * th * (1.f - tf * tf);
            dpre[r] = dh[r] * (1.f + tf * (1.f - th * th));
            gp[EB_B0 + 3 * l + r] += dpre[r];
        }
        float nh[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int k = EB_M1 + (l - 1) * 9 + r * 3 + j;
                gp[k] += dpre[r] * hin[l][j] * sg[k];
                nh[j] += q.sp[k] * dpre[r];
            }
        dh[0] = nh[0]; dh[1] = nh[1]; dh[2] = nh[2];
    }
    float dv = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float th = tanhf(pre[0][r]);
        const float tf = q.tf[r];
        gp[EB_F0 + r] += dh[r] * th * (1.f - tf * tf);
        const float dpre = dh[r] * (1.f + tf * (1.f - th * th));
        gp[EB_B0 + r] += dpre;
        gp[EB_M0 + r] += dpre * v * sg[EB_M0 + r];
        dv += dpre * q.sp[EB_M0 + r];
    }
    return dv;
}

// block = EB_CL channels x EB_PL pixel lanes (round 5: 16 x 16; rounds 1-4 ran 64 x 4 on at most 64 blocks -- 70 us on the 8 x 8 hyper-latents of
// a training step, ~6000 dependent VALU instructions per pixel and four serial pixels per thread on a quarter of the CUs).  The pixel lanes
// of a channel meet through two shuffles (a wave = 16 channels x 4 pixel lanes) and LDS before the block's 59 atomics per channel.
template <typename T>
__global__ __launch_bounds__(EB_CL * EB_PL) void eb_bwd_kernel(const T* __restrict__ z, const float* __restrict__ params, const T* __restrict__ noise,
                              const float* __restrict__ glik, const T* __restrict__ gzhat, T* __restrict__ dz,
                              float* __restrict__ dparams, int64_t P, int C) {
    constexpr int NW = EB_CL * EB_PL / 64;
    __shared__ float red[NW - 1][EB_CL][EB_NP + 1];
    const int cl = threadIdx.x & (EB_CL - 1), pl = threadIdx.x / EB_CL, wv = threadIdx.x >> 6;
    const int c0 = blockIdx.y * EB_CL, c = c0 + cl;
    const bool live = c < C;
    __shared__ float prep[EB_CL][EB_ROW], sgs[EB_CL][EB_ROW];
    eb_stage(params, c0, C, prep, sgs);
    __syncthreads();
    EBParams q;
    eb_load_prepared(prep[cl], q);
    const float* sgt = sgs[cl];
    const float med = prep[cl][EB_MED], bound = prep[cl][EB_BOUND];
    float gp[EB_NP + 1];
#pragma unroll
    for (int i = 0; i <= EB_NP; ++i) gp[i] = 0.f;
    if (live) {
        for (int64_t p = (int64_t)blockIdx.x * EB_PL + pl; p < P; p += (int64_t)gridDim.x * EB_PL) {
            const int64_t i = p * C + c;
            const float zv = elem<T>::ld(z + i);
            const float v = noise ? zv + elem<T>::ld(noise + i) : rintf(zv - med) + med;
            const float lo = eb_logits(q, v - 0.5f, nullptr, nullptr), up = eb_logits(q, v + 0.5f, nullptr, nullptr);
            const float s = -signf(lo + up);
            const float A = sigmoidf(s * up), Bv = sigmoidf(s * lo), dlt = A - Bv;
            float g = glik[i];
            if (!(fabsf(dlt) >= bound || g < 0.f)) g = 0.f;           // LowerBound rule (bound_ops.py:28-31)
            const float sg = signf(dlt) * g;
            const float gU = sg * A * (1.f - A) * s, gL = -sg * Bv * (1.f - Bv) * s;
            float dv = eb_logits_bwd(q, sgt, v + 0.5f, gU, gp) + eb_logits_bwd(q, sgt, v - 0.5f, gL, gp);
            if (gzhat) dv += elem<T>::ld(gzhat + i);
            if (noise) {
                elem<T>::st(dz + i, dv);
            } else {
                elem<T>::st(dz + i, 0.f);     // round() has zero gradient; "+ median" passes it to the median
                gp[EB_MED] += dv;
            }
        }
    }
    // a wave holds 4 pixel lanes (lane >> 4) of its 16 channels: fold them, then the waves through LDS
#pragma unroll
    for (int i = 0; i <= EB_NP; ++i) {
        gp[i] += __shfl_xor(gp[i], 16, 64);
        gp[i] += __shfl_xor(gp[i], 32, 64);
    }
    const bool lead = (threadIdx.x & 63) < EB_CL;
    if (wv > 0 && lead) {
#pragma unroll
        for (int i = 0; i <= EB_NP; ++i) red[wv - 1][cl][i] = gp[i];
    }
    __syncthreads();
    if (wv == 0 && lead && live) {
        float* out = dparams + (int64_t)c * HESIC_EB_PARAM_STRIDE;
#pragma unroll
        for (int i = 0; i <= EB_NP; ++i) {
            float v = gp[i];
#pragma unroll
            for (int l = 0; l < NW - 1; ++l) v += red[l][cl][i];
            atomicAdd(out + i, v);
        }
    }
}

// ------------------------------------------------------------------- Gaussian / Gaussian mixture
__device__ __forceinline__ float phi_cdf(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }
__device__ __forceinline__ float phi_pdf(float x) { return 0.39894228040143267794f * expf(-0.5f * x * x); }

// Branch-free erfc for the bf16 path: erfc(a) = exp(-a^2) * P((a-2)/(a+2)) / (1 + 2a) for a >= 0 (P: degree-10 least-squares
// fit of erfcx(a)(1+2a) on a in [0, 10.2], max relative error 1.8e-7 in fp32 Horner form), erfc(-a) = 2 - erfc(a); a^2 is
// split into its rounded value and the fma remainder so exp keeps ~2e-6 relative accuracy out to the 1e-9 likelihood
// floor.  ocml's erfcf is range-split (divergent lanes run several ranges) and cost ~45 VALU instructions per call, ten
// calls per latent: the kernel was VALU-bound at 2.7 TB/s.
__device__ __forceinline__ float erfc_fast(float z) {
    const float a = fabsf(z);
    const float p = (a - 2.f) * __builtin_amdgcn_rcpf(a + 2.f);
    float q = 5.535401624402612e-05f;
    q = fmaf(q, p, -0.0003277268339344692f);
    q = fmaf(q, p, -0.0012812873942078364f);
    q = fmaf(q, p, 0.001231548543911547f);
    q = fmaf(q, p, 0.008647743766865298f);
    q = fmaf(q, p, -0.008028522672854756f);
    q = fmaf(q, p, -0.054206538593428145f);
    q = fmaf(q, p, 0.16405084760016128f);
    q = fmaf(q, p, -0.1660312591225932f);
    q = fmaf(q, p, -0.0927638072951173f);
    q = fmaf(q, p, 1.2769783903081213f);
    const float s2 = a * a, el = fmaf(a, a, -s2);
    float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * s2);
    ex = fmaf(-el, ex, ex);
    const float ec = q * __builtin_amdgcn_rcpf(fmaf(2.f, a, 1.f)) * ex;
    return z >= 0.f ? ec : 2.f - ec;
}
__device__ __forceinline__ float phi_cdf_fast(float x) { return 0.5f * erfc_fast(-0.70710678118654752440f * x); }

constexpr int GMM_MAXK = 8;

template <typename T>
__global__ void gmm_fwd_kernel(const hesic_gmm_desc d, const T* __restrict__ y, const T* __restrict__ scales,
                               const T* __restrict__ means, const float* __restrict__ weights, const T* __restrict__ noise,
                               T* __restrict__ yhat, float* __restrict__ lik, int32_t* __restrict__ sym) {
    const int64_t total = (int64_t)d.B * d.HW * d.M;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = i % d.M;
        const int64_t p = i / d.M;
        const int b = p / d.HW;
        const float yv = elem<T>::ld(y + i);
        const int64_t sm = p * d.sm_pix_stride + m;
        float v;
        if (noise) {
            v = yv + elem<T>::ld(noise + i);
        } else if (d.use_means_in_quant) {
            const float mu = elem<T>::ld(means + sm + d.m_c_off);
            const float r = rintf(yv - mu);
            if (sym) sym[i] = (int32_t)r;
            v = r + mu;
        } else {
            v = rintf(yv);
            if (sym) sym[i] = (int32_t)v;
        }
        float acc = 0.f;
        for (int k = 0; k < d.K; ++k) {
            const float mu = elem<T>::ld(means + sm + d.m_c_off + k * d.M);
            const float s = fmaxf(elem<T>::ld(scales + sm + d.s_c_off + k * d.M), d.scale_bound);
            const float a = fabsf(v - mu);
            const float pk = phi_cdf((0.5f - a) / s) - phi_cdf((-0.5f - a) / s);
            acc += weights ? pk * weights[(int64_t)b * d.K * d.M + k * d.M + m] : pk;
        }
        elem<T>::st(yhat + i, v);
        lik[i] = fmaxf(acc, d.lik_bound);
    }
}

// ---------------------------------------------------------------- per-element CDF tables of HSIC.compress / decompress
// (ywz/mywork/newnet1.py:925-978, :1137-1175): for a listed channel m and every pixel, over the alphabet s = 0 .. 2*minmax,
//   pmf[s]  = sum_k w[k*M+m] * (Phi((.5 - |s - (mu_k + minmax)|)/sigma'_k) - Phi((-.5 - |..|)/sigma'_k))     (fp32, k ascending)
//   q[s]    = round_half_even(clip(pmf[s], 2^-16, 1) / sum(clip) * 65536)   with numpy's float32 pairwise summation order
//   cdf     = [0, cumsum(q)]                                                (exact integers in fp32, as np.add.accumulate)
// The reference does this with a Python loop per channel and pixel and a host round trip per channel; here it is one
// launch.  One thread per (channel, pixel); the row is first filled with the clipped pmf (as float bits), then rewritten.
__device__ float np_pairwise_sum(const float* a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// clipped pmf of symbol s under the row's mixture: ONE definition for both table kernels (their tables must agree bit for bit: an
// encoder may take one and a decoder the other only if both evaluate the same expression)
template <int DUMMY = 0>
__device__ __forceinline__ float cdf_pm(int s, const float* mu, const float* sg, const float* wk, int K) {
    float pm = 0.f;
    for (int k = 0; k < K; ++k) {
        const float a = fabsf((float)s - mu[k]);
        pm += (phi_cdf((0.5f - a) / sg[k]) - phi_cdf((-0.5f - a) / sg[k])) * wk[k];
    }
    return fminf(fmaxf(pm, 1.0f / 65536.0f), 1.0f);
}

template <typename T>
__global__ void gmm_cdf_kernel(const hesic_gmm_desc d, int b, const T* __restrict__ scales, const T* __restrict__ means,
                               const float* __restrict__ weights, const int32_t* __restrict__ channels, int n_ch, int minmax,
                               uint32_t* __restrict__ cdf, int pix_major) {
    const int A = 2 * minmax + 1;
    const int64_t total = (int64_t)n_ch * d.HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int hw = i % d.HW, j = i / d.HW;
        const int m = channels[j];
        const int64_t sm = ((int64_t)b * d.HW + hw) * d.sm_pix_stride + m;
        float mu[GMM_MAXK], sg[GMM_MAXK], wk[GMM_MAXK];
        for (int k = 0; k < d.K; ++k) {
            mu[k] = elem<T>::ld(means + sm + d.m_c_off + k * d.M) + (float)minmax;
            sg[k] = fmaxf(elem<T>::ld(scales + sm + d.s_c_off + k * d.M), d.scale_bound);
            wk[k] = weights ? weights[(int64_t)b * d.K * d.M + k * d.M + m] : 1.f;
        }
        uint32_t* row = cdf + (pix_major ? (int64_t)hw * n_ch + j : i) * (A + 1);
        float* frow = (float*)(row + 1);
        for (int s = 0; s < A; ++s) frow[s] = cdf_pm(s, mu, sg, wk, d.K);
        const float tot = np_pairwise_sum(frow, A);
        float run = 0.f;
        row[0] = 0u;
        for (int s = 0; s < A; ++s) {
            run += rintf(frow[s] / tot * 65536.0f);
            row[s + 1] = (uint32_t)run;
        }
    }
}

// The same tables, one WAVE per row (alphabets up to CDF_WAVE_MAX symbols; round 4): the 2 A error-function evaluations of a row -- the
// whole cost -- spread over the lanes, the clipped pmf staged in LDS, its sum taken by numpy's pairwise order on that array (every lane
// redundantly: LDS broadcasts), the cumulative counts by a wave scan (integer-valued floats below 2^24: any order is exact).  One thread
// per row took 15 us for the 2112 rows of a HESIC+ wavefront group (a latency-bound loop of ~40 symbols x 2 erfc); this form ~4 us.
constexpr int CDF_WAVE_MAX = 1024;
template <typename T>
__global__ __launch_bounds__(256) void gmm_cdf_wave_kernel(const hesic_gmm_desc d, int b, const T* __restrict__ scales, const T* __restrict__ means,
                                                           const float* __restrict__ weights, const int32_t* __restrict__ channels, int n_ch, int minmax,
                                                           uint32_t* __restrict__ cdf, const int32_t* __restrict__ dyn, int pix_major) {
    __shared__ float buf[4][CDF_WAVE_MAX];
    if (dyn) {          // channel count and alphabet of THIS image from device memory ({_, n_ch, minmax}): the launch sits in a graph captured for any image
        n_ch = dyn[1]; minmax = dyn[2];
        if (n_ch <= 0 || minmax < 1 || 2 * minmax + 1 > CDF_WAVE_MAX) return;
    }
    const int A = 2 * minmax + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* frow = buf[wave];
    const int64_t total = (int64_t)n_ch * d.HW;
    for (int64_t i = blockIdx.x * 4ll + wave; i < total; i += (int64_t)gridDim.x * 4) {
        const int hw = i % d.HW, j = i / d.HW;
        const int m = channels[j];
        const int64_t sm = ((int64_t)b * d.HW + hw) * d.sm_pix_stride + m;
        float mu[GMM_MAXK], sg[GMM_MAXK], wk[GMM_MAXK];
        for (int k = 0; k < d.K; ++k) {
            mu[k] = elem<T>::ld(means + sm + d.m_c_off + k * d.M) + (float)minmax;
            sg[k] = fmaxf(elem<T>::ld(scales + sm + d.s_c_off + k * d.M), d.scale_bound);
            wk[k] = weights ? weights[(int64_t)b * d.K * d.M + k * d.M + m] : 1.f;
        }
        for (int s = lane; s < A; s += 64) frow[s] = cdf_pm(s, mu, sg, wk, d.K);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float tot = np_pairwise_sum(frow, A);
        uint32_t* row = cdf + (pix_major ? (int64_t)hw * n_ch + j : i) * (A + 1);     // pixel-major: the order a pixel-by-pixel decoder walks them
        if (lane == 0) row[0] = 0u;
        float carry = 0.f;
        for (int s0 = 0; s0 < A; s0 += 64) {
            const int s = s0 + lane;
            float v = s < A ? rintf(frow[s] / tot * 65536.0f) : 0.f;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float u = __shfl_up(v, o, 64);
                if (lane >= o) v += u;
            }
            v += carry;
            if (s < A) row[s + 1] = (uint32_t)v;
            carry = __shfl(v, 63, 64);
        }
        __builtin_amdgcn_wave_barrier();          // the row buffer is rewritten by the next trip
    }
}

// block = 64 channels x 4 pixel lanes; grid = (M/64, pixel chunks, B): dweights reduced in-block first
// KT > 0: mixture count known at compile time (all 3K parameter loads of a pixel in flight at once, arrays in registers)
// and, for the bf16 path it is used on, the branch-free erfc / hardware reciprocal and exp2; KT == 0: run-time K, ocml math.
template <typename T, int KT>
__global__ __launch_bounds__(256) void gmm_bwd_kernel(const hesic_gmm_desc d, const T* __restrict__ y,
                                                      const T* __restrict__ scales, const T* __restrict__ means,
                                                      const float* __restrict__ weights, const T* __restrict__ noise,
                                                      const float* __restrict__ glik, const T* __restrict__ gyhat,
                                                      T* __restrict__ dy, T* __restrict__ dscales, T* __restrict__ dmeans,
                                                      float* __restrict__ dweights, int pix_per_block) {
    __shared__ float red[4][64][GMM_MAXK];
    const int ml = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + ml;
    const int b = blockIdx.z;
    const int hw0 = blockIdx.y * pix_per_block;
    const int KK = KT > 0 ? KT : d.K;
    constexpr int KA = KT > 0 ? KT : GMM_MAXK;
    float dw[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) dw[k] = 0.f;
    if (m < d.M) {
        for (int hw = hw0 + pl; hw < hw0 + pix_per_block && hw < d.HW; hw += 4) {
            const int64_t p = (int64_t)b * d.HW + hw;
            const int64_t i = p * d.M + m;
            const int64_t sm = p * d.sm_pix_stride + m;
            const float yv = elem<T>::ld(y + i);
            float v;
            if (noise) v = yv + elem<T>::ld(noise + i);
            else if (d.use_means_in_quant) { const float mu = elem<T>::ld(means + sm + d.m_c_off); v = rintf(yv - mu) + mu; }
            else v = rintf(yv);
            float pk[KA], mu[KA], s[KA], sraw[KA], wk[KA], is[KA];
            float L = 0.f;
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                if (k >= KK) break;
                mu[k] = elem<T>::ld(means + sm + d.m_c_off + k * d.M);
                sraw[k] = elem<T>::ld(scales + sm + d.s_c_off + k * d.M);
                wk[k] = weights ? weights[(int64_t)b * d.K * d.M + k * d.M + m] : 1.f;
            }
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                if (k >= KK) break;
                s[k] = fmaxf(sraw[k], d.scale_bound);
                const float a = fabsf(v - mu[k]);
                if constexpr (KT > 0) {
                    is[k] = __builtin_amdgcn_rcpf(s[k]);
                    pk[k] = phi_cdf_fast((0.5f - a) * is[k]) - phi_cdf_fast((-0.5f - a) * is[k]);
                } else {
                    is[k] = 1.f / s[k];
                    pk[k] = phi_cdf((0.5f - a) / s[k]) - phi_cdf((-0.5f - a) / s[k]);
                }
                L += wk[k] * pk[k];
            }
            float g = glik[i];
            if (!(L >= d.lik_bound || g < 0.f)) g = 0.f;
            float dv_sum = 0.f;
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                if (k >= KK) break;
                dw[k] += g * pk[k];
                const float df = v - mu[k];
                const float a = fabsf(df), sg = signf(df);
                float u, l, pu, pl_, dvk, dsk;
                if constexpr (KT > 0) {
                    u = (0.5f - a) * is[k]; l = (-0.5f - a) * is[k];
                    pu = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170f * u * u);
                    pl_ = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170f * l * l);
                } else {
                    u = (0.5f - a) / s[k]; l = (-0.5f - a) / s[k];
                    pu = phi_pdf(u); pl_ = phi_pdf(l);
                }
                const float ak = g * wk[k] * pu, ck = -g * wk[k] * pl_;
                if constexpr (KT > 0) { dvk = -(ak + ck) * is[k]; dsk = -(ak * u + ck * l) * is[k]; }
                else { dvk = -(ak + ck) / s[k]; dsk = -(ak * u + ck * l) / s[k]; }
                if (!(sraw[k] >= d.scale_bound || dsk < 0.f)) dsk = 0.f;  // LowerBound on the scale
                float dmu = -dvk * sg;
                dv_sum += dvk * sg;
                if (!noise && d.use_means_in_quant) dmu += dvk * sg;      // v = round(y-mu)+mu: dv/dmu = 1
                elem<T>::st(dscales + sm + d.s_c_off + k * d.M, dsk);
                elem<T>::st(dmeans + sm + d.m_c_off + k * d.M, dmu);
            }
            if (gyhat) {
                const float gy = elem<T>::ld(gyhat + i);
                dv_sum += gy;
                if (!noise && d.use_means_in_quant) {
                    // y_hat = round(y-mu)+mu also feeds mu directly (K == 1)
                    elem<T>::st(dmeans + sm + d.m_c_off, elem<T>::ld(dmeans + sm + d.m_c_off) + gy);
                }
            }
            elem<T>::st(dy + i, noise ? dv_sum : 0.f);
        }
    }
    if (dweights) {
#pragma unroll
        for (int k = 0; k < KA; ++k) red[pl][ml][k] = dw[k];
        __syncthreads();
        if (pl == 0 && m < d.M)
            for (int k = 0; k < KK; ++k)
                atomicAdd(dweights + (int64_t)b * d.K * d.M + k * d.M + m, red[0][ml][k] + red[1][ml][k] + red[2][ml][k] + red[3][ml][k]);
    }
}

int check_gmm(const hesic_gmm_desc* d, const char* who) {
    HESIC_CHECK_ARG(d && d->B > 0 && d->HW > 0 && d->M > 0 && d->K >= 1 && d->K <= GMM_MAXK, "%s: bad geometry (K <= %d)", who, GMM_MAXK);
    HESIC_CHECK_ARG(d->dtype == HESIC_H16 || d->dtype == HESIC_F32, "%s: bad dtype", who);
    HESIC_CHECK_ARG(!d->use_means_in_quant || d->K == 1, "%s: use_means_in_quant needs K == 1", who);
    return 0;
}

}  // namespace

extern "C" int hesic_eb_forward(const void* z, const float* params, const void* noise, void* z_hat, float* lik, int32_t* symbols,
                                int64_t P, int C, int dtype, void* stream) {
    HESIC_CHECK_ARG(z && params && z_hat && lik && P > 0 && C > 0, "eb_forward: bad arguments");
    const int64_t slices = (P + EB_PL - 1) / EB_PL;
    const int bx = EB_CL * EB_PL;
    const dim3 grid((unsigned)(slices < 256 ? slices : 256), (C + EB_CL - 1) / EB_CL);
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(eb_fwd_kernel<h16_t>, grid, dim3(bx), 0, (hipStream_t)stream, (const h16_t*)z, params,
                           (const h16_t*)noise, (h16_t*)z_hat, lik, symbols, P, C);
    else
        hipLaunchKernelGGL(eb_fwd_kernel<float>, grid, dim3(bx), 0, (hipStream_t)stream, (const float*)z, params,
                           (const float*)noise, (float*)z_hat, lik, symbols, P, C);
    HESIC_LAUNCH_RETURN("eb_forward");
}

extern "C" int hesic_eb_prepare_params(const float* params, float* prepared, int C, void* stream) {
    HESIC_CHECK_ARG(params && prepared && C > 0, "eb_prepare_params: bad arguments");
    hipLaunchKernelGGL(eb_prepare_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, params, prepared, C);
    HESIC_LAUNCH_RETURN("eb_prepare_params");
}

extern "C" int hesic_eb_backward(const void* z, const float* params, const void* noise, const float* g_lik, const void* g_zhat,
                                 void* dz, float* dparams, int64_t P, int C, int dtype, void* stream) {
    HESIC_CHECK_ARG(z && params && g_lik && dz && dparams && P > 0 && C > 0, "eb_backward: bad arguments");
    // every block ends in 59 atomics per channel on the gradient row, which serialise per address; fewer pixel slices mean fewer
    // contenders but more serial pixels (~6000 VALU instructions each) per thread.  Rounds 1-4 (64 channels x 4 pixel lanes per block), us per
    // launch on the 8 x 8 hyper-latents of a training step: 128 slices 111 | 64: 76 | 32: 70 | 16: 89 | 8: 147 | 4: 265; round 5 (16 x 16 per
    // block, C / 16 channel groups): 32 slices = one pixel per thread there, 256 blocks
    constexpr int max_slices_env = 32;
    const int max_slices = max_slices_env < 1 ? 1 : max_slices_env;              // 0 / negative would launch an empty grid
    const int64_t slices = (P + EB_PL - 1) / EB_PL;
    const dim3 grid((unsigned)(slices < max_slices ? slices : max_slices), (C + EB_CL - 1) / EB_CL), block(EB_CL * EB_PL);
    if (dtype == HESIC_H16)
        hipLaunchKernelGGL(eb_bwd_kernel<h16_t>, grid, block, 0, (hipStream_t)stream, (const h16_t*)z, params,
                           (const h16_t*)noise, g_lik, (const h16_t*)g_zhat, (h16_t*)dz, dparams, P, C);
    else
        hipLaunchKernelGGL(eb_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)z, params,
                           (const float*)noise, g_lik, (const float*)g_zhat, (float*)dz, dparams, P, C);
    HESIC_LAUNCH_RETURN("eb_backward");
}

// bf16 fast form of gmm_fwd_kernel: one thread = two neighbouring channels of a pixel (4-byte loads), K a template
// parameter so the 2K parameter loads of a thread are all in flight before the first erfc (the generic kernel's run-time
// K loop made them K serial HBM round trips: 44 % of its wave cycles were parked at s_waitcnt), 32-bit indexing.
// One channel pair of an NHWC row as two floats: 4-byte bf16 pairs or 8-byte fp32 pairs.
__device__ __forceinline__ f32x2 ld_pair(const h16_t* p) {
    const uint32_t r = *(const uint32_t*)p;
    return f32x2{h2f_lo(r), h2f_hi(r)};
}
__device__ __forceinline__ f32x2 ld_pair(const float* p) { return *(const f32x2*)p; }
__device__ __forceinline__ void st_pair(h16_t* p, float a, float b) { *(uint32_t*)p = pack_h2(a, b); }
__device__ __forceinline__ void st_pair(float* p, float a, float b) { *(f32x2*)p = f32x2{a, b}; }

// TI: storage of y / scales / means / noise, TO: storage of y_hat.  TI = float with TO = bf16 is the inference form of the
// bf16 mode: the latents and the entropy parameters come straight from the convs' fp32 accumulators
// (hesic_conv2d_forward_f32out), only the integer-valued y_hat that feeds the synthesis convs is bf16.
template <int K, typename TI = h16_t, typename TO = h16_t>
__global__ __launch_bounds__(256) void gmm_fwd_pair_kernel(const hesic_gmm_desc d, const TI* __restrict__ y, const TI* __restrict__ scales,
                                                           const TI* __restrict__ means, const float* __restrict__ weights,
                                                           const TI* __restrict__ noise, TO* __restrict__ yhat, float* __restrict__ lik,
                                                           int32_t* __restrict__ sym, FastDiv fd_m2) {
    const int M2 = d.M >> 1;
    const uint32_t total = (uint32_t)d.B * (uint32_t)d.HW * (uint32_t)M2;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const uint32_t p = fdiv(j, fd_m2);
        const int m = (int)(j - p * (uint32_t)M2) * 2;
        const int b = (int)(p / (uint32_t)d.HW);
        const int64_t i = (int64_t)p * d.M + m;
        const int64_t sm = (int64_t)p * d.sm_pix_stride + m;
        const f32x2 yr = ld_pair(y + i);
        f32x2 mr[K], sr[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            mr[k] = ld_pair(means + sm + d.m_c_off + k * d.M);
            sr[k] = ld_pair(scales + sm + d.s_c_off + k * d.M);
        }
        f32x2 wk[K];
#pragma unroll
        for (int k = 0; k < K; ++k) wk[k] = weights ? *(const f32x2*)(weights + (int64_t)b * d.K * d.M + k * d.M + m) : f32x2{1.f, 1.f};
        const f32x2 nr = noise ? ld_pair(noise + i) : f32x2{0.f, 0.f};
        float out_v[2], out_l[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float yv = e ? yr.y : yr.x;
            float v;
            if (noise) {
                v = yv + (e ? nr.y : nr.x);
            } else if (d.use_means_in_quant) {
                const float mu = e ? mr[0].y : mr[0].x;
                const float r = rintf(yv - mu);
                if (sym) sym[i + e] = (int32_t)r;
                v = r + mu;
            } else {
                v = rintf(yv);
                if (sym) sym[i + e] = (int32_t)v;
            }
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float mu = e ? mr[k].y : mr[k].x;
                const float sc = fmaxf(e ? sr[k].y : sr[k].x, d.scale_bound);
                const float a = fabsf(v - mu), inv = __builtin_amdgcn_rcpf(sc);
                const float pk = phi_cdf_fast((0.5f - a) * inv) - phi_cdf_fast((-0.5f - a) * inv);
                acc += weights ? pk * (e ? wk[k].y : wk[k].x) : pk;
            }
            out_v[e] = v;
            out_l[e] = fmaxf(acc, d.lik_bound);
        }
        st_pair(yhat + i, out_v[0], out_v[1]);
        *(f32x2*)(lik + i) = f32x2{out_l[0], out_l[1]};
    }
}

extern "C" int hesic_gmm_forward(const hesic_gmm_desc* d, const void* y, const void* scales, const void* means,
                                 const float* weights, const void* noise, void* y_hat, float* lik, int32_t* symbols,
                                 void* stream) {
    if (int e = check_gmm(d, "gmm_forward")) return e;
    HESIC_CHECK_ARG(y && scales && means && y_hat && lik, "gmm_forward: null pointer");
    HESIC_CHECK_ARG(weights || d->K == 1, "gmm_forward: weights required for K > 1");
    const int64_t total = (int64_t)d->B * d->HW * d->M;
    const dim3 grid(grid_for(total, 256));
    // pair form: even channel geometry (4-byte bf16 pairs, 8-byte fp32 pairs), 32-bit pair index
    constexpr bool no_pair = false;                  // A/B switch for profiling
    const bool pair = !no_pair && d->dtype == HESIC_H16 && (d->K == 5 || d->K == 1) && d->M % 2 == 0 && d->sm_pix_stride % 2 == 0 &&
                      d->s_c_off % 2 == 0 && d->m_c_off % 2 == 0 && total / 2 < (1ll << 31) && !((uintptr_t)y & 3) &&
                      !((uintptr_t)scales & 3) && !((uintptr_t)means & 3) && !((uintptr_t)noise & 3) && !((uintptr_t)y_hat & 3) &&
                      !((uintptr_t)lik & 7) && !((uintptr_t)weights & 7);
    if (pair) {
        const dim3 g2(grid_for(total / 2, 256));
        const FastDiv fd = make_fastdiv((uint32_t)(d->M / 2));
        if (d->K == 5)
            hipLaunchKernelGGL(gmm_fwd_pair_kernel<5>, g2, dim3(256), 0, (hipStream_t)stream, *d, (const h16_t*)y, (const h16_t*)scales,
                               (const h16_t*)means, weights, (const h16_t*)noise, (h16_t*)y_hat, lik, symbols, fd);
        else
            hipLaunchKernelGGL(gmm_fwd_pair_kernel<1>, g2, dim3(256), 0, (hipStream_t)stream, *d, (const h16_t*)y, (const h16_t*)scales,
                               (const h16_t*)means, weights, (const h16_t*)noise, (h16_t*)y_hat, lik, symbols, fd);
    } else if (d->dtype == HESIC_H16)
        hipLaunchKernelGGL(gmm_fwd_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, *d, (const h16_t*)y,
                           (const h16_t*)scales, (const h16_t*)means, weights, (const h16_t*)noise, (h16_t*)y_hat, lik, symbols);
    else
        hipLaunchKernelGGL(gmm_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *d, (const float*)y,
                           (const float*)scales, (const float*)means, weights, (const float*)noise, (float*)y_hat, lik, symbols);
    HESIC_LAUNCH_RETURN("gmm_forward");
}

extern "C" int hesic_gmm_cdf_rows(const hesic_gmm_desc* d, int b, const void* scales, const void* means, const float* weights,
                                  const int32_t* channels, int n_channels, int minmax, int pixel_major, uint32_t* cdf, void* stream);
extern "C" int hesic_gmm_cdf(const hesic_gmm_desc* d, int b, const void* scales, const void* means, const float* weights,
                             const int32_t* channels, int n_channels, int minmax, uint32_t* cdf, void* stream) {
    return hesic_gmm_cdf_rows(d, b, scales, means, weights, channels, n_channels, minmax, 0, cdf, stream);
}

// pixel_major: row (hw, j) at hw * n_channels + j instead of j * HW + hw -- the order in which the HESIC+ decoder consumes them (the
// host range decoder then streams through the table instead of striding: ~2x on its 30-40 us per wavefront group)
extern "C" int hesic_gmm_cdf_rows(const hesic_gmm_desc* d, int b, const void* scales, const void* means, const float* weights,
                                  const int32_t* channels, int n_channels, int minmax, int pixel_major, uint32_t* cdf, void* stream) {
    if (int e = check_gmm(d, "gmm_cdf")) return e;
    HESIC_CHECK_ARG(scales && means && channels && cdf && n_channels > 0 && minmax >= 1 && minmax < 32768 && b >= 0 && b < d->B,
                    "gmm_cdf: bad arguments");
    HESIC_CHECK_ARG(weights || d->K == 1, "gmm_cdf: weights required for K > 1");
    const int64_t total = (int64_t)n_channels * d->HW;
    constexpr bool thread_rows = false;          // A/B switch: the one-thread-per-row kernel for every alphabet
    if (2 * minmax + 1 <= CDF_WAVE_MAX && !thread_rows) {
        const dim3 gw(grid_for(total, 4, 256 * 16));
        if (d->dtype == HESIC_H16)
            hipLaunchKernelGGL(gmm_cdf_wave_kernel<h16_t>, gw, dim3(256), 0, (hipStream_t)stream, *d, b, (const h16_t*)scales, (const h16_t*)means, weights,
                               channels, n_channels, minmax, cdf, nullptr, pixel_major);
        else
            hipLaunchKernelGGL(gmm_cdf_wave_kernel<float>, gw, dim3(256), 0, (hipStream_t)stream, *d, b, (const float*)scales, (const float*)means, weights,
                               channels, n_channels, minmax, cdf, nullptr, pixel_major);
        HESIC_LAUNCH_RETURN("gmm_cdf");
    }
    const dim3 grid(grid_for(total, 128));
    if (d->dtype == HESIC_H16)
        hipLaunchKernelGGL(gmm_cdf_kernel<h16_t>, grid, dim3(128), 0, (hipStream_t)stream, *d, b, (const h16_t*)scales,
                           (const h16_t*)means, weights, channels, n_channels, minmax, cdf, pixel_major);
    else
        hipLaunchKernelGGL(gmm_cdf_kernel<float>, grid, dim3(128), 0, (hipStream_t)stream, *d, b, (const float*)scales,
                           (const float*)means, weights, channels, n_channels, minmax, cdf, pixel_major);
    HESIC_LAUNCH_RETURN("gmm_cdf");
}

// hesic_gmm_cdf with the channel count and the alphabet read on the device (state = {_, n_channels, minmax}, int32): the launch can then sit
// in a HIP graph that is replayed for images with other channel lists and alphabets (the HESIC+ wavefront step).  Rows are laid out with
// the image's own 2 * minmax + 2 stride, as hesic_gmm_cdf writes them; alphabets beyond the wave kernel's 1024 entries write nothing.
extern "C" int hesic_gmm_cdf_dyn(const hesic_gmm_desc* d, int b, const void* scales, const void* means, const float* weights,
                                 const int32_t* channels, int max_channels, const int32_t* state, int pixel_major, uint32_t* cdf, void* stream) {
    if (int e = check_gmm(d, "gmm_cdf_dyn")) return e;
    HESIC_CHECK_ARG(scales && means && channels && cdf && state && max_channels > 0 && b >= 0 && b < d->B, "gmm_cdf_dyn: bad arguments");
    HESIC_CHECK_ARG(weights || d->K == 1, "gmm_cdf_dyn: weights required for K > 1");
    const dim3 gw(grid_for((int64_t)max_channels * d->HW, 4, 256 * 16));
    if (d->dtype == HESIC_H16)
        hipLaunchKernelGGL(gmm_cdf_wave_kernel<h16_t>, gw, dim3(256), 0, (hipStream_t)stream, *d, b, (const h16_t*)scales, (const h16_t*)means, weights,
                           channels, max_channels, 1, cdf, state, pixel_major);
    else
        hipLaunchKernelGGL(gmm_cdf_wave_kernel<float>, gw, dim3(256), 0, (hipStream_t)stream, *d, b, (const float*)scales, (const float*)means, weights,
                           channels, max_channels, 1, cdf, state, pixel_major);
    HESIC_LAUNCH_RETURN("gmm_cdf_dyn");
}

extern "C" int hesic_gmm_backward(const hesic_gmm_desc* d, const void* y, const void* scales, const void* means,
                                  const float* weights, const void* noise, const float* g_lik, const void* g_yhat, void* dy,
                                  void* dscales, void* dmeans, float* dweights, void* stream) {
    if (int e = check_gmm(d, "gmm_backward")) return e;
    HESIC_CHECK_ARG(y && scales && means && g_lik && dy && dscales && dmeans, "gmm_backward: null pointer");
    HESIC_CHECK_ARG((weights && dweights) || d->K == 1, "gmm_backward: weights/dweights required for K > 1");
    // pixels per block: 64 on big maps, fewer (>= 4, one per pixel lane group) when that is what it takes to put ~1000 blocks
    // on the chip -- a 16x16 latent map at 64 pixels per block was 96 blocks of 16 serial iterations each
    const int other = ((d->M + 63) / 64) * d->B;
    const int want = (1024 + other - 1) / other;
    int ppb = (d->HW + want - 1) / want;
    ppb = (ppb + 3) / 4 * 4;
    if (ppb < 4) ppb = 4;
    if (ppb > 64) ppb = 64;
    const dim3 grid((d->M + 63) / 64, (d->HW + ppb - 1) / ppb, d->B);
    constexpr bool slow_bwd = false;                  // A/B switch for profiling
    if (d->dtype == HESIC_H16 && !slow_bwd && (d->K == 5 || d->K == 1)) {
        if (d->K == 5)
            hipLaunchKernelGGL((gmm_bwd_kernel<h16_t, 5>), grid, dim3(256), 0, (hipStream_t)stream, *d, (const h16_t*)y,
                               (const h16_t*)scales, (const h16_t*)means, weights, (const h16_t*)noise, g_lik,
                               (const h16_t*)g_yhat, (h16_t*)dy, (h16_t*)dscales, (h16_t*)dmeans, dweights, ppb);
        else
            hipLaunchKernelGGL((gmm_bwd_kernel<h16_t, 1>), grid, dim3(256), 0, (hipStream_t)stream, *d, (const h16_t*)y,
                               (const h16_t*)scales, (const h16_t*)means, weights, (const h16_t*)noise, g_lik,
                               (const h16_t*)g_yhat, (h16_t*)dy, (h16_t*)dscales, (h16_t*)dmeans, dweights, ppb);
    } else if (d->dtype == HESIC_H16)
        hipLaunchKernelGGL((gmm_bwd_kernel<h16_t, 0>), grid, dim3(256), 0, (hipStream_t)stream, *d, (const h16_t*)y,
                           (const h16_t*)scales, (const h16_t*)means, weights, (const h16_t*)noise, g_lik,
                           (const h16_t*)g_yhat, (h16_t*)dy, (h16_t*)dscales, (h16_t*)dmeans, dweights, ppb);
    else
        hipLaunchKernelGGL((gmm_bwd_kernel<float, 0>), grid, dim3(256), 0, (hipStream_t)stream, *d, (const float*)y,
                           (const float*)scales, (const float*)means, weights, (const float*)noise, g_lik,
                           (const float*)g_yhat, (float*)dy, (float*)dscales, (float*)dmeans, dweights, ppb);
    HESIC_LAUNCH_RETURN("gmm_backward");
}

// ------------------------------------------------------------------ training plumbing of the bottleneck parameters
// The 13 MLP tensors + quantiles of an EntropyBottleneck (entropy_models.py:262-300) <-> the [C][64] table of the kernels
// above, one launch each way (the torch route was cat + pad + fill on the way in and 14 slice copies on the way out, per
// bottleneck and step).  Entry j is a (C, width_j) row-major tensor occupying table columns [col_j, col_j + width_j);
// `stride` / `first` select a column subset of a wider tensor (the median = quantiles[:, 0, 1]: stride 3, first 1).
// thread = one table element (channel, column): one load and one store per thread (a thread per CHANNEL walked its 64 columns and
// ~58 dependent loads serially: 17.5 us for 128 channels)
__global__ void eb_pack_table_kernel(const hesic_eb_layout L, float* __restrict__ table, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = idx / HESIC_EB_PARAM_STRIDE, col = idx - c * HESIC_EB_PARAM_STRIDE;
    if (c >= C) return;
    float v = col == EB_BOUND ? L.lik_bound : 0.f;
    for (int j = 0; j < L.n; ++j) {
        const int k = col - L.col[j];
        if (k >= 0 && k < L.width[j]) v = L.ptr[j][(int64_t)c * L.stride[j] + L.first[j] + k];
    }
    table[idx] = v;
}

__global__ void eb_scatter_grads_kernel(const hesic_eb_layout L, const float* __restrict__ dtable, int C, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = idx / HESIC_EB_PARAM_STRIDE, col = idx - c * HESIC_EB_PARAM_STRIDE;
    if (c >= C) return;
    for (int j = 0; j < L.n; ++j) {
        const int k = col - L.col[j];
        if (k >= 0 && k < L.width[j]) {
            float* dst = L.ptr[j] + (int64_t)c * L.stride[j] + L.first[j] + k;
            *dst = accumulate ? *dst + dtable[idx] : dtable[idx];
        }
    }
}

// EntropyBottleneck.loss (entropy_models.py:345-348) forward + backward in one launch: loss += sum_c sum_q |c(quantiles[c,q]) -
// target[q]| with the cumulative's parameters detached, so the only gradient is d/dquantiles = sign(.) * dc/dv.
__global__ void eb_aux_loss_kernel(const float* __restrict__ params, const float* __restrict__ quantiles, float t_hi,
                                   float* __restrict__ loss, float* __restrict__ dq, int C, int accumulate) {
    // thread = (channel, quantile): the three evaluations of a channel side by side (one thread per channel walked them in turn: 28 us for 128 channels)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = idx / 3, j = idx - c * 3;
    float part = 0.f;
    if (c < C) {
        const float* raw = params + (int64_t)c * HESIC_EB_PARAM_STRIDE;
        EBParams q;
        eb_load(raw, q);
        float gp[EB_NP + 1];
#pragma unroll
        for (int i = 0; i <= EB_NP; ++i) gp[i] = 0.f;
        const float v = quantiles[idx], tgt = j == 0 ? -t_hi : (j == 1 ? 0.f : t_hi);
        const float diff = eb_logits(q, v, nullptr, nullptr) - tgt;
        part = fabsf(diff);
        if (dq) {
            const float g = eb_logits_bwd(q, raw, v, signf(diff), gp);
            dq[idx] = accumulate ? dq[idx] + g : g;
        }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0 && part != 0.f) atomicAdd(loss, part);
}

// ------------------------------------------------------------------ inference forms with fp32 latents (bf16 mode)
// In the bf16 mode the analysis convs hand y / z and the hyper-synthesis convs hand sigma / mu over as fp32 (straight from
// their accumulators, hesic_conv2d_forward_f32out): round() and the likelihoods then see the same precision as in the
// reference's fp32 pipeline, only the outputs that feed the next bf16 conv (z_hat, y_hat) are stored in `out_dtype`.
extern "C" int hesic_eb_forward_f32in(const float* z, const float* params, void* z_hat, int out_dtype, float* lik, int32_t* symbols,
                                      int64_t P, int C, void* stream) {
    HESIC_CHECK_ARG(z && params && z_hat && lik && P > 0 && C > 0, "eb_forward_f32in: bad arguments");
    HESIC_CHECK_ARG(out_dtype == HESIC_H16 || out_dtype == HESIC_F32, "eb_forward_f32in: bad dtype");
    const int64_t slices = (P + EB_PL - 1) / EB_PL;
    const int bx = EB_CL * EB_PL;
    const dim3 grid((unsigned)(slices < 256 ? slices : 256), (C + EB_CL - 1) / EB_CL);
    if (out_dtype == HESIC_H16)
        hipLaunchKernelGGL((eb_fwd_kernel<float, h16_t>), grid, dim3(bx), 0, (hipStream_t)stream, z, params, (const float*)nullptr,
                           (h16_t*)z_hat, lik, symbols, P, C);
    else
        hipLaunchKernelGGL((eb_fwd_kernel<float, float>), grid, dim3(bx), 0, (hipStream_t)stream, z, params, (const float*)nullptr,
                           (float*)z_hat, lik, symbols, P, C);
    HESIC_LAUNCH_RETURN("eb_forward_f32in");
}

extern "C" int hesic_gmm_forward_f32in(const hesic_gmm_desc* d, const float* y, const float* scales, const float* means,
                                       const float* weights, void* y_hat, int out_dtype, float* lik, int32_t* symbols, void* stream) {
    if (int e = check_gmm(d, "gmm_forward_f32in")) return e;
    HESIC_CHECK_ARG(y && scales && means && y_hat && lik, "gmm_forward_f32in: null pointer");
    HESIC_CHECK_ARG(weights || d->K == 1, "gmm_forward_f32in: weights required for K > 1");
    HESIC_CHECK_ARG(out_dtype == HESIC_H16 || out_dtype == HESIC_F32, "gmm_forward_f32in: bad dtype");
    const int64_t total = (int64_t)d->B * d->HW * d->M;
    HESIC_CHECK_ARG((d->K == 5 || d->K == 1) && d->M % 2 == 0 && d->sm_pix_stride % 2 == 0 && d->s_c_off % 2 == 0 && d->m_c_off % 2 == 0 &&
                        total / 2 < (1ll << 31) && !((uintptr_t)y & 7) && !((uintptr_t)scales & 7) && !((uintptr_t)means & 7) &&
                        !((uintptr_t)y_hat & 7) && !((uintptr_t)lik & 7) && !((uintptr_t)weights & 7),
                    "gmm_forward_f32in: K in {1, 5}, even channel geometry and 8-byte aligned buffers");
    const dim3 g2(grid_for(total / 2, 256));
    const FastDiv fd = make_fastdiv((uint32_t)(d->M / 2));
    hipStream_t st = (hipStream_t)stream;
#define GMM_F32IN(K_, TO_) hipLaunchKernelGGL((gmm_fwd_pair_kernel<K_, float, TO_>), g2, dim3(256), 0, st, *d, y, scales, means, weights, \
                                              (const float*)nullptr, (TO_*)y_hat, lik, symbols, fd)
    if (d->K == 5) { if (out_dtype == HESIC_H16) GMM_F32IN(5, h16_t); else GMM_F32IN(5, float); }
    else { if (out_dtype == HESIC_H16) GMM_F32IN(1, h16_t); else GMM_F32IN(1, float); }
#undef GMM_F32IN
    HESIC_LAUNCH_RETURN("gmm_forward_f32in");
}

extern "C" int hesic_eb_pack_table(const hesic_eb_layout* layout_host, float* table, int C, void* stream) {
    HESIC_CHECK_ARG(layout_host && table && C > 0 && layout_host->n > 0 && layout_host->n <= HESIC_EB_MAX_TENSORS, "eb_pack_table: bad arguments");
    for (int j = 0; j < layout_host->n; ++j)
        HESIC_CHECK_ARG(layout_host->ptr[j] && layout_host->width[j] > 0 && layout_host->col[j] >= 0 &&
                            layout_host->col[j] + layout_host->width[j] <= EB_READY, "eb_pack_table: entry %d out of range", j);
    hipLaunchKernelGGL(eb_pack_table_kernel, dim3((C * HESIC_EB_PARAM_STRIDE + 255) / 256), dim3(256), 0, (hipStream_t)stream, *layout_host, table, C);
    HESIC_LAUNCH_RETURN("eb_pack_table");
}

extern "C" int hesic_eb_scatter_grads(const hesic_eb_layout* layout_host, const float* dtable, int C, int accumulate, void* stream) {
    HESIC_CHECK_ARG(layout_host && dtable && C > 0 && layout_host->n > 0 && layout_host->n <= HESIC_EB_MAX_TENSORS, "eb_scatter_grads: bad arguments");
    hipLaunchKernelGGL(eb_scatter_grads_kernel, dim3((C * HESIC_EB_PARAM_STRIDE + 255) / 256), dim3(256), 0, (hipStream_t)stream, *layout_host, dtable, C, accumulate);
    HESIC_LAUNCH_RETURN("eb_scatter_grads");
}

extern "C" int hesic_eb_aux_loss(const float* params, const float* quantiles, float tail_mass, float* loss, float* dquantiles, int C,
                                 int accumulate, void* stream) {
    HESIC_CHECK_ARG(params && quantiles && loss && C > 0 && tail_mass > 0.f && tail_mass < 1.f, "eb_aux_loss: bad arguments");
    const float t_hi = logf(2.f / tail_mass - 1.f);
    hipLaunchKernelGGL(eb_aux_loss_kernel, dim3((3 * C + 63) / 64), dim3(64), 0, (hipStream_t)stream, params, quantiles, t_hi, loss, dquantiles, C, accumulate);
    HESIC_LAUNCH_RETURN("eb_aux_loss");
}
