// Round-4 probe: (1) does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  (2) MFMA-only rate, f16 vs bf16 operands, random data
// (the part is DVFS-limited: a wider multiplier array could cost clock).   hipcc --offload-arch=gfx950 -O3 f16_probe.hip -o f16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void denorm_kernel(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0; b[i] = (_Float16)0; }
    // k index 0 of lane group 0 only: A[row][k0] = aval, B[k0][col] = bval
    if ((threadIdx.x >> 5) == 0) { a[0] = (_Float16)aval; b[0] = (_Float16)bval; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

template <int F16>
__global__ __launch_bounds__(256) void rate_kernel(const u32x4* src, float* out, int iters) {
    u32x4 ra[4], rb[4];
    for (int i = 0; i < 4; ++i) { ra[i] = src[(threadIdx.x + 256 * i) & 1023]; rb[i] = src[(threadIdx.x + 256 * i + 77) & 1023]; }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (F16) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[(u + j) & 3]), __builtin_bit_cast(f16x8, rb[j]), acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[(u + j) & 3]), __builtin_bit_cast(bf16x8, rb[j]), acc[j], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; __builtin_memcpy(&u, &h, 2); return u; }
static uint16_t f2b(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main() {
    float* d; hipMalloc(&d, 4 * 2048 * 256);
    float h;
    struct { float a, b; } cases[] = {{1.f, 1.f}, {3e-6f, 1024.f}, {6e-8f, 16384.f}, {1024.f, 3e-6f}, {3e-6f, 3e-6f}};
    for (auto c : cases) {
        hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, d, c.a, c.b);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("denorm a=%g (fp16 %g) b=%g -> %g   expect %g\n", c.a, (float)(_Float16)c.a, c.b, h, (float)(_Float16)c.a * (float)(_Float16)c.b);
    }
    // operands: random normal-ish values in [-2, 2]
    uint16_t* hb = (uint16_t*)malloc(1024 * 16);
    u32x4* src; hipMalloc(&src, 1024 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 8;
    for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 4; ++mode) {   // 0 bf16 random, 1 f16 random, 2 bf16 zeros, 3 f16 zeros
        srand(1);
        for (int i = 0; i < 1024 * 8; ++i) {
            float v = mode >= 2 ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * 4.f;
            hb[i] = (mode & 1) ? f2h(v) : f2b(v);
        }
        hipMemcpy(src, hb, 1024 * 16, hipMemcpyHostToDevice);
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (mode & 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, src, d, iters);
            else hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, src, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
        printf("rate mode=%d (%s %s): %.3f ms  %.1f TFLOP/s\n", mode, (mode & 1) ? "f16" : "bf16", mode >= 2 ? "zeros" : "random", ms, fl / ms / 1e9);
    }
    return 0;
}
