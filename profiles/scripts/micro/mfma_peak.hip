// Power-limited matrix-core ceiling: a loop of independent v_mfma_f32_32x32x16_f16 on register operands (no LDS, no memory), every SIMD
// of the chip busy, for random and for all-zero operands.  Prints the rate per second of wall time; run rocm-smi next to it for clock and power.
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak [seconds] [zeros]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
// the 16x16x32 form (what the 32-channel enhancement kernels issue): 16 independent accumulators, same flops per launch
__global__ __launch_bounds__(256, 2) void k16(const h8* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    h8 a0 = src[t & 4095], a1 = src[(t + 64) & 4095], b0 = src[(t + 128) & 4095], b1 = src[(t + 192) & 4095];
    f4v c[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[1], 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[2], 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[3], 0, 0, 0);
            c[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[4], 0, 0, 0);
            c[5] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[5], 0, 0, 0);
            c[6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[6], 0, 0, 0);
            c[7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[7], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int u = 0; u < 8; ++u) for (int r = 0; r < 4; ++r) s += c[u][r];
    if (s == 12345.678f) out[t] = s;
}
__global__ __launch_bounds__(256, 2) void k(const h8* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    h8 a0 = src[t & 4095], a1 = src[(t + 64) & 4095], b0 = src[(t + 128) & 4095], b1 = src[(t + 192) & 4095];
    f16v c[4] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c[1], 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c[2], 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += c[u][r];
    if (s == 12345.678f) out[t] = s;
}
int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const bool zeros = argc > 2 && argv[2][0] == 'z';
    const char* tag = argc > 2 ? argv[2] : "r";
    const bool small = argc > 3;
    h8* src; float* out;
    hipMalloc(&src, 4096 * sizeof(h8)); hipMalloc(&out, 1 << 24);
    _Float16* h = (_Float16*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = zeros ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    // argv[2] = "m<bits>": clear the low <bits> mantissa bits of EVERY operand value (how much of the matrix pipe's power is operand toggling?)
    if (argc > 2 && argv[2][0] == 'm') {
        const unsigned short mask = (unsigned short)(0xffffu << atoi(argv[2] + 1));
        for (int i = 0; i < 4096 * 8; ++i) ((unsigned short*)h)[i] &= mask;
    }
    hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    const int iters = 4096, blocks = 256 * 2 * 4;          // 8 waves per CU x 4 rounds
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    double total_ms = 0; int n = 0; float last = 0;
    while (total_ms < seconds * 1e3) {
        hipEventRecord(e0);
        if (small) hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1); total_ms += last; ++n;
    }
    const double flops = (double)blocks * 4 * iters * 16 * 32768.0;
    printf("%s %s operands: %d launches, last %.3f ms, %.0f TFLOP/s (last launch), %.0f TFLOP/s (average)\n", small ? "16x16x32" : "32x32x16", zeros ? "zero" : tag, n, last,
           flops / last / 1e9, flops * n / total_ms / 1e9);
    return 0;
}
