// v_mfma_scale_f32_16x16x128_f8f6f4 vs v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8 and fp4 x fp8, unit scales), random and
// zero operands, register-resident: which shape sustains more under the power-limited clock?  (The 32x32 shape reads
// half the operand registers per flop.)  Same structure as mfma_peak_f8.hip: 32 / 16 independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int CBSZ>
__global__ void __launch_bounds__(512) k16(float* out, const int* seed, int iters) {
    v8i a[4], b[8];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) a[j][i] = seed[(threadIdx.x * 8 + i + j * 4096) & 65535];
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 8; ++i) b[j][i] = seed[(threadIdx.x * 8 + i + j * 5000 + 77) & 65535];
    v4f acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[j], b[i], acc[i][j], CBSZ, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    float s = 0; for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CBSZ>
__global__ void __launch_bounds__(512) k32(float* out, const int* seed, int iters) {
    v8i a[2], b[4];                                   // wave tile 128 x 64 = 4 x 2 tiles of 32 x 32
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 8; ++i) a[j][i] = seed[(threadIdx.x * 8 + i + j * 4096) & 65535];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) b[j][i] = seed[(threadIdx.x * 8 + i + j * 5000 + 77) & 65535];
    v16f acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)                // two 64-k instructions cover the 128 k of one 16x16x128
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j], b[i], acc[i][j], CBSZ, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; int* seed; hipMalloc(&d, (1 << 22) * 4); hipMalloc(&seed, 65536 * 4);
    int* h = (int*)malloc(65536 * 4);
    for (int zero = 0; zero < 2; ++zero) {
        srand(1);
        for (int i = 0; i < 65536; ++i) {
            uint32_t w = 0;
            for (int k = 0; k < 4; ++k) { uint32_t c = zero ? 0 : ((rand() & 1) << 7) | (((rand() % 6) + 5) << 3) | (rand() & 7); w |= c << (8 * k); }
            h[i] = (int)w;
        }
        hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
        const int blocks = 512, threads = 512, iters = 10000;
        for (int v = 0; v < 4; ++v) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int it) {
                if (v == 0) hipLaunchKernelGGL(k16<0>, dim3(blocks), dim3(threads), 0, 0, d, seed, it);
                else if (v == 1) hipLaunchKernelGGL(k32<0>, dim3(blocks), dim3(threads), 0, 0, d, seed, it);
                else if (v == 2) hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(threads), 0, 0, d, seed, it);
                else hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(threads), 0, 0, d, seed, it);
            };
            launch(100); hipDeviceSynchronize();
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * (threads / 64) * iters * 32.0 * 2.0 * 16 * 16 * 128;    // same work in all variants
            printf("%s data, %s, %s: %8.2f ms  %7.1f TFLOP/s\n", zero ? "zero  " : "random", v < 2 ? "fp8 x fp8" : "fp4 x fp8", (v & 1) ? "32x32x64 " : "16x16x128", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
