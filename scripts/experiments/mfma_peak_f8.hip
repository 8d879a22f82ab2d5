// wall-clock ceiling of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8, unit scales) with random / zero operands,
// next to the bf16 16x16x32 figure of mfma_peak.hip: is "2 fp8 MFMAs per 128 k" as fast as "4 bf16 MFMAs"?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) kf8(float* out, const int* seed, int iters) {
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
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[j], b[i], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    float s = 0; for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; int* seed; hipMalloc(&d, (1 << 22) * 4); hipMalloc(&seed, 65536 * 4);
    int* h = (int*)malloc(65536 * 4);
    for (int zero = 0; zero < 2; ++zero) {
        srand(1);
        for (int i = 0; i < 65536; ++i) {            // random e4m3 bytes with moderate exponents (no NaN codes)
            uint32_t w = 0;
            for (int k = 0; k < 4; ++k) { uint32_t c = zero ? 0 : ((rand() & 1) << 7) | (((rand() % 6) + 5) << 3) | (rand() & 7); w |= c << (8 * k); }
            h[i] = (int)w;
        }
        hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
        for (int blocks : {256, 512}) {
            const int threads = 512, iters = 10000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kf8, dim3(blocks), dim3(threads), 0, 0, d, seed, 100); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(kf8, dim3(blocks), dim3(threads), 0, 0, d, seed, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double macs128 = (double)blocks * (threads / 64) * iters * 32.0;      // 16x16x128 MFMAs
            printf("%s fp8 data, %4d blocks: %8.2f ms  %7.1f TFLOP/s fp8 = %7.1f 'bf16-equivalent' TFLOP/s when two MFMAs cover one bf16 operand\n",
                   zero ? "zero  " : "random", blocks, ms, macs128 * 2.0 * 16 * 16 * 128 / ms / 1e9, macs128 * 2.0 * 16 * 16 * 128 / 2.0 / ms / 1e9);
        }
    }
    return 0;
}
