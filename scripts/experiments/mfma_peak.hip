// wall-clock MFMA ceiling: every SIMD of the chip issues independent bf16 16x16x32 MFMAs back to back
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512) kpeak(float* out, const float* seed, int iters) {
    bf16x8 a[4], b[8];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) a[j][i] = (__bf16)(seed[(threadIdx.x * 8 + i + j * 4096) & 65535]);
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 8; ++i) b[j][i] = (__bf16)(seed[(threadIdx.x * 8 + i + j * 5000 + 77) & 65535]);
    v4f acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[i][j], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *d, *seed; hipMalloc(&d, (1 << 22) * 4); hipMalloc(&seed, 65536 * 4);
    float* h = (float*)malloc(65536 * 4);
    for (int zero = 0; zero < 2; ++zero) {
        srand(1); for (int i = 0; i < 65536; ++i) h[i] = zero ? 0.f : ((rand() % 2001) - 1000) / 500.0f;
        hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
        for (int threads : {256, 512}) for (int blocks : {256, 512, 1024}) {
            const int iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kpeak, dim3(blocks), dim3(threads), 0, 0, d, seed, 100); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(kpeak, dim3(blocks), dim3(threads), 0, 0, d, seed, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * (threads / 64) * iters * 32.0 * (2.0 * 16 * 16 * 32);
            printf("%s data, %4d blocks x %d threads: %8.2f ms  %7.1f TFLOP/s\n", zero ? "zero  " : "random", blocks, threads, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
