#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// cbsz / blgp format codes: 0 = fp8(e4m3), 1 = bf8, 2 = fp6, 3 = bf6, 4 = fp4
template <int FA, int FB>
__global__ void kscale(float* out, int iters) {
    v8i a, b; for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01010101 * i; b[i] = 0x3C343830 + threadIdx.x; }
    v4f acc[8]; for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], FA, FB, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
__global__ void kbf16(float* out, int iters) {
    bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 0.001f * i); b[i] = (__bf16)(0.5f + i); }
    v4f acc[8]; for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
int main() {
    float* d; hipMalloc(&d, ((1 << 20) + 4) * 4);
    const int iters = 2000; float cyc;
#define RUN(NAME, LAUNCH) LAUNCH; hipDeviceSynchronize(); hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost); printf("%-34s %.1f cycles / MFMA (1 wave per SIMD)\n", NAME, cyc / (iters * 8.0));
    dim3 g(256), b(256);
    RUN("bf16 16x16x32", hipLaunchKernelGGL(kbf16, g, b, 0, 0, d, iters))
    { auto kk = kscale<0, 0>; RUN("scaled 16x16x128 A=fp8 B=fp8", hipLaunchKernelGGL(kk, g, b, 0, 0, d, iters)) }
    { auto kk = kscale<4, 0>; RUN("scaled 16x16x128 A=fp4 B=fp8", hipLaunchKernelGGL(kk, g, b, 0, 0, d, iters)) }
    { auto kk = kscale<0, 4>; RUN("scaled 16x16x128 A=fp8 B=fp4", hipLaunchKernelGGL(kk, g, b, 0, 0, d, iters)) }
    { auto kk = kscale<4, 4>; RUN("scaled 16x16x128 A=fp4 B=fp4", hipLaunchKernelGGL(kk, g, b, 0, 0, d, iters)) }
    { auto kk = kscale<2, 0>; RUN("scaled 16x16x128 A=fp6 B=fp8", hipLaunchKernelGGL(kk, g, b, 0, 0, d, iters)) }
    return 0;
}
