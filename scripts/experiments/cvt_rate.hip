#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(uint32_t* out, uint32_t seed, float scale, int iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define STEP(x) \
        if (MODE == 0) x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x, scale, false)); \
        else if (MODE == 1) x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(x, scale, 0)); \
        else if (MODE == 2) x = x | (x >> 3); \
        else if (MODE == 3) x = __builtin_amdgcn_perm(x, seed, 0x01030205u + x); \
        else if (MODE == 4) x = x << 7;
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}
int main() {
    uint32_t* d; hipMalloc(&d, ((1 << 20) + 4) * 4);
    const int iters = 4096;
    const char* names[5] = {"cvt_scalef32_pk_bf16_fp8", "cvt_scalef32_pk_bf16_fp4", "v_or(+shift)", "v_perm_b32", "v_lshl"};
    for (int waves = 1; waves <= 2; ++waves) {
        for (int mode = 0; mode < 5; ++mode) {
            // 1 block per CU of `waves*4` waves -> `waves` waves per SIMD
            dim3 grid(256), blk(256 * waves);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, grid, blk, 0, 0, d, 12345u, 2.0f, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, grid, blk, 0, 0, d, 12345u, 2.0f, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, grid, blk, 0, 0, d, 12345u, 2.0f, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, grid, blk, 0, 0, d, 12345u, 2.0f, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, grid, blk, 0, 0, d, 12345u, 2.0f, iters); break;
            }
            hipDeviceSynchronize();
            uint32_t cyc; hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
            printf("waves/SIMD=%d %-28s %.2f cycles per wave-instruction (per wave)\n", waves, names[mode], (double)cyc / (iters * 16.0));
        }
    }
    return 0;
}
