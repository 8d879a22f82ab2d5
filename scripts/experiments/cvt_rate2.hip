// Issue cost of the scaled converts on gfx950, alone and as fillers behind MFMAs (one wave per SIMD, as k_qgemm256 runs):
//   cycles per wave-instruction back to back, and cycles per {MFMA, filler} pair against the MFMA alone (16 cycles).
// hipcc --offload-arch=gfx950 -O3 scripts/experiments/cvt_rate2.hip -o /tmp/cvt_rate2 && /tmp/cvt_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x32 __attribute__((ext_vector_type(32)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x6 __attribute__((ext_vector_type(6)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// MODE: 0 pk_bf16_fp8 (2 values), 1 pk_bf16_fp4 (2 values), 2 pk32_bf16_fp6 (32 values), 3 v_and_or (reference: one full-rate VALU op)
template <int MODE, bool WITH_MFMA>
__global__ void k(uint32_t* out, uint32_t seed, float scale, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (2 * i + 3) + threadIdx.x;
    u32x6 s6 = {a[0], a[1], a[2], a[3], a[4], a[5]};
    u32x16 acc6 = {};
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(threadIdx.x + i); fb[i] = (__bf16)(float)(i + 1); }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (WITH_MFMA) {
                __builtin_amdgcn_sched_barrier(0);
                if (u & 1) { if (u & 2) c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c3, 0, 0, 0); else c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c1, 0, 0, 0); }
                else { if (u & 2) c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c2, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c0, 0, 0, 0); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 0) a[u] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a[u], scale, false));
            else if (MODE == 1) a[u] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(a[u], scale, 0));
            else if (MODE == 2) {
                const bf16x32 r = __builtin_amdgcn_cvt_scalef32_pk32_bf16_fp6(s6, scale);
                const u32x16 ru = __builtin_bit_cast(u32x16, r);
                acc6 ^= ru;                                       // (16 xors: counted in; the loop of MODE 5 measures them alone)
                s6[u % 6] += 1u;
            }
            else if (MODE == 3) a[u] = (a[u] & 0x00080008u) | seed;
            else if (MODE == 5) { acc6 ^= __builtin_bit_cast(u32x16, acc6 + 1u); s6[u % 6] += 1u; }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t x = 0;
    for (int i = 0; i < 8; ++i) x ^= a[i];
    for (int i = 0; i < 16; ++i) x ^= acc6[i];
    x ^= __builtin_bit_cast(uint32_t, c0[0] + c1[1] + c2[2] + c3[3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}
template <int MODE, bool WM> static double run(uint32_t* d, int iters) {
    hipLaunchKernelGGL((k<MODE, WM>), dim3(256), dim3(256), 0, 0, d, 12345u, 2.0f, iters);    // one wave per SIMD
    hipDeviceSynchronize();
    uint32_t cyc; hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
    return (double)cyc / (iters * 8.0);
}
int main() {
    uint32_t* d; hipMalloc(&d, ((1 << 20) + 4) * 4);
    const int iters = 4096;
    run<3, false>(d, 64);
    printf("cycles per step of the loop body, one wave per SIMD (s_memtime)\n");
    printf("  v_and_or_b32 alone                              %6.2f\n", run<3, false>(d, iters));
    printf("  v_cvt_scalef32_pk_bf16_fp8 alone (2 values)     %6.2f\n", run<0, false>(d, iters));
    printf("  v_cvt_scalef32_pk_bf16_fp4 alone (2 values)     %6.2f\n", run<1, false>(d, iters));
    printf("  16 x v_xor + 1 add alone (harness of the next)  %6.2f\n", run<5, false>(d, iters));
    printf("  v_cvt_scalef32_pk32_bf16_fp6 + that harness (32 values) %6.2f\n", run<2, false>(d, iters));
    printf("  MFMA 16x16x32 bf16 + v_and_or                   %6.2f\n", run<3, true>(d, iters));
    printf("  MFMA + v_cvt_scalef32_pk_bf16_fp8               %6.2f\n", run<0, true>(d, iters));
    printf("  MFMA + v_cvt_scalef32_pk_bf16_fp4               %6.2f\n", run<1, true>(d, iters));
    printf("  MFMA + harness                                  %6.2f\n", run<5, true>(d, iters));
    printf("  MFMA + v_cvt_scalef32_pk32_bf16_fp6 + harness   %6.2f\n", run<2, true>(d, iters));
    return 0;
}
