// Issue cost of the gfx950 scaled converts, measured with inline asm (results kept alive by asm volatile, no consumer in the loop):
// back to back, and as the single filler between two MFMAs (one wave per SIMD, as k_qgemm256 runs).
// hipcc --offload-arch=gfx950 -O3 scripts/experiments/cvt_rate3.hip -o cvt_rate3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x6 __attribute__((ext_vector_type(6)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

#define FILL(MODE, j)                                                                                                     \
    if (MODE == 0) asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(r2[j]) : "v"(a[j]), "v"(scale));            \
    else if (MODE == 1) asm volatile("v_cvt_scalef32_pk_bf16_fp4 %0, %1, %2" : "=v"(r2[j]) : "v"(a[j]), "v"(scale));       \
    else if (MODE == 2) asm volatile("v_cvt_scalef32_pk32_bf16_fp6 %0, %1, %2" : "=v"(r16[j & 1]) : "v"(s6), "v"(scale));  \
    else if (MODE == 3) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r2[j]) : "v"(a[j]), "v"(msk), "v"(a[(j + 1) & 7])); \
    else if (MODE == 4) asm volatile("v_alignbit_b32 %0, %1, %1, 5" : "=v"(r2[j]) : "v"(a[j]));                           \
    else if (MODE == 5) asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2" : "=v"(r64[j & 1]) : "v"(a[j]), "v"(scale)); \
    else if (MODE == 6) asm volatile("v_alignbit_b32 %0, %1, %1, 5\n\ts_waitcnt lgkmcnt(0)" : "=v"(r2[j]) : "v"(a[j]));          \
    else if (MODE == 7) asm volatile("v_alignbit_b32 %0, %2, %2, 5\n\ts_add_u32 %1, %1, 1" : "=v"(r2[j]), "+s"(sacc) : "v"(a[j])); \
    else if (MODE == 8) asm volatile("v_alignbit_b32 %0, %1, %1, 5\n\ts_nop 0" : "=v"(r2[j]) : "v"(a[j]));                       \
    else if (MODE == 9) asm volatile("v_alignbit_b32 %0, %2, %2, 5\n\tv_alignbit_b32 %1, %2, %2, 7" : "=v"(r2[j]), "=v"(r2b[j]) : "v"(a[j])); \
    else if (MODE == 10) asm volatile("s_waitcnt lgkmcnt(0)\n\tv_alignbit_b32 %0, %1, %1, 5" : "=v"(r2[j]) : "v"(a[j]));         \
    else if (MODE == 11) asm volatile("ds_read_b128 %0, %1" : "=v"(rd[j & 1]) : "v"(lds_addr));                                    \
    else if (MODE == 12) asm volatile("s_waitcnt lgkmcnt(1)\n\tds_read_b128 %0, %1" : "=v"(rd[j & 1]) : "v"(lds_addr));

template <int MODE, int MFMA>   // MFMA: 0 none, 1 one MFMA in front of every filler, 2 MFMA only
__global__ void k(uint32_t* out, uint32_t seed, float scale, int iters) {
    __shared__ uint32_t lds[4096];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rd[2] = {};
    const uint32_t lds_addr = (uint32_t)(threadIdx.x * 16);
    if (seed == 0) lds[threadIdx.x] = seed;
    uint32_t a[8], r2[8] = {}, r2b[8] = {}, sacc = seed;
    u32x16 r16[2] = {};
    uint64_t r64[2] = {};
    const uint32_t msk = 0x00080008u;
    for (int i = 0; i < 8; ++i) a[i] = seed * (2 * i + 3) + threadIdx.x;
    const u32x6 s6 = {a[0], a[1], a[2], a[3], a[4], a[5]};
    f32x4 c[4] = {};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(threadIdx.x + i); fb[i] = (__bf16)(float)(i + 1); }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MFMA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[u & 3]) : "v"(fa), "v"(fb));
            if (MFMA != 2) { FILL(MODE, u) }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t x = 0;
    for (int i = 0; i < 8; ++i) x ^= r2[i] ^ r2b[i];
    x ^= sacc ^ rd[0][0] ^ rd[1][1] ^ lds[threadIdx.x & 1023];
    for (int i = 0; i < 16; ++i) x ^= r16[0][i] ^ r16[1][i];
    x ^= (uint32_t)r64[0] ^ (uint32_t)r64[1] ^ __builtin_bit_cast(uint32_t, c[0][0] + c[1][1] + c[2][2] + c[3][3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}
template <int MODE, int MF> static double run(uint32_t* d, int iters) {
    hipLaunchKernelGGL((k<MODE, MF>), dim3(256), dim3(256), 0, 0, d, 12345u, 2.0f, iters);    // one wave per SIMD
    hipDeviceSynchronize();
    uint32_t cyc; hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
    return (double)cyc / (iters * 8.0);
}
int main() {
    uint32_t* d; hipMalloc(&d, ((1 << 20) + 4) * 4);
    const int it = 4096;
    run<3, 0>(d, 64);
    printf("cycles per instruction (s_memtime ticks / instructions), one wave per SIMD; alone | behind one MFMA (pair - MFMA alone)\n");
    const double m = run<3, 2>(d, it);
    printf("  v_mfma_f32_16x16x32_bf16 alone                       %6.2f\n", m);
#define ROW(MODE, NAME) { const double a_ = run<MODE, 0>(d, it), p_ = run<MODE, 1>(d, it); printf("  %-52s %6.2f | pair %6.2f (exposed %5.2f)\n", NAME, a_, p_, p_ - m); }
    ROW(3, "v_and_or_b32")
    ROW(4, "v_alignbit_b32")
    ROW(0, "v_cvt_scalef32_pk_bf16_fp8 (2 values)")
    ROW(1, "v_cvt_scalef32_pk_bf16_fp4 (2 values)")
    ROW(5, "v_cvt_scalef32_pk_f32_fp8 (2 values)")
    ROW(2, "v_cvt_scalef32_pk32_bf16_fp6 (32 values, 16 VGPRs)")
    ROW(9, "two v_alignbit_b32")
    ROW(8, "v_alignbit_b32 + s_nop 0")
    ROW(6, "v_alignbit_b32 + s_waitcnt lgkmcnt(0) (satisfied)")
    ROW(10, "s_waitcnt lgkmcnt(0) + v_alignbit_b32")
    ROW(7, "v_alignbit_b32 + s_add_u32")
    ROW(11, "ds_read_b128")
    ROW(12, "s_waitcnt lgkmcnt(1) + ds_read_b128")
    return 0;
}
