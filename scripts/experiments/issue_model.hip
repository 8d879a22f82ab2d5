// issue_model.hip -- how do vector instructions of one or two waves per SIMD share the issue port with a stream of
// v_mfma_f32_16x16x32_bf16?  Register-only loop: per group 4 MFMAs (independent accumulators) + NC scaled converts
// (v_cvt_scalef32_pk_bf16_fp8) + NV plain VALU ops (v_and_or_b32) + ND ds_read_b128; 1 or 2 waves per SIMD.
// Prints shader cycles (s_memtime) per group, per wave: the matrix pipe alone needs 64 cycles per group and wave.
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/issue_model.hip -o /tmp/issue_model && /tmp/issue_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int NC, int NV, int ND, int THREADS>
__global__ void __launch_bounds__(512) k_issue(uint64_t* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int lane = threadIdx.x & 63;
    f32x4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    u32x4_t wa = {0x3f803f80u + lane, 0x3f803f80u, 0x40004000u, 0x3f803f80u};
    u32x4_t xb = {0x3f803f80u, 0x3f803f80u + 2 * lane, 0x3f803f80u, 0x40004000u};
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = lane * 0x01010101u + i;
    uint32_t code = lane * 0x9e3779b9u, ext = lane * 0x85ebca6bu, xtra = 0;
    float scale = __builtin_bit_cast(float, 0x3f800000u);
    for (int i = threadIdx.x; i < 4096; i += THREADS) reinterpret_cast<uint32_t*>(lds)[i] = i;
    __syncthreads();
    u32x4_t xf[3] = {xb, xb, xb};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            if (ND > 0) xf[(g + 2) % 3] = *reinterpret_cast<const u32x4_t*>(lds + ((lane * 16 + g * 1024) & 16383));   // consumed two groups later
#pragma unroll
            for (int d = 1; d < ND; ++d) {                                  // extra reads (consumed one group later through xtra)
                u32x4_t q = *reinterpret_cast<const u32x4_t*>(lds + ((lane * 16 + d * 1024 + g * 2048) & 16383));
                xtra ^= q[0];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[(g & 3) * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, ND > 0 ? xf[g % 3] : xb), acc[(g & 3) * 4 + i], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                r[c & 7] = (c & 1) ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[(c + 5) & 7], scale, true))
                                   : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[(c + 5) & 7], scale, false));
#pragma unroll
            for (int v = 0; v < NV / 2; ++v) {
                r[(v + 2) & 7] = (__builtin_amdgcn_alignbit(r[(v + 6) & 7], ext, v + 1) & 0x00080008u) | r[(v + 2) & 7];   // v_alignbit + v_and_or: count NV in pairs
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    uint32_t rr = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) rr ^= r[i];
    rr ^= xtra;
    if (s == 1.2345f || rr == 0x12345u) sink[0] = s;
    if (lane == 0) out[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NC, int NV, int ND, int THREADS>
static void run(const char* label) {
    const int blocks = 256, iters = 700;
    uint64_t* d; float* sink;
    hipMalloc(&d, blocks * (THREADS / 64) * 8); hipMalloc(&sink, 4);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_issue<NC, NV, ND, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_issue<NC, NV, ND, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, d, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * (THREADS / 64));
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[h.size() / 2] / (iters * 24.0);
    const double groups = (double)iters * 24.0;
    printf("%-34s waves/SIMD %d  cvt %d plain %d ds_read %d : %6.1f cycles per group and wave (pipe alone 64), wall %.1f ns per group -> %.2f GHz\n",
           label, THREADS / 256, NC, NV, ND, cyc, ms * 1e6 / groups, cyc / (ms * 1e6 / groups));
    hipFree(d); hipFree(sink);
}

int main() {
    run<0, 0, 0, 256>("mfma only");
    run<0, 0, 0, 512>("mfma only");
    run<2, 0, 0, 256>("U8 converts");
    run<2, 0, 0, 512>("U8 converts");
    run<2, 4, 0, 256>("U8X converts + ext");
    run<2, 4, 0, 512>("U8X converts + ext");
    run<0, 4, 0, 512>("4 plain");
    run<0, 8, 0, 512>("8 plain");
    run<4, 0, 0, 512>("4 cvt");
    run<2, 4, 1, 256>("U8X + 1 ds_read");
    run<2, 4, 1, 512>("U8X + 1 ds_read");
    run<0, 0, 1, 512>("1 ds_read");
    run<0, 0, 2, 512>("2 ds_read");
    run<1, 2, 1, 256>("half U8X + 1 ds_read (MF16 mix)");
    run<1, 2, 1, 512>("half U8X + 1 ds_read (MF16 mix)");
    return 0;
}
