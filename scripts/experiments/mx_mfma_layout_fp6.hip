// Which (row, k) does the 6-bit slot j of lane l hold in the E3M2 ("bf6", format code 3) A operand of
// v_mfma_scale_f32_16x16x128_f8f6f4?  One wave per (lane, slot): A has a single 1.0 there (code 0x0C at bits 6 j ..);
// the e4m3 B operand (layout known: mx_mfma_layout.hip) carries 2^f(k) in three runs, so the product names k.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int mode, int fmt) {
    const int probe = blockIdx.x, pl = probe / 32, pj = probe % 32;
    const int l = threadIdx.x, kg = l >> 4;
    uint32_t aw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l == pl) {
        const uint64_t code = (fmt == 3) ? 0x0Cull : 0x08ull;          // 1.0 in e3m2 (bias 3) / e2m3 (bias 1)
        const int bit = 6 * pj;
        aw[bit / 32] |= (uint32_t)(code << (bit % 32));
        if (bit % 32 > 26) aw[bit / 32 + 1] |= (uint32_t)(code >> (32 - bit % 32));
    }
    uint32_t bw[8];
    for (int i = 0; i < 8; ++i) {
        uint32_t w = 0;
        for (int by = 0; by < 4; ++by) {
            const int idx = i * 4 + by;
            const int kk = (idx < 16) ? 16 * kg + idx : 64 + 16 * kg + idx - 16;       // fp8 operand layout
            const int e = (mode == 0) ? (kk & 7) : (mode == 1 ? ((kk >> 3) & 7) : (kk >> 6));
            w |= (uint32_t)((7 + e) << 3) << (8 * by);
        }
        bw[i] = w;
    }
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)aw[i]; b[i] = (int)bw[i]; }
    v4f acc = {0, 0, 0, 0};
    if (fmt == 3) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 3, 0, 0, 127, 0, 127);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 2, 0, 0, 127, 0, 127);
    if ((l & 15) == 0) for (int e = 0; e < 4; ++e) out[(probe * 3 + mode) * 16 + 4 * (l >> 4) + e] = acc[e];
}
int main() {
    float* d; hipMalloc(&d, 2048 * 3 * 16 * 4);
    static float h[2048 * 3 * 16];
    for (int fmt = 3; fmt >= 2; --fmt) {
        for (int m = 0; m < 3; ++m) hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, 0, d, m, fmt);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int probe = 0; probe < 2048; ++probe) {
            const int pl = probe / 32, pj = probe % 32;
            int row = -1; float v[3] = {0, 0, 0};
            for (int i = 0; i < 16; ++i) if (h[(probe * 3 + 0) * 16 + i] != 0.f) { row = i; for (int m = 0; m < 3; ++m) v[m] = h[(probe * 3 + m) * 16 + i]; }
            const int kk = (row < 0) ? -1 : (int)lround(log2(v[0])) + 8 * (int)lround(log2(v[1])) + 64 * (int)lround(log2(v[2]));
            const int expect_row = pl & 15, expect_k = (pl >> 4) * 32 + pj;
            if (row != expect_row || kk != expect_k) { if (bad < 16) printf("fmt %d lane %2d slot %2d -> row %d k %d (contiguous hypothesis: row %d k %d)\n", fmt, pl, pj, row, kk, expect_row, expect_k); ++bad; }
        }
        printf("format code %d: %d of 2048 positions differ from the contiguous hypothesis\n", fmt, bad);
    }
    return 0;
}
