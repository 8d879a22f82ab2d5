// Which (row, k) does byte p of lane l hold in the fp8 A operand of v_mfma_scale_f32_16x16x128_f8f6f4?
// One wave per (lane, byte): A has a single 1.0 there; B[j][k] encodes k in three runs (values / block scales).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int mode) {
    const int probe = blockIdx.x, pl = probe / 32, pp = probe % 32;     // lane and byte under test
    const int l = threadIdx.x, r = l & 15, kg = l >> 4;
    v8i a = {0,0,0,0,0,0,0,0}, b;
    if (l == pl) a[pp / 4] = 0x38 << (8 * (pp % 4));                    // 1.0
    // B operand, assuming contiguous 32 k per lane (this layout is confirmed for fp4; the A side is what we test)
    uint32_t bw[8];
    for (int i = 0; i < 8; ++i) {
        uint32_t w = 0;
        for (int by = 0; by < 4; ++by) {
            const int kk = i * 4 + by;                                   // k % 32 if contiguous
            int e = (mode == 0) ? (kk / 4) : (mode == 1 ? (kk % 4) : 0); // power-of-two level
            w |= (uint32_t)((7 + e) << 3) << (8 * by);                  // 2^e
        }
        bw[i] = w;
    }
    for (int i = 0; i < 8; ++i) b[i] = bw[i];
    const int sb = (mode == 2) ? 127 + 8 * kg : 127;                    // block scale encodes the 32-k block
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 127, 0, sb);
    // every b-column holds the same pattern: read column 0 of all rows: lanes with c == 0 hold rows 4g + e
    if ((l & 15) == 0) for (int e = 0; e < 4; ++e) out[(probe * 3 + mode) * 16 + 4 * (l >> 4) + e] = acc[e];
}
int main() {
    float* d; hipMalloc(&d, 2048 * 3 * 16 * 4);
    for (int m = 0; m < 3; ++m) hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, 0, d, m);
    hipDeviceSynchronize();
    static float h[2048 * 3 * 16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int probe = 0; probe < 2048; ++probe) {
        const int pl = probe / 32, pp = probe % 32;
        int row = -1; float v[3] = {0, 0, 0};
        for (int i = 0; i < 16; ++i) if (h[(probe * 3 + 0) * 16 + i] != 0.f) { row = i; for (int m = 0; m < 3; ++m) v[m] = h[(probe * 3 + m) * 16 + i]; }
        const int kb = (int)lround(log2(v[2])) / 8, k4 = (int)lround(log2(v[0])), k1 = (int)lround(log2(v[1]));
        const int kk = kb * 32 + k4 * 4 + k1;
        const int expect_row = pl & 15, expect_k = (pl >> 4) * 32 + pp;
        if (row != expect_row || kk != expect_k) { if (bad < 24) printf("lane %2d byte %2d -> row %d k %d (contiguous hypothesis: row %d k %d)\n", pl, pp, row, kk, expect_row, expect_k); ++bad; }
    }
    printf("%d of 2048 positions differ from the contiguous hypothesis\n", bad);
    return 0;
}
