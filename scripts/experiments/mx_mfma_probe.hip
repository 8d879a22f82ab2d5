// semantics probe of v_mfma_scale_f32_16x16x128_f8f6f4: operand register layout (fp8 / fp4), scale operand, result layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
// A: [16 rows][128 k] codes (fp4: nibbles, 64 B per row; fp8: 128 B per row); B: [16 cols][128 k]
template <int FA, int FB>
__global__ void k(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* D) {
    const int l = threadIdx.x, r = l & 15, kg = l >> 4;
    v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
    // hypothesis: lane (r, kg) holds k = 32 kg .. 32 kg + 31 of row r, in ascending byte / nibble order
    if (FA == 4) { const uint32_t* p = (const uint32_t*)(A + r * 64 + kg * 16); for (int i = 0; i < 4; ++i) a[i] = p[i]; }
    else {   // fp8: bytes 0..15 of the lane = k 16 kg .. +15, bytes 16..31 = k 64 + 16 kg .. +15 (measured: mx_mfma_layout.hip)
        const uint32_t* p0 = (const uint32_t*)(A + r * 128 + kg * 16); const uint32_t* p1 = (const uint32_t*)(A + r * 128 + 64 + kg * 16);
        for (int i = 0; i < 4; ++i) { a[i] = p0[i]; a[4 + i] = p1[i]; } }
    if (FB == 4) { const uint32_t* p = (const uint32_t*)(B + r * 64 + kg * 16); for (int i = 0; i < 4; ++i) b[i] = p[i]; }
    else {
        const uint32_t* p0 = (const uint32_t*)(B + r * 128 + kg * 16); const uint32_t* p1 = (const uint32_t*)(B + r * 128 + 64 + kg * 16);
        for (int i = 0; i < 4; ++i) { b[i] = p0[i]; b[4 + i] = p1[i]; } }
    const int scale_a = sa[r * 4 + kg], scale_b = sb[r * 4 + kg];   // E8M0 byte of this lane's (row, 32-k block), in byte 0
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, FA, FB, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 4; ++i) D[l * 4 + i] = acc[i];
}
static float dec_e4m3(uint8_t c) { int s = c >> 7, e = (c >> 3) & 15, m = c & 7; float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6); return s ? -v : v; }
static float dec_e2m1(uint8_t c) { const float t[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6}; float v = t[c & 7]; return (c & 8) ? -v : v; }
template <int FA, int FB> void run(const char* name) {
    uint8_t A[16 * 128], B[16 * 128], sa[64], sb[64];
    srand(7);
    for (int i = 0; i < 16 * 128; ++i) { A[i] = rand() & 0xFF; B[i] = rand() & 0xFF; if ((A[i] & 0x7F) == 0x7F) A[i] = 0x3C; if ((B[i] & 0x7F) == 0x7F) B[i] = 0x3C; }
    for (int i = 0; i < 64; ++i) { sa[i] = 120 + rand() % 12; sb[i] = 122 + rand() % 8; }
    uint8_t *dA, *dB, *dsa, *dsb; float* dD;
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<FA, FB>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD); hipDeviceSynchronize();
    float D[256]; hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    // reference: R[i][j] = sum_k a(i,k) 2^(sa[i][k/32]-127) * b(j,k) 2^(sb[j][k/32]-127)
    double R[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int kk = 0; kk < 128; ++kk) {
            const float av = (FA == 4) ? dec_e2m1((A[i * 64 + kk / 2] >> (4 * (kk & 1))) & 15) : dec_e4m3(A[i * 128 + kk]);
            const float bv = (FB == 4) ? dec_e2m1((B[j * 64 + kk / 2] >> (4 * (kk & 1))) & 15) : dec_e4m3(B[j * 128 + kk]);
            s += (double)av * ldexp(1.0, sa[i * 4 + kk / 32] - 127) * (double)bv * ldexp(1.0, sb[j * 4 + kk / 32] - 127);
        }
        R[i][j] = s;
    }
    // hypotheses for the result layout: lane l, element e -> (i, j)
    double errA = 0, errB = 0, mag = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        const int c = l & 15, g = l >> 4;
        errA = fmax(errA, fabs(D[l * 4 + e] - R[4 * g + e][c]));       // D[i = 4g+e][j = c]
        errB = fmax(errB, fabs(D[l * 4 + e] - R[c][4 * g + e]));       // D[i = c][j = 4g+e]
        mag = fmax(mag, fabs(R[4 * g + e][c]));
    }
    printf("%-22s max|R| %.4g  err if D[lane,e] = R[a-row 4g+e][b-col c]: %.3g ; if R[a-row c][b-col 4g+e]: %.3g\n", name, mag, errA, errB);
}
int main() { run<0, 0>("A fp8 x B fp8"); run<4, 0>("A fp4 x B fp8"); run<0, 4>("A fp8 x B fp4"); run<4, 4>("A fp4 x B fp4"); return 0; }
