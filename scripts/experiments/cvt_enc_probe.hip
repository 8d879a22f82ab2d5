// semantics probe of the f32 -> fp4 / fp8 / bf8 scaled converts (saturation, ties, scale operand) and back
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef short v2s __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, const float* scl, float* out, uint32_t* codes, int n) {
    int i = threadIdx.x; if (i >= n) return;
    const float a = in[i], s = scl[i];
    uint32_t c4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, a, a, s, 0);
    v2f d4 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(c4, s, 0);
    v2s z = {0, 0};
    v2s c8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, a, a, s, false);
    v2f d8 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(__builtin_bit_cast(uint32_t, c8), s, false);
    v2s c5 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(z, a, a, s, false);
    v2f d5 = __builtin_amdgcn_cvt_scalef32_pk_f32_bf8(__builtin_bit_cast(uint32_t, c5), s, false);
    out[i * 3 + 0] = d4[0]; out[i * 3 + 1] = d8[0]; out[i * 3 + 2] = d5[0];
    codes[i * 3 + 0] = c4 & 0xF; codes[i * 3 + 1] = __builtin_bit_cast(uint32_t, c8) & 0xFF; codes[i * 3 + 2] = __builtin_bit_cast(uint32_t, c5) & 0xFF;
}
int main() {
    float vals[] = {0.f, -0.f, 0.25f, 0.2500001f, 0.2499999f, 0.75f, 1.25f, 1.75f, 2.5f, 3.5f, 5.0f, 5.0000005f, 6.0f, 7.0f, 7.99f, 100.f, 1e9f, -7.5f,
                    INFINITY, -INFINITY, NAN, 448.f, 464.f, 480.f, 500.f, 1e-3f, 0.001953125f /*2^-9*/, 0.0009765625f /*2^-10 tie to 0*/, 0.0009765626f,
                    57344.f, 61440.f, 65536.f, 1.0625f /* e4m3 tie 1+1/16 */, 1.1875f, 17.f, 18.f, 19.f, 20.f, 1.4e-45f, -1.4e-45f};
    const int n = sizeof(vals) / sizeof(float);
    float scl[64]; for (int i = 0; i < n; ++i) scl[i] = 1.0f;
    float *din, *dscl, *dout; uint32_t* dc;
    hipMalloc(&din, 4 * 64); hipMalloc(&dscl, 4 * 64); hipMalloc(&dout, 12 * 64); hipMalloc(&dc, 12 * 64);
    for (int pass = 0; pass < 2; ++pass) {
        float in[64];
        for (int i = 0; i < n; ++i) { in[i] = pass ? vals[i] * 0.0078125f : vals[i]; scl[i] = pass ? 0.0078125f * 1.7f /* mantissa garbage: only the exponent should count */ : 1.0f; }
        hipMemcpy(din, in, 4 * n, hipMemcpyHostToDevice); hipMemcpy(dscl, scl, 4 * n, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dscl, dout, dc, n); hipDeviceSynchronize();
        float out[192]; uint32_t c[192];
        hipMemcpy(out, dout, 12 * n, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 12 * n, hipMemcpyDeviceToHost);
        printf("pass %d (scale %g)\n", pass, scl[0]);
        for (int i = 0; i < n; ++i)
            printf("  x=%-14.9g fp4: code %x -> %-10g | e4m3: code %02x -> %-12g | e5m2: code %02x -> %-12g\n", in[i], c[i * 3], out[i * 3], c[i * 3 + 1], out[i * 3 + 1], c[i * 3 + 2], out[i * 3 + 2]);
    }
    return 0;
}
