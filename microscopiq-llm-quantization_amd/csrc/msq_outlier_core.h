// msq_outlier_core.h -- the per-block MicroScopiQ maths shared by the fake-quant and the
// pack kernels (register-resident block, one block per lane).
#pragma once
#include "../../include/msq.h"
#include "msq_device.h"

namespace msq {

// value * 2^e, exact (two-step so that pow2i stays in range)
MSQ_HD float scale_pow2(float v, int e) {
    while (e > 127) { v *= pow2i(127); e -= 127; }
    while (e < -126) { v *= pow2i(-126); e += 126; }
    return v * pow2i(e);
}
// codes of values that lie exactly on the target grid
MSQ_HD uint32_t encode_e2m1(float v) {                 // {0,.5,1,1.5,2,3,4,6}; zero is always +0 (code 0)
    const uint32_t u = f2u(v), m = u & 0x7FFFFFFFu;
    if (m == 0) return 0u;
    const int E = (int)(m >> 23) - 127;                // -1 .. 2
    const uint32_t mag = (E < 0) ? 1u : (uint32_t)(((E + 1) << 1) | ((m >> 22) & 1u));
    return ((u >> 28) & 8u) | mag;
}
MSQ_HD float decode_e2m1(uint32_t c) {
    const float t[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    const float v = t[c & 7];
    return (c & 8) ? -v : v;
}
MSQ_HD uint32_t encode_e4m3(float v) {                 // OCP e4m3fn, |v| <= 448 on the grid
    const uint32_t u = f2u(v), m = u & 0x7FFFFFFFu;
    if (m == 0) return 0u;
    if (v != v) return 0x7Fu;
    const int E = (int)(m >> 23) - 127;
    uint32_t mag;
    if (E >= -6) mag = (uint32_t)((E + 7) << 3) | ((m >> 20) & 7u);
    else mag = (uint32_t)(u2f(m) * 512.0f);            // subnormal: multiples of 2^-9
    return ((u >> 24) & 0x80u) | mag;
}
MSQ_HD uint32_t encode_e5m2(float v) {                 // e5m2 = truncated fp16
    const uint32_t u = f2u(v), m = u & 0x7FFFFFFFu;
    if (m == 0) return 0u;
    if (v != v) return 0x7Fu;
    const int E = (int)(m >> 23) - 127;
    uint32_t mag;
    if (E >= -14) mag = (uint32_t)((E + 15) << 2) | ((m >> 21) & 3u);
    else mag = (uint32_t)(u2f(m) * 65536.0f);          // subnormal: multiples of 2^-16
    return ((u >> 24) & 0x80u) | mag;
}

}  // namespace msq
using namespace msq;

// ===========================================================================
// MicroScopiQ outlier-aware fake-quant, one block per lane.
// ===========================================================================
struct OutlierArgs {
    Fmt fi, fo;
    int in_sb, out_sb;
    float k;          // std_dev as fp32 (python scalar * fp32 tensor)
    int rmode, flush, variant;
    int64_t pre, axis_len, post, nblk;
    uint8_t* mask;
    float* e_in;
    float* e_out;
    int8_t* n_out;
    int* status;
    const float* vmean;   // variant 1 statistics [pre, BS, post]
    const float* vstd;
};

// per-block maths on a register-resident block.  a[] in, result written back to a[];
// mk[] receives the 0/1 mask.  Returns status bits.
template <int BS, bool EMIT = false>
MSQ_D int outlier_block(float (&a)[BS], uint32_t (&mkw)[(BS + 31) / 32], float& se_in_o, float& se_out_o,
                        const OutlierArgs& A, int order, const float* vmean, const float* vstd,
                        int64_t vstride, uint32_t* codes = nullptr, int in_kind = 0, int out_kind = 0) {
    int status = 0;
    float lo, hi;
    if (A.variant == 0) {
        float ab[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) ab[b] = __builtin_fabsf(a[b]);
        float s;
        if (order == 1) s = sum_inner8<BS>(ab);
        else if (order == 2) s = sum_ilp4<BS>(ab);
        else s = sum_cascade<BS>(ab);
        const float mean = s / (float)BS;                    // utils/quant.py:477
        const float sd = std_welford<BS>(ab, 0);             // :478
        const float ks = A.k * sd;
        lo = mean - ks; hi = mean + ks;                      // :489-490
    }
#pragma unroll
    for (int w = 0; w < (BS + 31) / 32; ++w) mkw[w] = 0u;
    float mx_in = 0.f;
    float inl[BS];
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        if (A.variant != 0) {
            const float mean = vmean[b * vstride], sd = vstd[b * vstride];
            const float ks = A.k * sd;
            lo = mean - ks; hi = mean + ks;
        }
        const bool m = (a[b] < lo) || (a[b] > hi);           // :492 on the SIGNED value
        mkw[b >> 5] |= (m ? 1u : 0u) << (b & 31);
        const float mf = m ? 1.f : 0.f;
        inl[b] = a[b] * (1.0f - mf);                         // :192
        a[b] = a[b] * mf;                                    // :193 (a[] now holds the outlier part)
        const float t = __builtin_fabsf(inl[b]);
        mx_in = (t > mx_in || t != t) ? t : mx_in;
    }
    float se_in = shared_exp_of_max(mx_in);                  // :196-198
    const bool fl = A.flush && !(se_in > -127.f);            // :201-202
    se_in = se_in - (float)A.fi.emax;                        // :207
    se_in = clamp_scale_exp(se_in, A.in_sb, A.variant);      // :208-211
    const float sc_in = exp2f_int(se_in);
    const float rc_in = exp2f_int(-se_in);                   // x / 2^e == x * 2^-e exactly
    float mx_out = 0.f;
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        float v = inl[b];
        if (fl) v = v * 0.f;
        v = v * rc_in;                                       // :214
        a[b] = a[b] * sc_in;                                 // :216
        v = quant_elem(v, A.fi, A.rmode);                    // :218-221
        if (EMIT) codes[b] = (in_kind == MSQ_PLANE_FP4) ? encode_e2m1(v) : 0u;
        v = v * sc_in;                                       // :224
        if (v != v || a[b] != a[b]) status |= MSQ_STATUS_NAN; // :225-226
        inl[b] = v;
        const float t = __builtin_fabsf(a[b]);
        mx_out = (t > mx_out || t != t) ? t : mx_out;
    }
    float se_out = shared_exp_of_max(mx_out);                // :229-231
    if (se_out != se_out) status |= MSQ_STATUS_NAN;
    se_out = se_out - (float)A.fo.emax;                      // :237
    se_out = clamp_scale_exp(se_out, A.out_sb, A.variant);   // :239-242
    if (se_out != se_out) status |= MSQ_STATUS_NAN;          // :244
    const float sc_out = exp2f_int(se_out);
    const float rc_out = exp2f_int(-se_out);
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        float o = a[b] * rc_out;                             // :247
        if (o != o) status |= MSQ_STATUS_NAN;                // :250
        o = quant_elem(o, A.fo, A.rmode);                    // :252-255
        const float qo = o;
        o = (o * sc_out) * rc_in;                            // :258
        a[b] = inl[b] + o;                                   // :262
        if (EMIT) {
            // GEMM-ready planes hold HW-convertible codes; the fused kernel rebuilds
            // value = cvt(code) * 2^scale, so check that this model is exact here.
            uint32_t oc;
            if (in_kind == MSQ_PLANE_NONE) oc = f2u(a[b]) >> 16;             // whole value as bf16
            else if (out_kind == MSQ_PLANE_BF16) oc = f2u(o) >> 16;          // outlier part as bf16
            else oc = (out_kind == MSQ_PLANE_BF8) ? encode_e5m2(qo) : encode_e4m3(qo);
            codes[b] |= oc << 8;
            const float eff = se_out - se_in;
            bool exact;
            if (in_kind == MSQ_PLANE_NONE) exact = (u2f((f2u(a[b]) >> 16) << 16) == a[b]) || (a[b] != a[b]);
            else {
                exact = (u2f((f2u(o) >> 16) << 16) == o) || (o != o);
                if (out_kind != MSQ_PLANE_BF16 && qo != 0.f)
                    exact = exact && (eff >= -127.f) && (eff <= 127.f) && (scale_pow2(qo, (int)(eff == eff ? eff : 0.f)) == o);
                if (se_in == se_in)
                    exact = exact && (scale_pow2(decode_e2m1(codes[b] & 0xF), (int)se_in) == inl[b]);
            }
            if (!exact) status |= MSQ_STATUS_INEXACT;
        }
    }
    se_in_o = se_in; se_out_o = se_out;
    return status;
}

