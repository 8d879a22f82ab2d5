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


// ===========================================================================
// Fast form of the same maths (fake-quant kernels; float/int element formats).  Bit-identical to
// outlier_block<> / the oracle by construction:
//  * std: two-pass variance in double; the float result can differ from the sequential Welford
//    result only when the double lies within 2^-40 (relative) of a float rounding boundary -- those
//    blocks (about 1 in 65000) fall back to the Welford loop, so the float is always Welford's.
//  * both shared exponents are known before any element is rounded (max |o * 2^e_in| =
//    fl(max|o| * 2^e_in): the scaling is monotone), so every element takes ONE trip through the
//    codec with per-element selected parameters instead of an inlier and an outlier trip.
//  * 2^k scalings are applied with the same sequence of single-rounded multiplies as the reference.
// ===========================================================================
template <int RM>
MSQ_D float quant_core_fast(float a, int shift, int pe_lo, int pe_hi, float max_norm, int rmode) {
    // elemwise_ops.py:84-174 with saturate_normals, allow_denorm.  The private exponent is clamped to
    // [pe_lo, pe_hi]: (min_exp, 127) for float formats, (0, 0) for the integer formats.
    const uint32_t ua = f2u(a);
    const int rm = (RM >= 0) ? RM : rmode;
    int pe = __builtin_amdgcn_frexp_expf(a) - 1;           // exact floor(log2|a|), subnormals included; a == 0 -> -1 (harmless)
    // the reference's private exponent is floor(torch.log2(|a|)) (elemwise_ops.py:139-140): one too high for the floats just below a
    // power of two -- which changes the result only under truncation (msq_device.h ilog2f_torch)
    if (rm == 1 && (ua & 0x7FFFFFFFu) != 0u && (ua & 0x7F800000u) != 0x7F800000u) pe = ilog2f_torch(a);
    pe = pe < pe_lo ? pe_lo : pe;
    pe = pe > pe_hi ? pe_hi : pe;
    const float x = __builtin_ldexpf(a, shift - pe);        // == a / 2^pe * 2^(bits-2) (first product exact)
    const float m = __builtin_fabsf(x);
    float r;
    if (rm == 0) r = __builtin_floorf(m + 0.5f);
    else if (rm == 1) r = __builtin_floorf(m);
    else {
        float q = __builtin_fmodf(m - 0.5f, 2.0f);
        if (q != 0.f && q < 0.f) q += 2.0f;
        r = __builtin_floorf(m + 0.5f) - ((q == 0.f) ? 1.f : 0.f);
    }
    r = __builtin_copysignf(r, x);
    float out = __builtin_ldexpf(r, pe - shift);            // == r / 2^(bits-2) * 2^pe
    out = __builtin_fminf(__builtin_fmaxf(out, -max_norm), max_norm);
    // NaN in -> NaN out; +-Inf in -> +-Inf out (elemwise_ops.py:165-166)
    return ((ua & 0x7F800000u) == 0x7F800000u) ? a : out;
}

// posit<n,es> rounding of an fp32 value, fast path: when at least one FRACTION bit survives
// (fb >= 1) the posit's round-to-nearest-even on the bit pattern is an ordinary RNE of the
// significand to fb fraction bits; everything else (exponent bits cut, saturation, specials)
// goes through the generic posit_round().  Identical results (tests/test_gpu_a13_posit.py::test_posit_tables).
MSQ_D float posit_round_fast(float t, int n, int es) {
    const uint32_t ut = f2u(t) & 0x7FFFFFFFu;
    if (ut - 1u >= 0x7F7FFFFFu) return posit_round(t, n, es);          // 0, Inf, NaN
    const int s = __builtin_amdgcn_frexp_expf(t) - 1;
    const int k = s >> es;                                               // floor(s / 2^es)
    const int rl = (k >= 0) ? k + 2 : 1 - k;
    const int fb = n - 1 - rl - es;
    if (fb < 1) return posit_round(t, n, es);
    const float r = __builtin_rintf(__builtin_ldexpf(t, fb - s));       // RNE (default rounding mode)
    return __builtin_ldexpf(r, s - fb);
}

template <int BS>
MSQ_D float std_twopass_checked(const float (&x)[BS], int correction) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < BS; ++i) s += (double)x[i];
    const double mean = s / (double)BS;
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < BS; ++i) { const double d = (double)x[i] - mean; m2 = __builtin_fma(d, d, m2); }
    double den = (double)BS - (double)correction; den = den < 0 ? 0 : den;
    const double sd = __builtin_sqrt(m2 / den);
    // distance of the double from the nearest float rounding boundary, in units of the dropped 29 bits
    const uint64_t bits = __builtin_bit_cast(uint64_t, sd);
    const uint32_t dropped = (uint32_t)(bits & 0x1FFFFFFFull);
    const uint32_t dist = dropped > 0x10000000u ? dropped - 0x10000000u : 0x10000000u - dropped;
    const bool safe = (dist > 0x2000u) && (sd == sd) && (sd > 1e-300) && (sd < 1e300);
    if (__builtin_expect(safe, 1)) return (float)sd;
    return std_welford<BS>(x, correction);
}

// formats with a short exact codec: the gfx950 scaled converts (1 = OCP e4m3, 2 = OCP e5m2, 3 = e2m1) and
// 4 = int2 (values {-1, 0, 1} x scale: one compare and one select, formats.py:87-95); 0 = none
MSQ_HD int hw_codec_kind(const Fmt& f) {
    if (f.kind != 0) return 0;
    if (f.ebits == 0 && f.mbits == 2) return 4;
    if (f.ebits == 4 && f.mbits == 5) return 1;
    if (f.ebits == 5 && f.mbits == 4) return 2;
    if (f.ebits == 2 && f.mbits == 3) return 3;
    return 0;
}
// quantise-dequantise two values (mantissa LSB already set by the caller) on the grid `kind` with scale `s`
MSQ_D void hw_codec_pair(int kind, float x0, float x1, float s, float bound, float& v0, float& v1) {
    typedef short v2s_t __attribute__((ext_vector_type(2)));
    typedef float v2f_t __attribute__((ext_vector_type(2)));
    v2f_t v;
    if (kind == 4) {                                             // int2: round half away of |x| / s, clamped to 1
        const float h = 0.5f * s;                                // exact (s is a power of two)
        v[0] = (__builtin_fabsf(x0) >= h) ? __builtin_copysignf(s, x0) : 0.f;
        v[1] = (__builtin_fabsf(x1) >= h) ? __builtin_copysignf(s, x1) : 0.f;
    } else if (kind == 3) {                                      // saturates by itself
        const uint32_t c = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, x0, x1, s, 0);
        v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(c, s, 0);
    } else {
        const float y0 = __builtin_amdgcn_fmed3f(x0, -bound, bound), y1 = __builtin_amdgcn_fmed3f(x1, -bound, bound);
        const v2s_t z = {0, 0};
        if (kind == 1) {
            const v2s_t c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, y0, y1, s, false);
            v = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(__builtin_bit_cast(uint32_t, c), s, false);
        } else {
            const v2s_t c = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(z, y0, y1, s, false);
            v = __builtin_amdgcn_cvt_scalef32_pk_f32_bf8(__builtin_bit_cast(uint32_t, c), s, false);
        }
    }
    v0 = v[0]; v1 = v[1];
}

// HW: 0 = arithmetic codec only; 1 = inliers and outliers through the hardware converts (the launcher
// guarantees both formats are e2m1 / e4m3 / e5m2); 2 = inliers through the hardware converts, posit outliers
// through posit_round_fast.  Blocks that are not "safe" (see below) take the arithmetic codec in every mode.
template <int BS, int RM, bool EMIT = false, int HW = 0>
MSQ_D int outlier_block_fast(float (&a)[BS], uint32_t (&mkw)[(BS + 31) / 32], float& se_in_o, float& se_out_o,
                             const OutlierArgs& A, int order, const float* vmean, const float* vstd, int64_t vstride,
                             uint32_t* codes = nullptr, int in_kind = 0, int out_kind = 0, float* lds_row = nullptr) {
    // lds_row (HW == 2 only): the lane's own copy of the block's ORIGINAL values in LDS (BS consecutive floats); it lets the
    // posit outliers be rounded one per loop trip instead of for every element (see below) and is clobbered.
    int status = 0;
    float lo = 0.f, hi = 0.f;
    if (A.variant == 0) {
        float ab[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) ab[b] = __builtin_fabsf(a[b]);
        float s;
        if (order == 1) s = sum_inner8<BS>(ab);
        else if (order == 2) s = sum_ilp4<BS>(ab);
        else s = sum_cascade<BS>(ab);
        const float mean = s / (float)BS;
        const float sd = std_twopass_checked<BS>(ab, 0);
        const float ks = A.k * sd;
        lo = mean - ks; hi = mean + ks;
    }
#pragma unroll
    for (int w = 0; w < (BS + 31) / 32; ++w) mkw[w] = 0u;
    float mx_in = 0.f, mx_o = 0.f;
    bool nonfinite = false;
    bool quirky = false;            // an element that could be pred(half the smallest step): the converts round it to 0, the reference does not
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        if (A.variant != 0) {
            const float mean = vmean[b * vstride], sd = vstd[b * vstride];
            const float ks = A.k * sd;
            lo = mean - ks; hi = mean + ks;
        }
        const bool m = (a[b] < lo) || (a[b] > hi);
        mkw[b >> 5] |= (m ? 1u : 0u) << (b & 31);
        const float t = __builtin_fabsf(a[b]);
        nonfinite |= ((f2u(a[b]) & 0x7F800000u) == 0x7F800000u);
        if (RM == 0 && !EMIT && HW != 0) quirky |= mantissa_all_ones(f2u(a[b]));
        mx_in = (!m && t > mx_in) ? t : mx_in;
        mx_o = (m && t > mx_o) ? t : mx_o;
    }
    // a NaN / Inf element poisons inlier_val = A*(1-mask) or outlier_val = A*mask (x*0 = NaN) and one of
    // the reference's NaN asserts fires (utils/quant.py:225-250): only the flag matters then
    if (nonfinite) { status |= MSQ_STATUS_NAN; mx_in = u2f(0x7FC00000u); }
    float se_in = shared_exp_of_max(mx_in);
    const bool fl = A.flush && !(se_in > -127.f);
    se_in = clamp_scale_exp(se_in - (float)A.fi.emax, A.in_sb, A.variant);
    const float sc_in = exp2f_int(se_in), rc_in = exp2f_int(-se_in);
    const float mx_out = mx_o * sc_in;                         // = max |outlier_val * 2^e_in|
    float se_out = shared_exp_of_max(mx_out);
    if (se_out != se_out) status |= MSQ_STATUS_NAN;
    se_out = clamp_scale_exp(se_out - (float)A.fo.emax, A.out_sb, A.variant);
    if (se_out != se_out) status |= MSQ_STATUS_NAN;
    const float sc_out = exp2f_int(se_out), rc_out = exp2f_int(-se_out);
    if (sc_in != sc_in) status |= MSQ_STATUS_NAN;
    const int sh_i = A.fi.mbits - 2, sh_o = A.fo.mbits - 2;
    const int lo_i = A.fi.ebits ? 2 - (1 << (A.fi.ebits - 1)) : 0, hi_i = A.fi.ebits ? 127 : 0;
    const int lo_o = A.fo.ebits ? 2 - (1 << (A.fo.ebits - 1)) : 0, hi_o = A.fo.ebits ? 127 : 0;
    const float pre_in = fl ? 0.f : rc_in;
    // ---- hardware codec path (gfx950 v_cvt_scalef32_pk_{fp4,fp8,bf8}_f32 and back): e2m1 inliers with e4m3 /
    // e5m2 / e2m1 outliers, round-to-nearest.  The converts round to nearest EVEN; the reference rounds half
    // AWAY from zero (elemwise_ops.py:64-65).  Setting the mantissa LSB of the fp32 input first turns every
    // exact tie into "just above the tie" and never moves any other value across a tie or a grid point (ties
    // and grid points of these <= 4-bit grids have >= 19 trailing zero mantissa bits), so RNE(x | 1ulp) is the
    // half-away result.  e4m3 / e5m2 converts do not saturate (> 464 -> NaN / > 61440 -> Inf): clamp to
    // +-max_norm * 2^scale first (max_norm is a grid point, so clamp-then-round == round-then-clamp).  All
    // 2^k products of the reference are exact when both block exponents are moderate; elements too small to
    // be scaled exactly round to zero on both routes.  Anything else takes the arithmetic codec below.
    if (RM == 0 && !EMIT && HW != 0) {
        const int in_hw = hw_codec_kind(A.fi), out_hw = hw_codec_kind(A.fo);
        // a block without (non-zero) outliers never uses the outlier scale: every masked element is +-0 and comes
        // out as +0 on either route, so its se_out (-127 with 8 scale bits in variant 1) must not force the slow path
        const bool no_out = (mx_o == 0.f);
        const bool safe = !nonfinite && !quirky && !fl && status == 0 && se_in >= -60.f && se_in <= 60.f &&
                          (no_out || (se_out >= -60.f && se_out <= 60.f));
        if (safe) {
            const int ei = (int)se_in, eo = no_out ? 0 : (int)se_out - (int)se_in;
            const float s_in = u2f((uint32_t)(ei + 127) << 23), s_eff = u2f((uint32_t)(eo + 127) << 23);
            const float b_in = A.fi.max_norm * s_in, b_out = A.fo.max_norm * s_eff;   // exact (|exponent| <= 120)
            // the format kinds are wave-uniform run-time values: select a loop specialised on them once, outside
            // the element loop (a per-pair switch costs four scalar branches per pair and serialises the pairs)
            const int combo = in_hw * 4 + ((HW == 1) ? out_hw : 0);
#define MSQ_HW_LOOP(KI, KO)                                                                                   \
            _Pragma("unroll") for (int b = 0; b < BS; b += 2) {                                               \
                const float x0 = u2f(f2u(a[b]) | 1u), x1 = u2f(f2u(a[b + 1]) | 1u);                           \
                float vi0, vi1, vo0, vo1;                                                                     \
                hw_codec_pair(KI, x0, x1, s_in, b_in, vi0, vi1);                                              \
                if (HW == 1) {                                                                                \
                    hw_codec_pair(KO, x0, x1, s_eff, b_out, vo0, vo1);                                        \
                } else {                                 /* posit outliers, reference op order (:216,:247,:258) */ \
                    vo0 = (posit_round_fast((a[b] * sc_in) * rc_out, A.fo.mbits, A.fo.ebits) * sc_out) * rc_in; \
                    vo1 = (posit_round_fast((a[b + 1] * sc_in) * rc_out, A.fo.mbits, A.fo.ebits) * sc_out) * rc_in; \
                }                                                                                             \
                const bool m0 = (mkw[b >> 5] >> (b & 31)) & 1u, m1 = (mkw[(b + 1) >> 5] >> ((b + 1) & 31)) & 1u; \
                a[b] = (m0 ? vo0 : vi0) + 0.0f;                                                               \
                a[b + 1] = (m1 ? vo1 : vi1) + 0.0f;                                                           \
            }
            if (HW == 2 && combo == 3 * 4 + 0 && lds_row != nullptr) {
                // e2m1 inliers + posit outliers, sparse form: the posit rounding (~18 instructions) is worth running only
                // where the mask is set -- about one element in twenty.  Every element first takes the inlier convert; then
                // each lane walks its mask bits, one outlier per trip (the wave makes as many trips as its fullest lane has
                // outliers, typically 4-6 of 32), fetching the original value by its run-time position from the lane's LDS
                // row -- a register array cannot be indexed like that -- and leaving the result there; one last pass merges.
#pragma unroll
                for (int b = 0; b < BS; b += 2) {
                    const float x0 = u2f(f2u(a[b]) | 1u), x1 = u2f(f2u(a[b + 1]) | 1u);
                    float vi0, vi1;
                    hw_codec_pair(3, x0, x1, s_in, b_in, vi0, vi1);
                    a[b] = vi0 + 0.0f; a[b + 1] = vi1 + 0.0f;
                }
#pragma unroll
                for (int w = 0; w < (BS + 31) / 32; ++w) {
                    uint32_t m = mkw[w];
                    while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
                        if (m != 0u) {
                            const int b = w * 32 + __builtin_ctz(m);
                            m &= m - 1u;
                            const float x = lds_row[b];
                            lds_row[b] = ((posit_round_fast((x * sc_in) * rc_out, A.fo.mbits, A.fo.ebits) * sc_out) * rc_in) + 0.0f;
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < BS / 4; ++c) {
                    const float4 r = *reinterpret_cast<const float4*>(lds_row + c * 4);
                    const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int b = c * 4 + j;
                        a[b] = ((mkw[b >> 5] >> (b & 31)) & 1u) ? rv[j] : a[b];
                    }
                }
            }
            else if (combo == 3 * 4 + 1) { MSQ_HW_LOOP(3, 1) }   // e2m1 inliers, e4m3 outliers
            else if (combo == 3 * 4 + 0) { MSQ_HW_LOOP(3, 0) }   // e2m1 inliers, posit outliers (HW == 2)
            else if (combo == 1 * 4 + 1) { MSQ_HW_LOOP(1, 1) }   // e4m3 / e4m3 (activations)
            else if (combo == 3 * 4 + 2) { MSQ_HW_LOOP(3, 2) }   // e2m1 / e5m2
            else if (combo == 3 * 4 + 3) { MSQ_HW_LOOP(3, 3) }   // e2m1 / e2m1 (MXLinear weights)
            else if (combo == 4 * 4 + 3) { MSQ_HW_LOOP(4, 3) }   // int2 / fp4: the reference harness default
            else { MSQ_HW_LOOP(in_hw, out_hw) }                  // any other pair: kinds stay run-time values
#undef MSQ_HW_LOOP
            se_in_o = se_in; se_out_o = se_out;
            return status;
        }
    }
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        const bool m = (mkw[b >> 5] >> (b & 31)) & 1u;
        float t = a[b] * (m ? sc_in : 1.0f);                    // :216 (outliers only; x1 is exact)
        t = t * (m ? rc_out : pre_in);                          // :247 / :214 (+ flush :202)
        float q;
        if (A.fo.kind == 1) {                                   // posit outliers: per-element choice of codec
            q = m ? posit_round_fast(t, A.fo.mbits, A.fo.ebits)
                  : quant_core_fast<RM>(t, sh_i, lo_i, hi_i, A.fi.max_norm, A.rmode);
        } else {
            q = quant_core_fast<RM>(t, m ? sh_o : sh_i, m ? lo_o : lo_i, m ? hi_o : hi_i,
                                    m ? A.fo.max_norm : A.fi.max_norm, A.rmode);
        }
        float u = q * (m ? sc_out : sc_in);                     // :258 / :224
        u = u * (m ? rc_in : 1.0f);                             // :258
        if (u != u) status |= MSQ_STATUS_NAN;
        a[b] = u + 0.0f;                                        // inlier + outlier part: the other part is +0
        if (EMIT) {
            // plane codes + proof that value == code * 2^scale exactly (see outlier_block<>)
            uint32_t cd;
            bool exact = true;
            const float v = a[b];
            if (in_kind == MSQ_PLANE_NONE) {
                cd = (f2u(v) >> 16) << 8;
                exact = (u2f((f2u(v) >> 16) << 16) == v) || (v != v);
            } else if (!m) {
                cd = encode_e2m1(q);
                if (se_in == se_in) exact = (scale_pow2(decode_e2m1(cd), (int)se_in) == v);
            } else if (out_kind == MSQ_PLANE_BF16) {
                cd = (f2u(v) >> 16) << 8;
                exact = (u2f((f2u(v) >> 16) << 16) == v) || (v != v);
            } else {
                cd = ((out_kind == MSQ_PLANE_BF8) ? encode_e5m2(q) : encode_e4m3(q)) << 8;
                const float eff = se_out - se_in;
                if (q != 0.f) exact = (eff >= -127.f) && (eff <= 127.f) && (scale_pow2(q, (int)(eff == eff ? eff : 0.f)) == v);
            }
            codes[b] = cd;
            if (!exact) status |= MSQ_STATUS_INEXACT;
        }
    }
    se_in_o = se_in; se_out_o = se_out;
    return status;
}
