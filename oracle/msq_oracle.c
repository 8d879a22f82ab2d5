/*
 * msq_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar, single-threaded C restatement of the reference's CPU fake-quant
 * hot path (MicroScopiQ outlier-aware microscaling quant/dequant).  It is the
 * checker for the HIP product path: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product package never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function in
 * this file against the fixtures in tests/golden/ (npz, json), produced by importing
 * the reference (torch 2.10 CPU, this container) with tests/golden/make_golden.py
 * and by transcribing the input/expected vectors of the reference's own KATs
 * (number_system/mx/tests/test_corners_mx.py, test_fp8_e4m3_fix.py,
 * test_e5m0_scale.py, test_corners_elemwise.py, test_formats.py).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference root).  Arithmetic is done in IEEE binary32 exactly where the
 * reference does fp32 tensor ops, and in binary64 where ATen's CPU kernels do
 * (torch.std accumulates Welford in double: aten WelfordOps<float,double>).
 *
 * One documented deviation from the literal Python:
 *  (D1, REMOVED in round 4) floor(log2(x)) of the Python path is torch.log2 rounded to float32 and floored (floor_log2_torch
 *       below): the largest few floats under a power of two come out one binade high.  The native surfaces keep the exact
 *       exponent, as the reference's own native kernels do (cpp/mx.cuh:81-85, cpp/quantize.cuh:97).
 *  (D2) the `+1e-6` of mx_ops.py:444 is a reference defect (SURVEY.md section 4:
 *       it breaks 3 of the reference's own KATs); it is only applied when the
 *       caller passes plus_eps_defect=1 (used to pin the defect variant).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MSQ_RD_NEAREST 0 /* half away from zero; formats.py:15-18, common.cuh rd_away */
#define MSQ_RD_FLOOR 1   /* truncate toward zero */
#define MSQ_RD_EVEN 2

#define MSQ_VARIANT_QUANT 0 /* utils/quant.py:147-266 (canonical, a6) */
#define MSQ_VARIANT_MXOPS 1 /* number_system/mx/mx_ops.py:210-330 (a10) */

/* format kinds for the element codec */
#define MSQ_KIND_FLOAT 0 /* eXmY / intN through _quantize_elemwise_core */
#define MSQ_KIND_POSIT 1 /* posit<n,es> round-to-nearest-even (new; oracle = posit/Posit.py) */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* 2^e as an fp32 value; e outside [-149,127] saturates to 0 / +inf exactly as
 * torch.pow(2.0f, e) does.  NaN exponent -> NaN (utils/quant.py:210,214). */
static float exp2_float(float e) {
    if (e != e) return NAN;
    if (isinf(e)) return e > 0 ? INFINITY : 0.0f;
    return ldexpf(1.0f, (int)e);
}

/* exact floor(log2(x)) for finite x > 0, subnormals included: what the reference's NATIVE kernels compute from the exponent field
 * (cpp/mx.cuh:81-85, cpp/quantize.cuh:97) */
static int floor_log2_exact(float x) {
    int e;
    (void)frexpf(x, &e); /* x = m * 2^e, m in [0.5,1) */
    return e - 1;
}
/* floor(torch.log2(x)) as the reference's PYTHON path computes it on a float32 tensor (utils/quant.py:525-529 shared exponents,
 * elemwise_ops.py:139-140 private exponents): torch.log2 returns the float32 NEAREST to the true logarithm, so for the few floats
 * just below a power of two 2^u -- log2 x = u - d with d below half a float32 spacing next to u -- the result IS u and the floor
 * lands one binade too high.  x = m 2^e (m in [0.5, 1)), u = e: bumped iff -log2(m) < h, h = half the spacing of float32 on the
 * side of u the result approaches from (toward zero for u > 0: the binade below a power of two is finer).  That is the 1 (|u| 2..3),
 * 2 (4..7), 5 (8..15), 11 (16..31), 22 (32..63), 44, 88 largest floats below 2^u (one step less at u = 2, 4, 8, ... > 0).
 * Pinned by tests/golden/log2_f32.npz (made with torch, vector and scalar paths, every binade).  Round 1-3 of this build used the
 * exact exponent here ("deviation D1, measure-zero"); an OPT-125M-shaped fixture of 14 M weights hit it (two values of one block),
 * so it is restated exactly now. */
static float floor_log2_torch(float x) {
    if (x != x) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (isinf(x)) return INFINITY;
    int e;
    const double m = frexp((double)x, &e);                 /* x = m 2^e, m in [0.5, 1) */
    const int k = e - 1, u = e;
    if (u == 0) return (float)k;                           /* results in (-1, 0): every float there is exact enough */
    const int au = u < 0 ? -u : u;
    int jb = 0; while ((1 << (jb + 1)) <= au) jb++;        /* floor(log2 |u|) */
    if (u > 0 && (au & (au - 1)) == 0) jb -= 1;            /* approaching a positive power of two from below: the finer binade */
    const double h = ldexp(1.0, jb - 24);                  /* half a float32 spacing next to u */
    return (-log2(m) < h) ? (float)u : (float)k;
}

/* ------------------------------------------------------------------------
 * a1  formats.py:65-129  _get_format_params
 * returns 0 on success, -1 for an unknown name.
 * posit<n>_es<k> names are an extension: kind=POSIT,
 * ebits=es, mbits=n, emax=1 (block max is scaled into [2,4), the top of the
 * posit's full-precision region), max_norm=maxpos.
 * ---------------------------------------------------------------------- */
int msq_oracle_format_params(const char* name, int* ebits, int* mbits, int* emax,
                             float* max_norm, float* min_norm, int* kind) {
    int e = 0, m = 0, ex = 0, k = MSQ_KIND_FLOAT;
    double mx = 0.0, mn = 0.0;
    int is_e4m3 = 0;
    if (!strcmp(name, "int8")) { e = 0; m = 8; ex = 0; }
    else if (!strcmp(name, "int4")) { e = 0; m = 4; ex = 0; }
    else if (!strcmp(name, "int2")) { e = 0; m = 2; ex = 0; }
    else if (!strcmp(name, "fp8_e5m2")) { e = 5; m = 4; ex = 15; }
    else if (!strcmp(name, "fp8_e4m3")) { e = 4; m = 5; ex = 8; is_e4m3 = 1; }
    else if (!strcmp(name, "fp6_e3m2")) { e = 3; m = 4; ex = 4; }
    else if (!strcmp(name, "fp6_e2m3")) { e = 2; m = 5; ex = 2; }
    else if (!strcmp(name, "fp4") || !strcmp(name, "fp4_e2m1")) { e = 2; m = 3; ex = 2; }
    else if (!strcmp(name, "float16") || !strcmp(name, "fp16")) { e = 5; m = 12; ex = 15; }
    else if (!strcmp(name, "bfloat16") || !strcmp(name, "bf16")) { e = 8; m = 9; ex = 127; }
    else if (!strncmp(name, "posit", 5)) {
        int n = 0, es = 0;
        const char* p = name + 5;
        while (*p >= '0' && *p <= '9') n = n * 10 + (*p++ - '0');
        if (strncmp(p, "_es", 3)) return -1;
        p += 3;
        if (*p < '0' || *p > '9') return -1;
        while (*p >= '0' && *p <= '9') es = es * 10 + (*p++ - '0');
        if (*p || n < 3 || n > 16 || es > 3) return -1;
        k = MSQ_KIND_POSIT; e = es; m = n; ex = 1;
        mx = ldexp(1.0, (1 << es) * (n - 2)); /* maxpos = useed^(n-2) */
        mn = ldexp(1.0, -(1 << es) * (n - 2));
    } else return -1;
    if (k == MSQ_KIND_FLOAT) {
        if (!is_e4m3) mx = ldexp(1.0, ex) * (double)((1 << (m - 1)) - 1) / ldexp(1.0, m - 2);
        else mx = ldexp(1.0, ex) * 1.75; /* formats.py:122-123 */
        mn = (e == 0) ? 0.0 : ldexp(1.0, 2 - (1 << (e - 1))); /* formats.py:50-54 */
    }
    *ebits = e; *mbits = m; *emax = ex; *max_norm = (float)mx; *min_norm = (float)mn; *kind = k;
    return 0;
}

/* ------------------------------------------------------------------------
 * a2  elemwise_ops.py:47-78  _round_mantissa (no clamp)
 * ---------------------------------------------------------------------- */
static float round_mantissa(float a, int round_mode) {
    float s = (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f); /* torch.sign; NaN -> handled by caller */
    float absa = fabsf(a);
    if (a != a) return a;
    if (round_mode == MSQ_RD_FLOOR) return s * floorf(absa);
    if (round_mode == MSQ_RD_NEAREST) return s * floorf(absa + 0.5f);
    /* even: elemwise_ops.py:66-70 */
    float t = absa - 0.5f;
    float r = fmodf(t, 2.0f);
    if (r != 0.0f && ((r < 0.0f) != (2.0f < 0.0f))) r += 2.0f; /* python-style % */
    float maska = (r == 0.0f) ? 1.0f : 0.0f;
    return s * (floorf(absa + 0.5f) - maska);
}

/* a2  elemwise_ops.py:84-174  _quantize_elemwise_core, custom_cuda=False branch,
 * restated op for op in fp32.  Scalar form. */
static float quantize_elemwise_core_1(float a, int bits, int exp_bits, float max_norm,
                                      int round_mode, int saturate_normals, int allow_denorm) {
    float out = a;
    if (!allow_denorm && exp_bits > 0) { /* :132-134  (|A| >= min_norm).type(dtype) * A */
        float min_norm = ldexpf(1.0f, 2 - (1 << (exp_bits - 1)));
        out = ((fabsf(a) >= min_norm) ? 1.0f : 0.0f) * a;
    }
    int have_pe = (exp_bits != 0);
    float pe = 0.0f;
    if (have_pe) { /* :138-144 */
        float t = fabsf(a) + ((a == 0.0f) ? 1.0f : 0.0f);
        if (t != t) pe = NAN;
        else if (isinf(t)) pe = INFINITY;
        else pe = floor_log2_torch(t);
        float min_exp = (float)(-(1 << (exp_bits - 1)) + 2);
        if (pe == pe && pe < min_exp) pe = min_exp;
    }
    /* _safe_lshift :33-37 */
    float sh = ldexpf(1.0f, bits - 2);
    if (have_pe) out = out / exp2_float(pe) * sh;
    else out = out * sh;
    out = round_mantissa(out, round_mode);
    /* _safe_rshift :40-44 */
    if (have_pe) out = out / sh * exp2_float(pe);
    else out = out / sh;
    if (saturate_normals || exp_bits == 0) { /* :157-158 torch.clamp keeps NaN */
        if (out == out) { if (out < -max_norm) out = -max_norm; if (out > max_norm) out = max_norm; }
    } else { /* :160-161 */
        if (fabsf(out) > max_norm) out = (out > 0.0f) ? INFINITY : -INFINITY;
    }
    if (a == INFINITY) out = INFINITY; /* :165-167 */
    if (a == -INFINITY) out = -INFINITY;
    return out;
}

void msq_oracle_quantize_elemwise_core(const float* in, float* out, int64_t n, int bits, int exp_bits,
                                       float max_norm, int round_mode, int saturate_normals,
                                       int allow_denorm) {
    for (int64_t i = 0; i < n; ++i)
        out[i] = quantize_elemwise_core_1(in[i], bits, exp_bits, max_norm, round_mode,
                                          saturate_normals, allow_denorm);
}

/* ------------------------------------------------------------------------
 * a12  cpp/quantize.cuh:15-149  integer bit-manipulation codec (what
 * quantize_elemwise_func_cpp / _cuda run, cpp/funcs.cpp:98-133).
 * ---------------------------------------------------------------------- */
static float quantize_elemwise_bits_1(float input, int bits, int exp_bits, float max_norm,
                                      int round_mode, int saturate_normals, int allow_denorm) {
    uint32_t u = f2u(input);
    int biased_exp = (int)((u >> 23) & 0xFF);
    int sign = (int)(u >> 31);
    int tmant = (int)(u & 0x7FFFFF);
    const int mbits = bits - 1;
    const int is_int = (exp_bits == 0);
    const int new_bias = is_int ? 1 : (1 << (exp_bits - 1)) - 1;
    const int new_biased_exp = biased_exp - 127 + new_bias;
    if (!is_int && !allow_denorm && new_biased_exp < 1) return 0.0f; /* quantize.cuh:111-113 */
    int exp_diff = (new_biased_exp <= 0) ? 1 - new_biased_exp : 0;
    if (exp_diff > 24) exp_diff = 24;
    /* shift_right_round_mantissa quantize.cuh:15-56 */
    const int is_subnorm = (biased_exp == 0);
    int mant = is_subnorm ? tmant : tmant + (1 << 23);
    const int sig_bits = is_subnorm ? 23 : 24;
    int tie = 0, even = 0;
    if (round_mode == MSQ_RD_EVEN) {
        int tbits = exp_diff + (sig_bits - mbits);
        int mask = (1 << (tbits - 1)) - 1;
        tie = !(mant & mask);
        mask = (1 << tbits);
        even = !(mant & mask);
    }
    mant = mant >> exp_diff;
    mant = mant >> (sig_bits - mbits - 1);
    const int allow_overflow = !is_int;
    if ((round_mode == MSQ_RD_NEAREST || round_mode == MSQ_RD_EVEN) &&
        (allow_overflow || mant != ((1 << (mbits + 1)) - 1))) {
        if (!(tie && even)) mant = mant + 1;
    }
    mant = mant >> 1;
    if (mant == 0) return 0.0f;
    /* shift_left_mantissa quantize.cuh:64-79 */
    mant = mant << (sig_bits - mbits + exp_diff);
    const int overflow = (mant >= (1 << sig_bits));
    mant = (overflow && !is_subnorm) ? mant >> 1 : mant;
    mant = mant & ((1 << 23) - 1);
    if (overflow) biased_exp += 1;
    float output = u2f(((uint32_t)sign << 31) | ((uint32_t)biased_exp << 23) | (uint32_t)mant);
    if (fabsf(output) > max_norm) { /* quantize.cuh:142-147 */
        if (is_int || saturate_normals) output = sign ? -max_norm : max_norm;
        else output = u2f(((uint32_t)sign << 31) | (0xFFu << 23));
    }
    return output;
}

void msq_oracle_quantize_elemwise_bits(const float* in, float* out, int64_t n, int bits, int exp_bits,
                                       float max_norm, int round_mode, int saturate_normals,
                                       int allow_denorm) {
    for (int64_t i = 0; i < n; ++i)
        out[i] = quantize_elemwise_bits_1(in[i], bits, exp_bits, max_norm, round_mode,
                                          saturate_normals, allow_denorm);
}

/* ------------------------------------------------------------------------
 * a13  posit/Posit.py:221-385  standard posit<n,es> (rs=None): encode with
 * round-to-nearest-even on the bit pattern, never rounding to 0 / NaR
 * (Posit.py:266-272), decode (:337-385).  n<=16, es<=3.
 * ---------------------------------------------------------------------- */
double msq_oracle_posit_decode(uint32_t code, int n, int es) {
    uint32_t mask = (n == 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
    code &= mask;
    if (code == 0) return 0.0;
    if (code == (1u << (n - 1))) return NAN; /* NaR (Posit.py prints inf) */
    int sign = (code >> (n - 1)) & 1;
    uint32_t x = sign ? ((~code + 1u) & mask) : code;
    int regime_sign = (x >> (n - 2)) & 1;
    int rl = 0; /* run length */
    for (int b = n - 2; b >= 0; --b) {
        if ((int)((x >> b) & 1) == regime_sign) rl++; else break;
    }
    int k = regime_sign ? rl - 1 : -rl;
    int used = 1 + rl + 1; /* sign + run + terminator */
    int rem = n - used; if (rem < 0) rem = 0;
    uint32_t tail = x & ((rem >= 32) ? 0xFFFFFFFFu : ((1u << rem) - 1u));
    int ebits = es < rem ? es : rem;
    int fbits = rem - ebits;
    uint32_t e = (fbits >= 32 ? 0 : (tail >> fbits)) << (es - ebits);
    uint32_t f = tail & ((1u << fbits) - 1u);
    double v = ldexp(1.0 + (double)f / (double)(1u << fbits), (1 << es) * k + (int)e);
    return sign ? -v : v;
}

uint32_t msq_oracle_posit_encode(double v, int n, int es) {
    uint32_t mask = (1u << n) - 1u;
    if (v == 0.0) return 0;
    if (v != v || isinf(v)) return 1u << (n - 1);
    int sign = v < 0; double a = fabs(v);
    int ex; double m = frexp(a, &ex); /* a = m*2^ex, m in [0.5,1) */
    int scale = ex - 1; double frac = m * 2.0 - 1.0; /* a = (1+frac)*2^scale */
    int useed_log = 1 << es;
    int k = (scale >= 0) ? scale / useed_log : -((-scale + useed_log - 1) / useed_log);
    int e = scale - k * useed_log;
    int rl = (k >= 0) ? k + 2 : -k + 1; /* regime field length incl. terminator */
    const uint32_t maxpos = (1u << (n - 1)) - 1u, minpos = 1u;
    uint32_t body;
    if (rl >= n) { /* Posit.py:266-272 */
        body = (k >= 0) ? maxpos : minpos;
    } else {
        /* build an exact wide bit string: regime | exponent | fraction(52 bits) */
        /* value of the n-1 body bits as a real number, then RNE */
        int avail = n - 1 - rl; /* bits for exponent+fraction */
        uint64_t regime_bits = (k >= 0) ? (((1ull << (rl - 1)) - 1ull) << 1) : 1ull;
        /* exp_frac as integer with 52 fraction bits */
        uint64_t fr52 = (uint64_t)ldexp(frac, 52);
        /* combine exponent (es bits) and fraction: total es+52 bits */
        /* we need the top `avail` bits of [e | fr52] */
        int total = es + 52;
        uint64_t hi, rest; int restbits;
        /* use 128-bit via two parts: e fits in 3 bits, fr52 in 52 -> 55 bits fits in uint64 */
        uint64_t ef = ((uint64_t)e << 52) | fr52;
        if (avail >= total) { hi = ef << (avail - total); rest = 0; restbits = 0; }
        else { restbits = total - avail; hi = ef >> restbits; rest = ef & ((1ull << restbits) - 1ull); }
        uint64_t b = (regime_bits << avail) | hi;
        if (restbits > 0) {
            uint64_t half = 1ull << (restbits - 1);
            /* tie -> even on the last KEPT exponent/fraction bit; with no kept bit the
             * reference rounds the tie down (Posit.py:309-313 checkBit beyond exp_frac) */
            if (rest > half || (rest == half && avail > 0 && (hi & 1ull))) b += 1;
        }
        if (b == 0) b = minpos;            /* never round to zero */
        if (b > maxpos) b = maxpos;        /* never round to NaR */
        body = (uint32_t)b;
    }
    return sign ? ((~body + 1u) & mask) : body;
}

static float posit_round_1(float a, int n, int es) {
    if (a != a) return a;
    if (isinf(a)) return a;
    return (float)msq_oracle_posit_decode(msq_oracle_posit_encode((double)a, n, es), n, es);
}

/* element codec dispatch (float kinds via the fp32 Python restatement) */
typedef struct { int kind, ebits, mbits, emax; float max_norm; } fmt_t;

static float quant_elem(float a, const fmt_t* f, int round_mode) {
    if (f->kind == MSQ_KIND_POSIT) {
        float r = posit_round_1(a, f->mbits, f->ebits);
        return r;
    }
    /* MicroScopiQ always uses allow_denorm=True, saturate_normals=True (utils/quant.py:218-221) */
    return quantize_elemwise_core_1(a, f->mbits, f->ebits, f->max_norm, round_mode, 1, 1);
}

/* ------------------------------------------------------------------------
 * torch CPU reduction order for fp32 sum (what torch.mean divides), pinned
 * empirically against torch 2.10 CPU (tests/golden/make_golden.py, 'sum_order').
 *   outer (reduced dim not innermost): aten SumKernel multi_row_sum cascade,
 *     level_step = 2^max(4, ceil_log2(n)/4)
 *   inner (reduced dim contiguous): 8-lane (AVX2 Vectorized<float>) strided
 *     partials, 4-way ILP over vector rows, then lanes summed left to right.
 * ---------------------------------------------------------------------- */
static int ceil_log2_i(int64_t x) { int l = 0; while (((int64_t)1 << l) < x) l++; return l; }

static float sum_cascade(const float* x, int64_t n, int64_t stride) {
    int lp = ceil_log2_i(n) / 4; if (lp < 4) lp = 4;
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    float acc[4] = {0, 0, 0, 0};
    int64_t i = 0;
    while (i + step <= n) {
        for (int64_t j = 0; j < step; ++j, ++i) acc[0] += x[i * stride];
        for (int j = 1; j < 4; ++j) {
            acc[j] += acc[j - 1]; acc[j - 1] = 0;
            if ((i & (lmask << (j * lp))) != 0) break;
        }
    }
    for (; i < n; ++i) acc[0] += x[i * stride];
    for (int j = 1; j < 4; ++j) acc[0] += acc[j];
    return acc[0];
}

/* aten SumKernel row_sum<scalar>: 4 interleaved accumulators (ILP), each fed by
 * multi_row_sum's cascade over n/4 steps, remainder into acc[0], then
 * ((a0+a1)+a2)+a3.  Used by torch for the columns that do not fill a
 * 4-vector (32-column) group of an outer reduction. */
static float sum_ilp4(const float* x, int64_t n, int64_t stride) {
    const int64_t s = n / 4;
    int lp = ceil_log2_i(s > 0 ? s : 1) / 4; if (lp < 4) lp = 4;
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    float acc[4][4]; memset(acc, 0, sizeof(acc));
    int64_t i = 0;
    while (i + step <= s) {
        for (int64_t j = 0; j < step; ++j, ++i)
            for (int k = 0; k < 4; ++k) acc[0][k] += x[(i * 4 + k) * stride];
        for (int j = 1; j < 4; ++j) {
            for (int k = 0; k < 4; ++k) { acc[j][k] += acc[j - 1][k]; acc[j - 1][k] = 0; }
            if ((i & (lmask << (j * lp))) != 0) break;
        }
    }
    for (; i < s; ++i) for (int k = 0; k < 4; ++k) acc[0][k] += x[(i * 4 + k) * stride];
    for (int j = 1; j < 4; ++j) for (int k = 0; k < 4; ++k) acc[0][k] += acc[j][k];
    for (int64_t r = s * 4; r < n; ++r) acc[0][0] += x[r * stride];
    for (int k = 1; k < 4; ++k) acc[0][0] += acc[0][k];
    return acc[0][0];
}

/* which of the three orders torch uses for column q of an outer reduction with
 * `post` contiguous non-reduced columns (pinned empirically, torch 2.10 CPU):
 *   post == 1            -> inner (8-lane) order
 *   post >= 8            -> cascade for q < 32*floor(post/32), else ilp4
 *   2 <= post < 8        -> cascade for q < 4*floor(post/4),  else ilp4      */
#define MSQ_ORDER_CASCADE 0
#define MSQ_ORDER_INNER8 1
#define MSQ_ORDER_ILP4 2
static int sum_order_for(int64_t post, int64_t q) {
    if (post == 1) return MSQ_ORDER_INNER8;
    int64_t lim = (post >= 8) ? (post / 32) * 32 : (post / 4) * 4;
    return (q < lim) ? MSQ_ORDER_CASCADE : MSQ_ORDER_ILP4;
}

static float sum_inner_v8(const float* x, int64_t n) {
    enum { V = 8, ILP = 4 };
    const int64_t vec_size = n / V;
    const int64_t size_ilp = vec_size / ILP;
    float part[ILP][V];
    memset(part, 0, sizeof(part));
    {   /* multi_row_sum over size_ilp "rows" of ILP vectors, same cascade */
        int lp = ceil_log2_i(size_ilp > 0 ? size_ilp : 1) / 4; if (lp < 4) lp = 4;
        const int64_t step = (int64_t)1 << lp, lmask = step - 1;
        float acc[4][ILP][V]; memset(acc, 0, sizeof(acc));
        int64_t i = 0;
        while (i + step <= size_ilp) {
            for (int64_t j = 0; j < step; ++j, ++i)
                for (int k = 0; k < ILP; ++k)
                    for (int l = 0; l < V; ++l) acc[0][k][l] += x[(i * ILP + k) * V + l];
            for (int j = 1; j < 4; ++j) {
                for (int k = 0; k < ILP; ++k) for (int l = 0; l < V; ++l) {
                    acc[j][k][l] += acc[j - 1][k][l]; acc[j - 1][k][l] = 0; }
                if ((i & (lmask << (j * lp))) != 0) break;
            }
        }
        for (; i < size_ilp; ++i)
            for (int k = 0; k < ILP; ++k)
                for (int l = 0; l < V; ++l) acc[0][k][l] += x[(i * ILP + k) * V + l];
        for (int j = 1; j < 4; ++j)
            for (int k = 0; k < ILP; ++k) for (int l = 0; l < V; ++l) acc[0][k][l] += acc[j][k][l];
        memcpy(part, acc[0], sizeof(part));
    }
    for (int64_t i = size_ilp * ILP; i < vec_size; ++i)
        for (int l = 0; l < V; ++l) part[0][l] += x[i * V + l];
    for (int k = 1; k < ILP; ++k) for (int l = 0; l < V; ++l) part[0][l] += part[k][l];
    float fin = 0.0f;
    for (int64_t k = vec_size * V; k < n; ++k) fin += x[k];
    for (int l = 0; l < V; ++l) fin += part[0][l];
    return fin;
}

static float sum_ordered(const float* x, int64_t n, int order) {
    if (order == MSQ_ORDER_INNER8) return sum_inner_v8(x, n);
    if (order == MSQ_ORDER_ILP4) return sum_ilp4(x, n, 1);
    return sum_cascade(x, n, 1);
}

/* torch.std on CPU: sequential Welford in double (aten WelfordOps), sqrt in
 * double, one rounding to float.  correction = 0 (population) or 1. */
static float std_welford(const float* x, int64_t n, int64_t stride, int correction) {
    double mean = 0.0, m2 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double d = (double)x[i * stride];
        double delta = d - mean;
        mean = mean + delta / (double)(i + 1);
        double delta2 = d - mean;
        m2 = m2 + delta * delta2;
    }
    double denom = (double)n - (double)correction;
    if (denom < 0) denom = 0;
    return (float)sqrt(m2 / denom);
}

/* ------------------------------------------------------------------------
 * a5  utils/quant.py:498-541 _shared_exponents (method="max", ebits=0) for one
 * block already reduced to its max |.|.
 * ---------------------------------------------------------------------- */
static float shared_exp_of_max(float mx) {
    if (mx != mx) return NAN;
    float t = mx + ((mx == 0.0f) ? ldexpf(1.0f, -126) : 0.0f); /* FP32_MIN_NORMAL formats.py:12 */
    if (isinf(t)) return INFINITY;
    return floor_log2_torch(t);
}

/* clamp of utils/quant.py:207-211 / :237-242 (variant 0) and mx_ops.py:269-273 (variant 1) */
static float clamp_scale_exp(float e, int scale_bits, int variant) {
    const float scale_emax = (float)((1 << (scale_bits - 1)) - 1);
    if (e > scale_emax) return NAN;
    if (e < -scale_emax) {
        if (variant == MSQ_VARIANT_QUANT) return (-scale_emax < -20.0f) ? -20.0f : -scale_emax;
        return -scale_emax;
    }
    return e;
}

/* ------------------------------------------------------------------------
 * a3+a4+a5+a6 (+a7 num_outliers, +a10 variant)
 *
 * Tensor = [pre, axis_len, post] contiguous; blocks of `block` run along the
 * middle axis; the axis is zero-padded to a multiple of `block`
 * (utils/quant.py:563-583) and the padding takes part in the statistics.
 *
 * Outputs (any may be NULL): out (fake-quant result, same shape as in),
 * mask (uint8 0/1, same shape), e_in / e_out (float per block, layout
 * [pre, nblk, post], NaN kept), n_out (a7: utils/quant.py:66 num_outliers, int8,
 * one per every block-th block and post index, only when pre==1), status bit0 = a NaN assertion of the
 * reference would have fired (utils/quant.py:225-250).
 * Returns status.
 * ---------------------------------------------------------------------- */
int msq_oracle_outlier_fakequant(const float* in, float* out, uint8_t* mask, float* e_in_o,
                                 float* e_out_o, int8_t* n_out, int64_t pre, int64_t axis_len, int64_t post,
                                 int block, const char* inlier_fmt, const char* outlier_fmt,
                                 int inlier_scale_bits, int outlier_scale_bits, double std_dev,
                                 int round_mode, int flush_fp32_subnorms, int variant) {
    fmt_t fi, fo; float mn;
    if (msq_oracle_format_params(inlier_fmt, &fi.ebits, &fi.mbits, &fi.emax, &fi.max_norm, &mn, &fi.kind)) return -1;
    if (msq_oracle_format_params(outlier_fmt, &fo.ebits, &fo.mbits, &fo.emax, &fo.max_norm, &mn, &fo.kind)) return -1;
    if (block <= 0) block = (int)axis_len;
    const int64_t nblk = (axis_len + block - 1) / block;
    int status = 0;
    const float k = (float)std_dev; /* python scalar * fp32 tensor -> fp32 mul */

    /* variant 1 (mx_ops.py:248,62-66): statistics of the SIGNED values over the
     * block-COUNT axis, unbiased std: one (mean,std) per (p, intra-block pos, q) */
    float *vmean = NULL, *vstd = NULL, *col = NULL;
    if (variant == MSQ_VARIANT_MXOPS) {
        vmean = (float*)malloc(sizeof(float) * pre * block * post);
        vstd = (float*)malloc(sizeof(float) * pre * block * post);
        /* threads split the independent (p, b) statistics; every value is computed by one thread in the
         * reference's order, so the result does not depend on the thread count */
#pragma omp parallel reduction(| : status) private(col)
        {
        col = (float*)malloc(sizeof(float) * nblk);
#pragma omp for collapse(2) schedule(static)
        for (int64_t p = 0; p < pre; ++p) for (int b = 0; b < block; ++b) for (int64_t q = 0; q < post; ++q) {
            for (int64_t nb = 0; nb < nblk; ++nb) {
                int64_t ai = nb * block + b;
                col[nb] = (ai < axis_len) ? in[(p * axis_len + ai) * post + q] : 0.0f;
            }
            /* reduced dim = block count; the (block, post) dims behind it are one
             * contiguous run of block*post columns for ATen */
            float s = sum_ordered(col, nblk, sum_order_for((int64_t)block * post, (int64_t)b * post + q));
            vmean[(p * block + b) * post + q] = s / (float)nblk;
            vstd[(p * block + b) * post + q] = std_welford(col, nblk, 1, 1);
            if (vstd[(p * block + b) * post + q] != vstd[(p * block + b) * post + q]) status |= 1;
        }
        free(col);
        }
        col = NULL;
    }

    /* blocks are independent (utils/quant.py works on the whole tensor at once): OpenMP threads take whole
     * (p, nb) slabs with private scratch; per-element arithmetic and order are untouched */
#pragma omp parallel reduction(| : status)
    {
    float* a = (float*)malloc(sizeof(float) * block * 6);
    float *absa = a + block, *mk = a + 2 * block, *inl = a + 3 * block, *outl = a + 4 * block, *tmp = a + 5 * block;
#pragma omp for collapse(2) schedule(static)
    for (int64_t p = 0; p < pre; ++p) for (int64_t nb = 0; nb < nblk; ++nb) for (int64_t q = 0; q < post; ++q) {
        /* gather block with zero padding (a3) */
        for (int b = 0; b < block; ++b) {
            int64_t ai = nb * block + b;
            a[b] = (ai < axis_len) ? in[(p * axis_len + ai) * post + q] : 0.0f;
            absa[b] = fabsf(a[b]);
        }
        /* a4 mask */
        if (variant == MSQ_VARIANT_QUANT) {
            float s = sum_ordered(absa, block, sum_order_for(post, q));
            float mean = s / (float)block;
            float sd = std_welford(absa, block, 1, 0);
            float lo = mean - k * sd, hi = mean + k * sd; /* utils/quant.py:489-490 */
            for (int b = 0; b < block; ++b) mk[b] = ((a[b] < lo) || (a[b] > hi)) ? 1.0f : 0.0f;
        } else {
            for (int b = 0; b < block; ++b) {
                float mean = vmean[(p * block + b) * post + q], sd = vstd[(p * block + b) * post + q];
                float lo = mean - k * sd, hi = mean + k * sd;
                mk[b] = ((a[b] < lo) || (a[b] > hi)) ? 1.0f : 0.0f;
            }
        }
        /* utils/quant.py:192-193 */
        float mx_in = 0.0f;
        for (int b = 0; b < block; ++b) {
            inl[b] = a[b] * (1.0f - mk[b]);
            outl[b] = a[b] * mk[b];
            float t = fabsf(inl[b]); if (t > mx_in || t != t) mx_in = t;
        }
        float se_in = shared_exp_of_max(mx_in);                       /* :196-198 */
        if (flush_fp32_subnorms && !(se_in > -127.0f))                 /* :201-202 */
            for (int b = 0; b < block; ++b) inl[b] = inl[b] * 0.0f;
        se_in = se_in - (float)fi.emax;                                /* :207 */
        se_in = clamp_scale_exp(se_in, inlier_scale_bits, variant);    /* :208-211 */
        const float sc_in = exp2_float(se_in);
        float mx_out = 0.0f;
        for (int b = 0; b < block; ++b) {
            inl[b] = inl[b] / sc_in;                                   /* :214 */
            outl[b] = outl[b] * sc_in;                                 /* :216 */
            inl[b] = quant_elem(inl[b], &fi, round_mode);              /* :218-221 */
            inl[b] = inl[b] * sc_in;                                   /* :224 */
            if (inl[b] != inl[b] || outl[b] != outl[b]) status |= 1;   /* :225-226 */
            float t = fabsf(outl[b]); if (t > mx_out || t != t) mx_out = t;
        }
        float se_out = shared_exp_of_max(mx_out);                      /* :229-231 */
        if (se_out != se_out) status |= 1;
        se_out = se_out - (float)fo.emax;                              /* :237 */
        se_out = clamp_scale_exp(se_out, outlier_scale_bits, variant); /* :239-242 */
        if (se_out != se_out) status |= 1;                             /* :244 */
        const float sc_out = exp2_float(se_out);
        for (int b = 0; b < block; ++b) {
            outl[b] = outl[b] / sc_out;                                /* :247 */
            if (outl[b] != outl[b]) status |= 1;                       /* :250 */
            outl[b] = quant_elem(outl[b], &fo, round_mode);            /* :252-255 */
            outl[b] = (outl[b] * sc_out) / sc_in;                      /* :258 */
            tmp[b] = inl[b] + outl[b];                                 /* :262 */
        }
        for (int b = 0; b < block; ++b) {                              /* a3 undo */
            int64_t ai = nb * block + b;
            if (ai >= axis_len) continue;
            int64_t idx = (p * axis_len + ai) * post + q;
            if (out) out[idx] = tmp[b];
            if (mask) mask[idx] = (uint8_t)(mk[b] != 0.0f);
        }
        if (n_out && pre == 1 && (nb % block) == 0) { /* a7 utils/quant.py:66 (padding included) */
            int c = 0; for (int b = 0; b < block; ++b) c += (mk[b] != 0.0f);
            n_out[(nb / block) * post + q] = (int8_t)c;
        }
        if (e_in_o) e_in_o[(p * nblk + nb) * post + q] = se_in;
        if (e_out_o) e_out_o[(p * nblk + nb) * post + q] = se_out;
    }
    free(a);
    }
    free(vmean); free(vstd); free(col);
    return status;
}

/* ------------------------------------------------------------------------
 * a6 on HALF-PRECISION tensors: the RTN harness path quantises the checkpoint in its own dtype
 * (llm/llama.py:238  W = subset[name].weight.data -- fp16 for Llama-2 / OPT -- passed straight into
 * quantize_mx_outlier_v1), so every torch op of utils/quant.py:147-266 and elemwise_ops.py:84-174 runs on
 * Half / BFloat16 CPU tensors: ATen computes each op in fp32 and rounds the result back to the tensor dtype
 * (RNE), op by op.  This function restates exactly that sequence; R() is the per-op rounding.  Pinned against
 * the imported reference on fp16 and bf16 tensors (tests/golden/outlier_lowp.npz).  ATen specifics that matter:
 *   mean  = (sum of the fp32-upcast values in ATen's fp32 order / n) rounded once (ReduceOps.cpp mean: half
 *           types go cast_fp32 -> sum -> div -> cast back);
 *   std   = sequential Welford in double, sqrt in double, then double -> float -> T;
 *   floor(log2(x)) on a T tensor = floor(R(log2f(x))): log2 results within half a T-ulp below an integer round
 *           UP to it, so values just below a power of two get the next exponent (bf16: 21 % of all values).  The
 *           rule below (floor_log2_lowp) reproduces torch for EVERY positive fp16 / bf16 value (exhaustive
 *           fixture tests/golden/log2_lowp.npz);
 *   2**e  = R(powf(2, e)); FP32_MIN_NORMAL * mask underflows to 0 in fp16, so an all-zero block gives
 *           log2(0) = -inf -> clamp -> -20.
 * dtype: 1 = fp16, 2 = bf16.  `in` / `out` hold the values as floats (every input must be representable in T).
 * posit formats are not part of the reference: returns -2.
 * ---------------------------------------------------------------------- */
static float r_f16(float f) {
    uint32_t u = f2u(f), sg = u & 0x80000000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return f;                       /* NaN */
    if (a >= 0x477ff000u) return u2f(sg | 0x7f800000u);   /* >= 65520 -> Inf */
    if (a < 0x38800000u) {                                /* < 2^-14: fp16 subnormal grid, multiples of 2^-24 */
        float m = rintf(u2f(a) * 16777216.0f);            /* RNE (default rounding mode) */
        return u2f(sg | f2u(m * (1.0f / 16777216.0f)));
    }
    a += 0xfffu + ((a >> 13) & 1u);
    a &= ~0x1fffu;
    return u2f(sg | a);
}
static float r_bf16(float f) {
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    return u2f(u);
}
static float R_(float f, int dtype) { return dtype == 1 ? r_f16(f) : r_bf16(f); }

/* floor(R(log2f(v))) for a non-negative T value v */
static float floor_log2_lowp(float v, int dtype) {
    if (v != v) return NAN;
    if (v == 0.0f) return -INFINITY;
    if (isinf(v)) return INFINITY;
    const int p = (dtype == 1) ? 11 : 8;                   /* significand bits of T */
    int e;
    const double m = frexp((double)v, &e);                 /* v = m 2^e, m in [0.5, 1) */
    const int k = e - 1, u = e;                            /* exact floor, and the integer just above log2 v */
    if (u == 0) return (float)k;
    const int au = u < 0 ? -u : u;
    int jb = 0; while ((1 << (jb + 1)) <= au) jb++;        /* floor(log2 |u|) */
    if (u > 0 && (au & (au - 1)) == 0) jb -= 1;            /* just below a positive power of two: the binade under it */
    const double h = ldexp(1.0, jb - p);                   /* half a T-ulp of R(log2 v) next to u */
    return (m > exp2(-h)) ? (float)u : (float)k;
}
static float pow2_lowp(float e, int dtype) { return R_(exp2_float(e), dtype); }

static float round_mantissa_lowp(float a, int round_mode, int dt) {
    if (a != a) return a;
    const float s = (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f);
    const float absa = fabsf(a);
    if (round_mode == MSQ_RD_FLOOR) return R_(s * floorf(absa), dt);
    if (round_mode == MSQ_RD_NEAREST) return R_(s * floorf(R_(absa + 0.5f, dt)), dt);
    const float t = R_(absa - 0.5f, dt);
    float r = fmodf(t, 2.0f);
    if (r != 0.0f && r < 0.0f) r += 2.0f;
    r = R_(r, dt);
    const float maska = (r == 0.0f) ? 1.0f : 0.0f;
    return R_(s * R_(floorf(R_(absa + 0.5f, dt)) - maska, dt), dt);
}
static float quantize_elemwise_core_lowp(float a, int bits, int exp_bits, float max_norm, int round_mode, int dt) {
    float out = a, pe = 0.0f;
    const int have_pe = (exp_bits != 0);
    if (have_pe) {                                           /* elemwise_ops.py:138-144 */
        const float t = R_(fabsf(a) + ((a == 0.0f) ? 1.0f : 0.0f), dt);
        pe = floor_log2_lowp(t, dt);
        const float min_exp = (float)(-(1 << (exp_bits - 1)) + 2);
        if (pe == pe && pe < min_exp) pe = min_exp;
    }
    const float sh = ldexpf(1.0f, bits - 2);
    if (have_pe) out = R_(R_(out / pow2_lowp(pe, dt), dt) * sh, dt);   /* _safe_lshift :33-37 */
    else out = R_(out * sh, dt);
    out = round_mantissa_lowp(out, round_mode, dt);
    if (have_pe) out = R_(R_(out / sh, dt) * pow2_lowp(pe, dt), dt);   /* _safe_rshift :40-44 */
    else out = R_(out / sh, dt);
    const float mn = R_(max_norm, dt);                       /* clamp's scalar bounds are cast to the tensor dtype */
    if (out == out) { if (out < -mn) out = -mn; if (out > mn) out = mn; }
    if (a == INFINITY) out = INFINITY;
    if (a == -INFINITY) out = -INFINITY;
    return out;
}
static float shared_exp_lowp(float mx, int dt) {
    if (mx != mx) return NAN;
    const float t = R_(ldexpf(1.0f, -126) * ((mx == 0.0f) ? 1.0f : 0.0f), dt);   /* FP32_MIN_NORMAL * (x == 0).type(dtype) */
    return floor_log2_lowp(R_(mx + t, dt), dt);
}

int msq_oracle_outlier_fakequant_lowp(const float* in, float* out, uint8_t* mask, float* e_in_o, float* e_out_o,
                                      int dtype, int64_t pre, int64_t axis_len, int64_t post, int block,
                                      const char* inlier_fmt, const char* outlier_fmt, int inlier_scale_bits,
                                      int outlier_scale_bits, double std_dev, int round_mode,
                                      int flush_fp32_subnorms) {
    fmt_t fi, fo; float mn;
    if (dtype != 1 && dtype != 2) return -2;
    if (msq_oracle_format_params(inlier_fmt, &fi.ebits, &fi.mbits, &fi.emax, &fi.max_norm, &mn, &fi.kind)) return -1;
    if (msq_oracle_format_params(outlier_fmt, &fo.ebits, &fo.mbits, &fo.emax, &fo.max_norm, &mn, &fo.kind)) return -1;
    if (fi.kind == MSQ_KIND_POSIT || fo.kind == MSQ_KIND_POSIT) return -2;
    if (block <= 0) block = (int)axis_len;
    const int64_t nblk = (axis_len + block - 1) / block;
    const int dt = dtype;
    const float k = (float)std_dev;
    int status = 0;
#pragma omp parallel reduction(| : status)
    {
    float* a = (float*)malloc(sizeof(float) * block * 5);
    float *absa = a + block, *mk = a + 2 * block, *inl = a + 3 * block, *outl = a + 4 * block;
#pragma omp for collapse(2) schedule(static)
    for (int64_t p = 0; p < pre; ++p) for (int64_t nb = 0; nb < nblk; ++nb) for (int64_t q = 0; q < post; ++q) {
        for (int b = 0; b < block; ++b) {
            int64_t ai = nb * block + b;
            a[b] = (ai < axis_len) ? in[(p * axis_len + ai) * post + q] : 0.0f;
            absa[b] = fabsf(a[b]);
        }
        /* utils/quant.py:477-478, :489-492 */
        const float mean = R_(sum_ordered(absa, block, sum_order_for(post, q)) / (float)block, dt);
        const float sd = R_(std_welford(absa, block, 1, 0), dt);
        const float ks = R_(k * sd, dt);
        const float lo = R_(mean - ks, dt), hi = R_(mean + ks, dt);
        float mx_in = 0.0f;
        for (int b = 0; b < block; ++b) {
            mk[b] = ((a[b] < lo) || (a[b] > hi)) ? 1.0f : 0.0f;
            inl[b] = R_(a[b] * R_(1.0f - mk[b], dt), dt);             /* :192 */
            outl[b] = R_(a[b] * mk[b], dt);                          /* :193 */
            float t = fabsf(inl[b]); if (t > mx_in || t != t) mx_in = t;
        }
        float se_in = shared_exp_lowp(mx_in, dt);                    /* :196-198 */
        if (flush_fp32_subnorms && !(se_in > -127.0f))
            for (int b = 0; b < block; ++b) inl[b] = R_(inl[b] * 0.0f, dt);
        se_in = R_(se_in - (float)fi.emax, dt);                      /* :207 */
        se_in = clamp_scale_exp(se_in, inlier_scale_bits, MSQ_VARIANT_QUANT);
        const float sc_in = pow2_lowp(se_in, dt);
        float mx_out = 0.0f;
        for (int b = 0; b < block; ++b) {
            inl[b] = R_(inl[b] / sc_in, dt);                         /* :214 */
            outl[b] = R_(outl[b] * sc_in, dt);                       /* :216 */
            inl[b] = quantize_elemwise_core_lowp(inl[b], fi.mbits, fi.ebits, fi.max_norm, round_mode, dt);
            inl[b] = R_(inl[b] * sc_in, dt);                         /* :224 */
            if (inl[b] != inl[b] || outl[b] != outl[b]) status |= 1;
            float t = fabsf(outl[b]); if (t > mx_out || t != t) mx_out = t;
        }
        float se_out = shared_exp_lowp(mx_out, dt);                  /* :229-231 */
        if (se_out != se_out) status |= 1;
        se_out = R_(se_out - (float)fo.emax, dt);                    /* :237 */
        se_out = clamp_scale_exp(se_out, outlier_scale_bits, MSQ_VARIANT_QUANT);
        if (se_out != se_out) status |= 1;
        const float sc_out = pow2_lowp(se_out, dt);
        for (int b = 0; b < block; ++b) {
            float o = R_(outl[b] / sc_out, dt);                      /* :247 */
            if (o != o) status |= 1;
            o = quantize_elemwise_core_lowp(o, fo.mbits, fo.ebits, fo.max_norm, round_mode, dt);
            o = R_(R_(o * sc_out, dt) / sc_in, dt);                  /* :258 */
            const float r = R_(inl[b] + o, dt);                      /* :262 */
            int64_t ai = nb * block + b;
            if (ai >= axis_len) continue;
            int64_t idx = (p * axis_len + ai) * post + q;
            if (out) out[idx] = r;
            if (mask) mask[idx] = (uint8_t)(mk[b] != 0.0f);
        }
        if (e_in_o) e_in_o[(p * nblk + nb) * post + q] = se_in;
        if (e_out_o) e_out_o[(p * nblk + nb) * post + q] = se_out;
    }
    free(a);
    }
    return status;
}

/* test hooks of the pieces above */
/* floor(torch.log2(v)) on a float32 tensor (floor_log2_torch): exported for tests/test_oracle_golden.py (log2_f32.npz) */
void msq_oracle_floor_log2_f32(const float* v, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = floor_log2_torch(v[i]);
}
void msq_oracle_floor_log2_lowp(const float* v, float* out, int64_t n, int dtype) {
    for (int64_t i = 0; i < n; ++i) out[i] = floor_log2_lowp(v[i], dtype);
}
void msq_oracle_round_lowp(const float* v, float* out, int64_t n, int dtype) {
    for (int64_t i = 0; i < n; ++i) out[i] = R_(v[i], dtype);
}

/* ------------------------------------------------------------------------
 * a9  mx_ops.py:332-457  _quantize_mx (custom_cuda=False branch), one axis.
 * plus_eps_defect=1 reproduces `A / (2**e + 1e-6)` (mx_ops.py:444, D2).
 * Returns 1 if any NaN was produced by a scale overflow.
 * ---------------------------------------------------------------------- */
int msq_oracle_quantize_mx(const float* in, float* out, int64_t pre, int64_t axis_len, int64_t post,
                           int block, const char* elem_fmt, int scale_bits, int round_mode,
                           int flush_fp32_subnorms, int plus_eps_defect) {
    fmt_t f; float mn;
    if (msq_oracle_format_params(elem_fmt, &f.ebits, &f.mbits, &f.emax, &f.max_norm, &mn, &f.kind)) return -1;
    if (block <= 0) block = (int)axis_len;
    const int64_t nblk = (axis_len + block - 1) / block;
    int status = 0;
    for (int64_t p = 0; p < pre; ++p) for (int64_t nb = 0; nb < nblk; ++nb) for (int64_t q = 0; q < post; ++q) {
        float mx = 0.0f;
        for (int b = 0; b < block; ++b) {
            int64_t ai = nb * block + b;
            float v = (ai < axis_len) ? in[(p * axis_len + ai) * post + q] : 0.0f;
            float t = fabsf(v); if (t > mx || t != t) mx = t;
        }
        float se = shared_exp_of_max(mx);                              /* :428-430 */
        int flush = flush_fp32_subnorms && !(se > -127.0f);            /* :433-434 */
        se = se - (float)f.emax;                                       /* :438 */
        se = clamp_scale_exp(se, scale_bits, MSQ_VARIANT_MXOPS);       /* :440-442 */
        if (se != se) status |= 1;
        float sc = exp2_float(se);
        float den = plus_eps_defect ? (sc + 1e-6f) : sc;               /* :444 */
        for (int b = 0; b < block; ++b) {
            int64_t ai = nb * block + b;
            if (ai >= axis_len) continue;
            int64_t idx = (p * axis_len + ai) * post + q;
            float v = in[idx];
            if (flush) v = v * 0.0f;
            v = v / den;
            v = quant_elem(v, &f, round_mode);                         /* :446-449 */
            out[idx] = v * sc;                                         /* :451 */
        }
    }
    return status;
}

/* ------------------------------------------------------------------------
 * a9 on HALF-PRECISION tensors: number_system/mx/mx_ops.py:332-457 `_quantize_mx`, Python path (custom_cuda False: what the
 * CPU executes), run op by op on a Half / BFloat16 tensor -- the KV-cache variant of BASELINE config 4 and MXLinear inside a
 * half-precision model hand it such tensors.  R_() is ATen's per-op rounding to the tensor dtype (see the lowp block above).
 * What differs from "upcast, fp32, round once": floor(log2) of the block maximum and of every element's private exponent
 * rounds values just under a power of two UP (floor_log2_lowp); the `+ 1e-6` of :444 is absorbed by the rounding of
 * `2**e + 1e-6` for e > -9 in fp16 (e > -12 in bf16) -- above that the divisor IS the scale --, and 2^e underflows / overflows the
 * fp16 range.  Pinned by tests/golden/quantize_mx_lowp.npz (made by the imported reference).  dtype 1 = fp16, 2 = bf16.
 * ---------------------------------------------------------------------- */
int msq_oracle_quantize_mx_lowp(const float* in, float* out, int dtype, int64_t pre, int64_t axis_len, int64_t post,
                                int block, const char* elem_fmt, int scale_bits, int round_mode, int flush_fp32_subnorms) {
    fmt_t f; float mn;
    if (msq_oracle_format_params(elem_fmt, &f.ebits, &f.mbits, &f.emax, &f.max_norm, &mn, &f.kind)) return -1;
    if (f.kind != 0 || (dtype != 1 && dtype != 2)) return -2;
    const int dt = dtype;
    if (block <= 0) block = (int)axis_len;
    const int64_t nblk = (axis_len + block - 1) / block;
    int status = 0;
    for (int64_t p = 0; p < pre; ++p) for (int64_t nb = 0; nb < nblk; ++nb) for (int64_t q = 0; q < post; ++q) {
        float mx = 0.0f;
        for (int b = 0; b < block; ++b) {
            int64_t ai = nb * block + b;
            float v = (ai < axis_len) ? in[(p * axis_len + ai) * post + q] : 0.0f;
            float t = fabsf(v); if (t > mx || t != t) mx = t;
        }
        float se = shared_exp_lowp(mx, dt);                            /* :428-430 (_shared_exponents :525-529) */
        const int flush = flush_fp32_subnorms && !(se > -127.0f);      /* :433-434 */
        se = R_(se - (float)f.emax, dt);                               /* :438 */
        const float semax = (float)((1 << (scale_bits - 1)) - 1);      /* :440-442 */
        if (se > semax) se = NAN;
        if (se < -semax) se = -semax;
        if (se != se) status |= 1;
        const float sc = pow2_lowp(se, dt);                            /* 2**shared_exp, a T tensor */
        const float den = R_(sc + 1e-6f, dt);                          /* :444 */
        for (int b = 0; b < block; ++b) {
            int64_t ai = nb * block + b;
            if (ai >= axis_len) continue;
            int64_t idx = (p * axis_len + ai) * post + q;
            float v = in[idx];
            if (flush) v = R_(v * 0.0f, dt);
            v = R_(v / den, dt);
            v = quantize_elemwise_core_lowp(v, f.mbits, f.ebits, f.max_norm, round_mode, dt);   /* :446-449 */
            out[idx] = R_(v * sc, dt);                                 /* :451 */
        }
    }
    return status;
}

/* ------------------------------------------------------------------------
 * a12  cpp/shared_exp.cuh:14-53 + cpp/mx.cuh:107-170  native MX quant
 * (quantize_mx_by_tile / quantize_mx_func_cpp semantics: biased-exponent max,
 * NaN scale on overflow, ragged last tile NOT padded, integer codec).
 * ---------------------------------------------------------------------- */
static float mx_get_shared_scale(int shared_exp, int scale_bits, float elem_max_norm) {
    const int elem_emax = (int)((f2u(elem_max_norm) >> 23) & 0xFF) - 127;
    if (shared_exp != 255) shared_exp -= elem_emax;
    int emax = scale_bits != 0 ? (1 << (scale_bits - 1)) - 1 : 255;
    int ub = shared_exp - 127;
    if (ub > emax) shared_exp = 255;
    if (ub < -emax) shared_exp = 127 - emax;
    uint32_t mant = (shared_exp == 0 || shared_exp == 255) ? (1u << 22) : 0u;
    return u2f(((uint32_t)shared_exp << 23) | mant);
}

void msq_oracle_quantize_mx_native(const float* in, float* out, int64_t pre, int64_t axis_len,
                                   int64_t post, int tile, int scale_bits, int elem_ebits,
                                   int elem_mbits, float elem_max_norm, int flush_fp32_subnorms,
                                   int round_mode) {
    if (tile <= 0) tile = (int)axis_len;
    const int64_t ntiles = (axis_len + tile - 1) / tile;
    for (int64_t p = 0; p < pre; ++p) for (int64_t t = 0; t < ntiles; ++t) for (int64_t q = 0; q < post; ++q) {
        int64_t a0 = t * tile, a1 = a0 + tile; if (a1 > axis_len) a1 = axis_len;
        int se = 0;
        for (int64_t ai = a0; ai < a1; ++ai) {
            int e = (int)((f2u(in[(p * axis_len + ai) * post + q]) >> 23) & 0xFF);
            if (e > se) se = e;
        }
        int flush = (se == 0 && flush_fp32_subnorms);
        float scale = mx_get_shared_scale(se, scale_bits, elem_max_norm);
        for (int64_t ai = a0; ai < a1; ++ai) {
            int64_t idx = (p * axis_len + ai) * post + q;
            float si = flush ? 0.0f : in[idx] / scale;
            float so = quantize_elemwise_bits_1(si, elem_mbits, elem_ebits, elem_max_norm, round_mode, 1, 1);
            out[idx] = so * scale;
        }
    }
}

/* a12  cpp/reduce.cuh:154-210 semantics (values only): last-dim sum / max in fp32.
 * The reference kernel's fp32 summation order is a GPU tree; the oracle sums
 * in double and the test uses a tolerance, exactly as tests/test_reduce.py:17-46
 * compares against torch.sum. */
void msq_oracle_reduce_inner(const float* in, float* out, int64_t outer, int64_t inner, int is_max) {
    for (int64_t i = 0; i < outer; ++i) {
        if (is_max) { float m = -INFINITY; for (int64_t j = 0; j < inner; ++j) if (in[i * inner + j] > m) m = in[i * inner + j]; out[i] = m; }
        else { double s = 0; for (int64_t j = 0; j < inner; ++j) s += in[i * inner + j]; out[i] = (float)s; }
    }
}

/* plain fp32 reference of the dense Linear the reference executes after the
 * fake-quant (number_system/mx/linear.py:91  F.linear): y[M,N] = x[M,K] . w[N,K]^T,
 * accumulated in double (checker-grade, used with a tolerance). */
void msq_oracle_linear(const float* x, const float* w, const float* bias, float* y, int64_t M,
                       int64_t N, int64_t K) {
    /* 4 x 4 register blocks so that the x and w rows are reused out of cache; every output element is still the
     * same left-to-right double sum over k, so the result does not depend on the blocking or the thread count */
    const int64_t MB = (M + 3) / 4, NB = (N + 3) / 4;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t nb = 0; nb < NB; ++nb) for (int64_t mb = 0; mb < MB; ++mb) {
        const int64_t m0 = mb * 4, n0 = nb * 4;
        const int mr = (int)((M - m0) < 4 ? (M - m0) : 4), nr = (int)((N - n0) < 4 ? (N - n0) : 4);
        double acc[4][4] = {{0}};
        if (mr == 4 && nr == 4) {
            const float *x0 = x + m0 * K, *x1 = x0 + K, *x2 = x1 + K, *x3 = x2 + K;
            const float *w0 = w + n0 * K, *w1 = w0 + K, *w2 = w1 + K, *w3 = w2 + K;
            for (int64_t kk = 0; kk < K; ++kk) {
                const double a0 = x0[kk], a1 = x1[kk], a2 = x2[kk], a3 = x3[kk];
                const double b0 = w0[kk], b1 = w1[kk], b2 = w2[kk], b3 = w3[kk];
                acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[0][2] += a0 * b2; acc[0][3] += a0 * b3;
                acc[1][0] += a1 * b0; acc[1][1] += a1 * b1; acc[1][2] += a1 * b2; acc[1][3] += a1 * b3;
                acc[2][0] += a2 * b0; acc[2][1] += a2 * b1; acc[2][2] += a2 * b2; acc[2][3] += a2 * b3;
                acc[3][0] += a3 * b0; acc[3][1] += a3 * b1; acc[3][2] += a3 * b2; acc[3][3] += a3 * b3;
            }
        } else {
            for (int i = 0; i < mr; ++i) for (int j = 0; j < nr; ++j) {
                double s = 0.0;
                for (int64_t kk = 0; kk < K; ++kk) s += (double)x[(m0 + i) * K + kk] * (double)w[(n0 + j) * K + kk];
                acc[i][j] = s;
            }
        }
        for (int i = 0; i < mr; ++i) for (int j = 0; j < nr; ++j) {
            double s = acc[i][j];
            if (bias) s += bias[n0 + j];
            y[(m0 + i) * N + n0 + j] = (float)s;
        }
    }
}

/* ------------------------------------------------------------------------
 * f3  KV-cache group fake-quant of the GEAR tree (kv_quant/GEARLM/Simulated/compress_function.py):
 *   along_tokens = 0  :8-38   fake_groupwise_token_asymmetric_quantization   (groups along head.dim of one token)
 *   along_tokens = 1  :41-70  fake_groupwise_channel_asymmetric_quantization_new (groups along the tokens of a channel)
 * in / out: [B, H, S, D] cache tensors held as float; dtype 0 / 1 / 2 = the tensor was f32 / f16 / bf16 (the result
 * is cast back with .type(dtype), :34 / :65).  All arithmetic in fp32 in the reference's op order; torch.max / min
 * propagate NaN; a constant group gives scale 0 and 0 / 0 = NaN, as in the reference.
 * Returns 0, or -1 when group_size does not divide the grouped extent (the reference raises).
 * ---------------------------------------------------------------------- */
static float kv_nmax(float a, float b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
static float kv_nmin(float a, float b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
int msq_oracle_kv_group_quant(const float* in, float* out, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                              int quantize_bit, int64_t gs, int along_tokens) {
    const float levels = (float)((1 << quantize_bit) - 1);
    const int64_t HD = H * D;
    if (gs <= 0) return -1;
    if (along_tokens ? (S % gs) : (HD % gs)) return -1;
    const int64_t ng = along_tokens ? S / gs : HD / gs;
    for (int64_t b = 0; b < B; ++b) for (int64_t g = 0; g < ng; ++g)
        for (int64_t o = 0; o < (along_tokens ? HD : S); ++o) {          /* the non-grouped coordinate */
            float mx = -INFINITY, mn = INFINITY;
            for (int64_t i = 0; i < gs; ++i) {
                const int64_t hd = along_tokens ? o : g * gs + i, s = along_tokens ? g * gs + i : o;
                const float x = in[((b * H + hd / D) * S + s) * D + hd % D];
                mx = kv_nmax(mx, x); mn = kv_nmin(mn, x);
            }
            const float scale = (mx - mn) / levels;
            for (int64_t i = 0; i < gs; ++i) {
                const int64_t hd = along_tokens ? o : g * gs + i, s = along_tokens ? g * gs + i : o;
                const int64_t a = ((b * H + hd / D) * S + s) * D + hd % D;
                float v = (in[a] - mn) / scale;
                v = (v < 0.0f) ? 0.0f : v;                       /* F.relu */
                v = rintf(v);                                    /* round_(): half to even */
                v = v * scale + mn;
                out[a] = dtype ? R_(v, dtype) : v;
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------
 * f4  bfloat-rounded vector ops around the MX Linear (number_system/mx/vector_ops.py: every op is the torch op followed
 * by quantize_elemwise_op, elemwise_ops.py:237-266 -> _quantize_elemwise_core with saturate_normals=False):
 *   LayerNorm   layernorm.py:18-42 -> norm_utils.py:27-113 _norm_forward (axes = last)
 *   gelu        activations.py:460-512 (sigmoid form, bf16 coefficients; first_order variant)
 *   simd_add    simd_ops.py:85-106
 * (bits, exp_bits, max_norm, round_mode, allow_denorm) describe the rounding Q(): bfloat16 = (9, 8, bf16 max).
 * The row sums use ATen's order for a contiguous inner dimension (sum_inner_v8).
 * ---------------------------------------------------------------------- */
typedef struct { int bits, ebits, rmode, dn; float max_norm; } vq_t;
static float VQ(float a, const vq_t* q) {
    if (q->bits <= 0) return a;
    return quantize_elemwise_core_1(a, q->bits, q->ebits, q->max_norm, q->rmode, 0, q->dn);
}
void msq_oracle_vec_layernorm(const float* x, const float* w, const float* b, float* out, int64_t rows, int64_t H,
                              double eps, int bits, int exp_bits, float max_norm, int round_mode, int allow_denorm) {
    const vq_t q = {bits, exp_bits, round_mode, allow_denorm, max_norm};
    float* t = (float*)malloc(sizeof(float) * H * 2);
    float* p = t + H;
    for (int64_t r = 0; r < rows; ++r) {
        for (int64_t i = 0; i < H; ++i) t[i] = VQ(x[r * H + i], &q);               /* layernorm.py:24 */
        float mean = VQ(sum_inner_v8(t, H), &q);                                    /* vec_reduce_sum */
        mean = VQ(mean / (float)H, &q);                                             /* vec_div(s, denom) */
        for (int64_t i = 0; i < H; ++i) { t[i] = VQ(t[i] - mean, &q); p[i] = VQ(t[i] * t[i], &q); }
        float var = VQ(sum_inner_v8(p, H), &q);
        var = VQ(var / (float)H, &q);
        const float vare = VQ(var + (float)eps, &q);                                /* norm_utils.py:92 */
        const float sd = VQ(sqrtf(vare), &q);
        const float inv = VQ(1.0f / sd, &q);
        for (int64_t i = 0; i < H; ++i) {
            const float xn = VQ(t[i] * inv, &q);
            const float xs = VQ(VQ(w[i], &q) * xn, &q);
            out[r * H + i] = VQ(xs + VQ(b[i], &q), &q);
        }
    }
    free(t);
}
void msq_oracle_vec_gelu(const float* x, float* out, int64_t n, int first_order, int bits, int exp_bits, float max_norm,
                         int round_mode, int allow_denorm) {
    const vq_t q = {bits, exp_bits, round_mode, allow_denorm, max_norm};
    for (int64_t i = 0; i < n; ++i) {
        const float qi = VQ(x[i], &q);
        float s;
        if (first_order) s = VQ(1.703125f * qi, &q);
        else {
            s = VQ(qi * qi, &q); s = VQ(s * qi, &q); s = VQ(0.044677734f * s, &q);
            s = VQ(s + qi, &q); s = VQ(1.59375f * s, &q);
        }
        float phi = VQ(expf(-s), &q);                                               /* vec_exp, vec_use_exp2 False */
        phi = VQ(phi + 1.0f, &q);
        phi = VQ(1.0f / phi, &q);
        out[i] = VQ(qi * phi, &q);
    }
}
void msq_oracle_vec_add(const float* a, const float* b, float* out, int64_t n, int bits, int exp_bits, float max_norm,
                        int round_mode, int allow_denorm) {
    const vq_t q = {bits, exp_bits, round_mode, allow_denorm, max_norm};
    for (int64_t i = 0; i < n; ++i) out[i] = VQ(VQ(a[i], &q) + VQ(b[i], &q), &q);
}

/* RMSNorm forward, number_system/mx/layernorm.py:98-128 (RMSNormFunction.forward): x = Q(x); x2 = Q(x x); ms = Q(Q(sum x2) / H) (vec_reduce_mean,
 * vector_ops.py:121-130); rms = Q(sqrt(Q(ms + eps))); inv = Q(1 / rms); x_norm = Q(x inv); out = Q(Q(Q(w) x_norm) + Q(b)). */
void msq_oracle_vec_rmsnorm(const float* x, const float* w, const float* b, float* out, int64_t rows, int64_t H,
                            double eps, int bits, int exp_bits, float max_norm, int round_mode, int allow_denorm) {
    const vq_t q = {bits, exp_bits, round_mode, allow_denorm, max_norm};
    float* t = (float*)malloc(sizeof(float) * H * 2);
    float* p = t + H;
    for (int64_t r = 0; r < rows; ++r) {
        for (int64_t i = 0; i < H; ++i) { t[i] = VQ(x[r * H + i], &q); p[i] = VQ(t[i] * t[i], &q); }   /* :104, :107 */
        float ms = VQ(sum_inner_v8(p, H), &q);                                      /* vec_reduce_sum */
        ms = VQ(ms / (float)H, &q);                                                 /* vec_div(s, denom) */
        const float mse = VQ(ms + (float)eps, &q);                                  /* :114 */
        const float rms = VQ(sqrtf(mse), &q);                                       /* :116 */
        const float inv = VQ(1.0f / rms, &q);                                       /* :119 */
        for (int64_t i = 0; i < H; ++i) {
            const float xn = VQ(t[i] * inv, &q);                                    /* :120 */
            const float xs = VQ(VQ(w[i], &q) * xn, &q);                             /* :122, :126 */
            out[r * H + i] = VQ(xs + VQ(b[i], &q), &q);                             /* :124, :128 */
        }
    }
    free(t);
}
/* SiLU forward, activations.py:420-434: q = Q(x); e = Q(exp(-q)); p = Q(e + 1); s = Q(1 / p); out = Q(q s)  (vec_use_exp2 False) */
void msq_oracle_vec_silu(const float* x, float* out, int64_t n, int bits, int exp_bits, float max_norm, int round_mode, int allow_denorm) {
    const vq_t q = {bits, exp_bits, round_mode, allow_denorm, max_norm};
    for (int64_t i = 0; i < n; ++i) {
        const float qi = VQ(x[i], &q);
        float phi = VQ(expf(-qi), &q);
        phi = VQ(phi + 1.0f, &q);
        phi = VQ(1.0f / phi, &q);
        out[i] = VQ(qi * phi, &q);
    }
}
/* simd_mul, simd_ops.py:154-187 (tensor x tensor): Q(Q(a) Q(b)) */
void msq_oracle_vec_mul(const float* a, const float* b, float* out, int64_t n, int bits, int exp_bits, float max_norm,
                        int round_mode, int allow_denorm) {
    const vq_t q = {bits, exp_bits, round_mode, allow_denorm, max_norm};
    for (int64_t i = 0; i < n; ++i) out[i] = VQ(VQ(a[i], &q) * VQ(b[i], &q), &q);
}

/* thread count of the OpenMP regions above (0 = all cores); returns the count in effect */
#ifdef _OPENMP
#include <omp.h>
int msq_oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int msq_oracle_set_threads(int n) { (void)n; return 1; }
#endif

/* test hook: the per-block mean / std exactly as the a4 mask path computes them
 * for a tensor with `post` contiguous non-reduced columns. */
void msq_oracle_test_mean_std(const float* blocks, int64_t nblocks, int block, int64_t post, float* mean,
                              float* std) {
    /* blocks laid out [n, post, block]: block i belongs to column i % post */
    for (int64_t i = 0; i < nblocks; ++i) {
        const float* b = blocks + i * block;
        float s = sum_ordered(b, block, sum_order_for(post, i % post));
        mean[i] = s / (float)block;
        std[i] = std_welford(b, block, 1, 0);
    }
}
