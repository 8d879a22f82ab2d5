// msq_device.h -- scalar building blocks shared by the gfx950 kernels.
//
// Everything here is written so that the fp32 result is bit-identical to the
// reference's CPU fake-quant arithmetic (utils/quant.py, number_system/mx/
// elemwise_ops.py) -- the translation unit is compiled with -ffp-contract=off,
// f32 denormals are kept (hipcc default), and powers of two are applied with
// exact multiplies.  Reference citations are file:line under the reference root.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MSQ_HD __host__ __device__ __forceinline__
#define MSQ_D __device__ __forceinline__

namespace msq {

struct Fmt {      // formats.py:65-129 (+ posit extension)
    int kind;     // 0 = eXmY / intN, 1 = posit<mbits, ebits>
    int ebits;    // exponent bits (posit: es)
    int mbits;    // mantissa bits incl. sign and implicit one (posit: n)
    int emax;
    float max_norm;
};

MSQ_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
MSQ_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// 2^e for integer e in [-149, 127] as an exact fp32 value
MSQ_HD float pow2i(int e) {
    if (e >= -126) return u2f((uint32_t)(e + 127) << 23);
    return u2f(1u << (e + 149));
}

// exact floor(log2(x)) for finite x > 0 (subnormals included)
MSQ_HD int ilog2f(float x) {
    uint32_t u = f2u(x) & 0x7FFFFFFFu;
    int e = (int)(u >> 23);
    if (e) return e - 127;
    return 31 - __builtin_clz(u) - 149;   // subnormal: position of the leading one
}

// floor(torch.log2(x)) as the reference's PYTHON path computes it in float32 (utils/quant.py:525-529, elemwise_ops.py:139-140): torch.log2
// returns the float32 nearest to the true logarithm, so the K largest floats below a power of two 2^u give exactly u and the floor is one
// binade high.  K depends on the spacing of float32 next to u: 0 for |u| <= 1, 1 (2..3), 2 (4..7), 5 (8..15), 11 (16..31), 22 (32..63), 44
// (64..127), 88 (128..), and the count of the next smaller class when u is a POSITIVE power of two (the result approaches it from the finer
// binade).  K = ceil(2^jb ln 2) - 1.  The reference's native kernels (and ilog2f above) use the exact exponent instead.  Pinned by
// tests/golden/log2_f32.npz through the oracle (oracle/msq_oracle.c floor_log2_torch, an independent double-precision restatement).
MSQ_HD int ilog2f_torch(float x) {
    const uint32_t u = f2u(x) & 0x7FFFFFFFu;
    const int E = (int)(u >> 23);
    int k, dist;                                     // exact floor; distance (in units of 2^-24 relative) to the power of two above, minus one
    if (E) { k = E - 127; dist = (int)(0x7FFFFFu - (u & 0x7FFFFFu)); }
    else {                                           // subnormal: leading one at bit p, the value is (2^(p+1) - d) 2^-149: relative distance d 2^-(p+1)
        const int p = 31 - __builtin_clz(u);
        k = p - 149;
        const uint32_t d = (2u << p) - u;            // >= 1
        // as a count of 2^-24 steps: d 2^(23 - p) (rounded up: a coarser grid can only be farther away)
        dist = (p >= 23 - 7) ? (int)(d << (23 - p)) - 1 : 1 << 20;
    }
    const int up = k + 1;
    const int au = up < 0 ? -up : up;
    if (au <= 1) return k;
    int jb = 31 - __builtin_clz((uint32_t)au);
    if (up > 0 && (au & (au - 1)) == 0) jb -= 1;
    const int K = (jb <= 0) ? 0 : ((jb == 1) ? 1 : ((jb == 2) ? 2 : ((jb == 3) ? 5 : ((jb == 4) ? 11 : ((jb == 5) ? 22 : ((jb == 6) ? 44 : 88))))));
    return dist < K ? up : k;
}

// Biased exponent of a float as the reference's PYTHON path derives a block's shared exponent from it: floor(torch.log2(|v|)) + 127
// (mx_ops.py:66-77 _shared_exponents on the block maximum; the maximum of this over a block = this of the maximum: it is monotone).
// Only the 88 largest significands of a binade can round up to the next power of two (ilog2f_torch); zero / subnormals and Inf / NaN keep
// their exponent field (the scale clamps swallow the former, the latter mark the block).  The reference's NATIVE kernels read the
// exponent field alone (cpp/shared_exp.cuh:14-53): surfaces that replace those do not call this.
MSQ_HD int biased_exp_py(uint32_t bits) {
    const uint32_t u = bits & 0x7FFFFFFFu;
    const int e = (int)(u >> 23);
    if (e == 0 || e == 255 || (u & 0x7FFFFFu) < 0x7FFFA7u) return e;
    const int t = ilog2f_torch(u2f(u)) + 127;
    return t > 254 ? 254 : t;                        // (within 88 ulps of 2^128: no tensor holds such values; keep the block finite)
}

// ---------------------------------------------------------------------------
// elemwise_ops.py:47-78 _round_mantissa + :84-174 _quantize_elemwise_core with
// allow_denorm=True, saturate_normals=True (the only way MicroScopiQ calls it:
// utils/quant.py:218-221, 252-255), op for op in fp32.
// ---------------------------------------------------------------------------
MSQ_HD float round_mantissa(float a, int rmode) {
    if (a != a) return a;
    const float s = (a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f);
    const float m = __builtin_fabsf(a);
    if (rmode == 1) return s * __builtin_floorf(m);
    if (rmode == 0) return s * __builtin_floorf(m + 0.5f);
    float r = __builtin_fmodf(m - 0.5f, 2.0f);
    if (r != 0.f && r < 0.f) r += 2.0f;
    const float tie = (r == 0.f) ? 1.f : 0.f;
    return s * (__builtin_floorf(m + 0.5f) - tie);
}

// ---------------------------------------------------------------------------
// The one magnitude on which the reference's "nearest" is not round-half-away.  elemwise_ops.py:64-65 rounds with
// floor(|x| + 0.5) IN float32 on the value scaled to the integer grid.  Below the grid's first step that sum is inexact for
// exactly one input: |x| = pred(0.5) -- 0.5 + (0.5 - 2^-25) = 1 - 2^-25 is a tie of the float32 grid under 1 and rounds
// (to even) to 1.0 -- so the reference returns the grid's SMALLEST step where true rounding gives 0 (`even`, :66-72, takes
// the same sum).  In unscaled terms: |x| = pred(h), h = half the smallest subnormal of the element format x the block scale.
// The reference's native kernel (cpp/quantize.cuh:88-149, integer rounding) does not have it.  The arithmetic codecs here
// (quant_core_sat, quant_core_fast) reproduce it by construction; the bit / hardware-convert codecs call these helpers.
// Found in round 6 by planted maxima one ulp under a power of two meeting a block scale set by a larger neighbour
// (tests/test_gpu_a9_quantize_mx.py); ~1e-8 of Gaussian float32 weights.
// ---------------------------------------------------------------------------
// bit pattern of pred(2^t) (t = exponent of HALF the smallest step, after scaling); 0xFFFFFFFF when 2^t is not a float
MSQ_HD uint32_t half_away_quirk_bits(int t) {
    if (t >= -126 && t <= 127) return ((uint32_t)(t + 127) << 23) - 1u;
    if (t >= -148) return (1u << (t + 149)) - 1u;
    return 0xFFFFFFFFu;
}
// input for a round-to-nearest-EVEN convert that yields the reference's result: an exact tie becomes "just above the tie"
// (| 1: ties and grid points of the <= 8-bit grids have >= 15 trailing zero mantissa bits, nothing else moves across one),
// the quirk magnitude becomes "just above h" (+ 2 ulps)
MSQ_HD float sticky_half_away(float a, uint32_t quirk_bits) {
    const uint32_t u = f2u(a);
    return u2f(((u & 0x7FFFFFFFu) == quirk_bits) ? u + 2u : (u | 1u));
}
// conservative block-level test for the fast paths that fall back to the arithmetic codec: some element has an all-ones
// mantissa (every pred(2^t) has); 32 x 2^-23 of the blocks
MSQ_HD bool mantissa_all_ones(uint32_t bits) { return (bits & 0x7FFFFFu) == 0x7FFFFFu; }

MSQ_HD float quant_core_sat(float a, int bits, int ebits, float max_norm, int rmode) {
    float out;
    const float up = pow2i(bits - 2);
    const float dn = pow2i(2 - bits);
    if (ebits != 0) {
        const float t = __builtin_fabsf(a) + ((a == 0.f) ? 1.f : 0.f);
        if (t != t) return t;                       // NaN in -> NaN out
        if (__builtin_isinf(t)) return a;           // elemwise_ops.py:165-166
        // the Python path's private exponent is floor(torch.log2(t)): for the floats just below a power of two it is one too high, which
        // changes the result only under truncation (nearest / even: both exponents round such a value to the power of two itself)
        int pe = (rmode == 1) ? ilog2f_torch(t) : ilog2f(t);
        const int min_exp = 2 - (1 << (ebits - 1));
        pe = pe < min_exp ? min_exp : pe;
        const float inv = pow2i(-pe), fwd = pow2i(pe);
        out = a * inv * up;                          // _safe_lshift (x / 2^pe * 2^(bits-2))
        out = round_mantissa(out, rmode);
        out = out * dn * fwd;                        // _safe_rshift
    } else {
        if (a != a) return a;
        out = a * up;
        out = round_mantissa(out, rmode);
        out = out * dn;
        if (__builtin_isinf(a)) return a;
    }
    out = out < -max_norm ? -max_norm : out;         // torch.clamp
    out = out > max_norm ? max_norm : out;
    return out;
}

// ---------------------------------------------------------------------------
// Integer codec with the semantics of the reference's native path
// (cpp/quantize.cuh:88-149, what custom_cuda=True executes).  Own formulation:
// the 24-bit significand is divided by 2^shift with the requested rounding.
// ---------------------------------------------------------------------------
MSQ_HD float quant_bits(float x, int bits, int ebits, float max_norm, int rmode, bool saturate,
                        bool allow_denorm) {
    const uint32_t u = f2u(x);
    const uint32_t sign = u & 0x80000000u;
    const int E = (int)((u >> 23) & 0xFF);
    const uint32_t M = u & 0x7FFFFFu;
    const int mbits = bits - 1;                      // magnitude significand bits incl. implicit one
    const bool is_int = (ebits == 0);
    if (E == 0xFF) {                                 // Inf / NaN
        if (M) return x;
        return (is_int || saturate) ? u2f(sign | f2u(max_norm)) : x;
    }
    const int e_true = (E ? E : 1) - 127;            // exponent of the significand's leading position
    const int e_min = is_int ? 0 : 2 - (1 << (ebits - 1));
    if (!is_int && !allow_denorm && (E - 127) < e_min) return 0.0f;
    int extra = e_min - e_true;                      // target-format subnormal: coarser quantum
    extra = extra < 0 ? 0 : extra;
    int shift = 24 - mbits + extra;
    const uint32_t sig = E ? (M | 0x800000u) : M;    // value = sig * 2^(e_true - 23)
    uint32_t q;
    if (shift > 25) {
        q = 0;
    } else if (shift <= 0) {
        q = sig; shift = 0;
    } else {
        const uint32_t keep = sig >> shift;
        const uint32_t rem = sig & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        q = keep;
        if (rmode == 0) q += (rem >= half);
        else if (rmode == 2) q += (rem > half) || (rem == half && (keep & 1u));
    }
    if (q == 0) return 0.0f;
    // q * 2^(shift + e_true - 23): exact (q < 2^25 and the result is representable or overflows)
    float mag = __builtin_ldexpf((float)q, shift + e_true - 23);     // one v_ldexp_f32
    if (mag > max_norm) mag = (is_int || saturate) ? max_norm : u2f(0x7F800000u);
    return u2f(sign | f2u(mag));
}

// ---------------------------------------------------------------------------
// posit<n,es> round-to-nearest (ties to even on the kept exponent/fraction bits,
// never to 0 / NaR), pinned against number_system/posit/Posit.py:221-385 through
// the oracle.  Input is an fp32 value.
// ---------------------------------------------------------------------------
MSQ_HD float posit_decode(uint32_t body, int n, int es) {   // body: magnitude pattern, 1..2^(n-1)-1
    const int top = n - 2;
    const int rs = (body >> top) & 1;
    // run length of bits equal to rs starting at bit `top`
    uint32_t x = rs ? (~body) : body;
    x &= (1u << (n - 1)) - 1u;
    int rl = x ? (top - (31 - __builtin_clz(x))) : (n - 1);
    int k = rs ? rl - 1 : -rl;
    int rem = n - 1 - rl - 1; rem = rem < 0 ? 0 : rem;
    uint32_t tail = body & ((1u << rem) - 1u);
    int eb = es < rem ? es : rem;
    int fb = rem - eb;
    int e = (int)((tail >> fb) << (es - eb));
    uint32_t f = tail & ((1u << fb) - 1u);
    float mant = 1.0f + (float)f * pow2i(-fb);
    int sc = (1 << es) * k + e;
    return mant * pow2i(sc < -126 ? -126 : (sc > 127 ? 127 : sc));
}

MSQ_HD float posit_round(float a, int n, int es) {
    if (a != a || a == 0.f || __builtin_isinf(a)) return a;
    uint32_t u = f2u(a);
    const uint32_t sign = u & 0x80000000u;
    u &= 0x7FFFFFFFu;
    int E = (int)(u >> 23);
    uint32_t M = u & 0x7FFFFFu;
    int scale;
    if (E == 0) {                                   // normalise an fp32 subnormal
        int lz = __builtin_clz(M) - 8;              // leading one -> bit 23
        M = (M << lz) & 0x7FFFFFu;
        scale = -126 - lz;
    } else scale = E - 127;
    const int ul = 1 << es;
    int k = (scale >= 0) ? (scale >> es) : -((-scale + ul - 1) >> es);
    int e = scale - k * ul;
    int rl = (k >= 0) ? k + 2 : 1 - k;
    const uint32_t maxpos = (1u << (n - 1)) - 1u;
    uint32_t body;
    if (rl >= n) {
        body = (k >= 0) ? maxpos : 1u;
    } else {
        const int avail = n - 1 - rl;
        const uint32_t regime = (k >= 0) ? (((1u << (rl - 1)) - 1u) << 1) : 1u;
        const uint32_t ef = ((uint32_t)e << 23) | M;         // es + 23 bits
        const int rest_bits = es + 23 - avail;               // > 0 for n <= 16
        const uint32_t hi = ef >> rest_bits;
        const uint32_t rest = ef & ((1u << rest_bits) - 1u);
        const uint32_t half = 1u << (rest_bits - 1);
        uint32_t b = (regime << avail) | hi;
        if (rest > half || (rest == half && avail > 0 && (hi & 1u))) b += 1;
        b = b == 0 ? 1u : b;
        b = b > maxpos ? maxpos : b;
        body = b;
    }
    return u2f(sign | f2u(posit_decode(body, n, es)));
}

MSQ_HD float quant_elem(float a, const Fmt& f, int rmode) {
    if (f.kind == 1) return posit_round(a, f.mbits, f.ebits);
    return quant_core_sat(a, f.mbits, f.ebits, f.max_norm, rmode);
}

// ---------------------------------------------------------------------------
// utils/quant.py:498-541 _shared_exponents for a block reduced to max|.|, and the
// clamps of :207-211 / :237-242 (variant 0) / mx_ops.py:269-273 (variant 1).
// Exponents travel as float so that the reference's NaN-on-overflow is kept.
// ---------------------------------------------------------------------------
MSQ_HD float shared_exp_of_max(float mx) {
    if (mx != mx) return mx;
    if (__builtin_isinf(mx)) return mx;
    const float t = (mx == 0.f) ? pow2i(-126) : mx;    // + FP32_MIN_NORMAL * (max == 0)
    return (float)ilog2f_torch(t);                     // floor(torch.log2(.)), utils/quant.py:525-529
}

MSQ_HD float clamp_scale_exp(float e, int scale_bits, int variant) {
    const float lim = (float)((1 << (scale_bits - 1)) - 1);
    if (e > lim) return u2f(0x7FC00000u);
    if (e < -lim) return (variant == 0 && -lim < -20.f) ? -20.f : -lim;
    return e;
}

// 2^e and 2^-e for a (possibly NaN / inf) float exponent
MSQ_HD float exp2f_int(float e) {
    if (e != e) return e;
    if (e > 127.f) return u2f(0x7F800000u);
    if (e < -149.f) return 0.f;
    return pow2i((int)e);
}

// ---------------------------------------------------------------------------
// torch CPU fp32 summation orders (aten SumKernel) -- see oracle/msq_oracle.c.
// x is a register array of N block elements.
// ---------------------------------------------------------------------------
template <int N>
MSQ_HD float sum_cascade(const float (&x)[N]) {
    // multi_row_sum with level_step 16 (N < 256): every full run of 16 is summed
    // left to right into acc0 and folded into acc1; the tail stays in acc0.
    static_assert(N < 256, "cascade levels 2/3 not needed below 256");
    float acc1 = 0.f;
    constexpr int FULL = (N / 16) * 16;
#pragma unroll
    for (int i = 0; i < FULL; i += 16) {
        float acc0 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc0 += x[i + j];
        acc1 += acc0;
    }
    float acc0 = 0.f;
#pragma unroll
    for (int j = FULL; j < N; ++j) acc0 += x[j];
    return acc0 + acc1;
}

template <int N>
MSQ_HD float sum_ilp4(const float (&x)[N]) {
    // row_sum: 4 interleaved accumulators over N/4 rows (cascade step 16 over rows),
    // remainder into a[0], then ((a0+a1)+a2)+a3.
    static_assert(N / 4 < 256, "");
    constexpr int S = N / 4, FULL = (S / 16) * 16;
    float b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FULL; i += 16) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += x[(i + j) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] += a[k];
    }
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = FULL; r < S; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += x[r * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += b[k];
#pragma unroll
    for (int r = S * 4; r < N; ++r) a[0] += x[r];
    a[0] += a[1]; a[0] += a[2]; a[0] += a[3];
    return a[0];
}

template <int N>
MSQ_HD float sum_inner8(const float (&x)[N]) {
    constexpr int V = 8, VS = N / V, SI = VS / 4;
    float p[4][V];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < V; ++l) p[k][l] = 0.f;
    // SI <= 4 for N <= 128: plain sequential rows
#pragma unroll
    for (int r = 0; r < SI; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < V; ++l) p[k][l] += x[(r * 4 + k) * V + l];
#pragma unroll
    for (int r = SI * 4; r < VS; ++r)
#pragma unroll
        for (int l = 0; l < V; ++l) p[0][l] += x[r * V + l];
#pragma unroll
    for (int k = 1; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < V; ++l) p[0][l] += p[k][l];
    float fin = 0.f;
#pragma unroll
    for (int r = VS * V; r < N; ++r) fin += x[r];
#pragma unroll
    for (int l = 0; l < V; ++l) fin += p[0][l];
    return fin;
}

// aten WelfordOps<float,double>: sequential, sqrt in double, one rounding to float
template <int N>
MSQ_HD float std_welford(const float (&x)[N], int correction) {
    double mean = 0.0, m2 = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double d = (double)x[i];
        const double delta = d - mean;
        mean = mean + delta / (double)(i + 1);
        const double delta2 = d - mean;
        m2 = m2 + delta * delta2;
    }
    double den = (double)N - (double)correction;
    den = den < 0 ? 0 : den;
    return (float)__builtin_sqrt(m2 / den);
}

}  // namespace msq
