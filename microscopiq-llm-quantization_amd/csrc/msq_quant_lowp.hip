// msq_quant_lowp.hip -- the MicroScopiQ fake-quant computed IN the tensor dtype (fp16 / bf16).
//
// The reference's RTN harness hands the checkpoint weight to quantize_mx_outlier_v1 in its own dtype
// (llm/llama.py:238, fp16 for Llama-2 / OPT), so every torch op of utils/quant.py:147-266 and
// number_system/mx/elemwise_ops.py:84-174 runs on a Half / BFloat16 CPU tensor: ATen evaluates each op in fp32 and
// rounds the result back to the tensor dtype, op by op.  The kernels here reproduce that sequence literally -- R()
// is the per-op rounding (v_cvt_f16_f32 / bf16 RNE) -- so that the result is bit-identical to the reference's for
// fp16 / bf16 inputs as well (oracle: msq_oracle_outlier_fakequant_lowp; goldens: tests/golden/outlier_lowp.npz).
// What differs from "compute in fp32, round once":
//   * mean = R(fp32 sum in ATen's order / n); std = R((float) double Welford);  bounds R(mean -+ R(k std));
//   * floor(log2(x)) = floor(R(log2f(x))): results within half a T-ulp below an integer round UP to it, i.e. values
//     just under a power of two get the next exponent (21 % of all bf16 values).  floor_log2_lowp() is a closed rule
//     that matches torch for every positive fp16 / bf16 value (tests/golden/log2_lowp.npz, exhaustive);
//   * 2^e, every scaling, the +0.5 of the rounding and the final recombination are rounded to T;
//   * FP32_MIN_NORMAL underflows to 0 in fp16: an all-zero block has log2(0) = -inf -> clamp -> -20.
// One quantisation block per lane, register resident; the element loop is a straight-line float pipeline (VALU
// bound, ~200 instructions per element), read once / written once.  Offline step of the harness, not a GEMM-path
// kernel: 6.5 G weights of Llama-2-7B take well under a second.
// Compiled with -ffp-contract=off (no fused multiply-adds: every op rounds on its own).
//
// Round 6: the kernels above are the FALL-BACK now.  Weight-like tensors (round to nearest, the format pairs with a hardware codec, whole
// blocks of 8 ... 64) run k_outlier_lowp_pk / _pk2 further down -- two values per dword from load to store, ~30 instructions per element,
// 3-4 x faster, same bits -- and only the waves outside their exponent bounds come back here through a list (k_outlier_lowp_list).  The same
// packed machinery with FLOAT32 semantics (outlier_block_pk32) serves dtype 1 / 2 of msq_outlier_fakequant (the MicroScopiQ KV cache), and the
// MX quantiser (k_mx_lowp_*) takes e2m1 through the scaled converts next to e4m3.  DESIGN.md 5.005.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"
#include "msq_outlier_core.h"

using namespace msq;

template <int DT> MSQ_D float Rr(float x) {
    if (DT == 1) {
        // v_cvt_f16_f32: RNE, fp16 subnormals kept, overflow -> Inf.  Through asm: written as (float)(_Float16)(a * b) hipcc fuses the
        // product and the conversion into v_fma_mixlo_f16 a, b, +0 -- and (-1)(+0) + (+0) = +0 loses the sign of a zero product
        // (the reference keeps -0 where a negative value rounds to zero: sign * floor(...), elemwise_ops.py:64-70).
        uint32_t h;
        float r;
        asm("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(x));
        asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(h));
        return r;
    }
    return (float)(__bf16)x;                         // RNE on the upper 16 bits
}
template <int DT> MSQ_D float ld16(const uint16_t* p, int64_t i) {
    if (DT == 1) return (float)__builtin_bit_cast(_Float16, p[i]);
    return u2f((uint32_t)p[i] << 16);
}
template <int DT> MSQ_D void st16(uint16_t* p, int64_t i, float v) {   // v is already a T value
    if (DT == 1) p[i] = __builtin_bit_cast(uint16_t, (_Float16)v);
    else p[i] = (uint16_t)(f2u(v) >> 16);
}

// floor(R(log2f(v))) for a non-negative T value v (see the header comment).  c[jb + 1] = 2^(-2^(jb - p)), p = 11 / 8
// significand bits: the significand m in [0.5, 1) of v rounds up to the next integer exponent iff m > c.
template <int DT> MSQ_D float floor_log2_lowp(float v) {
    if (v != v) return v;
    if (v == 0.f) return u2f(0xFF800000u);
    if (__builtin_isinf(v)) return v;
    const int e = __builtin_amdgcn_frexp_expf(v);       // v = m 2^e, m in [0.5, 1)
    const float m = __builtin_amdgcn_frexp_mantf(v);
    const int k = e - 1, u = e;
    if (u == 0) return (float)k;
    const int au = u < 0 ? -u : u;
    int jb = 31 - __builtin_clz((unsigned)au);
    if (u > 0 && (au & (au - 1)) == 0) jb -= 1;
    float c;
    if (DT == 1) {
        const float t[9] = {0.9998307824134827f, 0.9996616244316101f, 0.999323308467865f, 0.9986470937728882f, 0.9972960352897644f,
                            0.9945994019508362f, 0.9892280101776123f, 0.9785720705986023f, 0.9576032757759094f};
        c = t[jb + 1];
    } else {
        const float t[9] = {0.9986470937728882f, 0.9972960352897644f, 0.9945994019508362f, 0.9892280101776123f, 0.9785720705986023f,
                            0.9576032757759094f, 0.9170040488243103f, 0.8408964276313782f, 0.7071067690849304f};
        c = t[jb + 1];
    }
    return (m > c) ? (float)u : (float)k;
}
// x / d for the divisors of this file, which are R(2^e): a normal fp32 power of two has an exact reciprocal and
// x * (1 / d) is the same correctly rounded quotient (one v_mul instead of the ~11-instruction IEEE division); 0, Inf,
// NaN, subnormal or non-power divisors (possible only through under- / overflow of R(2^e)) take the real division.
struct Div { float d, r; bool fast; };
MSQ_D Div make_div(float d) {
    const uint32_t u = f2u(d);
    const uint32_t e = (u >> 23) & 0xFFu;
    Div v; v.d = d; v.fast = ((u & 0x807FFFFFu) == 0u) && e >= 1u && e <= 253u;
    v.r = u2f((254u - e) << 23);
    return v;
}
MSQ_D float divp(float x, const Div& v) {
    if (__builtin_expect(v.fast, 1)) return x * v.r;
    return x / v.d;
}
template <int DT> MSQ_D float pow2_lowp(float e) { return Rr<DT>(exp2f_int(e)); }     // R(powf(2, e))

// elemwise_ops.py:47-78
template <int DT> MSQ_D float round_mantissa_lowp(float a, int rmode) {
    if (a != a) return a;
    const float s = (a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f);
    const float m = __builtin_fabsf(a);
    if (rmode == 1) return Rr<DT>(s * __builtin_floorf(m));
    if (rmode == 0) return Rr<DT>(s * __builtin_floorf(Rr<DT>(m + 0.5f)));
    const float t = Rr<DT>(m - 0.5f);
    float r = __builtin_fmodf(t, 2.0f);
    if (r != 0.f && r < 0.f) r += 2.0f;
    r = Rr<DT>(r);
    const float tie = (r == 0.f) ? 1.f : 0.f;
    return Rr<DT>(s * Rr<DT>(__builtin_floorf(Rr<DT>(m + 0.5f)) - tie));
}
// elemwise_ops.py:84-174 (saturate_normals, allow_denorm: utils/quant.py:218-221), every op rounded to T
template <int DT> MSQ_D float core_lowp(float a, const Fmt& f, int rmode) {
    float out = a, pe = 0.f, p2 = 1.f;
    const bool have_pe = f.ebits != 0;
    if (have_pe) {
        const float t = Rr<DT>(__builtin_fabsf(a) + ((a == 0.f) ? 1.f : 0.f));
        pe = floor_log2_lowp<DT>(t);
        const float min_exp = (float)(2 - (1 << (f.ebits - 1)));
        pe = (pe == pe && pe < min_exp) ? min_exp : pe;
        p2 = pow2_lowp<DT>(pe);
    }
    const float sh = pow2i(f.mbits - 2), rsh = pow2i(2 - f.mbits);      // out / sh == out * rsh exactly
    if (have_pe) out = Rr<DT>(Rr<DT>(divp(out, make_div(p2))) * sh);
    else out = Rr<DT>(out * sh);
    out = round_mantissa_lowp<DT>(out, rmode);
    if (have_pe) out = Rr<DT>(Rr<DT>(out * rsh) * p2);
    else out = Rr<DT>(out * rsh);
    const float mn = Rr<DT>(f.max_norm);
    if (out == out) { out = out < -mn ? -mn : out; out = out > mn ? mn : out; }
    if (__builtin_isinf(a)) out = a;
    return out;
}
template <int DT> MSQ_D float shared_exp_lowp(float mx) {
    if (mx != mx) return mx;
    const float t = Rr<DT>(pow2i(-126) * ((mx == 0.f) ? 1.f : 0.f));
    return floor_log2_lowp<DT>(Rr<DT>(mx + t));
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path of one block (round 4; the op-by-op path below is ~200 vector instructions per element: 358 GB/s on a 16384 x 4096 fp16
// weight).  Same values, fewer instructions -- nothing is approximated:
//  * every scaling of utils/quant.py:214-258 is by R(2^e) with an integer e: inside the bounds checked below the scale IS 2^e in T
//    and the product / quotient is exact in float32, so `R(x / R(2^e))` is `R(ldexp(x, -e))` (one v_ldexp + the rounding pair);
//  * roundings to T that cannot change the value are dropped: a T value scaled UP by a power of two, the quotient by the private
//    exponent's power of two, the short significand coming out of the element codec (see the derivation next to each line);
//    those that can are kept: into the element domain (:214, :216, :247), R(|v| + 0.5) in front of the floor, back out (:224, :258);
//  * an element is an inlier or an outlier, never both: ONE trip through the codec with the parameters of its side (the other side
//    of the reference computes on a zero and contributes +0; the final `inl + o` only turns a -0 into +0);
//  * max |R(o 2^e_in)| over the outliers is R(max |o| 2^e_in) (both maps are monotone): no second array, no second pass;
//  * floor(R(log2 t)): the integer rule floor_log2_fast (pinned exhaustively by tests/golden/log2_lowp.npz through k_mx_lowp).
// Taken when the WHOLE wave qualifies: round to nearest, no flush, finite block, both shared exponents inside T's exact range.
// Everything else -- NaN / Inf members, scales at the edge of fp16, other rounding modes -- takes the op-by-op path.
// ---------------------------------------------------------------------------------------------------------------
template <int DT> MSQ_D int floor_log2_fast(float t);                  // defined with the MX kernels below
template <int DT> MSQ_D int floor_log2_tab(float t, const uint8_t* tab);
template <int DT> MSQ_D void floor_log2_tab_init(uint8_t* tab);
struct LowpSide { int mine, shexp; float mn; bool has_pe; };
MSQ_D LowpSide lowp_side(const Fmt& f) {
    LowpSide s;
    s.has_pe = f.ebits != 0;
    s.mine = s.has_pe ? 2 - (1 << (f.ebits - 1)) : 0;
    s.shexp = f.mbits - 2;
    s.mn = f.max_norm;
    return s;
}
// UPX: both shared exponents are <= 0 for every lane of the wave (weights below 1: the usual case) -- then the outlier's second scaling
// into its domain (x 2^-eo) and the final scaling back (x 2^-ei) go UP by a power of two: exact for a T value, no rounding pair
template <int DT, bool UPX>
MSQ_D float outlier_elem_fast(float x, bool m, int ei, int eo, const LowpSide& si, const LowpSide& so, const uint8_t* tab) {
    // into the element domain: inlier R(x / sc_in) (:214); outlier R(R(x sc_in) / sc_out) (:216, :247)
    float t = Rr<DT>(__builtin_ldexpf(x, m ? ei : -ei));
    t = __builtin_ldexpf(t, m ? -eo : 0);                              // (an inlier is a T value already: R is the identity)
    if (!UPX) t = Rr<DT>(t);
    // elemwise_ops.py:84-174 with the side's parameters
    const float av = __builtin_fabsf(t);
    int pe = floor_log2_tab<DT>(av, tab);                              // floor(R(log2 |t|)); t == 0: some exponent, the mantissa below is 0
    const int mine = m ? so.mine : si.mine;
    pe = pe < mine ? mine : pe;
    pe = (m ? so.has_pe : si.has_pe) ? pe : 0;
    const int sa = (m ? so.shexp : si.shexp) - pe;
    const float mm = __builtin_ldexpf(av, sa);                         // |t| / 2^pe * 2^(bits - 2): a T value scaled by powers of two, exact
    const float q = __builtin_floorf(Rr<DT>(mm + 0.5f));               // the rounding that matters (ties just below n + 1/2)
    float r = __builtin_ldexpf(q, -sa);                                // q 2^(2 - bits) 2^pe: at most bits - 1 significant bits, exact in T
    const float mn = m ? so.mn : si.mn;
    r = r > mn ? mn : r;                                               // saturate_normals
    r = __builtin_copysignf(r, t);
    // back: inlier R(v sc_in) (:224); outlier R(R(o sc_out) / sc_in) (:258)
    r = Rr<DT>(__builtin_ldexpf(r, m ? eo : ei));
    r = __builtin_ldexpf(r, m ? -ei : 0);
    if (!UPX) r = Rr<DT>(r);
    return r + 0.0f;                                                   // :262 inl + o with the other side +0: a -0 becomes +0
}

template <int BS, int DT>
MSQ_D int outlier_block_lowp(float (&a)[BS], uint32_t (&mkw)[(BS + 31) / 32], float& se_in_o, float& se_out_o,
                             const OutlierArgs& A, int order, const uint8_t* l2tab) {
    int status = 0;
    float lo, hi;
    {
        float ab[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) ab[b] = __builtin_fabsf(a[b]);
        float s;
        if (order == 1) s = sum_inner8<BS>(ab);
        else if (order == 2) s = sum_ilp4<BS>(ab);
        else s = sum_cascade<BS>(ab);
        const float mean = Rr<DT>(s / (float)BS);                       // utils/quant.py:477 (fp32 sum, one rounding)
        const float sd = Rr<DT>(std_twopass_checked<BS>(ab, 0));        // :478 (double Welford -> float -> T)
        const float ks = Rr<DT>(A.k * sd);
        lo = Rr<DT>(mean - ks); hi = Rr<DT>(mean + ks);                 // :489-490
    }
#pragma unroll
    for (int w = 0; w < (BS + 31) / 32; ++w) mkw[w] = 0u;
    {
        // ---- fast path: qualification on block-level quantities only
        float mxi = 0.f, mxo = 0.f;
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            const bool m = (a[b] < lo) || (a[b] > hi);                  // :492
            mkw[b >> 5] |= (m ? 1u : 0u) << (b & 31);
            const float t = __builtin_fabsf(a[b]);
            mxi = (!m && (t > mxi || t != t)) ? t : mxi;
            mxo = (m && (t > mxo || t != t)) ? t : mxo;
        }
        constexpr int ELO = (DT == 1) ? -24 : -100, EHI = (DT == 1) ? 15 : 100;     // R(2^e) is exactly 2^e and nothing leaves float32
        float se_in = shared_exp_lowp<DT>(mxi);                         // :196-198
        se_in = Rr<DT>(se_in - (float)A.fi.emax);                       // :207
        se_in = clamp_scale_exp(se_in, A.in_sb, 0);                     // :208-211
        bool fast = A.rmode == 0 && !A.flush && se_in == se_in && mxi < 3.0e38f && mxo < 3.0e38f && se_in >= (float)ELO && se_in <= (float)EHI;
        const int ei = fast ? (int)se_in : 0;
        const float mx_out = Rr<DT>(__builtin_ldexpf(mxo, ei));         // max over the block of R(o sc_in) (:216): monotone maps
        float se_out = shared_exp_lowp<DT>(mx_out);                     // :229-231
        se_out = Rr<DT>(se_out - (float)A.fo.emax);                     // :237
        se_out = clamp_scale_exp(se_out, A.out_sb, 0);                  // :239-242
        fast = fast && se_out == se_out && mx_out < 3.0e38f && se_out >= (float)ELO && se_out <= (float)EHI;
        // ... and neither side overflows T when it enters its element domain (a maximum just under a power of two whose exponent the
        // log2 rule lifted can reach 2^(emax + 1): 65536 for an e5m2 side is Inf in fp16 -- the op-by-op path carries that Inf)
        const int eo_ = fast ? (int)se_out : 0;
        // ... nor does R(2^pe) of the largest private exponent (the rule may lift it to emax + 1 = 16 for e5m2: Inf in fp16, NaN results
        // in the reference); floor_log2 is monotone, so the block maxima decide
        constexpr int PEMAX = (DT == 1) ? 15 : 127;
        const float ti_ = Rr<DT>(__builtin_ldexpf(mxi, -ei)), to_ = Rr<DT>(__builtin_ldexpf(mx_out, -eo_));
        fast = fast && ti_ < 3.0e38f && to_ < 3.0e38f && (ti_ == 0.f || floor_log2_fast<DT>(ti_) <= PEMAX) && (to_ == 0.f || floor_log2_fast<DT>(to_) <= PEMAX);
        if (__builtin_amdgcn_ballot_w64(!fast) == 0ull) {
            const int eo = (int)se_out;
            const LowpSide si = lowp_side(A.fi), so = lowp_side(A.fo);
            if (__builtin_amdgcn_ballot_w64(ei > 0 || eo > 0) == 0ull) {
#pragma unroll
                for (int b = 0; b < BS; ++b)
                    a[b] = outlier_elem_fast<DT, true>(a[b], (mkw[b >> 5] >> (b & 31)) & 1u, ei, eo, si, so, l2tab);
            } else {
#pragma unroll
                for (int b = 0; b < BS; ++b)
                    a[b] = outlier_elem_fast<DT, false>(a[b], (mkw[b >> 5] >> (b & 31)) & 1u, ei, eo, si, so, l2tab);
            }
            se_in_o = se_in; se_out_o = se_out;
            return 0;
        }
    }
    // ---- op-by-op path (utils/quant.py:147-266 literally, every op rounded to T)
    float mx_in = 0.f;
    float inl[BS];
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        const bool m = (mkw[b >> 5] >> (b & 31)) & 1u;
        const float mf = m ? 1.f : 0.f;
        inl[b] = a[b] * (1.0f - mf);                                    // :192 (exact: x * 1 or x * 0)
        a[b] = a[b] * mf;                                               // :193
        const float t = __builtin_fabsf(inl[b]);
        mx_in = (t > mx_in || t != t) ? t : mx_in;
    }
    float se_in = shared_exp_lowp<DT>(mx_in);                           // :196-198
    const bool fl = A.flush && !(se_in > -127.f);
    se_in = Rr<DT>(se_in - (float)A.fi.emax);                           // :207
    se_in = clamp_scale_exp(se_in, A.in_sb, 0);                         // :208-211
    const float sc_in = pow2_lowp<DT>(se_in);
    const Div d_in = make_div(sc_in);
    float mx_out = 0.f;
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        float v = inl[b];
        if (fl) v = v * 0.f;
        v = Rr<DT>(divp(v, d_in));                                      // :214
        a[b] = Rr<DT>(a[b] * sc_in);                                    // :216
        v = core_lowp<DT>(v, A.fi, A.rmode);                            // :218-221
        v = Rr<DT>(v * sc_in);                                          // :224
        if (v != v || a[b] != a[b]) status |= MSQ_STATUS_NAN;           // :225-226
        inl[b] = v;
        const float t = __builtin_fabsf(a[b]);
        mx_out = (t > mx_out || t != t) ? t : mx_out;
    }
    float se_out = shared_exp_lowp<DT>(mx_out);                         // :229-231
    if (se_out != se_out) status |= MSQ_STATUS_NAN;
    se_out = Rr<DT>(se_out - (float)A.fo.emax);                         // :237
    se_out = clamp_scale_exp(se_out, A.out_sb, 0);                      // :239-242
    if (se_out != se_out) status |= MSQ_STATUS_NAN;
    const float sc_out = pow2_lowp<DT>(se_out);
    const Div d_out = make_div(sc_out);
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        float o = Rr<DT>(divp(a[b], d_out));                            // :247
        if (o != o) status |= MSQ_STATUS_NAN;                           // :250
        o = core_lowp<DT>(o, A.fo, A.rmode);                            // :252-255
        o = Rr<DT>(divp(Rr<DT>(o * sc_out), d_in));                     // :258
        a[b] = Rr<DT>(inl[b] + o);                                      // :262
    }
    se_in_o = se_in; se_out_o = se_out;
    return status;
}

// lane <-> (p, nb, q), q fastest.  post > 1: each of the BS row accesses of a wave is one contiguous 128-byte segment;
// post == 1: every lane walks its own contiguous block (the cache lines are shared by neighbouring lanes).
// the work of lane `t` (one block); every lane of a wave calls it with consecutive t (the transposition slice is the wave's)
template <int BS, int DT>
MSQ_D void outlier_lowp_lane(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, const OutlierArgs& A, int64_t t, const uint8_t* l2tab, char* xsm) {
    const int64_t total = A.pre * A.nblk * A.post;
    if (t >= total) return;
    const int64_t q = t % A.post;
    const int64_t nb = (t / A.post) % A.nblk;
    const int64_t p = t / (A.post * A.nblk);
    const int64_t a0 = nb * BS;
    const int64_t base = (p * A.axis_len + a0) * A.post + q;
    float a[BS];
    // blocks along the contiguous axis: the lane's block is BS * 2 contiguous bytes -> 16-byte accesses
    const bool vec = (BS % 8 == 0) && A.post == 1 && (A.axis_len % BS) == 0 &&
                     ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    // ... coalesced: the 64 blocks of a full wave are one contiguous run of 64 * BS * 2 bytes -- lane l loads 16-byte chunk c * 64 + l
    // of it (a wave instruction = 1 KiB contiguous; lane-per-block loads touched every 128-byte line BS / 8 times, 34 % of the wave
    // cycles were memory waits: profiles/r04_pmc_outlier_lowp.txt), the chunks cross to their block's lane through the wave's LDS
    // slice (row stride BS * 2 + 16 bytes: conflict-free ds_read_b128), and back the same way for the stores
    constexpr int CH = BS / 8;                                            // 16-byte chunks per block
    constexpr int ROWB = BS * 2 + 16;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave_t0 = t - lane;
    const bool xpose = vec && (BS % 8 == 0) && (wave_t0 + 64 <= total);   // a full wave (wave-uniform)
    char* const wsl = xsm + wv * (64 * ROWB);
    const int64_t wave_base = wave_t0 * BS;                               // post == 1: block index t starts at element t * BS
    if (xpose) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = c * 64 + lane;                                  // chunk of the wave's run
            const uint4 u = *reinterpret_cast<const uint4*>(in + wave_base + (int64_t)j * 8);
            *reinterpret_cast<uint4*>(wsl + (j / CH) * ROWB + (j % CH) * 16) = u;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                               // this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            union { uint4 u; uint16_t h[8]; } v;
            v.u = *reinterpret_cast<const uint4*>(wsl + lane * ROWB + c * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[c * 8 + j] = ld16<DT>(v.h, j);
        }
    } else if (vec) {
#pragma unroll
        for (int c = 0; c < BS / 8; ++c) {
            union { uint4 u; uint16_t h[8]; } v;
            v.u = *reinterpret_cast<const uint4*>(in + base + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[c * 8 + j] = ld16<DT>(v.h, j);
        }
    } else {
#pragma unroll
        for (int b = 0; b < BS; ++b)
            a[b] = (a0 + b < A.axis_len) ? ld16<DT>(in, base + (int64_t)b * A.post) : 0.f;     // zero padding, :563-583
    }
    int order = 1;                                           // reduced dim contiguous
    if (A.post > 1) {
        const int64_t lim = (A.post >= 8) ? (A.post / 32) * 32 : (A.post / 4) * 4;
        order = (q < lim) ? 0 : 2;
    }
    uint32_t mkw[(BS + 31) / 32];
    float se_in, se_out;
    const int status = outlier_block_lowp<BS, DT>(a, mkw, se_in, se_out, A, order, l2tab);
    if (xpose) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                               // (the reads of the input image are long done)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            union { uint4 u; uint16_t h[8]; } v;
#pragma unroll
            for (int j = 0; j < 8; ++j) st16<DT>(v.h, j, a[c * 8 + j]);
            *reinterpret_cast<uint4*>(wsl + lane * ROWB + c * 16) = v.u;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = c * 64 + lane;
            *reinterpret_cast<uint4*>(out + wave_base + (int64_t)j * 8) = *reinterpret_cast<const uint4*>(wsl + (j / CH) * ROWB + (j % CH) * 16);
        }
    } else if (vec) {
#pragma unroll
        for (int c = 0; c < BS / 8; ++c) {
            union { uint4 u; uint16_t h[8]; } v;
#pragma unroll
            for (int j = 0; j < 8; ++j) st16<DT>(v.h, j, a[c * 8 + j]);
            *reinterpret_cast<uint4*>(out + base + c * 8) = v.u;
        }
    }
    if (vec) {
        if (A.mask) {
#pragma unroll
            for (int b = 0; b < BS; ++b) A.mask[base + b] = (uint8_t)((mkw[b >> 5] >> (b & 31)) & 1u);
        }
    } else {
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            if (a0 + b < A.axis_len) {
                st16<DT>(out, base + (int64_t)b * A.post, a[b]);
                if (A.mask) A.mask[base + (int64_t)b * A.post] = (uint8_t)((mkw[b >> 5] >> (b & 31)) & 1u);
            }
        }
    }
    if (A.e_in) A.e_in[(p * A.nblk + nb) * A.post + q] = se_in;
    if (A.e_out) A.e_out[(p * A.nblk + nb) * A.post + q] = se_out;
    if (status && A.status) atomicOr(A.status, status);
}
template <int BS, int DT>
__global__ void __launch_bounds__(256)
k_outlier_lowp(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, OutlierArgs A) {
    __shared__ uint8_t l2tab[256];                          // bump allowance of floor(R(log2 .)) by biased exponent (fast path)
    __shared__ __attribute__((aligned(16))) char xsm[(BS % 8 == 0) ? 4 * 64 * (BS * 2 + 16) : 16];   // one transposition slice per wave
    floor_log2_tab_init<DT>(l2tab);
    __syncthreads();
    outlier_lowp_lane<BS, DT>(in, out, A, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, l2tab, xsm);
}
// NaN-keeping maximum over the BS lanes of a block (the sequential `t > mx || t != t ? t : mx` ends on NaN iff one member is NaN)
template <int BS> MSQ_D float group_max_nan(float t) {
#pragma unroll
    for (int o = 1; o < BS; o <<= 1) {
        const float u = __shfl_xor(t, o, 64);
        t = (u > t || u != u) ? u : t;
    }
    return t;
}
// The op-by-op path of outlier_block_lowp with ONE ELEMENT per lane (the BS lanes of a block are neighbours inside a wave; every lane holds
// the whole block for the statistics, which it computes redundantly).  A handed-back wave is one instruction stream of ~250 instructions per
// element whatever the number of lanes that need it: a lane per block walks BS elements of it (~30 us for a few entries, longer than the
// packed kernel itself), a lane per element one.  blk[]: the block's BS values; x = blk[i], this lane's.
template <int BS, int DT>
MSQ_D float outlier_elem_lowp(const float (&blk)[BS], float x, bool& mask, float& se_in_o, float& se_out_o, int& status, const OutlierArgs& A, int order) {
    float lo, hi;
    {
        float ab[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) ab[b] = __builtin_fabsf(blk[b]);
        float s;
        if (order == 1) s = sum_inner8<BS>(ab);
        else if (order == 2) s = sum_ilp4<BS>(ab);
        else s = sum_cascade<BS>(ab);
        const float mean = Rr<DT>(s / (float)BS);                       // utils/quant.py:477
        const float sd = Rr<DT>(std_twopass_checked<BS>(ab, 0));        // :478
        const float ks = Rr<DT>(A.k * sd);
        lo = Rr<DT>(mean - ks); hi = Rr<DT>(mean + ks);                 // :489-490
    }
    const bool m = (x < lo) || (x > hi);                                // :492
    mask = m;
    const float mf = m ? 1.f : 0.f;
    float inl = x * (1.0f - mf);                                        // :192
    float a = x * mf;                                                   // :193
    const float mx_in = group_max_nan<BS>(__builtin_fabsf(inl));
    float se_in = shared_exp_lowp<DT>(mx_in);                           // :196-198
    const bool fl = A.flush && !(se_in > -127.f);
    se_in = Rr<DT>(se_in - (float)A.fi.emax);                           // :207
    se_in = clamp_scale_exp(se_in, A.in_sb, 0);                         // :208-211
    const float sc_in = pow2_lowp<DT>(se_in);
    const Div d_in = make_div(sc_in);
    float v = inl;
    if (fl) v = v * 0.f;
    v = Rr<DT>(divp(v, d_in));                                          // :214
    a = Rr<DT>(a * sc_in);                                              // :216
    v = core_lowp<DT>(v, A.fi, A.rmode);                                // :218-221
    v = Rr<DT>(v * sc_in);                                              // :224
    if (v != v || a != a) status |= MSQ_STATUS_NAN;                     // :225-226
    const float mx_out = group_max_nan<BS>(__builtin_fabsf(a));
    float se_out = shared_exp_lowp<DT>(mx_out);                         // :229-231
    if (se_out != se_out) status |= MSQ_STATUS_NAN;
    se_out = Rr<DT>(se_out - (float)A.fo.emax);                         // :237
    se_out = clamp_scale_exp(se_out, A.out_sb, 0);                      // :239-242
    if (se_out != se_out) status |= MSQ_STATUS_NAN;
    const float sc_out = pow2_lowp<DT>(se_out);
    const Div d_out = make_div(sc_out);
    float o = Rr<DT>(divp(a, d_out));                                   // :247
    if (o != o) status |= MSQ_STATUS_NAN;                               // :250
    o = core_lowp<DT>(o, A.fo, A.rmode);                                // :252-255
    o = Rr<DT>(divp(Rr<DT>(o * sc_out), d_in));                         // :258
    se_in_o = se_in; se_out_o = se_out;
    return Rr<DT>(v + o);                                               // :262
}

// The waves the packed kernels below could not take (their list: ws[0] = count, ws[8 ...] = first lane of each, in k_outlier_lowp's lane
// numbering).  A short list (the usual case: a few blocks without inliers per tensor) is done one element per lane -- a thread block takes
// 256 / BS of an entry's 64 blocks per trip; a long one (a tensor outside the packed form's bounds altogether) one block per lane, as
// k_outlier_lowp would have: the redundant statistics of the first form cost throughput, the second costs latency.
template <int BS, int DT>
__global__ void __launch_bounds__(256)
k_outlier_lowp_list(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, OutlierArgs A, const int64_t* __restrict__ ws, const uint8_t* __restrict__ marks, int64_t cap) {
    __shared__ uint8_t l2tab[256];
    __shared__ __attribute__((aligned(16))) char xsm[4 * 64 * (BS * 2 + 16)];
    const int64_t count = ws[0];
    if (count == 0) return;
    const int64_t total = A.pre * A.nblk * A.post;
    if (count < cap) {
        constexpr int BPT = 256 / BS, TRIPS = 64 / BPT;                 // blocks per trip; trips per entry
        uint16_t* const sx = reinterpret_cast<uint16_t*>(xsm);
        const int lb = threadIdx.x / BS, i = threadIdx.x % BS;
        for (int64_t w = blockIdx.x; w < count * TRIPS; w += gridDim.x) {
            const int64_t t = ws[8 + w / TRIPS] + (w % TRIPS) * BPT + lb;
            const bool live = t < total;
            const int64_t tt = live ? t : 0;
            const int64_t q = tt % A.post, nb = (tt / A.post) % A.nblk, p = tt / (A.post * A.nblk);
            const int64_t at = ((p * A.axis_len + nb * BS + i) * A.post) + q;
            __syncthreads();
            sx[threadIdx.x] = in[at];
            __syncthreads();
            float blk[BS];
#pragma unroll
            for (int b = 0; b < BS; ++b) blk[b] = ld16<DT>(sx, lb * BS + b);
            int order = 1;                                           // reduced dim contiguous
            if (A.post > 1) {
                const int64_t lim = (A.post >= 8) ? (A.post / 32) * 32 : (A.post / 4) * 4;
                order = (q < lim) ? 0 : 2;
            }
            bool m; float se_in, se_out; int status = 0;
            const float r = outlier_elem_lowp<BS, DT>(blk, ld16<DT>(sx, threadIdx.x), m, se_in, se_out, status, A, order);
            if (live) {
                st16<DT>(out, at, r);
                if (A.mask) A.mask[at] = (uint8_t)m;
                if (i == 0) {
                    if (A.e_in) A.e_in[(p * A.nblk + nb) * A.post + q] = se_in;
                    if (A.e_out) A.e_out[(p * A.nblk + nb) * A.post + q] = se_out;
                }
                if (status && A.status) atomicOr(A.status, status);
            }
        }
        return;
    }
    floor_log2_tab_init<DT>(l2tab);
    __syncthreads();
    // the list stopped growing at `cap`: the marks say which waves (a tensor mostly outside the bounds; every wave has written its own)
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (total + 63) / 64, step = (int64_t)gridDim.x * 4;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nwaves; w += step)
        if (marks[w]) outlier_lowp_lane<BS, DT>(in, out, A, w * 64 + lane, l2tab, xsm);
}

// ---------------------------------------------------------------------------------------------------------------
// `_quantize_mx` (number_system/mx/mx_ops.py:332-457, Python path) on a half-precision tensor, op by op as ATen's CPU half
// kernels evaluate it (oracle: msq_oracle_quantize_mx_lowp; fixtures tests/golden/quantize_mx_lowp.npz).  Same thread <->
// block map and access pattern as k_outlier_lowp: one lane per block of BS values along the axis.
// ---------------------------------------------------------------------------------------------------------------
// floor(R(log2f(t))) for a finite T value t > 0, integer form of floor_log2_lowp: the exact exponent E, plus one when the
// significand sits within dmax T-ulps below 2.0 (d = 2^(p-1) - fraction <= dmax), dmax = what c[jb + 1] of floor_log2_lowp
// allows for the binade of |E + 1| (checked against the reference-made fixtures through k_mx_lowp).
// bump allowance of exponent E (see floor_log2_fast): a function of E alone -- the vectorised kernels keep it as a byte table in LDS
// (index = biased exponent) and look it up instead of recomputing it per element
template <int DT> MSQ_D uint32_t floor_log2_dmax(int E) {
    const int up = E + 1;
    int au = up < 0 ? -up : up;
    au |= (up == 0) ? 1 : 0;
    int jb = 31 - __builtin_clz((unsigned)au);
    jb -= (up > 0 && (au & (au - 1)) == 0) ? 1 : 0;
    const uint32_t lo = 0x02010000u;
    const uint32_t hi = (DT == 1) ? 0x2B160B05u : 0x28150A05u;
    const uint32_t top = (DT == 1) ? 86u : 74u;
    const int idx = jb + 1;
    const uint32_t w = idx < 4 ? lo : hi;
    uint32_t dmax = (w >> (8 * (idx & 3))) & 0xFFu;
    return idx >= 8 ? top : dmax;
}
template <int DT> MSQ_D int floor_log2_tab(float t, const uint8_t* tab) {
    const uint32_t u = f2u(t);
    const uint32_t e8 = (u >> 23) & 0xFFu;
    constexpr int SH = (DT == 1) ? 13 : 16;
    const uint32_t d = (0x800000u - (u & 0x7FFFFFu)) >> SH;
    return (int)e8 - 127 + ((d <= (uint32_t)tab[e8]) ? 1 : 0);
}
template <int DT> MSQ_D void floor_log2_tab_init(uint8_t* tab) {       // 256 entries; call from every thread, then __syncthreads()
    for (int e = threadIdx.x; e < 256; e += blockDim.x) tab[e] = (uint8_t)floor_log2_dmax<DT>(e - 127);
}
template <int DT> MSQ_D int floor_log2_fast(float t) {                   // branch-free
    const uint32_t u = f2u(t);
    const int E = (int)((u >> 23) & 0xFFu) - 127;
    const int up = E + 1;
    int au = up < 0 ? -up : up;
    au |= (up == 0) ? 1 : 0;                                        // up == 0: jb + 1 = 1 -> dmax 0 -> no bump
    int jb = 31 - __builtin_clz((unsigned)au);
    jb -= (up > 0 && (au & (au - 1)) == 0) ? 1 : 0;
    const uint32_t lo = 0x02010000u;                                // jb + 1 = 0..3: 0, 0, 1, 2 (both dtypes)
    const uint32_t hi = (DT == 1) ? 0x2B160B05u : 0x28150A05u;      // jb + 1 = 4..7: fp16 5, 11, 22, 43; bf16 5, 10, 21, 40
    const uint32_t top = (DT == 1) ? 86u : 74u;                     // jb + 1 = 8
    const int idx = jb + 1;
    const uint32_t w = idx < 4 ? lo : hi;
    uint32_t dmax = (w >> (8 * (idx & 3))) & 0xFFu;
    dmax = idx >= 8 ? top : dmax;
    constexpr int SH = (DT == 1) ? 13 : 16;                         // fraction bits below the T significand
    const uint32_t d = (0x800000u - (u & 0x7FFFFFu)) >> SH;         // T-ulps below 2.0 (1 ... 2^(p-1))
    return E + ((d <= dmax) ? 1 : 0);
}

struct MxLowpArgs {
    int64_t pre, axis_len, post, nblk;
    Fmt f;
    int scale_bits, rmode, flush;
    int* status;
};

// block-level quantities of one MX block (maximum mx already reduced) and the per-element arithmetic
template <int DT> struct MxBlk {
    float sc, mn, sh, rsh, dr, mnsc;
    Div dv;
    int min_exp, status, shexp, sei;
    bool fast, fl, pow2den, has_pe;
};
template <int DT> MSQ_D MxBlk<DT> mx_block_setup(float mx, const MxLowpArgs& A) {
    MxBlk<DT> B;
    float se = shared_exp_lowp<DT>(mx);                                  // :428-430
    B.fl = A.flush && !(se > -127.f);                                    // :433-434
    se = Rr<DT>(se - (float)A.f.emax);                                   // :438
    const float semax = (float)((1 << (A.scale_bits - 1)) - 1);          // :440-442
    if (se > semax) se = u2f(0x7FC00000u);
    if (se < -semax) se = -semax;
    B.status = (se != se) ? MSQ_STATUS_NAN : 0;
    B.sc = pow2_lowp<DT>(se);                                            // 2**shared_exp as a T tensor
    const float den = Rr<DT>(B.sc + 1e-6f);                              // :444
    B.dv = make_div(den);
    B.dr = B.dv.r;
    // Fast path (round to nearest, a finite block whose scale is a normal power of two in T, element formats with exponent bits):
    // the scalings by powers of two are exact in T as long as nothing leaves T's range -- guaranteed here by the bounds on se --,
    // so only the roundings that can change a value are kept: R(|x| + 0.5) before the floor, and the exponent rule.
    constexpr int TMINE = (DT == 1) ? -24 : -133, TMAXE = (DT == 1) ? 15 : 127;     // smallest subnormal / largest binade of T
    B.has_pe = A.f.ebits > 0;
    B.min_exp = B.has_pe ? 2 - (1 << (A.f.ebits - 1)) : 0;
    const int sei = (int)se;
    // den is the scale itself (a power of two: the division is an exact multiply) or, for small scales, the scale plus 1e-6
    // rounded to T: then x / den is a real division, rounded to T, and everything after it is unchanged
    B.pow2den = B.dv.fast && den == B.sc;
    const uint32_t de = (f2u(den) >> 23) & 0xFFu;
    B.fast = A.rmode == 0 && !B.fl && se == se && mx < 3.0e38f && de >= 1u && de <= 253u && f2u(B.sc) == ((uint32_t)(sei + 127) << 23) &&
             sei + B.min_exp + 2 - A.f.mbits >= TMINE && sei + A.f.emax + 2 <= TMAXE && sei - 1 >= TMINE + 11 && mx != 0.f;
    B.mn = A.f.max_norm;
    B.sh = pow2i(A.f.mbits - 2); B.rsh = pow2i(2 - A.f.mbits);
    B.shexp = A.f.mbits - 2; B.sei = sei;
    B.mnsc = A.f.max_norm * B.sc;                                        // exact inside the fast path's bounds (only used there)
    return B;
}
// fast path of one element (straight-line; the caller has made sure the whole wave takes it)
template <int DT> MSQ_D float mx_elem_fast(float x, const MxBlk<DT>& B, const uint8_t* tab = nullptr) {
    const float v = B.pow2den ? x * B.dr : Rr<DT>(x / B.dv.d);           // exact, or the rounded quotient by scale + 1e-6 (:444)
    const float av = __builtin_fabsf(v);
    int pe = tab ? floor_log2_tab<DT>(av, tab) : floor_log2_fast<DT>(av);   // v == 0: some small exponent, m = 0 below
    pe = pe < B.min_exp ? B.min_exp : pe;
    pe = B.has_pe ? pe : 0;                                              // integer element formats: no private exponent
    const int sa = B.shexp - pe;
    const float m = __builtin_ldexpf(av, sa);                            // |v| 2^-pe 2^(mbits - 2): exact (one v_ldexp_f32)
    const float q = __builtin_floorf(Rr<DT>(m + 0.5f));                  // the one rounding that matters (ties just below n + 1/2)
    float r = __builtin_ldexpf(q, B.sei - sa);                           // q 2^(2 - mbits) 2^pe 2^se: exact (bounds on se)
    r = r > B.mnsc ? B.mnsc : r;                                         // clamp to max_norm (scaled: the scale is a power of two)
    r = __builtin_copysignf(r, v);                                       // sign(v) * floor(...): -0 where a negative value rounds to zero
    r = (v == 0.f) ? 0.f : r;                                            // sign(+-0) = 0: 0 * floor(...) = +0
    return r;
}
template <int DT> MSQ_D float mx_elem(float x, const MxBlk<DT>& B, const MxLowpArgs& A) {
    if (B.fast) return mx_elem_fast<DT>(x, B);
    float v = x;
    if (B.fl) v = Rr<DT>(v * 0.f);
    v = Rr<DT>(divp(v, B.dv));
    v = core_lowp<DT>(v, A.f, A.rmode);                                  // :446-449
    return Rr<DT>(v * B.sc);                                             // :451
}


// ---------------------------------------------------------------------------------------------------------------------------
// MX-FP8 (e4m3) fast path through the gfx950 scaled converts: inside mx_elem_fast's conditions with a power-of-two divisor
// (MxBlk::fast && pow2den) the element step is "round x / 2^se half away from zero onto the e4m3 grid (subnormals kept), clamp to
// +-448, times 2^se" -- which is what v_cvt_scalef32_pk_fp8_{f16,f32} + v_cvt_scalef32_pk_{f16,bf16}_fp8 compute once (a) the
// lowest significand bit of the source is set (a sticky bit: the converts round to nearest EVEN, and with it no input is a tie
// any more while nothing else changes sides: T has >= 4 more significand bits than e4m3, and the bounds on se keep T's subnormal
// spacing 8 times finer than the grid's), (b) the magnitude is clamped first (the converts do not saturate) and (c) -0 inputs are
// made +0 (sign(+-0) = 0 in elemwise_ops.py:146-149; negative values that ROUND to zero keep -0 there and here).  Two values of ONE
// block per call (one dword of T pairs): ~3 instructions per element instead of ~20 (the KV-cache MX quantiser was VALU-bound at
// 0.3 of the HBM rate).  Pinned bit for bit by the half-tensor goldens (tests/golden/quantize_mx_lowp.npz) and the oracle tests.
// ---------------------------------------------------------------------------------------------------------------------------
typedef short lp_v2s_t __attribute__((ext_vector_type(2)));
typedef _Float16 lp_h2_t __attribute__((ext_vector_type(2)));
typedef __bf16 lp_b2_t __attribute__((ext_vector_type(2)));
template <int DT> MSQ_D uint32_t mx_e4m3_hw_pair(uint32_t w, float sc, float bound) {
    if (DT == 1) {
        const lp_h2_t z = {(_Float16)0.f, (_Float16)0.f};
        const lp_h2_t x = __builtin_bit_cast(lp_h2_t, w) + z;                                   // -0 -> +0
        lp_h2_t y = __builtin_bit_cast(lp_h2_t, __builtin_bit_cast(uint32_t, x) | 0x00010001u);  // sticky bit
        const lp_h2_t b = {(_Float16)bound, (_Float16)bound};
        y = __builtin_elementwise_max(__builtin_elementwise_min(y, b), -b);
        lp_v2s_t c = {0, 0};
        c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c, y, sc, false);
        return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(__builtin_bit_cast(uint32_t, c), sc, false));
    }
    float x0 = u2f(w << 16) + 0.f, x1 = u2f(w & 0xFFFF0000u) + 0.f;
    x0 = __builtin_amdgcn_fmed3f(u2f(f2u(x0) | 1u), -bound, bound);
    x1 = __builtin_amdgcn_fmed3f(u2f(f2u(x1) | 1u), -bound, bound);
    lp_v2s_t c = {0, 0};
    c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(c, x0, x1, sc, false);
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(__builtin_bit_cast(uint32_t, c), sc, false));
}
// MX-FP4 (e2m1), same contract.  The grid is four binades wide, so the excepted magnitude (pred_T of half the smallest step = a quarter of
// 2^se) sits in the bulk of every block: it is lifted over the tie (pk_lift: q = its T bits or 0xFFFF) instead of sending the wave away.
// The e2m1 converts saturate by themselves.
MSQ_D uint32_t pk_add_u16(uint32_t a, uint32_t b);                       // (defined with the packed fake-quant below)
MSQ_D uint32_t pk_lift(uint32_t a, uint32_t q);
template <int DT> MSQ_D uint32_t mx_e2m1_hw_pair(uint32_t w, float sc, uint32_t q) {
    if (DT == 1) {
        const lp_h2_t z = {(_Float16)0.f, (_Float16)0.f};
        const uint32_t x = __builtin_bit_cast(uint32_t, (lp_h2_t)(__builtin_bit_cast(lp_h2_t, w) + z));      // -0 -> +0
        const uint32_t y = pk_lift(x & 0x7FFF7FFFu, q) | (x & 0x80008000u);                                    // lifted, sticky bit set, sign back
        return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(__builtin_amdgcn_cvt_scalef32_pk_fp4_f16(y, __builtin_bit_cast(lp_h2_t, y), sc, 0), sc, 0));
    }
    const uint32_t a = w & 0x7FFF7FFFu;
    const uint32_t sg = (pk_add_u16(a, 0x7FFF7FFFu) & w) & 0x80008000u;                                        // the sign of the non-zero values: -0 -> +0
    const uint32_t y = pk_lift(a, q) | sg;
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(__builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(y, __builtin_bit_cast(lp_b2_t, y), sc, 0), sc, 0));
}
// largest magnitude of the two T values of a dword as T bits (non-negative T values order like their bit patterns; a NaN is > Inf's)
typedef unsigned short lp_us2_t __attribute__((ext_vector_type(2)));
template <int DT> MSQ_D uint32_t pk_absmax(uint32_t acc, uint32_t w) {    // v_and + v_pk_max_u16
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(lp_us2_t, acc), __builtin_bit_cast(lp_us2_t, w & 0x7FFF7FFFu)));
}
template <int DT> MSQ_D float absmax_to_float(uint32_t bits16) {          // T magnitude bits -> float (NaN stays NaN)
    if (DT == 1) return (float)__builtin_bit_cast(_Float16, (uint16_t)bits16);
    return u2f(bits16 << 16);
}
MSQ_D bool fmt_is_e4m3(const Fmt& f) { return f.kind == 0 && f.ebits == 4 && f.mbits == 5; }
// element formats the convert path of the MX kernels takes: 1 = e4m3, 3 = e2m1 (MX-FP4: round 6), 0 = none
MSQ_D int mx_hw_kind(const Fmt& f) { return fmt_is_e4m3(f) ? 1 : ((f.kind == 0 && f.ebits == 2 && f.mbits == 3) ? 3 : 0); }
template <int DT> MSQ_D uint32_t pk_excepted(int k);                     // (defined with the packed fake-quant below)
// The ONE magnitude on which the reference's in-dtype arithmetic differs from exact half-away rounding inside the fast path's bounds:
// the largest T value under the tie between zero and the smallest e4m3 subnormal, |x| = pred_T(2^(se - 10)).  There m = |x| 2^(9 - se)
// = 1/2 - ulp, and `floor(m + 0.5)` (elemwise_ops.py:59-62) computes m + 0.5 IN T: 1 - ulp/2 is a tie of T's grid under 1 and rounds
// (to even) to 1.0 -- the element becomes one subnormal step instead of zero (reference-made fixture bf16|tinyh|fp8e4m3_sb4).  Every other
// m + 0.5 is exact in T or does not cross an integer.  The converts know nothing of it: a wave that holds such a value takes
// mx_elem_fast instead (one value per block scale and sign: practically never).  T bits of that magnitude:
template <int DT> MSQ_D uint32_t e4m3_quirk_bits(int sei) {
    const int k = sei - 10;                                               // the tie is 2^k
    if (DT == 1) return ((k >= -14) ? (uint32_t)(k + 15) << 10 : 1u << (k + 24)) - 1u;
    return ((uint32_t)(k + 127) << 7) - 1u;
}
template <int DT> MSQ_D uint32_t pk_min_xor(uint32_t acc, uint32_t w, uint32_t cc) {     // min over halfwords of (|w| ^ cc): 0 = that magnitude is present
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(lp_us2_t, acc), __builtin_bit_cast(lp_us2_t, (w & 0x7FFF7FFFu) ^ cc)));
}

// Block quantities of the e4m3 convert path from the T bits of the block maximum, without mx_block_setup's general arithmetic (~150
// instructions a block; the vectorised kernels run it per LANE: it was most of their instruction count).  ok <=> mx_block_setup would
// give fast && pow2den with status 0 for this maximum (round to nearest, e4m3):
//   se0 = floor(R(log2 mx)) (the table rule), se = se0 - emax (an exact small integer in T), |se| inside the scale range;
//   sc = 2^se exact in T and normal in float32; den = R(sc + 1e-6) == sc (evaluated, not assumed); the three range bounds of
//   MxBlk::fast with e4m3's min_exp = -6, mbits = 5, emax = 8: se - 9 >= TMINE, se + 10 <= TMAXE, se - 1 >= TMINE + 11; mx finite, != 0.
// flush_fp32_subnorms never applies (se0 > -127 inside these bounds).  A wave in which any lane says no runs the general set-up.
// (e2m1, round 6: the same with its min_exp = 0, mbits = 3, emax = 2 -- the bounds below are written on the format's parameters)
template <int DT> MSQ_D bool mx_setup_e4m3_lean(uint32_t mb, const MxLowpArgs& A, const uint8_t* tab, float& sc, float& bound, int& sei) {
    constexpr int TMINE = (DT == 1) ? -24 : -133, TMAXE = (DT == 1) ? 15 : 127;
    constexpr uint32_t INFB = (DT == 1) ? 0x7C00u : 0x7F80u;
    const float mx = absmax_to_float<DT>(mb);
    const int se = (tab ? floor_log2_tab<DT>(mx, tab) : floor_log2_fast<DT>(mx)) - A.f.emax;
    const int semax = (1 << (A.scale_bits - 1)) - 1;
    const int min_exp = 2 - (1 << (A.f.ebits - 1));
    int lo = TMINE + 12; lo = lo < TMINE - min_exp - 2 + A.f.mbits ? TMINE - min_exp - 2 + A.f.mbits : lo; lo = lo < -semax ? -semax : lo; lo = lo < -126 ? -126 : lo;
    int hi = TMAXE - A.f.emax - 2; hi = hi > semax ? semax : hi;
    const int sc_e = se < lo ? lo : (se > hi ? hi : se);                 // (keeps the bit pattern below a normal power of two whatever se is)
    sc = u2f((uint32_t)(sc_e + 127) << 23);
    bound = A.f.max_norm * sc;
    sei = sc_e;
    const float den = Rr<DT>(sc + 1e-6f);                                // mx_ops.py:444
    return mb != 0u && mb < INFB && se >= lo && se <= hi && den == sc;
}

template <int BS, int DT>
__global__ void __launch_bounds__(256)
k_mx_lowp(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, MxLowpArgs A) {
    const int64_t total = A.pre * A.nblk * A.post;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t q = t % A.post;
    const int64_t nb = (t / A.post) % A.nblk;
    const int64_t p = t / (A.post * A.nblk);
    const int64_t a0 = nb * BS;
    const int64_t base = (p * A.axis_len + a0) * A.post + q;
    float a[BS];
#pragma unroll
    for (int b = 0; b < BS; ++b)
        a[b] = (a0 + b < A.axis_len) ? ld16<DT>(in, base + (int64_t)b * A.post) : 0.f;         // zero padding (_reshape_to_blocks)
    float mx = 0.f;
    bool nan = false;
#pragma unroll
    for (int b = 0; b < BS; ++b) { const float v = __builtin_fabsf(a[b]); nan |= (v != v); mx = v > mx ? v : mx; }
    if (nan) mx = u2f(0x7FC00000u);
    const MxBlk<DT> B = mx_block_setup<DT>(mx, A);
#pragma unroll
    for (int b = 0; b < BS; ++b)
        if (a0 + b < A.axis_len) st16<DT>(out, base + (int64_t)b * A.post, mx_elem<DT>(a[b], B, A));
    if (B.status && A.status) atomicOr(A.status, B.status);
}

// Contiguous axis (post == 1, axis_len % BS == 0, 16-byte aligned): BS / 8 lanes per block, 8 values = one 16-byte access per lane;
// the block maximum crosses the lanes of a block with xor-shuffles (the lanes of a block are neighbours inside a wave).
template <int BS, int DT>
__global__ void __launch_bounds__(256)
k_mx_lowp_vec(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, MxLowpArgs A, int64_t nchunks) {
    constexpr int LPB = BS / 8;                                          // lanes per block: 1, 2, 4, 8, 16
    __shared__ uint8_t s_tab[256];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < nchunks;
    union { uint4 u; uint16_t h[8]; } v;
    v.u = live ? *reinterpret_cast<const uint4*>(in + t * 8) : make_uint4(0, 0, 0, 0);
    // e4m3, round to nearest (kernel-uniform): one exponent look-up per lane -- computed, no table, no barrier in front of the data
    const int hk = (A.rmode == 0) ? mx_hw_kind(A.f) : 0;
    const bool e4 = hk == 1, e2 = hk == 3;
    const uint8_t* tab = hk ? nullptr : s_tab;
    if (!hk) { floor_log2_tab_init<DT>(s_tab); __syncthreads(); }
    // block maximum on the T bit patterns (non-negative T values order like their bits; a NaN's magnitude bits exceed Inf's, so a NaN
    // element IS the maximum and stays one): four packed ops per lane instead of eight converts + compares
    uint32_t am = pk_absmax<DT>(pk_absmax<DT>(pk_absmax<DT>(pk_absmax<DT>(0u, v.u.x), v.u.y), v.u.z), v.u.w);
    uint32_t mb = (am & 0xFFFFu) > (am >> 16) ? (am & 0xFFFFu) : (am >> 16);
#pragma unroll
    for (int o = 1; o < LPB; o <<= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)mb, o, 64);
        mb = other > mb ? other : mb;
    }
    if (e4) {
        float sc, bound; int sei;
        bool hw = mx_setup_e4m3_lean<DT>(mb, A, tab, sc, bound, sei);
        const uint32_t qb = e4m3_quirk_bits<DT>(sei), cc = qb | (qb << 16);
        const uint32_t tq = pk_min_xor<DT>(pk_min_xor<DT>(pk_min_xor<DT>(pk_min_xor<DT>(0xFFFFFFFFu, v.u.x, cc), v.u.y, cc), v.u.z, cc), v.u.w, cc);
        hw = hw && (tq & 0xFFFFu) != 0u && (tq >> 16) != 0u;
        if (__builtin_amdgcn_ballot_w64(!hw) == 0) {                     // the whole wave: the scaled converts
            v.u.x = mx_e4m3_hw_pair<DT>(v.u.x, sc, bound); v.u.y = mx_e4m3_hw_pair<DT>(v.u.y, sc, bound);
            v.u.z = mx_e4m3_hw_pair<DT>(v.u.z, sc, bound); v.u.w = mx_e4m3_hw_pair<DT>(v.u.w, sc, bound);
            if (live) *reinterpret_cast<uint4*>(out + t * 8) = v.u;
            return;
        }
    }
    if (e2) {                                                            // MX-FP4: the excepted magnitude (first tie 2^(se - 2)) is lifted, not looked for
        float sc, bound; int sei;
        const bool hw = mx_setup_e4m3_lean<DT>(mb, A, tab, sc, bound, sei);
        const uint32_t q = pk_excepted<DT>(sei - 2);
        if (__builtin_amdgcn_ballot_w64(!hw) == 0) {
            v.u.x = mx_e2m1_hw_pair<DT>(v.u.x, sc, q); v.u.y = mx_e2m1_hw_pair<DT>(v.u.y, sc, q);
            v.u.z = mx_e2m1_hw_pair<DT>(v.u.z, sc, q); v.u.w = mx_e2m1_hw_pair<DT>(v.u.w, sc, q);
            if (live) *reinterpret_cast<uint4*>(out + t * 8) = v.u;
            return;
        }
    }
    const float mx = absmax_to_float<DT>(mb);
    MxBlk<DT> B = mx_block_setup<DT>(mx, A);
    if (__builtin_amdgcn_ballot_w64(!B.fast) == 0) {              // the whole wave: straight-line code
#pragma unroll
        for (int j = 0; j < 8; ++j) st16<DT>(v.h, j, mx_elem_fast<DT>(ld16<DT>(v.h, j), B, tab));
    } else {
        B.fast = false;                                                  // the general path is right for every block
#pragma unroll
        for (int j = 0; j < 8; ++j) st16<DT>(v.h, j, mx_elem<DT>(ld16<DT>(v.h, j), B, A));     // (unrolled: a run-time index would put v.h[] in scratch)
    }
    if (live) *reinterpret_cast<uint4*>(out + t * 8) = v.u;
    if (live && B.status && A.status) atomicOr(A.status, B.status);
}

// Strided axis with an even contiguous extent (K cache: blocks of BS tokens of one channel, post = head_dim): one lane per PAIR of
// neighbouring channels, 4-byte accesses (a wave moves 256 contiguous bytes per token row instead of 128).
template <int BS, int DT>
__global__ void __launch_bounds__(256)
k_mx_lowp_pair(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, MxLowpArgs A) {
    __shared__ uint8_t s_tab[256];
    const int hk = (A.rmode == 0 && (BS % 2) == 0 && (A.axis_len % BS) == 0) ? mx_hw_kind(A.f) : 0;      // kernel-uniform
    const bool e4 = hk == 1, e2 = hk == 3;
    const uint8_t* tab = hk ? nullptr : s_tab;
    if (!hk) { floor_log2_tab_init<DT>(s_tab); __syncthreads(); }
    const int64_t hp = A.post / 2;
    const int64_t total = A.pre * A.nblk * hp;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t q = (t % hp) * 2;
    const int64_t nb = (t / hp) % A.nblk;
    const int64_t p = t / (hp * A.nblk);
    const int64_t a0 = nb * BS;
    const int64_t base = (p * A.axis_len + a0) * A.post + q;
    uint32_t raw[BS];
    uint32_t am = 0u;                                                    // magnitude maxima of the two channels as T bits
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        raw[b] = (a0 + b < A.axis_len) ? *reinterpret_cast<const uint32_t*>(in + base + (int64_t)b * A.post) : 0u;
        am = pk_absmax<DT>(am, raw[b]);
    }
    if (e4) {
        float sc0, sc1, bd0, bd1; int se0, se1;
        bool hw = mx_setup_e4m3_lean<DT>(am & 0xFFFFu, A, tab, sc0, bd0, se0);
        hw = mx_setup_e4m3_lean<DT>(am >> 16, A, tab, sc1, bd1, se1) && hw;
        const uint32_t cc = e4m3_quirk_bits<DT>(se0) | (e4m3_quirk_bits<DT>(se1) << 16);
        uint32_t tq = 0xFFFFFFFFu;
#pragma unroll
        for (int b = 0; b < BS; ++b) tq = pk_min_xor<DT>(tq, raw[b], cc);
        hw = hw && (tq & 0xFFFFu) != 0u && (tq >> 16) != 0u;
        if (__builtin_amdgcn_ballot_w64(!hw) == 0) {
            // the scaled converts take two values of ONE block: pair the tokens b, b + 1 of each channel (v_perm), convert, pair back
#pragma unroll
            for (int b = 0; b < BS; b += 2) {
                const uint32_t p0 = __builtin_amdgcn_perm(raw[b + 1], raw[b], 0x05040100u), p1 = __builtin_amdgcn_perm(raw[b + 1], raw[b], 0x07060302u);
                const uint32_t q0 = mx_e4m3_hw_pair<DT>(p0, sc0, bd0), q1 = mx_e4m3_hw_pair<DT>(p1, sc1, bd1);
                *reinterpret_cast<uint32_t*>(out + base + (int64_t)b * A.post) = __builtin_amdgcn_perm(q1, q0, 0x05040100u);
                *reinterpret_cast<uint32_t*>(out + base + (int64_t)(b + 1) * A.post) = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
            }
            return;
        }
    }
    if (e2) {
        float sc0, sc1, bd0, bd1; int se0, se1;
        bool hw = mx_setup_e4m3_lean<DT>(am & 0xFFFFu, A, tab, sc0, bd0, se0);
        hw = mx_setup_e4m3_lean<DT>(am >> 16, A, tab, sc1, bd1, se1) && hw;
        const uint32_t x0 = pk_excepted<DT>(se0 - 2), x1 = pk_excepted<DT>(se1 - 2);
        if (__builtin_amdgcn_ballot_w64(!hw) == 0) {
#pragma unroll
            for (int b = 0; b < BS; b += 2) {
                const uint32_t p0 = __builtin_amdgcn_perm(raw[b + 1], raw[b], 0x05040100u), p1 = __builtin_amdgcn_perm(raw[b + 1], raw[b], 0x07060302u);
                const uint32_t q0 = mx_e2m1_hw_pair<DT>(p0, sc0, x0), q1 = mx_e2m1_hw_pair<DT>(p1, sc1, x1);
                *reinterpret_cast<uint32_t*>(out + base + (int64_t)b * A.post) = __builtin_amdgcn_perm(q1, q0, 0x05040100u);
                *reinterpret_cast<uint32_t*>(out + base + (int64_t)(b + 1) * A.post) = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
            }
            return;
        }
    }
    const float m0 = absmax_to_float<DT>(am & 0xFFFFu), m1 = absmax_to_float<DT>(am >> 16);     // (a NaN element is the maximum: NaN stays NaN)
    MxBlk<DT> B0 = mx_block_setup<DT>(m0, A), B1 = mx_block_setup<DT>(m1, A);
    float a0v[BS], a1v[BS];
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        union { uint32_t u; uint16_t h[2]; } w;
        w.u = raw[b];
        a0v[b] = ld16<DT>(w.h, 0); a1v[b] = ld16<DT>(w.h, 1);
    }
    if (__builtin_amdgcn_ballot_w64(!(B0.fast && B1.fast)) == 0) {
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            if (a0 + b >= A.axis_len) continue;
            union { uint32_t u; uint16_t h[2]; } w;
            st16<DT>(w.h, 0, mx_elem_fast<DT>(a0v[b], B0, tab));
            st16<DT>(w.h, 1, mx_elem_fast<DT>(a1v[b], B1, tab));
            *reinterpret_cast<uint32_t*>(out + base + (int64_t)b * A.post) = w.u;
        }
    } else {
        B0.fast = false; B1.fast = false;
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            if (a0 + b >= A.axis_len) continue;
            union { uint32_t u; uint16_t h[2]; } w;
            st16<DT>(w.h, 0, mx_elem<DT>(a0v[b], B0, A));
            st16<DT>(w.h, 1, mx_elem<DT>(a1v[b], B1, A));
            *reinterpret_cast<uint32_t*>(out + base + (int64_t)b * A.post) = w.u;
        }
    }
    const int st = B0.status | B1.status;
    if (st && A.status) atomicOr(A.status, st);
}

// The same, blocks of 32 along the strided axis with no ragged tail (the K cache of a prefill), cut FOUR ways: a thread block owns one
// MX block row (32 tokens x 64 channel pairs), wave w its tokens 8 w ... 8 w + 7; the packed magnitude maxima meet in LDS (one dword per
// lane, one barrier).  k_mx_lowp_pair holds 32 tokens per lane: 127 VGPRs, 4096 waves for a 7B layer cache = ONE round of four waves per
// SIMD, in which the whole chip reads, then computes, then writes (23 us for 67 MB); eight values per lane give four times the waves at
// half the registers, and the rounds overlap each other's reads and writes.
template <int DT>
__global__ void __launch_bounds__(256)
k_mx_lowp_pair4(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, MxLowpArgs A, int chunks) {
    constexpr int BS = 32, TW = 8;
    __shared__ uint8_t s_tab[256];
    __shared__ uint32_t s_am[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t bi = blockIdx.x;
    const int chunk = (int)(bi % chunks);
    const int64_t r = bi / chunks, nb = r % A.nblk, p = r / A.nblk;
    const int64_t hp = A.post / 2, cp = (int64_t)chunk * 64 + lane;
    const bool live = cp < hp;
    const int64_t base = (p * A.axis_len + nb * BS + w * TW) * A.post + 2 * (live ? cp : 0);
    uint32_t raw[TW];
    uint32_t am = 0u;
#pragma unroll
    for (int b = 0; b < TW; ++b) {
        raw[b] = *reinterpret_cast<const uint32_t*>(in + base + (int64_t)b * A.post);
        am = pk_absmax<DT>(am, raw[b]);
    }
    const int hk = (A.rmode == 0) ? mx_hw_kind(A.f) : 0;                                  // kernel-uniform
    const bool e4 = hk == 1, e2 = hk == 3;
    const uint8_t* tab = hk ? nullptr : s_tab;
    if (!hk) floor_log2_tab_init<DT>(s_tab);
    s_am[w][lane] = am;
    __syncthreads();
    am = pk_absmax<DT>(pk_absmax<DT>(pk_absmax<DT>(s_am[0][lane], s_am[1][lane]), s_am[2][lane]), s_am[3][lane]);
    if (e4) {
        float sc0, sc1, bd0, bd1; int se0, se1;
        bool hw = mx_setup_e4m3_lean<DT>(am & 0xFFFFu, A, tab, sc0, bd0, se0);
        hw = mx_setup_e4m3_lean<DT>(am >> 16, A, tab, sc1, bd1, se1) && hw;
        const uint32_t cc = e4m3_quirk_bits<DT>(se0) | (e4m3_quirk_bits<DT>(se1) << 16);
        uint32_t tq = 0xFFFFFFFFu;
#pragma unroll
        for (int b = 0; b < TW; ++b) tq = pk_min_xor<DT>(tq, raw[b], cc);
        hw = (hw && (tq & 0xFFFFu) != 0u && (tq >> 16) != 0u) || !live;
        if (__builtin_amdgcn_ballot_w64(!hw) == 0) {
#pragma unroll
            for (int b = 0; b < TW; b += 2) {
                const uint32_t p0 = __builtin_amdgcn_perm(raw[b + 1], raw[b], 0x05040100u), p1 = __builtin_amdgcn_perm(raw[b + 1], raw[b], 0x07060302u);
                const uint32_t q0 = mx_e4m3_hw_pair<DT>(p0, sc0, bd0), q1 = mx_e4m3_hw_pair<DT>(p1, sc1, bd1);
                if (live) {
                    *reinterpret_cast<uint32_t*>(out + base + (int64_t)b * A.post) = __builtin_amdgcn_perm(q1, q0, 0x05040100u);
                    *reinterpret_cast<uint32_t*>(out + base + (int64_t)(b + 1) * A.post) = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
                }
            }
            return;
        }
    }
    if (e2) {
        float sc0, sc1, bd0, bd1; int se0, se1;
        bool hw = mx_setup_e4m3_lean<DT>(am & 0xFFFFu, A, tab, sc0, bd0, se0);
        hw = (mx_setup_e4m3_lean<DT>(am >> 16, A, tab, sc1, bd1, se1) && hw) || !live;
        const uint32_t x0 = pk_excepted<DT>(se0 - 2), x1 = pk_excepted<DT>(se1 - 2);
        if (__builtin_amdgcn_ballot_w64(!hw) == 0) {
#pragma unroll
            for (int b = 0; b < TW; b += 2) {
                const uint32_t p0 = __builtin_amdgcn_perm(raw[b + 1], raw[b], 0x05040100u), p1 = __builtin_amdgcn_perm(raw[b + 1], raw[b], 0x07060302u);
                const uint32_t q0 = mx_e2m1_hw_pair<DT>(p0, sc0, x0), q1 = mx_e2m1_hw_pair<DT>(p1, sc1, x1);
                if (live) {
                    *reinterpret_cast<uint32_t*>(out + base + (int64_t)b * A.post) = __builtin_amdgcn_perm(q1, q0, 0x05040100u);
                    *reinterpret_cast<uint32_t*>(out + base + (int64_t)(b + 1) * A.post) = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
                }
            }
            return;
        }
    }
    const float m0 = absmax_to_float<DT>(am & 0xFFFFu), m1 = absmax_to_float<DT>(am >> 16);     // (a NaN element is the maximum: NaN stays NaN)
    MxBlk<DT> B0 = mx_block_setup<DT>(m0, A), B1 = mx_block_setup<DT>(m1, A);
    const bool fastw = __builtin_amdgcn_ballot_w64(!(B0.fast && B1.fast)) == 0;
    if (!fastw) { B0.fast = false; B1.fast = false; }
#pragma unroll
    for (int b = 0; b < TW; ++b) {
        union { uint32_t u; uint16_t h[2]; } wi, wo;
        wi.u = raw[b];
        const float x0 = ld16<DT>(wi.h, 0), x1 = ld16<DT>(wi.h, 1);
        if (fastw) { st16<DT>(wo.h, 0, mx_elem_fast<DT>(x0, B0, tab)); st16<DT>(wo.h, 1, mx_elem_fast<DT>(x1, B1, tab)); }
        else { st16<DT>(wo.h, 0, mx_elem<DT>(x0, B0, A)); st16<DT>(wo.h, 1, mx_elem<DT>(x1, B1, A)); }
        if (live) *reinterpret_cast<uint32_t*>(out + base + (int64_t)b * A.post) = wo.u;
    }
    const int st = B0.status | B1.status;
    if (live && w == 0 && st && A.status) atomicOr(A.status, st);
}


// ---------------------------------------------------------------------------------------------------------------------------
// The MicroScopiQ fake-quant on PACKED half values (round 6): two T values of one block per dword, the statistics in float32
// with a proof of the rounding, the element step through the gfx950 scaled converts.  k_outlier_lowp above spends ~70 vector
// instructions per element (VALU bound, 0.2 of the HBM rate); this form ~20.  Same results, bit for bit -- the goldens
// (tests/golden/outlier_lowp.npz) and the oracle pin it, and tests/test_gpu_a6_lowp_packed.py holds it against the op-by-op
// kernel above on adversarial inputs.  What makes it legal:
//  * mask: x < lo or x > hi on packed values -- fp16: the sign of the correctly rounded differences x - lo, hi - x (v_pk_add_f16;
//    fp16 denormals are kept, so a difference of unequal values never rounds to zero); bf16: saturating differences of the
//    order-preserving integer keys of the bit patterns.  Both are wrong only for a zero against a zero bound of the other sign:
//    blocks whose bound is +-0 (or NaN) are not taken here.
//  * std: the reference's value is R_T((float) double Welford).  Here: sd' = sqrt(sum (|x| - c)^2 / n) in float32 with c = the
//    fp32 mean already summed in ATen's order; |sd' - sd| <= n / 8 + 4 float32 ulps (four fma chains of n / 4, the rounding of each
//    difference, the square root, and the term n (mean - c)^2 that the shifted centre leaves out, negligible once
//    sd' >= n 2^-12 c).  When sd' lies further than that from every rounding boundary of T, R_T(sd') is the reference's value;
//    the lanes where it does not (one block in ~400 for fp16, ~3000 for bf16) run the double two-pass form.
//  * the element step: inside the exponent bounds checked per block every scaling of utils/quant.py:214-258 is exact in T (or
//    is the one correctly rounded product the reference also forms: fp16 outliers, v_pk_mul_f16), and the codec
//    (elemwise_ops.py:84-174, nearest) is "half away from zero on the format's grid, subnormals kept, saturating" except on ONE
//    magnitude per side (pred_T of half the smallest step: DESIGN.md 5.002).  The converts round to nearest even; with the lowest
//    significand bit of the source set no input is a tie and nothing else changes sides (T has >= 3 more significand bits than the
//    grids, and the bounds keep T's own spacing four times finer than the smallest step).  Magnitudes only go through the
//    converts; the sign comes back where the result is not zero (the reference's final `inl + o` turns every -0 into +0).
//    The excepted magnitude sits in the bulk of a weight block (a quarter to a half of the inlier maximum: every other wave holds
//    one), so it is lifted over the tie on its way into the convert (pk_lift) rather than sent elsewhere.
//    A wave with a block outside the bounds takes outlier_block_lowp.
// ---------------------------------------------------------------------------------------------------------------------------
typedef short lp_s2_t __attribute__((ext_vector_type(2)));
MSQ_D uint32_t pk_max_u16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(lp_us2_t, a), __builtin_bit_cast(lp_us2_t, b))); }
MSQ_D uint32_t pk_min_u16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(lp_us2_t, a), __builtin_bit_cast(lp_us2_t, b))); }
MSQ_D uint32_t pk_subsat_u16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(lp_us2_t, a), __builtin_bit_cast(lp_us2_t, b))); }
MSQ_D uint32_t pk_subsat_i16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(lp_s2_t, a), __builtin_bit_cast(lp_s2_t, b))); }
MSQ_D uint32_t pk_add_u16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (lp_us2_t)(__builtin_bit_cast(lp_us2_t, a) + __builtin_bit_cast(lp_us2_t, b))); }
MSQ_D uint32_t pk_mul_u16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (lp_us2_t)(__builtin_bit_cast(lp_us2_t, a) * __builtin_bit_cast(lp_us2_t, b))); }
MSQ_D uint32_t pk_sign_fill(uint32_t a) { const lp_s2_t f = {15, 15}; return __builtin_bit_cast(uint32_t, (lp_s2_t)(__builtin_bit_cast(lp_s2_t, a) >> f)); }   // 0xFFFF where bit 15 is set
MSQ_D uint32_t dup16(uint32_t h) { return h | (h << 16); }                  // h <= 0xFFFF
template <int DT> MSQ_D uint32_t t_bits(float v) {                       // bit pattern of a float that IS a T value
    if (DT == 1) return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)v);
    return f2u(v) >> 16;
}
template <int DT> MSQ_D uint32_t pow2_bits(int k) {                      // T bits of 2^k (fp16: subnormals included, k >= -24)
    if (DT == 1) return (k >= -14) ? (uint32_t)(k + 15) << 10 : 1u << ((k + 24) & 31);
    return (uint32_t)(k + 127) << 7;
}
MSQ_D float pow2f(int k) { return u2f((uint32_t)(k + 127) << 23); }      // -126 <= k <= 127

// quantise-dequantise two T magnitudes of one block on the grid K (1 e4m3, 2 e5m2, 3 e2m1; lowest significand bit set by the caller):
// round(m / sf) on the grid, times sb.  bound = T bits of max_norm * sf (the fp8 converts do not saturate)
#ifndef MSQ_LOWP_FP4_CLAMP
#define MSQ_LOWP_FP4_CLAMP 0          // the e2m1 converts saturate by themselves (as their float32-source forms do in msq_outlier_core.h)
#endif
template <int DT, int K> MSQ_D uint32_t hw_mag_pair(uint32_t m, float sf, float sb, uint32_t bound) {
    if (K != 3 || MSQ_LOWP_FP4_CLAMP) m = pk_min_u16(m, bound);
    // (the converts write one byte / one half of their destination and keep the rest: the source register itself serves -- the part
    //  that is read back is the part just written)
    if (DT == 1) {
        const lp_h2_t x = __builtin_bit_cast(lp_h2_t, m);
        if (K == 3) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(__builtin_amdgcn_cvt_scalef32_pk_fp4_f16(m, x, sf, 0), sb, 0));
        lp_v2s_t c = __builtin_bit_cast(lp_v2s_t, m);
        if (K == 1) { c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c, x, sf, false); return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(__builtin_bit_cast(uint32_t, c), sb, false)); }
        c = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(c, x, sf, false);
        return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(__builtin_bit_cast(uint32_t, c), sb, false));
    } else {
        const lp_b2_t x = __builtin_bit_cast(lp_b2_t, m);
        if (K == 3) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(__builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(m, x, sf, 0), sb, 0));
        lp_v2s_t c = __builtin_bit_cast(lp_v2s_t, m);
        if (K == 1) { c = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(c, x, sf, false); return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(__builtin_bit_cast(uint32_t, c), sb, false)); }
        c = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(c, x, sf, false);
        return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(__builtin_bit_cast(uint32_t, c), sb, false));
    }
}

// fp16 outliers: x 2^e_in is often an fp16 SUBNORMAL (e4m3 spans 17 binades: 2^e_out sits near 2^-18 for a weight block), where the
// lowest bit of the half is worth a whole step of the grid and cannot serve as the sticky bit.  The pair goes to float32 (exact), takes
// the bit there, and enters the float32-source convert; the way back is the f16 convert as before.  m: clamped, the excepted magnitude
// already lifted
template <int K> MSQ_D uint32_t hw_mag_pair_f32src(uint32_t m, float s) {
    const lp_h2_t h = __builtin_bit_cast(lp_h2_t, m);
    const float x0 = u2f(f2u((float)h[0]) | 1u), x1 = u2f(f2u((float)h[1]) | 1u);
    if (K == 3) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(__builtin_amdgcn_cvt_scalef32_pk_fp4_f32(m, x0, x1, s, 0), s, 0));
    lp_v2s_t c = __builtin_bit_cast(lp_v2s_t, m);
    if (K == 1) { c = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(c, x0, x1, s, false); return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(__builtin_bit_cast(uint32_t, c), s, false)); }
    c = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(c, x0, x1, s, false);
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(__builtin_bit_cast(uint32_t, c), s, false));
}

// The excepted magnitude of a side, q = pred_T(first tie) (its lowest bit is set), becomes "just above the tie" on its way into the
// convert: |x| == q -> q + 2 ulps.  (|x| == q - 1 -> q, which the converts round down like q - 1; everything else: |x| | 1.)
// q = 0xFFFF where the side has none.
MSQ_D uint32_t pk_lift(uint32_t a, uint32_t q) { return pk_add_u16(a, pk_subsat_u16(0x00020002u, a ^ q)) | 0x00010001u; }
// T bits (both halves) of pred_T(2^k), or 0xFFFF: an fp16 SUBNORMAL pred (k <= -14) is further below the tie than half an ulp of
// 1/2 -- its spacing is 2^-24, not 2^(k - 11) --, R(m + 0.5) stays below 1 and there is no exception
template <int DT> MSQ_D uint32_t pk_excepted(int k) {
    if (DT == 1) return (k >= -13) ? dup16(((uint32_t)(k + 15) << 10) - 1u) : 0xFFFFFFFFu;
    return dup16(((uint32_t)(k + 127) << 7) - 1u);
}

// The element loop of one block.  e_in / e_out: the block's two scale exponents; tie_i / tie_o: exponent of the first tie (half the
// smallest step) of the inlier grid in x's domain, of the outlier grid in the domain of x 2^e_in.
template <int BS, int DT, int KI, int KO>
MSQ_D void pk_codec_loop(const uint32_t (&pw)[BS / 2], const uint32_t (&mk)[BS / 2], uint32_t (&res)[BS / 2], const OutlierArgs& A, int ei, int eo,
                         int tie_i, int tie_o) {
    const float s_in = pow2f(ei);
    // fp16: outliers are rounded into the domain of x 2^e_in first (:216, v_pk_mul_f16 -- a real rounding where the product is an
    // fp16 subnormal), converted with 2^e_out, and leave through R(. / 2^e_in) (:258).  bf16: x 2^e_in is exact (or far below the first
    // tie), so ONE scale 2^(e_out - e_in) takes x to the outlier grid and back
    const float s_out = (DT == 1) ? pow2f(eo) : pow2f(eo - ei);
    const uint32_t b_out = dup16(t_bits<DT>(A.fo.max_norm * s_out));
    const uint32_t q_in = pk_excepted<DT>(tie_i), q_out = pk_excepted<DT>((DT == 1) ? tie_o : tie_o - ei);
    const uint32_t b_in = (KI == 4) ? 0u : dup16(t_bits<DT>(A.fi.max_norm * s_in));
    // int2 inliers: |x| >= step / 2 ? step : 0 -- and the excepted magnitude, one ulp below step / 2, joins the upper side
    const uint32_t h_in = (KI == 4) ? dup16(pow2_bits<DT>(ei - 1)) - ((q_in != 0xFFFFFFFFu) ? 0x00020002u : 0x00010001u) : 0u;
    const uint32_t st_in = (KI == 4) ? dup16(pow2_bits<DT>(ei)) : 0u;
    const uint32_t m_dn = dup16((uint32_t)(ei + 15) << 10), m_up = dup16((uint32_t)(15 - ei) << 10);       // fp16 only: |e_in| <= 14
#pragma unroll
    for (int j = 0; j < BS / 2; ++j) {
        const uint32_t w = pw[j], a = w & 0x7FFF7FFFu, m = mk[j];
        uint32_t ri, ro;
        if (DT == 1) {
            if (KI == 4) ri = pk_mul_u16(pk_min_u16(pk_subsat_u16(a, h_in), 0x00010001u), st_in);
            else ri = hw_mag_pair<DT, KI>(pk_lift(a, q_in), s_in, s_in, b_in);
            const uint32_t t1 = __builtin_bit_cast(uint32_t, (lp_h2_t)(__builtin_bit_cast(lp_h2_t, a) * __builtin_bit_cast(lp_h2_t, m_dn)));
            uint32_t q;
            if (KO == 3) {
                // e2m1 spans four binades: with the first tie at 2^-22 or above (checked per block) the half's own spacing is fine enough
                q = hw_mag_pair<DT, KO>(pk_lift(t1, q_out), s_out, s_out, b_out);
            } else {
                const uint32_t tl = pk_min_u16(pk_add_u16(t1, pk_subsat_u16(0x00020002u, t1 ^ q_out)), b_out);
                q = hw_mag_pair_f32src<KO>(tl, s_out);
            }
            ro = __builtin_bit_cast(uint32_t, (lp_h2_t)(__builtin_bit_cast(lp_h2_t, q) * __builtin_bit_cast(lp_h2_t, m_up)));
        } else {
            // one lift serves both converts: each element uses the result of its own side only
            const uint32_t in = pk_lift(a, (q_out & m) | (q_in & ~m));
            if (KI == 4) ri = pk_mul_u16(pk_min_u16(pk_subsat_u16(a, h_in), 0x00010001u), st_in);
            else ri = hw_mag_pair<DT, KI>(in, s_in, s_in, b_in);
            ro = hw_mag_pair<DT, KO>(in, s_out, s_out, b_out);
        }
        const uint32_t rm = (ro & m) | (ri & ~m);
        if (DT == 1) {
            // the sign: rm * (+-1) + 0 -- a zero comes out as +0 whatever its factor's sign (v_pk_fma_f16; exact)
            const lp_h2_t z = {(_Float16)0.f, (_Float16)0.f};
            res[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(lp_h2_t, rm), __builtin_bit_cast(lp_h2_t, (w & 0x80008000u) | 0x3C003C00u), z));
        } else {
            // the sign where the magnitude is not zero (rm <= 0x7FFF: rm + 0x7FFF has bit 15 set iff rm != 0)
            res[j] = ((pk_add_u16(rm, 0x7FFF7FFFu) & w) & 0x80008000u) | rm;
        }
    }
}

// floor(R(log2 max)) - emax and the clamp of utils/quant.py:207-211 / :237-242 on integers: the exponent is an integer in
// [-133, 127] and emax <= 15, so the reference's float chain (every step rounded to T) is exact.  mb: T bits of the block maximum
// (finite); a zero maximum: fp16 R(2^-126) = 0 -> log2 = -inf -> the lower clamp; bf16 keeps 2^-126.  false: the exponent exceeds
// the scale range (the reference stores NaN)
template <int DT> MSQ_D bool scale_exp_int(uint32_t mb, float mx, int emax, int sb, const uint8_t* tab, int& se) {
    const int lim = (1 << (sb - 1)) - 1, neg = (-lim < -20) ? -20 : -lim;
    int e = floor_log2_tab<DT>(mx, tab) - emax;
    e = (mb == 0u) ? ((DT == 1) ? -100000 : -126 - emax) : e;
    se = (e < -lim) ? neg : e;
    return e <= lim;
}

// one block: pw[] = the BS values, two per dword in axis order.  Returns true when res[] / mk[] (0xFFFF per outlier) / the
// two exponents are the block's result; false: nothing is, the caller runs outlier_block_lowp.  Straight-line apart from the
// (divergent, rare) double-precision std.
#ifdef MSQ_LOWP_WHY
__device__ unsigned long long g_lowp_why[16];
extern "C" void msq_lowp_why_(unsigned long long* out16, int reset) {
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lowp_why), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_lowp_why), z, sizeof(z)); }
}
#define MSQ_WHY(cond, slot) do { if (!(cond)) atomicAdd(&g_lowp_why[slot], 1ull); } while (0)
#else
#define MSQ_WHY(cond, slot) do { } while (0)
#endif
template <int BS, int DT>
MSQ_D bool outlier_block_pk(const uint32_t (&pw)[BS / 2], uint32_t (&res)[BS / 2], uint32_t (&mk)[BS / 2], float& se_in_o, float& se_out_o,
                            const OutlierArgs& A, int order, const uint8_t* l2tab, int kin, int kout) {
    bool ok = true;
    float lo, hi;
    {
        float ab[BS];
#pragma unroll
        for (int j = 0; j < BS / 2; ++j) {
            const uint32_t a = pw[j] & 0x7FFF7FFFu;
            if (DT == 1) { const lp_h2_t h = __builtin_bit_cast(lp_h2_t, a); ab[2 * j] = (float)h[0]; ab[2 * j + 1] = (float)h[1]; }
            else { ab[2 * j] = u2f(a << 16); ab[2 * j + 1] = u2f(a & 0xFFFF0000u); }
        }
        float s;
        if (order == 1) s = sum_inner8<BS>(ab);
        else if (order == 2) s = sum_ilp4<BS>(ab);
        else s = sum_cascade<BS>(ab);
        const float c = s * (1.0f / (float)BS);                         // exact: BS is a power of two
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
        for (int b = 0; b < BS; b += 4) {
            const float d0 = ab[b] - c, d1 = ab[b + 1] - c, d2 = ab[b + 2] - c, d3 = ab[b + 3] - c;
            q0 = __builtin_fmaf(d0, d0, q0); q1 = __builtin_fmaf(d1, d1, q1); q2 = __builtin_fmaf(d2, d2, q2); q3 = __builtin_fmaf(d3, d3, q3);
        }
        const float var = ((q0 + q1) + (q2 + q3)) * (1.0f / (float)BS);
        float sdp = __builtin_amdgcn_sqrtf(var);                        // v_sqrt_f32: 1 ulp (inside the margin below)
        constexpr int DROP = (DT == 1) ? 13 : 16;
        // |sd' - (float) sd_double| <= (BS / 8 + 2.5) [four fma chains of BS / 4, two adds, the rounded differences, half of it through the root]
        // + 1 [v_sqrt_f32] + 0.5 [the double's own rounding to float] ulps; one more for good measure
        constexpr uint32_t MARG = BS / 8 + 5, HALF = 1u << (DROP - 1);
        const uint32_t fr = f2u(sdp) & ((1u << DROP) - 1u);
        const uint32_t dist = fr > HALF ? fr - HALF : HALF - fr;
        const float sdmin = (DT == 1) ? 6.103515625e-05f : 8.8817841970012523e-16f, sdmax = (DT == 1) ? 60000.f : 1.125899906842624e15f;
        const bool std_ok = dist > MARG && sdp >= sdmin && sdp <= sdmax && sdp >= (float)BS * 0.000244140625f * c;
#ifndef MSQ_LOWP_NOSTD64
        if (__builtin_amdgcn_ballot_w64(!std_ok) != 0ull) {
            if (!std_ok) sdp = std_twopass_checked<BS>(ab, 0);
        }
#endif
        const float mean = Rr<DT>(c);                                   // utils/quant.py:477
        const float sd = Rr<DT>(sdp);                                   // :478
        const float ks = Rr<DT>(A.k * sd);
        lo = Rr<DT>(mean - ks); hi = Rr<DT>(mean + ks);                 // :489-490
    }
    ok = ok && lo == lo && hi == hi;
    MSQ_WHY(lo == lo && hi == hi, 0);
    // ---- mask and the two masked maxima, on the bit patterns
    uint32_t mi = 0u, mo = 0u;
    {
        // a zero bound against a zero of the other sign: x < +-0 and x > +-0 are false for both zeros.  fp16: (-0) - (+0) = -0 would say
        // "below" -- the lower bound becomes -0 (x - (-0) = x + 0 >= +0 for both zeros), the upper one +0; bf16: the key of -0 is -1, of +0
        // is 0 -- "below zero" is key < -1, "above zero" is key > 0
        const uint32_t lob = dup16((lo == 0.f) ? 0x8000u : t_bits<DT>(lo)), hib = dup16((hi == 0.f) ? 0u : t_bits<DT>(hi));
        const uint32_t lok = lob ^ (pk_sign_fill(lob) & 0x7FFF7FFFu), hik = hib ^ (pk_sign_fill(hib) & 0x7FFF7FFFu);
#pragma unroll
        for (int j = 0; j < BS / 2; ++j) {
            const uint32_t w = pw[j];
            uint32_t sg;
            if (DT == 1) {
                const lp_h2_t x = __builtin_bit_cast(lp_h2_t, w);
                sg = __builtin_bit_cast(uint32_t, (lp_h2_t)(x - __builtin_bit_cast(lp_h2_t, lob))) | __builtin_bit_cast(uint32_t, (lp_h2_t)(__builtin_bit_cast(lp_h2_t, hib) - x));
            } else {
                const uint32_t key = w ^ (pk_sign_fill(w) & 0x7FFF7FFFu);
                sg = pk_subsat_i16(key, lok) | pk_subsat_i16(hik, key);
            }
            const uint32_t m = pk_sign_fill(sg);                        // :492
            mk[j] = m;
            mo = pk_max_u16(mo, w & 0x7FFF7FFFu & m);
            mi = pk_max_u16(mi, w & 0x7FFF7FFFu & ~m);
        }
    }
    const uint32_t mib = (mi & 0xFFFFu) > (mi >> 16) ? (mi & 0xFFFFu) : (mi >> 16), mob = (mo & 0xFFFFu) > (mo >> 16) ? (mo & 0xFFFFu) : (mo >> 16);
    constexpr uint32_t INFB = (DT == 1) ? 0x7C00u : 0x7F80u;
    ok = ok && mib < INFB && mob < INFB;
    MSQ_WHY(mib < INFB && mob < INFB, 1);
    // ---- the two shared exponents (:196-211, :229-242)
    // e_in: fp16 multiplies by 2^e_in and 2^-e_in as halves (normal for |e| <= 14).  e_out: the reference's R(2^e_out) must BE 2^e_out --
    // below 2^-24 it is 0 in fp16 and every outlier of the block becomes NaN (x / 0), which is the op-by-op kernel's business
    constexpr int ELO = (DT == 1) ? -14 : -50, EHI = (DT == 1) ? 14 : 50, EOLO = (DT == 1) ? -24 : -50, EOHI = (DT == 1) ? 14 : 50;
    int ei, eo;
    ok = scale_exp_int<DT>(mib, absmax_to_float<DT>(mib), A.fi.emax, A.in_sb, l2tab, ei) && ok;
    se_in_o = (float)ei;
    MSQ_WHY(ok, 2);
    MSQ_WHY(ei >= ELO, 3); MSQ_WHY(ei <= EHI, 4);
    ok = ok && ei >= ELO && ei <= EHI;
    ei = ok ? ei : 0;
    const float mx_out = Rr<DT>(__builtin_ldexpf(absmax_to_float<DT>(mob), ei));   // max over the block of R(o sc_in) (:216): monotone maps
    // a block without (non-zero) outliers never uses the outlier scale: every masked element is +-0 and comes out as +0
    const bool no_out = mob == 0u;
    const bool eo_ok = scale_exp_int<DT>(t_bits<DT>(mx_out), mx_out, A.fo.emax, A.out_sb, l2tab, eo);
    se_out_o = (float)eo;
    MSQ_WHY(eo_ok, 5); MSQ_WHY(no_out || eo >= EOLO, 6); MSQ_WHY(no_out || eo <= EOHI, 7);
    ok = ok && eo_ok && (no_out || (eo >= EOLO && eo <= EOHI));
    eo = (ok && !no_out) ? eo : 0;
    // ---- bounds (see the header): first tie, top binade and the raw tie of either side
    const int mine_i = A.fi.ebits ? 2 - (1 << (A.fi.ebits - 1)) : 0, mine_o = A.fo.ebits ? 2 - (1 << (A.fo.ebits - 1)) : 0;
    const int rawt_i = mine_i - A.fi.mbits + 1, rawt_o = mine_o - A.fo.mbits + 1;
    const int tie_i = ei + rawt_i, tie_o = eo + rawt_o;                 // inliers: in x's domain; outliers: in the domain of x 2^e_in
    if (DT == 1) {
        // inliers enter their convert as halves: fp16's own spacing (2^-24) must be four times finer than the first tie; outliers enter
        // theirs as float32 and need no such bound
        ok = ok && tie_i >= -22 && ei + A.fi.emax + 1 <= 15 && (ei <= 0 || rawt_i >= -13)
                && eo + A.fo.emax + 1 <= 15 && (eo <= 0 || rawt_o >= -13)
                && eo + A.fo.emax >= -22                                // the clamp bound max_norm 2^e_out is an fp16 value (1.75 2^k: k >= -22)
                && (kout != 3 || tie_o >= -22);                         // e2m1 outliers enter as halves too (pk_codec_loop)
        // nothing overflows fp16 on its way into either element domain: |t| < 2^(emax + 2) there (the exponent rule lifts by one at
        // most; a clamped exponent only makes t smaller) -- only e5m2 (emax 15) can get past 65504
        // -- nor does R(2^pe) of the largest private exponent (the rule may lift it to 16: Inf in fp16, NaN results in the reference)
        if (A.fi.emax + 2 > 15) { const float ti = Rr<DT>(__builtin_ldexpf(absmax_to_float<DT>(mib), -ei)); ok = ok && ti < 3.0e38f && (ti == 0.f || floor_log2_tab<DT>(ti, l2tab) <= 15); }
        if (A.fo.emax + 2 > 15) { const float to = Rr<DT>(__builtin_ldexpf(mx_out, -eo)); ok = ok && to < 3.0e38f && (to == 0.f || floor_log2_tab<DT>(to, l2tab) <= 15); }
    }
    if (DT == 1) { MSQ_WHY(tie_i >= -22, 8); MSQ_WHY(ei + A.fi.emax + 1 <= 15, 9); MSQ_WHY(ei <= 0 || rawt_i >= -13, 10);
                   MSQ_WHY(eo + A.fo.emax + 1 <= 15, 11); MSQ_WHY(eo <= 0 || rawt_o >= -13, 12); MSQ_WHY(eo + A.fo.emax >= -22, 13); }
    MSQ_WHY(ok, 14); MSQ_WHY(false, 15);
    const int combo = kin * 4 + kout;
    if (combo == 4 * 4 + 3) pk_codec_loop<BS, DT, 4, 3>(pw, mk, res, A, ei, eo, tie_i, tie_o);          // int2 / fp4: the harness default
    else if (combo == 3 * 4 + 1) pk_codec_loop<BS, DT, 3, 1>(pw, mk, res, A, ei, eo, tie_i, tie_o);     // e2m1 / e4m3
    else if (combo == 3 * 4 + 3) pk_codec_loop<BS, DT, 3, 3>(pw, mk, res, A, ei, eo, tie_i, tie_o);
    else if (combo == 3 * 4 + 2) pk_codec_loop<BS, DT, 3, 2>(pw, mk, res, A, ei, eo, tie_i, tie_o);
    else pk_codec_loop<BS, DT, 1, 1>(pw, mk, res, A, ei, eo, tie_i, tie_o);                             // e4m3 / e4m3 (the launcher admits these five)
    return ok;
}

template <int BS, int DT>
MSQ_D bool outlier_block_pk32(const uint32_t (&pw)[BS / 2], uint32_t (&res)[BS / 2], uint32_t (&mk)[BS / 2], float& se_in_o, float& se_out_o,
                              const OutlierArgs& A, int order, int kin, int kout);      // float32 semantics: below the kernels
// Blocks along the contiguous axis (post == 1, whole blocks, 16-byte aligned): one lane per block; the 64 blocks of a wave are one
// contiguous run that crosses to the lanes through the wave's LDS slice, as in k_outlier_lowp.  A wave with a block outside the packed
// form's bounds writes nothing and puts itself on the list for k_outlier_lowp_list (keeping that path's ~250 registers out of this kernel:
// inlined as a branch it halved the occupancy and cost a factor of three).
template <int BS, int DT, bool F32SEM = false>
__global__ void __launch_bounds__(256)
k_outlier_lowp_pk(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, OutlierArgs A, int kin, int kout, int64_t* __restrict__ ws, uint8_t* __restrict__ marks, int64_t cap) {
    __shared__ uint8_t l2tab[256];
    constexpr int CH = BS / 8, ROWB = BS * 2 + 16;
    __shared__ __attribute__((aligned(16))) char xsm[4 * 64 * ROWB];
    floor_log2_tab_init<DT>(l2tab);
    __syncthreads();
    const int64_t total = A.pre * A.nblk;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave_t0 = t - lane;
    if (wave_t0 >= total) return;
    const bool live = t < total, full = wave_t0 + 64 <= total;
    char* const wsl = xsm + wv * (64 * ROWB);
    const int64_t wave_base = wave_t0 * BS;
    uint32_t pw[BS / 2];
    if (full) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = c * 64 + lane;
            *reinterpret_cast<uint4*>(wsl + (j / CH) * ROWB + (j % CH) * 16) = *reinterpret_cast<const uint4*>(in + wave_base + (int64_t)j * 8);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(wsl + lane * ROWB + c * 16);
            pw[4 * c] = u.x; pw[4 * c + 1] = u.y; pw[4 * c + 2] = u.z; pw[4 * c + 3] = u.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint4 u = live ? *reinterpret_cast<const uint4*>(in + t * BS + c * 8) : make_uint4(0, 0, 0, 0);
            pw[4 * c] = u.x; pw[4 * c + 1] = u.y; pw[4 * c + 2] = u.z; pw[4 * c + 3] = u.w;
        }
    }
    uint32_t res[BS / 2], mk[BS / 2];
    float se_in, se_out;
    bool ok;
    if constexpr (F32SEM) ok = outlier_block_pk32<BS, DT>(pw, res, mk, se_in, se_out, A, 1, kin, kout) || !live;
    else ok = outlier_block_pk<BS, DT>(pw, res, mk, se_in, se_out, A, 1, l2tab, kin, kout) || !live;
    // handed back: every wave leaves its mark; the list takes entries only while it is short (one counter for 32 768 waves that all fail
    // serialises on its cache line: 0.5 ms -- a tensor outside the bounds altogether is found through the marks instead)
    const bool back = __builtin_amdgcn_ballot_w64(!ok) != 0ull;
    if (lane == 0) {
        marks[wave_t0 >> 6] = back ? 1 : 0;
        if (back && *reinterpret_cast<volatile int64_t*>(ws) < cap) ws[8 + atomicAdd(reinterpret_cast<unsigned long long*>(ws), 1ull)] = wave_t0;
    }
    if (back) return;
    if (full) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CH; ++c)
            *reinterpret_cast<uint4*>(wsl + lane * ROWB + c * 16) = make_uint4(res[4 * c], res[4 * c + 1], res[4 * c + 2], res[4 * c + 3]);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = c * 64 + lane;
            *reinterpret_cast<uint4*>(out + wave_base + (int64_t)j * 8) = *reinterpret_cast<const uint4*>(wsl + (j / CH) * ROWB + (j % CH) * 16);
        }
    } else if (live) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            *reinterpret_cast<uint4*>(out + t * BS + c * 8) = make_uint4(res[4 * c], res[4 * c + 1], res[4 * c + 2], res[4 * c + 3]);
    }
    if (!live) return;
    if (A.mask) {
#pragma unroll
        for (int j = 0; j < BS / 2; ++j)
            *reinterpret_cast<uint16_t*>(A.mask + t * BS + 2 * j) = (uint16_t)((mk[j] & 1u) | ((mk[j] >> 8) & 0x100u));
    }
    if (A.e_in) A.e_in[t] = se_in;
    if (A.e_out) A.e_out[t] = se_out;
}

// Blocks along a strided axis (post even, whole blocks, 4-byte aligned): one lane per PAIR of neighbouring columns -- a wave moves
// 256 contiguous bytes per row --, the rows b, b + 1 of each column paired into one dword (v_perm) for the packed arithmetic.
template <int BS, int DT, bool F32SEM = false>
__global__ void __launch_bounds__(256)
k_outlier_lowp_pk2(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, OutlierArgs A, int kin, int kout, int64_t* __restrict__ ws, uint8_t* __restrict__ marks, int64_t cap) {
    __shared__ uint8_t l2tab[256];
    floor_log2_tab_init<DT>(l2tab);
    __syncthreads();
    const int64_t hp = A.post / 2;
    const int64_t total = A.pre * A.nblk * hp;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (t - lane >= total) return;
    const bool live = t < total;
    const int64_t tt = live ? t : t - lane;                              // (the wave's first lane is live)
    const int64_t q = (tt % hp) * 2;
    const int64_t nb = (tt / hp) % A.nblk;
    const int64_t p = tt / (hp * A.nblk);
    const int64_t base = (p * A.axis_len + nb * BS) * A.post + q;
    uint32_t p0[BS / 2], p1[BS / 2];
    // (lanes past the end re-read lane 0's columns and store nothing: no per-load branches)
    const uint16_t* src = in + base;
#pragma unroll
    for (int j = 0; j < BS / 2; ++j) {
        const uint32_t r0 = *reinterpret_cast<const uint32_t*>(src);
        const uint32_t r1 = *reinterpret_cast<const uint32_t*>(src + A.post);
        src += 2 * A.post;
        p0[j] = __builtin_amdgcn_perm(r1, r0, 0x05040100u);
        p1[j] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);
    }
    // ATen's order of the block sums (see outlier_lowp_lane): the same for both columns of a pair (the limits are even)
    const int64_t lim = (A.post >= 8) ? (A.post / 32) * 32 : (A.post / 4) * 4;
    const int order = (q < lim) ? 0 : 2;
    uint32_t r0[BS / 2], r1[BS / 2], m0[BS / 2], m1[BS / 2];
    float si0, so0, si1, so1;
    bool ok0, ok1;
    if constexpr (F32SEM) {
        ok0 = outlier_block_pk32<BS, DT>(p0, r0, m0, si0, so0, A, order, kin, kout);
        ok1 = outlier_block_pk32<BS, DT>(p1, r1, m1, si1, so1, A, order, kin, kout);
    } else {
        ok0 = outlier_block_pk<BS, DT>(p0, r0, m0, si0, so0, A, order, l2tab, kin, kout);
        ok1 = outlier_block_pk<BS, DT>(p1, r1, m1, si1, so1, A, order, l2tab, kin, kout);
    }
    // this wave's columns are the lanes 2 (t - lane) ... + 127 of k_outlier_lowp's numbering: two of its waves (marks and list: see k_outlier_lowp_pk)
    const bool back = __builtin_amdgcn_ballot_w64(!(ok0 && ok1)) != 0ull;
    if (lane == 0) {
        marks[(t - lane) >> 5] = back ? 1 : 0; marks[((t - lane) >> 5) + 1] = back ? 1 : 0;
        if (back && *reinterpret_cast<volatile int64_t*>(ws) < cap) {
            const unsigned long long i = atomicAdd(reinterpret_cast<unsigned long long*>(ws), 2ull);
            ws[8 + i] = 2 * (t - lane); ws[8 + i + 1] = 2 * (t - lane) + 64;
        }
    }
    if (back) return;
    if (!live) return;
    uint16_t* dst = out + base;
#pragma unroll
    for (int j = 0; j < BS / 2; ++j) {
        *reinterpret_cast<uint32_t*>(dst) = __builtin_amdgcn_perm(r1[j], r0[j], 0x05040100u);
        *reinterpret_cast<uint32_t*>(dst + A.post) = __builtin_amdgcn_perm(r1[j], r0[j], 0x07060302u);
        dst += 2 * A.post;
    }
    if (A.mask) {
#pragma unroll
        for (int j = 0; j < BS / 2; ++j) {
            *reinterpret_cast<uint16_t*>(A.mask + base + (int64_t)(2 * j) * A.post) = (uint16_t)((m0[j] & 1u) | ((m1[j] & 1u) << 8));
            *reinterpret_cast<uint16_t*>(A.mask + base + (int64_t)(2 * j + 1) * A.post) = (uint16_t)(((m0[j] >> 16) & 1u) | (((m1[j] >> 16) & 1u) << 8));
        }
    }
    const int64_t eidx = (p * A.nblk + nb) * A.post + q;
    if (A.e_in) { A.e_in[eidx] = si0; A.e_in[eidx + 1] = si1; }
    if (A.e_out) { A.e_out[eidx] = so0; A.e_out[eidx + 1] = so1; }
}


// ===========================================================================================================================
// The same packed machinery for FLOAT32 semantics on 16-bit tensors (dtype 1 / 2 of msq_outlier_fakequant: an fp16 / bf16 tensor
// read as float32 values, utils/quant.py:147-266 computed in float32, ONE rounding to the tensor dtype on the way out -- what the
// MicroScopiQ KV cache runs, kvcache.py).  The float32 kernels (msq_outlier_kernels.h) spend ~45 instructions per element on it and one
// lane walks a whole block; here the block's values stay packed as they were loaded.  What differs from the in-dtype form above:
//  * mean, k std, lo, hi are float32 values; the mask compares T values with them, so the bounds are moved to the nearest T value
//    inside (x < lo <=> x < ceil_T(lo), x > hi <=> x > floor_T(hi)).  The std must be ATen's to the last float32 bit for the
//    reference's bounds -- but the mask only needs the two T values: both ends of sd' -+ its error bound give the same pair in all
//    but one block in a few hundred, and those take the double-precision std in a rare branch;
//  * exponents: floor(log2) of a T-valued float is its exponent field (torch.log2's float32 quirk needs a value within 2^-22 of a
//    power of two); 2^e scalings are exact in float32, so x goes to either grid with ONE scale and no intermediate rounding;
//  * no excepted magnitude: float32's `floor(|x| + 0.5)` errs only on pred_f32(half a step), which no 16-bit value is;
//  * the results are grid values times a power of two: at most four significant bits, exact in T inside the exponent bounds that
//    are checked per block -- outside, the block's wave goes on the list and the float32 routine redoes it.
// ===========================================================================================================================
template <int DT> MSQ_D float t_from_bits(uint32_t b) {
    if (DT == 1) return (float)__builtin_bit_cast(_Float16, (uint16_t)b);
    return u2f(b << 16);
}
template <int DT> MSQ_D float t_ceil(float v) {                          // smallest T value >= v (v not NaN)
    const float r = Rr<DT>(v);
    if (!(r < v)) return r;
    uint32_t b = t_bits<DT>(r);
    b = ((b & 0x7FFFu) == 0u) ? 1u : ((b & 0x8000u) ? b - 1u : b + 1u);
    return t_from_bits<DT>(b);
}
template <int DT> MSQ_D float t_floor(float v) {                         // largest T value <= v
    const float r = Rr<DT>(v);
    if (!(r > v)) return r;
    uint32_t b = t_bits<DT>(r);
    b = ((b & 0x7FFFu) == 0u) ? 0x8001u : ((b & 0x8000u) ? b + 1u : b - 1u);
    return t_from_bits<DT>(b);
}
// x < lo or x > hi on two packed T values -> 0xFFFF per outlier (lo, hi: T values; see outlier_block_pk for the zero bounds)
struct PkBounds { uint32_t lob, hib, lok, hik; };
template <int DT> MSQ_D PkBounds pk_bounds(float lo, float hi) {
    PkBounds B;
    B.lob = dup16((lo == 0.f) ? 0x8000u : t_bits<DT>(lo)); B.hib = dup16((hi == 0.f) ? 0u : t_bits<DT>(hi));
    B.lok = B.lob ^ (pk_sign_fill(B.lob) & 0x7FFF7FFFu); B.hik = B.hib ^ (pk_sign_fill(B.hib) & 0x7FFF7FFFu);
    return B;
}
template <int DT> MSQ_D uint32_t pk_mask(uint32_t w, const PkBounds& B) {
    uint32_t sg;
    if (DT == 1) {
        const lp_h2_t x = __builtin_bit_cast(lp_h2_t, w);
        sg = __builtin_bit_cast(uint32_t, (lp_h2_t)(x - __builtin_bit_cast(lp_h2_t, B.lob))) | __builtin_bit_cast(uint32_t, (lp_h2_t)(__builtin_bit_cast(lp_h2_t, B.hib) - x));
    } else {
        const uint32_t key = w ^ (pk_sign_fill(w) & 0x7FFF7FFFu);
        sg = pk_subsat_i16(key, B.lok) | pk_subsat_i16(B.hik, key);
    }
    return pk_sign_fill(sg);
}
template <int BS, int DT, int KI, int KO>
MSQ_D void pk_codec_loop32(const uint32_t (&pw)[BS / 2], const uint32_t (&mk)[BS / 2], uint32_t (&res)[BS / 2], const OutlierArgs& A, int ei, int eoi) {
    const float s_in = pow2f(ei), s_out = pow2f(eoi);
    const uint32_t b_in = (KI == 4) ? 0u : dup16(t_bits<DT>(A.fi.max_norm * s_in)), b_out = dup16(t_bits<DT>(A.fo.max_norm * s_out));
    const uint32_t h_in = (KI == 4) ? dup16(pow2_bits<DT>(ei - 1)) - 0x00010001u : 0u, st_in = (KI == 4) ? dup16(pow2_bits<DT>(ei)) : 0u;
#pragma unroll
    for (int j = 0; j < BS / 2; ++j) {
        const uint32_t w = pw[j], a = w & 0x7FFF7FFFu, m = mk[j], in = a | 0x00010001u;
        // fp16, e4m3 / e5m2 grids: their small binades reach under 2^-22, where the half's lowest bit is no sticky bit any more (see
        // hw_mag_pair_f32src): those pairs go through the float32-source convert; e2m1 (four binades) stays on the half-source one
        uint32_t ri, ro;
        if (KI == 4) ri = pk_mul_u16(pk_min_u16(pk_subsat_u16(a, h_in), 0x00010001u), st_in);
        else if (DT == 1 && KI != 3) ri = hw_mag_pair_f32src<KI>(pk_min_u16(a, b_in), s_in);
        else ri = hw_mag_pair<DT, KI>(in, s_in, s_in, b_in);
        if (DT == 1 && KO != 3) ro = hw_mag_pair_f32src<KO>(pk_min_u16(a, b_out), s_out);
        else ro = hw_mag_pair<DT, KO>(in, s_out, s_out, b_out);
        const uint32_t rm = (ro & m) | (ri & ~m);
        if (DT == 1) {
            const lp_h2_t z = {(_Float16)0.f, (_Float16)0.f};
            res[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(lp_h2_t, rm), __builtin_bit_cast(lp_h2_t, (w & 0x80008000u) | 0x3C003C00u), z));
        } else {
            res[j] = ((pk_add_u16(rm, 0x7FFF7FFFu) & w) & 0x80008000u) | rm;
        }
    }
}
// exponent of the block maximum minus emax, clamped as utils/quant.py:207-211 (float32 route: shared_exp_of_max + clamp_scale_exp)
MSQ_D bool scale_exp_f32(float mx, int emax, int sb, int& se) {
    const int lim = (1 << (sb - 1)) - 1, neg = (-lim < -20) ? -20 : -lim;
    const int e = ((mx == 0.f) ? -126 : (int)((f2u(mx) >> 23) & 0xFFu) - 127) - emax;       // + FP32_MIN_NORMAL * (max == 0)
    se = (e < -lim) ? neg : e;
    return e <= lim;
}
template <int BS, int DT>
MSQ_D bool outlier_block_pk32(const uint32_t (&pw)[BS / 2], uint32_t (&res)[BS / 2], uint32_t (&mk)[BS / 2], float& se_in_o, float& se_out_o,
                              const OutlierArgs& A, int order, int kin, int kout) {
    bool ok = true, exact;
    float lo, hi, c;
    {
        float ab[BS];
#pragma unroll
        for (int j = 0; j < BS / 2; ++j) {
            const uint32_t a = pw[j] & 0x7FFF7FFFu;
            if (DT == 1) { const lp_h2_t h = __builtin_bit_cast(lp_h2_t, a); ab[2 * j] = (float)h[0]; ab[2 * j + 1] = (float)h[1]; }
            else { ab[2 * j] = u2f(a << 16); ab[2 * j + 1] = u2f(a & 0xFFFF0000u); }
        }
        float s;
        if (order == 1) s = sum_inner8<BS>(ab);
        else if (order == 2) s = sum_ilp4<BS>(ab);
        else s = sum_cascade<BS>(ab);
        c = s / (float)BS;                                              // the reference's float32 mean (:477)
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
        for (int b = 0; b < BS; b += 4) {
            const float d0 = ab[b] - c, d1 = ab[b + 1] - c, d2 = ab[b + 2] - c, d3 = ab[b + 3] - c;
            q0 = __builtin_fmaf(d0, d0, q0); q1 = __builtin_fmaf(d1, d1, q1); q2 = __builtin_fmaf(d2, d2, q2); q3 = __builtin_fmaf(d3, d3, q3);
        }
        const float sdp = __builtin_amdgcn_sqrtf(((q0 + q1) + (q2 + q3)) * (1.0f / (float)BS));
        constexpr uint32_t EULP = BS / 8 + 5;                            // (the bound derived in outlier_block_pk)
        exact = !(sdp >= 8.8817841970012523e-16f && sdp <= 1.125899906842624e15f && sdp >= (float)BS * 0.000244140625f * c && A.k > 0.f);
        const float ks_a = A.k * u2f(f2u(sdp) - EULP), ks_b = A.k * u2f(f2u(sdp) + EULP);
        lo = t_ceil<DT>(c - ks_b); hi = t_floor<DT>(c + ks_b);          // :489-490 with the larger std ...
        const float lo2 = t_ceil<DT>(c - ks_a), hi2 = t_floor<DT>(c + ks_a);   // ... and with the smaller one: the same T values, or the exact std decides
        exact = exact || !(lo == lo2 && hi == hi2);
    }
    if (__builtin_amdgcn_ballot_w64(exact) != 0ull) {
        if (exact) {
            float ab[BS];
#pragma unroll
            for (int j = 0; j < BS / 2; ++j) {
                const uint32_t a = pw[j] & 0x7FFF7FFFu;
                if (DT == 1) { const lp_h2_t h = __builtin_bit_cast(lp_h2_t, a); ab[2 * j] = (float)h[0]; ab[2 * j + 1] = (float)h[1]; }
                else { ab[2 * j] = u2f(a << 16); ab[2 * j + 1] = u2f(a & 0xFFFF0000u); }
            }
            const float ks = A.k * std_twopass_checked<BS>(ab, 0);      // :478
            const float lof = c - ks, hif = c + ks;
            ok = ok && lof == lof && hif == hif;
            lo = t_ceil<DT>(lof); hi = t_floor<DT>(hif);
        }
    }
    uint32_t mi = 0u, mo = 0u;
    {
        const PkBounds Bn = pk_bounds<DT>(lo, hi);
#pragma unroll
        for (int j = 0; j < BS / 2; ++j) {
            const uint32_t w = pw[j];
            const uint32_t m = pk_mask<DT>(w, Bn);                      // :492
            mk[j] = m;
            mo = pk_max_u16(mo, w & 0x7FFF7FFFu & m);
            mi = pk_max_u16(mi, w & 0x7FFF7FFFu & ~m);
        }
    }
    const uint32_t mib = (mi & 0xFFFFu) > (mi >> 16) ? (mi & 0xFFFFu) : (mi >> 16), mob = (mo & 0xFFFFu) > (mo >> 16) ? (mo & 0xFFFFu) : (mo >> 16);
    constexpr uint32_t INFB = (DT == 1) ? 0x7C00u : 0x7F80u;
    ok = ok && mib < INFB && mob < INFB;
    int ei, eo;
    ok = scale_exp_f32(absmax_to_float<DT>(mib), A.fi.emax, A.in_sb, ei) && ok;
    se_in_o = (float)ei;
    ok = ok && ei >= -60 && ei <= 60;
    ei = ok ? ei : 0;
    const bool no_out = mob == 0u;
    const float mxo = absmax_to_float<DT>(mob);
    const float mx_out = mxo * pow2f(ei);                               // max |o 2^e_in| (:216): exact in float32 inside the bound below
    ok = ok && (no_out || mx_out >= 1.1754943508222875e-38f);
    const bool eo_ok = scale_exp_f32(mx_out, A.fo.emax, A.out_sb, eo);
    se_out_o = (float)eo;
    ok = ok && eo_ok && (no_out || (eo >= -60 && eo <= 60));
    const int eoi = (ok && !no_out) ? eo - ei : 0;                      // one scale takes x to the outlier grid and back
    // the results q 2^e (q on the grid: at most four significant bits) and the clamp bounds must be T values, and T's own spacing four
    // times finer than the first tie (the sticky bit is the half's lowest)
    const int mine_i = A.fi.ebits ? 2 - (1 << (A.fi.ebits - 1)) : 0, mine_o = A.fo.ebits ? 2 - (1 << (A.fo.ebits - 1)) : 0;
    constexpr int TMINE = (DT == 1) ? -24 : -133, TMAXE = (DT == 1) ? 15 : 127;
    // (first tie at 2^(TMINE + 2) or above where the values enter a convert as halves / bf16; the smallest step a T value where they enter
    //  as float32: fp16 with the e4m3 / e5m2 grids, see pk_codec_loop32)
    const int slack_i = (DT == 1 && kin != 3 && kin != 4) ? -1 : 2, slack_o = (DT == 1 && kout != 3) ? -1 : 2;
    ok = ok && ei + mine_i - A.fi.mbits + 1 >= TMINE + slack_i && ei + A.fi.emax + 1 <= TMAXE && ei + A.fi.emax >= TMINE + 2
            && (no_out || (eoi + mine_o - A.fo.mbits + 1 >= TMINE + slack_o && eoi + A.fo.emax + 1 <= TMAXE && eoi + A.fo.emax >= TMINE + 2 && eoi >= -126 && eoi <= 126));
    const int eoc = ok ? eoi : 0;
    const int combo = kin * 4 + kout;
    if (combo == 4 * 4 + 3) pk_codec_loop32<BS, DT, 4, 3>(pw, mk, res, A, ei, eoc);
    else if (combo == 3 * 4 + 1) pk_codec_loop32<BS, DT, 3, 1>(pw, mk, res, A, ei, eoc);
    else if (combo == 3 * 4 + 3) pk_codec_loop32<BS, DT, 3, 3>(pw, mk, res, A, ei, eoc);
    else if (combo == 3 * 4 + 2) pk_codec_loop32<BS, DT, 3, 2>(pw, mk, res, A, ei, eoc);
    else pk_codec_loop32<BS, DT, 1, 1>(pw, mk, res, A, ei, eoc);
    return ok;
}


// the waves the float32-semantics packed kernels hand back: one block per lane through the float32 routine (outlier_block_fast with the
// hardware codecs: the route of k_outlier_contig / _strided for these tensors), written with one rounding to T
template <int BS, int DT>
__global__ void __launch_bounds__(256)
k_outlier_f32sem_list(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, OutlierArgs A, const int64_t* __restrict__ ws, const uint8_t* __restrict__ marks, int64_t cap) {
    const int64_t count = ws[0];
    if (count == 0) return;
    const int64_t total = A.pre * A.nblk * A.post;
    const int lane = threadIdx.x & 63;
    const bool by_marks = count >= cap;                                  // (the list stopped growing there: see k_outlier_lowp_pk)
    const int64_t items = by_marks ? (total + 63) / 64 : count, step = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < items; i += step) {
        if (by_marks && !marks[i]) continue;
        const int64_t t = (by_marks ? i * 64 : ws[8 + i]) + lane;
        if (t >= total) continue;
        const int64_t q = t % A.post, nb = (t / A.post) % A.nblk, p = t / (A.post * A.nblk);
        const int64_t base = (p * A.axis_len + nb * BS) * A.post + q;
        float a[BS];
        const bool vec = A.post == 1 && (BS % 8) == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
        if (vec) {
#pragma unroll
            for (int c = 0; c < BS / 8; ++c) {
                union { uint4 u; uint16_t h[8]; } v;
                v.u = *reinterpret_cast<const uint4*>(in + base + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[c * 8 + j] = ld16<DT>(v.h, j);
            }
        } else {
#pragma unroll
            for (int b = 0; b < BS; ++b) a[b] = ld16<DT>(in, base + (int64_t)b * A.post);
        }
        int order = 1;
        if (A.post > 1) {
            const int64_t lim = (A.post >= 8) ? (A.post / 32) * 32 : (A.post / 4) * 4;
            order = (q < lim) ? 0 : 2;
        }
        uint32_t mkw[(BS + 31) / 32];
        float se_in, se_out;
        const int status = outlier_block_fast<BS, 0, false, 1>(a, mkw, se_in, se_out, A, order, nullptr, nullptr, 1);
        if (vec) {
#pragma unroll
            for (int c = 0; c < BS / 8; ++c) {
                union { uint4 u; uint16_t h[8]; } v;
#pragma unroll
                for (int j = 0; j < 8; ++j) st16<DT>(v.h, j, Rr<DT>(a[c * 8 + j]));
                *reinterpret_cast<uint4*>(out + base + c * 8) = v.u;
            }
        } else {
#pragma unroll
            for (int b = 0; b < BS; ++b) st16<DT>(out, base + (int64_t)b * A.post, Rr<DT>(a[b]));
        }
        if (A.mask) {
#pragma unroll
            for (int b = 0; b < BS; ++b) A.mask[base + (int64_t)b * A.post] = (uint8_t)((mkw[b >> 5] >> (b & 31)) & 1u);
        }
        if (A.e_in) A.e_in[(p * A.nblk + nb) * A.post + q] = se_in;
        if (A.e_out) A.e_out[(p * A.nblk + nb) * A.post + q] = se_out;
        if (status && A.status) atomicOr(A.status, status);
    }
}

extern "C" void msq_set_error_(const char* msg);
#include <atomic>
#include <string.h>
static std::atomic<int> g_mx_pair4{1};
static std::atomic<int> g_outlier_pk{1};
extern "C" void msq_set_tuning_lowp_(const char* key, int value) {
    if (key && !strcmp(key, "mx_lowp_pair4")) g_mx_pair4.store(value);
    if (key && !strcmp(key, "outlier_lowp_pk")) g_outlier_pk.store(value);
}

// in / out: fp16 (dtype 1) or bf16 (dtype 2) tensors [pre, axis_len, post]; blocks of `block` (8 ... 128) along the axis, the
// last one zero padded; status_flag (device int, may be NULL) receives MSQ_STATUS_NAN when a shared exponent exceeds the
// scale range (the reference stores NaN there).
extern "C" int msq_quantize_mx_lowp(const void* in, void* out, int dtype, int64_t pre, int64_t axis_len, int64_t post, int block,
                                    int scale_bits, int elem_fmt, int rmode, int flush_fp32_subnorms, int* status_flag, void* stream) {
    if (pre <= 0 || axis_len <= 0 || post <= 0) return (pre == 0 || axis_len == 0 || post == 0) ? MSQ_OK : MSQ_ERR_BAD_ARG;
    if (!in || !out) { msq_set_error_("msq_quantize_mx_lowp: null buffer"); return MSQ_ERR_BAD_ARG; }
    if (dtype != 1 && dtype != 2) { msq_set_error_("msq_quantize_mx_lowp: dtype must be 1 (fp16) or 2 (bf16)"); return MSQ_ERR_BAD_ARG; }
    if (scale_bits <= 0 || scale_bits > 8 || rmode < 0 || rmode > 2) { msq_set_error_("msq_quantize_mx_lowp: bad scale bits / rounding mode"); return MSQ_ERR_BAD_ARG; }
    msq_host::FmtInfo fi;
    if (!msq_host::format_info(elem_fmt, &fi)) { msq_set_error_("msq_quantize_mx_lowp: unknown element format"); return MSQ_ERR_BAD_ARG; }
    if (fi.kind != 0) { msq_set_error_("msq_quantize_mx_lowp: float / int element formats only"); return MSQ_ERR_UNSUPPORTED; }
    MxLowpArgs A;
    A.pre = pre; A.axis_len = axis_len; A.post = post; A.nblk = (axis_len + block - 1) / block;
    A.f = Fmt{fi.kind, fi.ebits, fi.mbits, fi.emax, fi.max_norm};
    A.scale_bits = scale_bits; A.rmode = rmode; A.flush = flush_fp32_subnorms; A.status = status_flag;
    const int64_t n = A.pre * A.nblk * A.post;
    hipStream_t st = (hipStream_t)stream;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    const bool vec = post == 1 && (axis_len % block) == 0 && aligned;                    // contiguous axis: 8 values per lane
    const bool pair = !vec && post >= 2 && (post % 2) == 0 && (((uintptr_t)in | (uintptr_t)out) & 3) == 0 && block <= 32;   // strided axis: two channels per lane
    // ... cut four ways when the axis is whole blocks of 32 (MSQ_TUNE mx_lowp_pair4 = 0 keeps the one-lane-per-block-pair kernel)
    const int chunks4 = (int)((post / 2 + 63) / 64);
    const int64_t blocks4 = pre * A.nblk * chunks4;
    const bool pair4 = pair && block == 32 && (axis_len % 32) == 0 && blocks4 < (int64_t)0x7FFFFFFF && g_mx_pair4.load(std::memory_order_relaxed) != 0;
    const dim3 grid4((unsigned)(pair4 ? blocks4 : 1));
    const int64_t nchunks = pre * axis_len / 8;
    const int64_t nthreads = vec ? nchunks : (pair ? n / 2 : n);
    const dim3 grid((unsigned)((nthreads + 255) / 256)), blk(256);
#define MSQ_MXLP1(BS, DTV)                                                                                              \
        if (vec) hipLaunchKernelGGL((k_mx_lowp_vec<BS, DTV>), grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A, nchunks); \
        else if (pair4) hipLaunchKernelGGL((k_mx_lowp_pair4<DTV>), grid4, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A, chunks4); \
        else if (pair) hipLaunchKernelGGL((k_mx_lowp_pair<(BS <= 32 ? BS : 32), DTV>), grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A); \
        else hipLaunchKernelGGL((k_mx_lowp<BS, DTV>), grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A);
#define MSQ_MXLP(BS)                                                                                                    \
    case BS:                                                                                                           \
        if (dtype == 1) { MSQ_MXLP1(BS, 1) } else { MSQ_MXLP1(BS, 2) }                                                  \
        break;
    switch (block) { MSQ_MXLP(8) MSQ_MXLP(16) MSQ_MXLP(32) MSQ_MXLP(64) MSQ_MXLP(128)
        default: msq_set_error_("msq_quantize_mx_lowp: block must be 8, 16, 32, 64 or 128"); return MSQ_ERR_UNSUPPORTED; }
#undef MSQ_MXLP
#undef MSQ_MXLP1
    return hipGetLastError() == hipSuccess ? MSQ_OK : MSQ_ERR_LAUNCH;
}

// test hook kernels: the two rules the exhaustive fixtures pin
template <int DT>
__global__ void __launch_bounds__(256) k_floor_log2_lowp(const float* __restrict__ v, float* __restrict__ o, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = floor_log2_lowp<DT>(v[i]);
}

// called from msq_outlier_fakequant (msq_quant.hip) for dtype MSQ_DTYPE_F16_NATIVE / _BF16_NATIVE
// clears the head of the list (hipMemsetAsync goes through a generic fill kernel that takes 4.5 us for these 8 bytes; this one ~2)
__global__ void k_lowp_list_clear(int64_t* __restrict__ ws) { if (threadIdx.x < 8) ws[threadIdx.x] = 0; }
// bytes of the list of handed-back waves: a 64-byte head (the count) + one 8-byte entry per wave of k_outlier_lowp's lane numbering + one
// mark byte per wave (the list stops taking entries at 1 / 16 of the waves: beyond that the marks are read)
extern "C" int64_t msq_outlier_lowp_ws_bytes_(int64_t pre, int64_t axis_len, int64_t post, int block) {
    if (block != 8 && block != 16 && block != 32 && block != 64) return 0;
    const int64_t lanes = pre * ((axis_len + block - 1) / block) * post, nwv = (lanes + 63) / 64;
    return 64 + 8 * (nwv + 2) + (nwv + 2 + 63) / 64 * 64;               // head + list entries + one mark byte per wave
}
extern "C" int msq_launch_outlier_lowp_(const void* in, void* out, const void* args, int block, int dt, void* ws, int64_t ws_bytes, void* stream) {
    const OutlierArgs& A = *(const OutlierArgs*)args;
    const int64_t n = A.pre * A.nblk * A.post;
    int64_t g = (n + 255) / 256; if (g < 1) g = 1;
    const dim3 grid((unsigned)g), blk(256);
    hipStream_t st = (hipStream_t)stream;
    // the packed form: round to nearest, the format pairs with a hardware codec, whole blocks of 8 ... 64, aligned tensors, and a
    // workspace for the list of waves it hands back (msq_outlier_workspace_bytes)
    const int kin = hw_codec_kind(A.fi), kout = hw_codec_kind(A.fo), combo = kin * 4 + kout;
    const bool pk_fmt = A.rmode == 0 && !A.flush && (combo == 4 * 4 + 3 || combo == 3 * 4 + 1 || combo == 3 * 4 + 3 || combo == 3 * 4 + 2 || combo == 1 * 4 + 1);
    const uintptr_t al = (uintptr_t)in | (uintptr_t)out | (uintptr_t)A.mask;
    if (pk_fmt && (block == 8 || block == 16 || block == 32 || block == 64) && (A.axis_len % block) == 0 && g_outlier_pk.load(std::memory_order_relaxed) != 0 &&
        ws && ws_bytes >= msq_outlier_lowp_ws_bytes_(A.pre, A.axis_len, A.post, block) && ((uintptr_t)ws & 7) == 0) {
        const bool contig = A.post == 1 && (al & 15) == 0;
        const bool strided = A.post >= 2 && (A.post % 2) == 0 && (al & 3) == 0;
        if (contig || strided) {
            const int64_t lanes = contig ? A.pre * A.nblk : A.pre * A.nblk * (A.post / 2);
            const dim3 gp((unsigned)((lanes + 255) / 256));
            int64_t gl = (n + 1023) / 1024; if (gl > 512) gl = 512; if (gl < 1) gl = 1;
            const dim3 glist((unsigned)gl);
            const int64_t nwv = (n + 63) / 64, cap = nwv / 16 + 1;           // list entries while fewer than 1 / 16 of the waves; marks beyond
            uint8_t* marks = (uint8_t*)ws + 64 + 8 * (nwv + 2);
            hipLaunchKernelGGL(k_lowp_list_clear, dim3(1), dim3(64), 0, st, (int64_t*)ws);
#define MSQ_LPK(BS, DTV)                                                                                               \
            if (contig) hipLaunchKernelGGL((k_outlier_lowp_pk<BS, DTV>), gp, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A, kin, kout, (int64_t*)ws, marks, cap); \
            else hipLaunchKernelGGL((k_outlier_lowp_pk2<BS, DTV>), gp, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A, kin, kout, (int64_t*)ws, marks, cap); \
            hipLaunchKernelGGL((k_outlier_lowp_list<BS, DTV>), glist, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A, (const int64_t*)ws, (const uint8_t*)marks, cap);
            if (block == 8) { if (dt == 1) { MSQ_LPK(8, 1) } else { MSQ_LPK(8, 2) } }
            else if (block == 16) { if (dt == 1) { MSQ_LPK(16, 1) } else { MSQ_LPK(16, 2) } }
            else if (block == 32) { if (dt == 1) { MSQ_LPK(32, 1) } else { MSQ_LPK(32, 2) } }
            else { if (dt == 1) { MSQ_LPK(64, 1) } else { MSQ_LPK(64, 2) } }
#undef MSQ_LPK
            return 1;
        }
    }
#define MSQ_LP(BS)                                                                                                     \
    case BS:                                                                                                           \
        if (dt == 1) hipLaunchKernelGGL((k_outlier_lowp<BS, 1>), grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A); \
        else hipLaunchKernelGGL((k_outlier_lowp<BS, 2>), grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A);   \
        return 1;
    switch (block) { MSQ_LP(8) MSQ_LP(16) MSQ_LP(32) MSQ_LP(64) MSQ_LP(128) default: return 0; }
#undef MSQ_LP
}

// dtype 1 / 2 of msq_outlier_fakequant (fp16 / bf16 tensors computed in float32) on the packed kernels; 0 = not this call (the float32
// kernels of msq_outlier_kernels.h take it): other formats / rounding modes / blocks, no workspace, odd strides
extern "C" int msq_launch_outlier_f32sem_(const void* in, void* out, const void* args, int block, int dt, void* ws, int64_t ws_bytes, void* stream) {
    const OutlierArgs& A = *(const OutlierArgs*)args;
    const int kin = hw_codec_kind(A.fi), kout = hw_codec_kind(A.fo), combo = kin * 4 + kout;
    const bool pk_fmt = A.rmode == 0 && !A.flush && A.variant == 0 && !A.n_out &&
                        (combo == 4 * 4 + 3 || combo == 3 * 4 + 1 || combo == 3 * 4 + 3 || combo == 3 * 4 + 2 || combo == 1 * 4 + 1);
    if (!pk_fmt || (block != 8 && block != 16 && block != 32) || (A.axis_len % block) != 0 || g_outlier_pk.load(std::memory_order_relaxed) == 0) return 0;
    if (!ws || ws_bytes < msq_outlier_lowp_ws_bytes_(A.pre, A.axis_len, A.post, block) || ((uintptr_t)ws & 7) != 0) return 0;
    const uintptr_t al = (uintptr_t)in | (uintptr_t)out | (uintptr_t)A.mask;
    const bool contig = A.post == 1 && (al & 15) == 0;
    const bool strided = A.post >= 2 && (A.post % 2) == 0 && (al & 3) == 0;
    if (!contig && !strided) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = A.pre * A.nblk * A.post;
    const int64_t lanes = contig ? A.pre * A.nblk : A.pre * A.nblk * (A.post / 2);
    const dim3 gp((unsigned)((lanes + 255) / 256)), blk(256);
    int64_t gl = (n + 1023) / 1024; if (gl > 512) gl = 512; if (gl < 1) gl = 1;
    const dim3 glist((unsigned)gl);
    const int64_t nwv = (n + 63) / 64, cap = nwv / 16 + 1;
    uint8_t* marks = (uint8_t*)ws + 64 + 8 * (nwv + 2);
    hipLaunchKernelGGL(k_lowp_list_clear, dim3(1), dim3(64), 0, st, (int64_t*)ws);
#define MSQ_LPK32(BS, DTV)                                                                                             \
    if (contig) hipLaunchKernelGGL((k_outlier_lowp_pk<BS, DTV, true>), gp, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A, kin, kout, (int64_t*)ws, marks, cap); \
    else hipLaunchKernelGGL((k_outlier_lowp_pk2<BS, DTV, true>), gp, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A, kin, kout, (int64_t*)ws, marks, cap); \
    hipLaunchKernelGGL((k_outlier_f32sem_list<BS, DTV>), glist, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, A, (const int64_t*)ws, (const uint8_t*)marks, cap);
    if (block == 8) { if (dt == 1) { MSQ_LPK32(8, 1) } else { MSQ_LPK32(8, 2) } }
    else if (block == 16) { if (dt == 1) { MSQ_LPK32(16, 1) } else { MSQ_LPK32(16, 2) } }
    else { if (dt == 1) { MSQ_LPK32(32, 1) } else { MSQ_LPK32(32, 2) } }
#undef MSQ_LPK32
    return 1;
}

extern "C" int msq_floor_log2_lowp(const float* v, float* out, int64_t n, int dtype, void* stream) {
    if (n <= 0) return MSQ_OK;
    if (!v || !out || (dtype != 1 && dtype != 2)) return MSQ_ERR_BAD_ARG;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (dtype == 1) hipLaunchKernelGGL(k_floor_log2_lowp<1>, grid, blk, 0, (hipStream_t)stream, v, out, n);
    else hipLaunchKernelGGL(k_floor_log2_lowp<2>, grid, blk, 0, (hipStream_t)stream, v, out, n);
    return hipGetLastError() == hipSuccess ? MSQ_OK : MSQ_ERR_LAUNCH;
}
