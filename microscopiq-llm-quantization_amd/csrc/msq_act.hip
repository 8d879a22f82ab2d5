// msq_act.hip -- activation side of the W4A8 Linear (number_system/mx/linear.py:66-73): MicroScopiQ
// outlier-aware MX quantisation of X [M, K] along K, result written as bf16 for the fused GEMM.
// Every quantised value is code * 2^scale with an <= 8-bit code, so bf16 holds it exactly (checked:
// MSQ_STATUS_INEXACT otherwise).  HBM-bound: 4 B read + 2 B written per element.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"
#include "msq_outlier_core.h"

extern "C" void msq_set_error_(const char* msg);
extern "C" int msq_mxops_stats_x_(const void* in, int x_bf16, float* vmean, float* vstd, int64_t pre, int64_t axis_len, int64_t post,
                                  int block, int* status, void* stream);

namespace {

MSQ_D uint32_t bf16_bits_exact(float v, int& status) {
    const uint32_t u = f2u(v);
    if ((u & 0xFFFFu) && v == v) status |= MSQ_STATUS_INEXACT;
    return u >> 16;
}

// Lean form of outlier_block_fast<BS, 0, false, 1> + the bf16 packing for the activation kernel, same results bit
// for bit (tests: test_act_quant_*): about a quarter fewer vector instructions per block (the kernel is VALU-issue
// bound: ~1300 instructions per 64 blocks measured with SQ_INSTS_VALU, profiles/r02_act_quant_pmc.txt).
//  * the inlier / outlier maxima are integer maxima of the |a| bit patterns (v_max3_u32: one instruction per two
//    elements and class); a NaN / Inf element shows up as a maximum >= 0x7F800000;
//  * variant 0 recomputes the mask bit from (lo, hi) in the codec loop instead of building and unpacking a mask word;
//  * two results are packed with one v_perm_b32; the "exact in bf16" proof ORs the dropped halves of the whole block.
// Blocks with a NaN / Inf element, a flushed or NaN scale, or an exponent outside +-60 return false untouched and take
// the general routine.
// variant 1 bounds: lane_tab = true -> lane l of the wave holds (lo, hi) of block position l in (tlo, thi) (the 64
// blocks of the wave share one table row: 3 instructions per wave + 2 v_readlane per position instead of 2 scalar
// loads + 3 instructions per position); otherwise they are computed from the per-lane table rows vm / vs.
template <int BS, int KI, int KO>
MSQ_D bool act_block_lean(const float (&a)[BS], uint32_t (&h)[BS / 2], const OutlierArgs& A, const float* vm, const float* vs,
                          bool lane_tab, float tlo, float thi, int& status) {
    float lo = 0.f, hi = 0.f;
    const bool v1 = (A.variant != 0);
    if (!v1) {
        float ab[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) ab[b] = __builtin_fabsf(a[b]);
        const float mean = sum_inner8<BS>(ab) / (float)BS;
        const float ks = A.k * std_twopass_checked<BS>(ab, 0);
        lo = mean - ks; hi = mean + ks;
    }
    uint32_t mkw[(BS + 31) / 32];
#pragma unroll
    for (int w = 0; w < (BS + 31) / 32; ++w) mkw[w] = 0u;
    uint32_t ui = 0u, uo = 0u;
    bool quirky = false;
#pragma unroll
    for (int b = 0; b < BS; b += 2) {
        bool m[2];
        uint32_t ti[2], to[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (v1) {
                if (lane_tab) {
                    lo = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(tlo), b + j));
                    hi = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(thi), b + j));
                } else {
                    const float ks = A.k * vs[b + j];
                    lo = vm[b + j] - ks; hi = vm[b + j] + ks;
                }
            }
            m[j] = (a[b + j] < lo) || (a[b + j] > hi);
            if (v1) mkw[(b + j) >> 5] |= (m[j] ? 1u : 0u) << ((b + j) & 31);
            const uint32_t t = f2u(a[b + j]) & 0x7FFFFFFFu;
            quirky |= mantissa_all_ones(t);
            ti[j] = m[j] ? 0u : t; to[j] = m[j] ? t : 0u;
        }
        ui = max(ui, max(ti[0], ti[1])); uo = max(uo, max(to[0], to[1]));
    }
    if (max(ui, uo) >= 0x7F800000u) return false;                      // NaN / Inf element (a NaN compares false: unmasked)

    const float mx_in = u2f(ui), mx_o = u2f(uo);
    float se_in = shared_exp_of_max(mx_in);
    if (A.flush && !(se_in > -127.f)) return false;
    se_in = clamp_scale_exp(se_in - (float)A.fi.emax, A.in_sb, A.variant);
    if (!(se_in >= -60.f && se_in <= 60.f)) return false;               // also NaN
    const float sc_in = exp2f_int(se_in);
    const bool no_out = (uo == 0u);
    float se_out = shared_exp_of_max(mx_o * sc_in);
    se_out = clamp_scale_exp(se_out - (float)A.fo.emax, A.out_sb, A.variant);
    if (!(no_out || (se_out >= -60.f && se_out <= 60.f))) return false;
    const int ei = (int)se_in, eo = no_out ? 0 : (int)se_out - (int)se_in;
    const float s_in = u2f((uint32_t)(ei + 127) << 23), s_eff = u2f((uint32_t)(eo + 127) << 23);
    const float b_in = A.fi.max_norm * s_in, b_out = A.fo.max_norm * s_eff;
    uint32_t dropped = 0u;
    // The converts round half away from zero exactly; the reference's float32 `floor(|x| + 0.5)` does not for ONE magnitude per grid (msq_device.h
    // half_away_quirk_bits): |x| = pred(half the smallest step) comes back as one step.  A block that holds an all-ones mantissa (every such
    // magnitude has one; ~1 block in 260 000 of Gaussian data) runs the same loop with the exact per-element fix-up -- NOT the general routine:
    // one wave in it stretched this 19 us kernel by 8 us (round 6, profiles/r06_act_quirk_ab.txt).
    const int hexp_i = ei + (A.fi.ebits ? 2 - (1 << (A.fi.ebits - 1)) : 0) - A.fi.mbits + 1;      // exponent of half the smallest inlier step x the block scale
    const int hexp_o = (ei + eo) + (A.fo.ebits ? 2 - (1 << (A.fo.ebits - 1)) : 0) - A.fo.mbits + 1;
#pragma unroll
    for (int b = 0; b < BS; b += 2) {
        const float x0 = u2f(f2u(a[b]) | 1u), x1 = u2f(f2u(a[b + 1]) | 1u);
        float vi0, vi1, vo0, vo1;
        hw_codec_pair(KI, x0, x1, s_in, b_in, vi0, vi1);
        hw_codec_pair(KO, x0, x1, s_eff, b_out, vo0, vo1);
        bool m0, m1;
        if (v1) { m0 = (mkw[b >> 5] >> (b & 31)) & 1u; m1 = (mkw[(b + 1) >> 5] >> ((b + 1) & 31)) & 1u; }
        else { m0 = (a[b] < lo) || (a[b] > hi); m1 = (a[b + 1] < lo) || (a[b + 1] > hi); }
        const uint32_t r0 = f2u((m0 ? vo0 : vi0) + 0.0f), r1 = f2u((m1 ? vo1 : vi1) + 0.0f);
        dropped |= r0 | r1;
        h[b >> 1] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);           // (r0 >> 16) | (r1 & 0xFFFF0000)
    }
    if (quirky) {                                                          // the fix-up pass: compares only, patches the packed halves in place
        const uint32_t qi = half_away_quirk_bits(hexp_i), qo = half_away_quirk_bits(hexp_o);
        const uint32_t fi_ = f2u(pow2i(hexp_i + 1)) >> 16, fo_ = f2u(pow2i(hexp_o + 1)) >> 16;     // one step, as bf16 bits
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            bool m;
            if (v1) m = (mkw[b >> 5] >> (b & 31)) & 1u;
            else m = (a[b] < lo) || (a[b] > hi);
            const uint32_t u = f2u(a[b]);
            if ((u & 0x7FFFFFFFu) == (m ? qo : qi)) {
                const uint32_t f = ((u >> 16) & 0x8000u) | (m ? fo_ : fi_);
                h[b >> 1] = (b & 1) ? ((h[b >> 1] & 0x0000FFFFu) | (f << 16)) : ((h[b >> 1] & 0xFFFF0000u) | f);
            }
        }
    }
    if (dropped & 0xFFFFu) status |= MSQ_STATUS_INEXACT;                   // no NaN can reach this point
    return true;
}

// A wave owns 64 consecutive blocks = one contiguous run of 64*BS floats: coalesced 16-byte loads,
// transpose through LDS (row stride BS+4 words), one block per lane, and back the same way as bf16.
// XBF16: X holds bfloat16 (the activations of a bf16 model: every bf16 is an fp32 value, same results as casting first)
template <int BS, int RM, int HW, bool XBF16 = false>
__global__ void __launch_bounds__(256)
k_act_quant(const float* __restrict__ X, uint16_t* __restrict__ Xq, OutlierArgs A) {
    constexpr int LDS_STRIDE = BS + 4;
    __shared__ __attribute__((aligned(16))) float tile[4][64 * LDS_STRIDE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t nblocks = A.pre * A.nblk;
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (g0 >= nblocks) return;
    const bool full = (g0 + 64 <= nblocks);
    const int64_t gidx = g0 + lane;
    float* tl = tile[wv];
    float a[BS];
    const uint16_t* Xh = reinterpret_cast<const uint16_t*>(X);
    if (full) {
        if (XBF16) {
            const uint4* s8 = reinterpret_cast<const uint4*>(Xh + g0 * BS);          // 8 bf16 per 16-byte load
#pragma unroll
            for (int t = 0; t < BS / 8; ++t) {
                const int f = lane + 64 * t;
                const int row = f / (BS / 8), c8 = f % (BS / 8);
                const uint4 v = s8[f];
                *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c8 * 8) = make_float4(u2f(v.x << 16), u2f(v.x & 0xFFFF0000u), u2f(v.y << 16), u2f(v.y & 0xFFFF0000u));
                *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c8 * 8 + 4) = make_float4(u2f(v.z << 16), u2f(v.z & 0xFFFF0000u), u2f(v.w << 16), u2f(v.w & 0xFFFF0000u));
            }
        } else {
            const float4* src = reinterpret_cast<const float4*>(X + g0 * BS);
#pragma unroll
            for (int t = 0; t < BS / 4; ++t) {
                const int f = lane + 64 * t;
                const int row = f / (BS / 4), c4 = f % (BS / 4);
                *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c4 * 4) = src[f];
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int c = 0; c < BS / 4; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(tl + lane * LDS_STRIDE + c * 4);
            a[c * 4 + 0] = v.x; a[c * 4 + 1] = v.y; a[c * 4 + 2] = v.z; a[c * 4 + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int b = 0; b < BS; ++b) a[b] = (gidx < nblocks) ? (XBF16 ? u2f((uint32_t)Xh[gidx * BS + b] << 16) : X[gidx * BS + b]) : 0.f;
    }
    uint32_t h[BS / 2];
    int status = 0;
    if (gidx < nblocks) {
        const bool uni_tab = A.vmean && full && (A.nblk % 64 == 0);
        const float *vm = nullptr, *vs = nullptr;
        float tlo = 0.f, thi = 0.f;
        if (uni_tab) {
            // the 64 blocks of this wave lie in one row: the statistics tables are wave-uniform, read them
            // through scalar loads (the mean / std of position b are the same for every lane)
            const int64_t prow = g0 / A.nblk;
            const uint64_t pm = (uint64_t)(A.vmean + prow * BS), ps = (uint64_t)(A.vstd + prow * BS);
            auto uni64 = [](uint64_t v) -> uint64_t {            // readfirstlane returns a signed int: widen through uint32_t
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
                return ((uint64_t)hi << 32) | (uint64_t)lo;
            };
            vm = (const float*)uni64(pm); vs = (const float*)uni64(ps);
            if (BS <= 64) {
                const int l = lane < BS ? lane : 0;
                const float ks = A.k * vs[l];
                tlo = vm[l] - ks; thi = vm[l] + ks;
            }
        } else if (A.vmean) {
            vm = A.vmean + (gidx / A.nblk) * BS; vs = A.vstd + (gidx / A.nblk) * BS;
        }
        bool done = false;
        if (HW == 1 && RM == 0) {
            const int ki = hw_codec_kind(A.fi), ko = hw_codec_kind(A.fo);          // wave-uniform
            if (ki == 1 && ko == 1) done = act_block_lean<BS, 1, 1>(a, h, A, vm, vs, uni_tab, tlo, thi, status);
            else if (ki == 3 && ko == 3) done = act_block_lean<BS, 3, 3>(a, h, A, vm, vs, uni_tab, tlo, thi, status);
            else if (ki == 3 && ko == 1) done = act_block_lean<BS, 3, 1>(a, h, A, vm, vs, uni_tab, tlo, thi, status);
        }
        if (!done) {
            uint32_t mkw[(BS + 31) / 32];
            float se_in = 0.f, se_out = 0.f;
            status |= outlier_block_fast<BS, RM, false, HW>(a, mkw, se_in, se_out, A, /*inner order*/ 1, vm, vs, 1);
#pragma unroll
            for (int b = 0; b < BS / 2; ++b) h[b] = bf16_bits_exact(a[2 * b], status) | (bf16_bits_exact(a[2 * b + 1], status) << 16);
        }
    }
    if (full) {
        // bf16 rows: BS/2 words per block, LDS row stride BS/2 + 4 words (16-byte aligned, conflict-free)
        constexpr int HS = BS / 2 + 4;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < BS / 8; ++c)
            *reinterpret_cast<uint4*>(tl + lane * HS + c * 4) = make_uint4(h[c * 4], h[c * 4 + 1], h[c * 4 + 2], h[c * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        uint4* dst = reinterpret_cast<uint4*>(Xq + g0 * BS);
#pragma unroll
        for (int t = 0; t < BS / 8; ++t) {
            const int f = lane + 64 * t;
            const int row = f / (BS / 8), c4 = f % (BS / 8);
            dst[f] = *reinterpret_cast<const uint4*>(tl + row * HS + c4 * 4);
        }
    } else if (gidx < nblocks) {
#pragma unroll
        for (int b = 0; b < BS / 2; ++b) reinterpret_cast<uint32_t*>(Xq + gidx * BS)[b] = h[b];
    }
    if (status && A.status) atomicOr(A.status, status);
}

// ---------------------------------------------------------------------------
// variant 1 (mx_ops statistics, number_system/mx/mx_ops.py:225-233 on utils/quant.py:460-495's column mean / std) in ONE pass over X
// for rows of at most 4096 elements: a wave owns a row, keeps it in registers (64 values per lane), computes the column statistics
// exactly as k_mxops_stats_rows does (msq_quant.hip: torch's cascade order for the fp32 mean -- sixteen-block runs summed from zero,
// folded in order; the std as a two-pass variance in double, rounded to float only when it provably lies on the same side of the
// rounding boundary as torch's sequential Welford result, else redone sequentially), then runs the
// quantiser of the kernel above over the row (64 blocks per round, handed over through a 9 KB LDS tile).  X is read once (4 B in, 2 B out per element) and the two launches
// (statistics, quantiser) become one: [2048, 4096] fp32 35.0 -> see profiles/r04_side_kernels.txt.
// Lane (j, q), j = lane / LPB, q = lane % LPB, LPB = BS / 4: columns 4 q ... 4 q + 3 of the blocks of run j (blocks 16 j ... 16 j + 15).
// ---------------------------------------------------------------------------
template <int BS, int RM, int HW, bool XBF16>
__global__ void __launch_bounds__(256, 2)
k_act_quant_rows(const float* __restrict__ X, uint16_t* __restrict__ Xq, OutlierArgs A) {
    constexpr int LPB = BS / 4, RW = 64 / LPB;                      // lanes per block row, runs per wave (16 RW blocks = 4096 elements)
    constexpr int STRIDE = BS + 4, HS = BS / 2 + 4;
    __shared__ __attribute__((aligned(16))) float tile[4][64 * STRIDE];
    __shared__ __attribute__((aligned(16))) float tab[4][2][BS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 4 + wv;
    if (p >= A.pre) return;
    const int nblk = (int)A.nblk;
    const int j = lane / LPB, q = lane % LPB;
    const int nruns = nblk / 16, tail = nblk % 16;
    const int len = j < nruns ? 16 : (j == nruns ? tail : 0);       // blocks of this lane's run
    float* tl = tile[wv];
    const float* rowf = X + p * A.axis_len;
    const uint16_t* rowh = reinterpret_cast<const uint16_t*>(X) + p * A.axis_len;
    int status = 0;
    float4 v[16];                                                   // the lane's part of the row: live until its round of the quantiser
    {
        // ---- the row: 16 x 16 bytes per lane, all requested before the first use
#pragma unroll
        for (int i = 0; i < 16; ++i) {          // (unconditional loads: a load under a lane-dependent branch is waited for at the join)
            int blk = 16 * j + i; blk = blk < nblk ? blk : nblk - 1;
            const int64_t at = (int64_t)blk * BS + 4 * q;
            if (XBF16) {
                const uint2 u = *reinterpret_cast<const uint2*>(rowh + at);
                v[i] = make_float4(u2f(u.x << 16), u2f(u.x & 0xFFFF0000u), u2f(u.y << 16), u2f(u.y & 0xFFFF0000u));
            } else v[i] = *reinterpret_cast<const float4*>(rowf + at);
        }
        // ---- statistics of columns 4 q ... 4 q + 3 (every lane ends with the full-column values)
        float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f, rs3 = 0.f;
        double ds0 = 0.0, ds1 = 0.0, ds2 = 0.0, ds3 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < len) {
                rs0 += v[i].x; rs1 += v[i].y; rs2 += v[i].z; rs3 += v[i].w;
                ds0 += (double)v[i].x; ds1 += (double)v[i].y; ds2 += (double)v[i].z; ds3 += (double)v[i].w;
            }
        auto xsum = [&](double d) {
#pragma unroll
            for (int o = LPB; o < 64; o <<= 1) d += __shfl_xor(d, o, 64);
            return d;
        };
        const double n = (double)nblk;
        const double m0 = xsum(ds0) / n, m1 = xsum(ds1) / n, m2_ = xsum(ds2) / n, m3 = xsum(ds3) / n;
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < len) {
                const double e0 = (double)v[i].x - m0, e1 = (double)v[i].y - m1, e2 = (double)v[i].z - m2_, e3 = (double)v[i].w - m3;
                q0 = __builtin_fma(e0, e0, q0); q1 = __builtin_fma(e1, e1, q1); q2 = __builtin_fma(e2, e2, q2); q3 = __builtin_fma(e3, e3, q3);
                d0 += e0; d1 += e1; d2 += e2; d3 += e3;
            }
        // one column: cascade fold of the run sums (ATen: level-1 accumulator folded every 256 elements; <= 128 blocks never fold twice),
        // two-pass std with the rounding-boundary check
        auto column = [&](float rs, double dq, double dd, float& mean_out, double& sdd, bool& safe) {
            float acc1 = 0.f, acc2 = 0.f, acc0 = 0.f;
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                const float r = __shfl(rs, q + LPB * k, 64);
                if (k < nruns) { acc1 += r; if ((((k + 1) * 16) & (15 << 4)) == 0) { acc2 += acc1; acc1 = 0.f; } }
                else if (k == nruns) acc0 = r;                      // the tail run (0 when empty)
            }
            acc0 += acc1; acc0 += acc2; acc0 += 0.f;
            mean_out = acc0 / (float)nblk;
            dq = xsum(dq); dd = xsum(dd);
            double den = n - 1.0; den = den < 0 ? 0 : den;
            double m2 = dq - dd * dd / n; m2 = m2 < 0 ? 0 : m2;
            sdd = __builtin_sqrt(m2 / den);
            const uint64_t bits = __builtin_bit_cast(uint64_t, sdd);
            const uint32_t dropped = (uint32_t)(bits & 0x1FFFFFFFull);
            const uint32_t dist = dropped > 0x10000000u ? dropped - 0x10000000u : 0x10000000u - dropped;
            safe = (dist > 4u * (uint32_t)nblk + 64u) && (sdd == sdd) && (sdd > 1e-150) && (sdd < 1e150);
        };
        float4 mean4, sd4;
        double s0, s1, s2, s3;
        bool f0, f1, f2, f3;
        column(rs0, q0, d0, mean4.x, s0, f0); column(rs1, q1, d1, mean4.y, s1, f1);
        column(rs2, q2, d2, mean4.z, s2, f2); column(rs3, q3, d3, mean4.w, s3, f3);
        // A column whose two-pass value lies too close to a float rounding boundary (about one in 500 000) is redone as torch does it:
        // sequential Welford in double over the blocks in order.  The whole wave walks the row together (the values come from the
        // owning lanes' registers by ds_bpermute, which needs every lane active), only the flagged one of the four columns per lane.
        // The chain of nblk dependent steps is the tail of the whole launch when it happens, so the division by the count -- a small
        // integer b -- is the fma form: y = RN(1 / b) (one true division per lane, up front), q = RN(g y), two rounds of
        // r = RN(g - q b) (exact), q = RN(q + r y): correctly rounded (Markstein: y is the correctly rounded reciprocal and b's
        // significand is short), 5 dependent operations instead of the ~12 of the expanded IEEE division; checked against true
        // division on the host (tests/test_host_logic.py).
        const uint64_t unsafe = __builtin_amdgcn_ballot_w64(!(f0 && f1 && f2 && f3));
        if (unsafe != 0) {
            const double yA = 1.0 / (double)(lane + 1), yB = 1.0 / (double)(lane + 65);
            auto rdl = [&](double a, int l) {
                const uint64_t u = __builtin_bit_cast(uint64_t, a);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
                return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
            };
            auto welford = [&](auto comp) {
                double w = 0.0, t = 0.0;
                for (int jj = 0; jj * 16 < nblk; ++jj) {
                    double e[16];
#pragma unroll
                    for (int ii = 0; ii < 16; ++ii) e[ii] = (double)__shfl(comp(v[ii]), q + LPB * jj, 64);     // all sixteen in flight, then the chain
#pragma unroll
                    for (int ii = 0; ii < 16; ++ii) {
                        const int s = jj * 16 + ii;
                        if (s < nblk) {
                            const double cnt = (double)(s + 1), y = s < 64 ? rdl(yA, s) : rdl(yB, s - 64);
                            const double g = e[ii] - w;
                            double qt = g * y;
                            qt = __builtin_fma(__builtin_fma(-qt, cnt, g), y, qt);
                            qt = __builtin_fma(__builtin_fma(-qt, cnt, g), y, qt);
                            w = w + qt;
                            t = t + g * (e[ii] - w);
                        }
                    }
                }
                double den = n - 1.0; den = den < 0 ? 0 : den;
                return __builtin_sqrt(t / den);
            };
            if (__builtin_amdgcn_ballot_w64(!f0) != 0) { const double r = welford([](const float4& x) { return x.x; }); if (!f0) s0 = r; }
            if (__builtin_amdgcn_ballot_w64(!f1) != 0) { const double r = welford([](const float4& x) { return x.y; }); if (!f1) s1 = r; }
            if (__builtin_amdgcn_ballot_w64(!f2) != 0) { const double r = welford([](const float4& x) { return x.z; }); if (!f2) s2 = r; }
            if (__builtin_amdgcn_ballot_w64(!f3) != 0) { const double r = welford([](const float4& x) { return x.w; }); if (!f3) s3 = r; }
        }
        sd4 = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
        if (sd4.x != sd4.x || sd4.y != sd4.y || sd4.z != sd4.z || sd4.w != sd4.w) status |= MSQ_STATUS_NAN;
        if (j == 0) {
            *reinterpret_cast<float4*>(&tab[wv][0][4 * q]) = mean4;
            *reinterpret_cast<float4*>(&tab[wv][1][4 * q]) = sd4;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const float* vm = tab[wv][0];
    const float* vs = tab[wv][1];
    float tlo, thi;
    { const int l = lane < BS ? lane : 0; const float ks = A.k * vs[l]; tlo = vm[l] - ks; thi = vm[l] + ks; }
    // ---- the quantiser of k_act_quant over the row, 64 blocks at a time, fed from the registers that hold the row
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int nb = nblk - b0 < 64 ? nblk - b0 : 64;             // blocks of this round
        const int blk = b0 + lane;
        float a[BS];
        // the 64 blocks of this round are the runs 4 r .. 4 r + 3: their lanes put their registers into the tile (block-major rows)
        if ((j >> 2) == (b0 >> 6)) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < len) *reinterpret_cast<float4*>(tl + (16 * (j & 3) + i) * STRIDE + 4 * q) = v[i];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int c = 0; c < BS / 4; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(tl + (lane < nb ? lane : 0) * STRIDE + c * 4);
            a[c * 4 + 0] = t.x; a[c * 4 + 1] = t.y; a[c * 4 + 2] = t.z; a[c * 4 + 3] = t.w;
        }
        uint32_t h[BS / 2];
        bool done = false;
        if (HW == 1 && RM == 0) {
            const int ki = hw_codec_kind(A.fi), ko = hw_codec_kind(A.fo);              // wave-uniform
            if (ki == 1 && ko == 1) done = act_block_lean<BS, 1, 1>(a, h, A, vm, vs, true, tlo, thi, status);
            else if (ki == 3 && ko == 3) done = act_block_lean<BS, 3, 3>(a, h, A, vm, vs, true, tlo, thi, status);
            else if (ki == 3 && ko == 1) done = act_block_lean<BS, 3, 1>(a, h, A, vm, vs, true, tlo, thi, status);
        }
        if (!done) {
            uint32_t mkw[(BS + 31) / 32];
            float se_in = 0.f, se_out = 0.f;
            int st = outlier_block_fast<BS, RM, false, HW>(a, mkw, se_in, se_out, A, /*inner order*/ 1, vm, vs, 1);
#pragma unroll
            for (int b = 0; b < BS / 2; ++b) h[b] = bf16_bits_exact(a[2 * b], st) | (bf16_bits_exact(a[2 * b + 1], st) << 16);
            if (lane < nb) status |= st;
        }
        // bf16 rows back through LDS (the input tile is consumed), coalesced 16-byte stores
        __builtin_amdgcn_wave_barrier();
        if (lane < nb) {
#pragma unroll
            for (int c = 0; c < BS / 8; ++c)
                *reinterpret_cast<uint4*>(tl + lane * HS + c * 4) = make_uint4(h[c * 4], h[c * 4 + 1], h[c * 4 + 2], h[c * 4 + 3]);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        uint4* dst = reinterpret_cast<uint4*>(Xq + p * A.axis_len + (int64_t)b0 * BS);
#pragma unroll
        for (int t = 0; t < BS / 8; ++t) {
            const int f = lane + 64 * t;
            if (f < nb * (BS / 8)) dst[f] = *reinterpret_cast<const uint4*>(tl + (f / (BS / 8)) * HS + (f % (BS / 8)) * 4);
        }
        __builtin_amdgcn_wave_barrier();
        (void)blk;
    }
    if (status && A.status) atomicOr(A.status, status);
}

}  // namespace

static msq_host::TuneKey g_act_rows("MSQ_ACT_ROWS");        // 0: statistics + quantiser as two launches (msq_set_tuning / environment, per call)
extern "C" int msq_set_tuning_act_(const char* key, int value) { return g_act_rows.set_if(key, value); }

extern "C" int64_t msq_act_quant_workspace_bytes(int64_t M, int64_t K, int block, int variant) {
    if (variant != MSQ_VARIANT_MXOPS || M <= 0 || K <= 0 || block <= 0) return 0;
    return 2 * (int64_t)sizeof(float) * M * block;
}

static int act_quant_impl(const void* Xv, int x_bf16, void* Xq, int* status_flag, void* workspace, int64_t workspace_bytes,
                          int64_t M, int64_t K, int block, int inlier_fmt, int outlier_fmt, int inlier_scale_bits,
                          int outlier_scale_bits, float std_dev, int rmode, int flush_fp32_subnorms, int variant,
                          void* stream) {
    const float* X = (const float*)Xv;
    if (M < 0 || K < 0) { msq_set_error_("msq_act_quant_bf16: negative size"); return MSQ_ERR_BAD_ARG; }
    if (M == 0 || K == 0) return MSQ_OK;
    if (!X || !Xq) { msq_set_error_("msq_act_quant_bf16: null buffer"); return MSQ_ERR_BAD_ARG; }
    msq_host::FmtInfo fi, fo;
    if (!msq_host::format_info(inlier_fmt, &fi) || !msq_host::format_info(outlier_fmt, &fo)) {
        msq_set_error_("msq_act_quant_bf16: unknown element format"); return MSQ_ERR_BAD_ARG; }
    if (fi.kind != 0 || fo.kind != 0 || fi.mbits > 9 || fo.mbits > 9) {
        msq_set_error_("msq_act_quant_bf16: element formats must be float/int formats of at most 8 bits"); return MSQ_ERR_UNSUPPORTED; }
    if (inlier_scale_bits <= 0 || outlier_scale_bits <= 0 || inlier_scale_bits > 8 || outlier_scale_bits > 8 || rmode < 0 || rmode > 2) {
        msq_set_error_("msq_act_quant_bf16: bad scale bits / rounding mode"); return MSQ_ERR_BAD_ARG; }
    if (variant != MSQ_VARIANT_QUANT && variant != MSQ_VARIANT_MXOPS) { msq_set_error_("msq_act_quant_bf16: bad variant"); return MSQ_ERR_BAD_ARG; }
    if (!(block == 16 || block == 32 || block == 64) || (K % block)) {
        msq_set_error_("msq_act_quant_bf16: block must be 16, 32 or 64 and divide K"); return MSQ_ERR_UNSUPPORTED; }
    OutlierArgs A;
    A.fi = Fmt{fi.kind, fi.ebits, fi.mbits, fi.emax, fi.max_norm};
    A.fo = Fmt{fo.kind, fo.ebits, fo.mbits, fo.emax, fo.max_norm};
    A.in_sb = inlier_scale_bits; A.out_sb = outlier_scale_bits; A.k = std_dev; A.rmode = rmode;
    A.flush = flush_fp32_subnorms; A.variant = variant;
    A.pre = M; A.axis_len = K; A.post = 1; A.nblk = K / block;
    A.mask = nullptr; A.e_in = nullptr; A.e_out = nullptr; A.n_out = nullptr; A.status = status_flag;
    A.vmean = nullptr; A.vstd = nullptr;
    if (variant == MSQ_VARIANT_MXOPS) {
        if (!workspace || workspace_bytes < msq_act_quant_workspace_bytes(M, K, block, variant)) {
            msq_set_error_("msq_act_quant_bf16: workspace too small (msq_act_quant_workspace_bytes)"); return MSQ_ERR_BAD_ARG; }
        float* vmean = (float*)workspace;
        float* vstd = vmean + M * block;
        if (x_bf16 && block == 16) { msq_set_error_("msq_act_quant_bf16_x16: the mx_ops variant on bfloat16 input needs block 32 or 64"); return MSQ_ERR_UNSUPPORTED; }
        // rows of <= 4096 elements, blocks of 32, round-to-nearest: statistics and quantiser in one pass over X (k_act_quant_rows).
        // MSQ_ACT_ROWS=0 (tuning and A / B, read per call) keeps the two launches.
        if (K <= 4096 && block == 32 && rmode == 0 && K / block >= 2 && g_act_rows.value(1) != 0) {
            const bool hw1 = hw_codec_kind(A.fi) && hw_codec_kind(A.fo);
            const dim3 grid((unsigned)((M + 3) / 4)), blk(256);
            hipStream_t st = (hipStream_t)stream;
#define MSQ_AQR(BS) do { if (x_bf16) { if (hw1) hipLaunchKernelGGL((k_act_quant_rows<BS, 0, 1, true>), grid, blk, 0, st, X, (uint16_t*)Xq, A); \
                                          else hipLaunchKernelGGL((k_act_quant_rows<BS, 0, 0, true>), grid, blk, 0, st, X, (uint16_t*)Xq, A); } \
                         else if (hw1) hipLaunchKernelGGL((k_act_quant_rows<BS, 0, 1, false>), grid, blk, 0, st, X, (uint16_t*)Xq, A); \
                         else hipLaunchKernelGGL((k_act_quant_rows<BS, 0, 0, false>), grid, blk, 0, st, X, (uint16_t*)Xq, A); } while (0)
            MSQ_AQR(32);      // (block 64: 64 + 32 live values per lane on top of the row do not fit four waves per SIMD -- two launches)
#undef MSQ_AQR
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
            return MSQ_OK;
        }
        const int rc = msq_mxops_stats_x_(Xv, x_bf16, vmean, vstd, M, K, 1, block, status_flag, stream);
        if (rc) return rc;
        A.vmean = vmean; A.vstd = vstd;
    }
    const int64_t nblocks = M * A.nblk;
    const dim3 grid((unsigned)((nblocks + 255) / 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    // RM = 0: round-to-nearest (half away) specialised, through the hardware converts when both formats have
    // one (e4m3 / e5m2 / e2m1); -1: any rounding mode, arithmetic codec
    const bool hw = (rmode == 0) && hw_codec_kind(A.fi) && hw_codec_kind(A.fo);
#define MSQ_AQ(BS) do { if (x_bf16) { if (hw) hipLaunchKernelGGL((k_act_quant<BS, 0, 1, true>), grid, blk, 0, st, X, (uint16_t*)Xq, A); \
                                         else if (rmode == 0) hipLaunchKernelGGL((k_act_quant<BS, 0, 0, true>), grid, blk, 0, st, X, (uint16_t*)Xq, A); \
                                         else hipLaunchKernelGGL((k_act_quant<BS, -1, 0, true>), grid, blk, 0, st, X, (uint16_t*)Xq, A); } \
                        else if (hw) hipLaunchKernelGGL((k_act_quant<BS, 0, 1>), grid, blk, 0, st, X, (uint16_t*)Xq, A); \
                        else if (rmode == 0) hipLaunchKernelGGL((k_act_quant<BS, 0, 0>), grid, blk, 0, st, X, (uint16_t*)Xq, A); \
                        else hipLaunchKernelGGL((k_act_quant<BS, -1, 0>), grid, blk, 0, st, X, (uint16_t*)Xq, A); } while (0)
    switch (block) {
        case 16: MSQ_AQ(16); break;
        case 32: MSQ_AQ(32); break;
        default: MSQ_AQ(64); break;
    }
#undef MSQ_AQ
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

extern "C" int msq_act_quant_bf16(const float* X, void* Xq, int* status_flag, void* workspace, int64_t workspace_bytes,
                                  int64_t M, int64_t K, int block, int inlier_fmt, int outlier_fmt, int inlier_scale_bits,
                                  int outlier_scale_bits, float std_dev, int rmode, int flush_fp32_subnorms, int variant,
                                  void* stream) {
    return act_quant_impl(X, 0, Xq, status_flag, workspace, workspace_bytes, M, K, block, inlier_fmt, outlier_fmt, inlier_scale_bits,
                          outlier_scale_bits, std_dev, rmode, flush_fp32_subnorms, variant, stream);
}

// the same with bfloat16 activations (no cast pass in front: 2 + 2 bytes per element instead of 2 + 4 + 4 + 2)
extern "C" int msq_act_quant_bf16_x16(const void* X, void* Xq, int* status_flag, void* workspace, int64_t workspace_bytes,
                                      int64_t M, int64_t K, int block, int inlier_fmt, int outlier_fmt, int inlier_scale_bits,
                                      int outlier_scale_bits, float std_dev, int rmode, int flush_fp32_subnorms, int variant,
                                      void* stream) {
    return act_quant_impl(X, 1, Xq, status_flag, workspace, workspace_bytes, M, K, block, inlier_fmt, outlier_fmt, inlier_scale_bits,
                          outlier_scale_bits, std_dev, rmode, flush_fp32_subnorms, variant, stream);
}
