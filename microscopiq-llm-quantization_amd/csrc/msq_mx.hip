// msq_mx.hip -- operand packing for the MX-native W4A8 GEMM (msq_gemm.hip k_mxgemm): plain OCP-MX block
// quantisation (number_system/mx/mx_ops.py:332-457 _quantize_mx as its PYTHON path computes it: shared exponent = floor(torch.log2(max)),
// block 32 along K, round to nearest = half away) emitting CODES + E8M0 scale bytes instead of fake-quant values:
//   activations  X [M,K] f32 -> e4m3 codes [M][K] (row-major bytes) + scales [M][K/32]
//   weights      W [N,K] f32 -> e2m1 codes in the operand order of v_mfma_scale_f32_16x16x128_f8f6f4
//                (tile = 64 n x 128 k; slot nf = 16 n: lane (n % 16, (k % 128) / 32) holds 32 k = 16 bytes)
//                + one scale byte per (lane, nf)
// The codes come from the hardware converts (RNE of x | 1ulp == round half away, see msq_outlier_core.h).
//   fake-quant values Wq [N,K] f32 (MicroScopiQ inliers + outliers, GPTQ output, ...) -> e4m3 codes in the fp8
//                operand order (same tile; lane (n % 16, kg) holds k = 16 kg .. +15 and 64 + 16 kg .. +15 = 2 x 16
//                bytes, stored as two 1 KiB half-slots) + one scale byte per (lane, nf), scale rule of MSQ-U1
//                (msq_pack_unified.hip); every code is decoded back and compared: MSQ_STATUS_INEXACT otherwise.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_mx_pack_core.h"
#include "msq_host.h"

using namespace msq;

extern "C" void msq_set_error_(const char* msg);

namespace {

// A wave owns 64 consecutive blocks of 32 floats = one contiguous 8 KiB run: coalesced 16-byte loads, transpose
// through LDS (row stride 36 words), one block per lane.  FP4: weights, codes scattered into the MFMA tile order;
// otherwise activations, e4m3 codes written back row-major through the same LDS tile.
// MODE 0: activations (e4m3, row-major), 1: weights (e2m1, operand order), 2: exact values (e4m3, operand order)
// XBF16: the source holds bfloat16 (activations of a bf16 model: every bf16 is an fp32 value, same results as casting first)
// AF6 (activations only): 3 / 2 = quantise to fp6_e3m2 / fp6_e2m3 VALUES with the fp6 block scale and store them as e4m3
// codes (every fp6 value is an e4m3 normal: exact) -- the W6A6 product of the fp6 spec on the fp8 activation operand.
template <int MODE, bool XBF16 = false, int AF6 = 0>
__global__ void __launch_bounds__(256)
k_mx_pack(const float* __restrict__ src, uint8_t* __restrict__ codes, uint8_t* __restrict__ scales, int64_t rows, int64_t K,
          int flush, int* status_flag) {
    constexpr bool FP4 = (MODE == 1), VAL = (MODE == 2);
    constexpr int BS = 32, LDS_STRIDE = BS + 4;
    __shared__ __attribute__((aligned(16))) float tile[4][64 * LDS_STRIDE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t nblk = K / BS, nblocks = rows * nblk;
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (g0 >= nblocks) return;
    const bool full = (g0 + 64 <= nblocks);
    const int64_t g = g0 + lane;
    float* tl = tile[wv];
    float a[BS];
    const uint16_t* srch = reinterpret_cast<const uint16_t*>(src);
    if (full) {
        if (XBF16) {
            const uint4* s8 = reinterpret_cast<const uint4*>(srch + g0 * BS);     // 8 bf16 per 16-byte load
#pragma unroll
            for (int t = 0; t < BS / 8; ++t) {
                const int f = lane + 64 * t;
                const int row = f / (BS / 8), c8 = f % (BS / 8);
                const uint4 v = s8[f];
                *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c8 * 8) = make_float4(u2f(v.x << 16), u2f(v.x & 0xFFFF0000u), u2f(v.y << 16), u2f(v.y & 0xFFFF0000u));
                *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c8 * 8 + 4) = make_float4(u2f(v.z << 16), u2f(v.z & 0xFFFF0000u), u2f(v.w << 16), u2f(v.w & 0xFFFF0000u));
            }
        } else {
        const float4* s4 = reinterpret_cast<const float4*>(src + g0 * BS);
#pragma unroll
        for (int t = 0; t < BS / 4; ++t) {
            const int f = lane + 64 * t;
            const int row = f / (BS / 4), c4 = f % (BS / 4);
            *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c4 * 4) = s4[f];
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
        for (int b = 0; b < BS; ++b) a[b] = (g < nblocks) ? (XBF16 ? u2f((uint32_t)srch[g * BS + b] << 16) : src[g * BS + b]) : 0.f;
    }
    int status = 0;
    // shared exponent of the block as the reference's PYTHON path derives it -- floor(torch.log2(max|a|)), mx_ops.py:66-77: these
    // operands stand in for `_quantize_mx` of the CPU fake-quant (custom_cuda = False) -- which is the exponent field of the maximum
    // except just under a power of two (msq_device.h biased_exp_py)
    uint32_t mbits_ = 0u;
#pragma unroll
    for (int b = 0; b < BS; ++b) { const uint32_t t = f2u(a[b]) & 0x7FFFFFFFu; mbits_ = t > mbits_ ? t : mbits_; }
    const int se = biased_exp_py(mbits_);
    const bool fl = (se == 0) && flush && !VAL;
    int sb = mx_scale_byte(se, FP4 ? 2 : (AF6 == 3 ? 4 : (AF6 == 2 ? 2 : 8)), status);
    if (VAL) {                                                   // MSQ-U1 rule: max |v| 2^-s in [256, 448] or (448, 512) -> s + 1
        float mx = 0.f;
#pragma unroll
        for (int b = 0; b < BS; ++b) { const float t = __builtin_fabsf(a[b]); mx = t > mx ? t : mx; }
        int su = 0;
        if (mx > 0.f && se != 255) {
            su = ilog2f(mx) - 8;
            if (__builtin_ldexpf(mx, -su) > 448.f) su += 1;
        }
        su = su < -126 ? -126 : su;
        if (su > 127) { su = 127; status |= MSQ_STATUS_INEXACT; }
        sb = su + 127;
    }
    const float s_op = AF6 ? 1.0f : u2f((uint32_t)sb << 23);    // the converts read the exponent field only
    const float bound = __builtin_ldexpf(448.f, sb - 127);       // e4m3 max_norm x scale (exact)
    uint32_t cw[FP4 ? 4 : 8];
    // the reference's floor(|x| + 0.5) in float32 (msq_device.h half_away_quirk_bits): half the smallest step of the grid x the scale
    // (e2m1: 2^-2, e4m3: 2^-10; the fp6 grids are met after the exact scaling: e3m2 2^-5, e2m3 2^-4)
    const uint32_t qb = VAL ? 0xFFFFFFFFu : half_away_quirk_bits(AF6 ? (AF6 == 3 ? -5 : -4) : sb - 127 - (FP4 ? 2 : 10));
#pragma unroll
    for (int p = 0; p < BS / 2; ++p) {
        typedef short v2s_t __attribute__((ext_vector_type(2)));
        float x0 = fl ? 0.f : (VAL ? a[2 * p] : sticky_half_away(a[2 * p], qb)), x1 = fl ? 0.f : (VAL ? a[2 * p + 1] : sticky_half_away(a[2 * p + 1], qb));
        if (AF6) {                                               // fp6 value of a / 2^(sb - 127) (arithmetic codec), then an exact convert
            const int sh = 127 - (sb == 255 ? 127 : sb);
            const float y0 = __builtin_ldexpf(a[2 * p], sh), y1 = __builtin_ldexpf(a[2 * p + 1], sh);
            x0 = fl ? 0.f : quant_bits(y0, AF6 == 3 ? 4 : 5, AF6, AF6 == 3 ? 28.0f : 7.5f, 0, true, true);
            x1 = fl ? 0.f : quant_bits(y1, AF6 == 3 ? 4 : 5, AF6, AF6 == 3 ? 28.0f : 7.5f, 0, true, true);
            if (!fl && (f2u(y0) & 0x7FFFFFFFu) == qb) x0 = __builtin_copysignf(AF6 == 3 ? 0.0625f : 0.125f, y0);
            if (!fl && (f2u(y1) & 0x7FFFFFFFu) == qb) x1 = __builtin_copysignf(AF6 == 3 ? 0.0625f : 0.125f, y1);
        }
        if (FP4) {
            uint32_t w = (p & 3) ? cw[p >> 2] : 0u;
            if ((p & 3) == 0) w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, x0, x1, s_op, 0);
            else if ((p & 3) == 1) w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, x0, x1, s_op, 1);
            else if ((p & 3) == 2) w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, x0, x1, s_op, 2);
            else w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, x0, x1, s_op, 3);
            cw[p >> 2] = w;
        } else {
            x0 = __builtin_amdgcn_fmed3f(x0, -bound, bound); x1 = __builtin_amdgcn_fmed3f(x1, -bound, bound);   // e4m3 does not saturate
            v2s_t cur = __builtin_bit_cast(v2s_t, (p & 1) ? cw[p >> 1] : 0u);
            if ((p & 1) == 0) cur = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(cur, x0, x1, s_op, false);
            else cur = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(cur, x0, x1, s_op, true);
            cw[p >> 1] = __builtin_bit_cast(uint32_t, cur);
            if (VAL) {                                           // decode as the hardware does and compare
                const uint32_t u0 = f2u(a[2 * p]), u1 = f2u(a[2 * p + 1]);
                const uint32_t d = ((p & 1) == 0) ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(cw[p >> 1], s_op, false))
                                                  : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(cw[p >> 1], s_op, true));
                if ((((u0 | u1) & 0xFFFFu) != 0u || d != ((u0 >> 16) | (u1 & 0xFFFF0000u))) && se != 255) status |= MSQ_STATUS_INEXACT;
            }
        }
    }
    // -0 codes are fine for the MFMA (no OR-combination here)
    if (g < nblocks) {
        const int64_t r = g / nblk, kb = g % nblk;
        if (FP4) {
            const int64_t KT = K / 128;
            const int64_t t = (r / 64) * KT + kb / 4;
            const int nf = (int)((r % 64) / 16), ln = (int)((kb % 4) * 16 + (r % 16));
            *reinterpret_cast<uint4*>(codes + ((t * 4 + nf) * 64 + ln) * 16) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            scales[(t * 64 + ln) * 4 + nf] = (uint8_t)sb;
        } else if (VAL) {
            // block kb % 4 = b holds k = 32 b .. 32 b + 31 of the 128-k tile: its first 16 codes belong to lane
            // (r, kg = 2 (b & 1)), the next 16 to kg + 1, both in half b >> 1; the scale byte to lane (r, b)
            const int64_t KT = K / 128;
            const int64_t t = (r / 64) * KT + kb / 4;
            const int nf = (int)((r % 64) / 16), b = (int)(kb % 4), r16 = (int)(r % 16);
            uint8_t* slot = codes + (((t * 4 + nf) * 2 + (b >> 1)) * 64) * 16;
            *reinterpret_cast<uint4*>(slot + ((2 * (b & 1)) * 16 + r16) * 16) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            *reinterpret_cast<uint4*>(slot + ((2 * (b & 1) + 1) * 16 + r16) * 16) = make_uint4(cw[4], cw[5], cw[6], cw[7]);
            scales[(t * 64 + b * 16 + r16) * 4 + nf] = (uint8_t)sb;
        } else {
            scales[g] = (uint8_t)sb;
        }
    }
    if (MODE == 0) {
        if (full) {                                             // 8 words per block back through LDS, coalesced 16-byte stores
            constexpr int HS = 12;                              // words per row (8 used; 48-byte rows keep 16-byte alignment)
            __builtin_amdgcn_wave_barrier();
            uint32_t* tw = reinterpret_cast<uint32_t*>(tl);
            *reinterpret_cast<uint4*>(tw + lane * HS) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            *reinterpret_cast<uint4*>(tw + lane * HS + 4) = make_uint4(cw[FP4 ? 0 : 4], cw[FP4 ? 1 : 5], cw[FP4 ? 2 : 6], cw[FP4 ? 3 : 7]);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            uint4* dst = reinterpret_cast<uint4*>(codes + g0 * BS);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int f = lane + 64 * t;                    // 16-byte piece index: block f / 2, half f % 2
                dst[f] = *reinterpret_cast<const uint4*>(tw + (f >> 1) * HS + (f & 1) * 4);
            }
        } else if (g < nblocks) {
#pragma unroll
            for (int w = 0; w < 8; ++w) reinterpret_cast<uint32_t*>(codes + g * BS)[w] = cw[FP4 ? 0 : w];
        }
    }
    if (status && status_flag) atomicOr(status_flag, status);
}

// MX-FP6 weights (fp6_e3m2 / fp6_e2m3, formats.py:76-79) in the fp6 operand order of the scaled MFMA: lane (n % 16, kg) of
// slot nf holds k = 32 kg .. +31 of the 128-k tile as 32 six-bit codes, little endian, in 24 bytes -- stored as a 16-byte
// piece and an 8-byte piece, 1.5 KiB per (tile, nf) = exactly 6 bits per weight (+ the E8M0 scale byte per 32:
// 6.25 bits per weight).  The codes are those of the arithmetic element codec (quant_bits, the routine
// msq_quantize_mx_by_tile uses): decoded, they ARE the oracle's quantize_mx values.
template <int EB> MSQ_D uint32_t fp6_code(float q) {
    constexpr int MB = 5 - EB, BIAS = (1 << (EB - 1)) - 1;
    const uint32_t u = f2u(q), s = (u >> 31) << 5, a = u & 0x7FFFFFFFu;
    if (a == 0u) return s;
    const int E = (int)(a >> 23) - 127;
    if (E >= 1 - BIAS) return s | ((uint32_t)(E + BIAS) << MB) | ((a >> (23 - MB)) & ((1u << MB) - 1u));
    return s | (uint32_t)__builtin_ldexpf(u2f(a), BIAS - 1 + MB);       // subnormal: m x 2^(1 - BIAS - MB), m = 1 .. 2^MB - 1
}

template <int EB>
__global__ void __launch_bounds__(256)
k_mx_pack_w6(const float* __restrict__ src, uint8_t* __restrict__ codes, uint8_t* __restrict__ scales, int64_t rows, int64_t K,
             int flush, int* status_flag) {
    constexpr int BS = 32, LDS_STRIDE = BS + 4;
    constexpr int EMAX = (EB == 3) ? 4 : 2, MBITS = (EB == 3) ? 4 : 5;
    constexpr float MAXN = (EB == 3) ? 28.0f : 7.5f;
    __shared__ __attribute__((aligned(16))) float tile[4][64 * LDS_STRIDE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t nblk = K / BS, nblocks = rows * nblk;            // rows % 64 == 0 and K % 128 == 0: every wave is full
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (g0 >= nblocks) return;
    const int64_t g = g0 + lane;
    float* tl = tile[wv];
    float a[BS];
    const float4* s4 = reinterpret_cast<const float4*>(src + g0 * BS);
#pragma unroll
    for (int t = 0; t < BS / 4; ++t) {
        const int f = lane + 64 * t;
        const int row = f / (BS / 4), c4 = f % (BS / 4);
        *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c4 * 4) = s4[f];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int c = 0; c < BS / 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(tl + lane * LDS_STRIDE + c * 4);
        a[c * 4 + 0] = v.x; a[c * 4 + 1] = v.y; a[c * 4 + 2] = v.z; a[c * 4 + 3] = v.w;
    }
    int status = 0;
    uint32_t mbits_ = 0u;                                        // (shared exponent: Python-path rule, as k_mx_pack)
#pragma unroll
    for (int b = 0; b < BS; ++b) { const uint32_t t = f2u(a[b]) & 0x7FFFFFFFu; mbits_ = t > mbits_ ? t : mbits_; }
    const int se = biased_exp_py(mbits_);
    const bool fl = (se == 0) && flush;
    const int sb = mx_scale_byte(se, EMAX, status);
    uint32_t w[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    if (sb != 255) {
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            const float si = fl ? 0.f : __builtin_ldexpf(a[b], 127 - sb);               // a / 2^(sb - 127), exact
            float q = quant_bits(si, MBITS, EB, MAXN, 0, true, true);
            // the reference's floor(|x| + 0.5) in float32 (msq_device.h half_away_quirk_bits): pred(half the smallest step) -> one step
            if ((f2u(si) & 0x7FFFFFFFu) == half_away_quirk_bits(EB == 3 ? -5 : -4)) q = __builtin_copysignf(EB == 3 ? 0.0625f : 0.125f, si);
            const uint32_t cd = fp6_code<EB>(q);
            const int bit = 6 * b;
            w[bit >> 5] |= cd << (bit & 31);
            if ((bit & 31) > 26) w[(bit >> 5) + 1] |= cd >> (32 - (bit & 31));
        }
    }
    const int64_t r = g / nblk, kb = g % nblk;
    const int64_t KT = K / 128;
    const int64_t t = (r / 64) * KT + kb / 4;
    const int nf = (int)((r % 64) / 16), ln = (int)((kb % 4) * 16 + (r % 16));
    uint8_t* slot = codes + (t * 4 + nf) * 1536;
    *reinterpret_cast<uint4*>(slot + ln * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint2*>(slot + 1024 + ln * 8) = make_uint2(w[4], w[5]);
    scales[(t * 64 + ln) * 4 + nf] = (uint8_t)sb;
    if (status && status_flag) atomicOr(status_flag, status);
}

// Activations (MODE 0 of k_mx_pack: e4m3 codes row-major + one scale byte per block), one lane per EIGHT consecutive values:
// four neighbouring lanes hold a block, its largest exponent crosses them by two quad permutes, every lane converts its own
// eight values and stores 8 bytes -- 32-byte loads, 8-byte stores, no LDS transpose.  Same scale rule, same converts, same
// flags as k_mx_pack<0> (tests/test_gpu_n3_mx_w4a8.py::test_mx_act_pack_vec_equals_block_kernel).
template <int XS>      // source: 0 float32, 1 bfloat16, 2 float16 (every half value is an fp32 value: the codes of casting first)
__global__ void __launch_bounds__(256)
k_mx_pack_a8_vec(const void* __restrict__ src, uint8_t* __restrict__ codes, uint8_t* __restrict__ scales, int64_t n8, int flush,
                 int* status_flag) {
    typedef short v2s_t __attribute__((ext_vector_type(2)));
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;                                         // n8 is a multiple of 4: quads are never split
    float a[8];
    if (XS == 1) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[i];
        a[0] = u2f(v.x << 16); a[1] = u2f(v.x & 0xFFFF0000u); a[2] = u2f(v.y << 16); a[3] = u2f(v.y & 0xFFFF0000u);
        a[4] = u2f(v.z << 16); a[5] = u2f(v.z & 0xFFFF0000u); a[6] = u2f(v.w << 16); a[7] = u2f(v.w & 0xFFFF0000u);
    } else if (XS == 2) {
        union { uint4 u; _Float16 h[8]; } r;
        r.u = reinterpret_cast<const uint4*>(src)[i];
#pragma unroll
        for (int b = 0; b < 8; ++b) a[b] = (float)r.h[b];
    } else {
        const float4 v0 = reinterpret_cast<const float4*>(src)[2 * i], v1 = reinterpret_cast<const float4*>(src)[2 * i + 1];
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    }
    uint32_t cw[2];
    int sb, status = 0;
    mx_pack8_e4m3_quad(a, cw, sb, flush, status);                // msq_mx_pack_core.h
    reinterpret_cast<uint2*>(codes)[i] = make_uint2(cw[0], cw[1]);
    if ((threadIdx.x & 3) == 0) scales[i >> 2] = (uint8_t)sb;
    if (status && status_flag) atomicOr(status_flag, status);
}

}  // namespace

static msq_host::TuneKey g_pack_block("MSQ_MX_PACK_BLOCK");   // != 0: one lane per block (k_mx_pack<0>) instead of the 8-values-per-lane packer
extern "C" int msq_set_tuning_mx_(const char* key, int value) { return g_pack_block.set_if(key, value); }

extern "C" int msq_mx_pack_a8(const float* X, void* codes, void* scales, int* status_flag, int64_t M, int64_t K,
                              int flush_fp32_subnorms, void* stream) {
    if (M < 0 || K < 0) { msq_set_error_("msq_mx_pack_a8: negative size"); return MSQ_ERR_BAD_ARG; }
    if (M == 0 || K == 0) return MSQ_OK;
    if (K % 128) { msq_set_error_("msq_mx_pack_a8: K must be a multiple of 128"); return MSQ_ERR_UNSUPPORTED; }
    if (!X || !codes || !scales) { msq_set_error_("msq_mx_pack_a8: null buffer"); return MSQ_ERR_BAD_ARG; }
    const int64_t nblocks = M * (K / 32);
    if ((((uintptr_t)X | (uintptr_t)codes) & 15) == 0 && g_pack_block.value(0) == 0)
        hipLaunchKernelGGL((k_mx_pack_a8_vec<0>), dim3((unsigned)((nblocks * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const void*)X,
                           (uint8_t*)codes, (uint8_t*)scales, nblocks * 4, flush_fp32_subnorms, status_flag);
    else
    hipLaunchKernelGGL((k_mx_pack<0>), dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X,
                       (uint8_t*)codes, (uint8_t*)scales, M, K, flush_fp32_subnorms, status_flag);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

extern "C" int msq_mx_pack_a6(const float* X, void* codes, void* scales, int* status_flag, int64_t M, int64_t K, int a_format,
                              int flush_fp32_subnorms, void* stream) {
    if (M < 0 || K < 0) { msq_set_error_("msq_mx_pack_a6: negative size"); return MSQ_ERR_BAD_ARG; }
    if (M == 0 || K == 0) return MSQ_OK;
    if (K % 128) { msq_set_error_("msq_mx_pack_a6: K must be a multiple of 128"); return MSQ_ERR_UNSUPPORTED; }
    if (!X || !codes || !scales) { msq_set_error_("msq_mx_pack_a6: null buffer"); return MSQ_ERR_BAD_ARG; }
    if (a_format != MSQ_FMT_FP6_E3M2 && a_format != MSQ_FMT_FP6_E2M3) { msq_set_error_("msq_mx_pack_a6: a_format must be MSQ_FMT_FP6_E3M2 or MSQ_FMT_FP6_E2M3"); return MSQ_ERR_BAD_ARG; }
    const int64_t nblocks = M * (K / 32);
    const dim3 grid((unsigned)((nblocks + 255) / 256)), blk(256);
    if (a_format == MSQ_FMT_FP6_E3M2)
        hipLaunchKernelGGL((k_mx_pack<0, false, 3>), grid, blk, 0, (hipStream_t)stream, X, (uint8_t*)codes, (uint8_t*)scales, M, K, flush_fp32_subnorms, status_flag);
    else
        hipLaunchKernelGGL((k_mx_pack<0, false, 2>), grid, blk, 0, (hipStream_t)stream, X, (uint8_t*)codes, (uint8_t*)scales, M, K, flush_fp32_subnorms, status_flag);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

extern "C" int msq_mx_pack_w4(const float* W, void* codes, void* scales, int* status_flag, int64_t N, int64_t K,
                              int flush_fp32_subnorms, void* stream) {
    if (N <= 0 || K <= 0 || (N % 64) || (K % 128)) { msq_set_error_("msq_mx_pack_w4: N must be a multiple of 64 and K of 128"); return MSQ_ERR_UNSUPPORTED; }
    if (!W || !codes || !scales) { msq_set_error_("msq_mx_pack_w4: null buffer"); return MSQ_ERR_BAD_ARG; }
    const int64_t nblocks = N * (K / 32);
    hipLaunchKernelGGL((k_mx_pack<1>), dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       (uint8_t*)codes, (uint8_t*)scales, N, K, flush_fp32_subnorms, status_flag);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

extern "C" int msq_mx_pack_w8(const float* Wq, void* codes, void* scales, int* status_flag, int64_t N, int64_t K, void* stream) {
    if (N <= 0 || K <= 0 || (N % 64) || (K % 128)) { msq_set_error_("msq_mx_pack_w8: N must be a multiple of 64 and K of 128"); return MSQ_ERR_UNSUPPORTED; }
    if (!Wq || !codes || !scales) { msq_set_error_("msq_mx_pack_w8: null buffer"); return MSQ_ERR_BAD_ARG; }
    const int64_t nblocks = N * (K / 32);
    hipLaunchKernelGGL((k_mx_pack<2>), dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Wq,
                       (uint8_t*)codes, (uint8_t*)scales, N, K, 0, status_flag);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

extern "C" int msq_mx_pack_w6(const float* W, void* codes, void* scales, int* status_flag, int64_t N, int64_t K, int w_format,
                              int flush_fp32_subnorms, void* stream) {
    if (N <= 0 || K <= 0 || (N % 64) || (K % 128)) { msq_set_error_("msq_mx_pack_w6: N must be a multiple of 64 and K of 128"); return MSQ_ERR_UNSUPPORTED; }
    if (!W || !codes || !scales) { msq_set_error_("msq_mx_pack_w6: null buffer"); return MSQ_ERR_BAD_ARG; }
    if (w_format != MSQ_FMT_FP6_E3M2 && w_format != MSQ_FMT_FP6_E2M3) { msq_set_error_("msq_mx_pack_w6: w_format must be MSQ_FMT_FP6_E3M2 or MSQ_FMT_FP6_E2M3"); return MSQ_ERR_BAD_ARG; }
    const int64_t nblocks = N * (K / 32);
    const dim3 grid((unsigned)((nblocks + 255) / 256)), blk(256);
    if (w_format == MSQ_FMT_FP6_E3M2)
        hipLaunchKernelGGL((k_mx_pack_w6<3>), grid, blk, 0, (hipStream_t)stream, W, (uint8_t*)codes, (uint8_t*)scales, N, K, flush_fp32_subnorms, status_flag);
    else
        hipLaunchKernelGGL((k_mx_pack_w6<2>), grid, blk, 0, (hipStream_t)stream, W, (uint8_t*)codes, (uint8_t*)scales, N, K, flush_fp32_subnorms, status_flag);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

// as msq_mx_pack_a8 with bfloat16 activations (x_dtype of include/msq.h: 2)
extern "C" int msq_mx_pack_a8_bf16(const void* X, void* codes, void* scales, int* status_flag, int64_t M, int64_t K,
                                   int flush_fp32_subnorms, void* stream) {
    if (M < 0 || K < 0) { msq_set_error_("msq_mx_pack_a8_bf16: negative size"); return MSQ_ERR_BAD_ARG; }
    if (M == 0 || K == 0) return MSQ_OK;
    if (K % 128) { msq_set_error_("msq_mx_pack_a8_bf16: K must be a multiple of 128"); return MSQ_ERR_UNSUPPORTED; }
    if (!X || !codes || !scales) { msq_set_error_("msq_mx_pack_a8_bf16: null buffer"); return MSQ_ERR_BAD_ARG; }
    const int64_t nblocks = M * (K / 32);
    if ((((uintptr_t)X | (uintptr_t)codes) & 15) == 0 && g_pack_block.value(0) == 0)
        hipLaunchKernelGGL((k_mx_pack_a8_vec<1>), dim3((unsigned)((nblocks * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X,
                           (uint8_t*)codes, (uint8_t*)scales, nblocks * 4, flush_fp32_subnorms, status_flag);
    else
    hipLaunchKernelGGL((k_mx_pack<0, true>), dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)X,
                       (uint8_t*)codes, (uint8_t*)scales, M, K, flush_fp32_subnorms, status_flag);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

// as msq_mx_pack_a8 with float16 activations (an fp16 model, llm/llama.py:33): no cast pass in front of the packer
extern "C" int msq_mx_pack_a8_f16(const void* X, void* codes, void* scales, int* status_flag, int64_t M, int64_t K,
                                  int flush_fp32_subnorms, void* stream) {
    if (M < 0 || K < 0) { msq_set_error_("msq_mx_pack_a8_f16: negative size"); return MSQ_ERR_BAD_ARG; }
    if (M == 0 || K == 0) return MSQ_OK;
    if (K % 128) { msq_set_error_("msq_mx_pack_a8_f16: K must be a multiple of 128"); return MSQ_ERR_UNSUPPORTED; }
    if (!X || !codes || !scales) { msq_set_error_("msq_mx_pack_a8_f16: null buffer"); return MSQ_ERR_BAD_ARG; }
    if ((((uintptr_t)X | (uintptr_t)codes) & 15) != 0) { msq_set_error_("msq_mx_pack_a8_f16: buffers must be 16-byte aligned"); return MSQ_ERR_UNSUPPORTED; }
    const int64_t nblocks = M * (K / 32);
    hipLaunchKernelGGL((k_mx_pack_a8_vec<2>), dim3((unsigned)((nblocks * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X,
                       (uint8_t*)codes, (uint8_t*)scales, nblocks * 4, flush_fp32_subnorms, status_flag);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

namespace {
// float16 -> bfloat16 (round to nearest even through fp32: what Tensor.to(torch.bfloat16) gives), 16 bytes per lane in and out
__global__ void __launch_bounds__(256) k_cast_f16_bf16(const uint4* __restrict__ x, uint4* __restrict__ y, int64_t n8) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        union { uint4 u; _Float16 h[8]; } r;
        union { uint4 u; __bf16 b[8]; } o;
        r.u = x[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) o.b[e] = (__bf16)(float)r.h[e];
        y[i] = o.u;
    }
}
__global__ void __launch_bounds__(256) k_cast_f16_bf16_tail(const _Float16* __restrict__ x, __bf16* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = (__bf16)(float)x[i];
}
}  // namespace

// the activation cast in front of the bf16 MFMA for an fp16 model at prefill sizes (the decode kernels convert while loading)
extern "C" int msq_cast_f16_bf16(const void* x, void* y, int64_t n, void* stream) {
    if (n < 0) { msq_set_error_("msq_cast_f16_bf16: negative size"); return MSQ_ERR_BAD_ARG; }
    if (n == 0) return MSQ_OK;
    if (!x || !y) { msq_set_error_("msq_cast_f16_bf16: null buffer"); return MSQ_ERR_BAD_ARG; }
    int64_t done = 0;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) == 0 && n >= 8) {
        const int64_t n8 = n / 8;
        int64_t g = (n8 + 255) / 256; if (g > 256 * 32) g = 256 * 32;
        hipLaunchKernelGGL(k_cast_f16_bf16, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, n8);
        done = n8 * 8;
    }
    if (done < n)
        hipLaunchKernelGGL(k_cast_f16_bf16_tail, dim3((unsigned)((n - done + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)x + done, (__bf16*)y + done, n - done);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Measurement aid (bench.py `decode_cold.frac_of_read_stream`): the plain read stream of this chip over `bytes` of a buffer -- 16-byte
// loads, `inflight` (4 or 8) loads in flight per lane, grid-stride over `blocks` workgroups of 256 -- so that a weight-streaming
// kernel's rate can be put beside what a read of the same bytes with the same grid reaches on the same box at the same moment.  The
// loaded words are xor-folded; `sink` (4 bytes) is written only if the fold hits a constant, i.e. never in practice.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ void __launch_bounds__(256) k_read_probe(const probe_u32x4* __restrict__ p, uint32_t* __restrict__ sink, int64_t n16) {
    const int64_t nthreads = (int64_t)gridDim.x * 256;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    int64_t i = tid;
    for (; i + (U - 1) * nthreads < n16; i += U * nthreads) {
        probe_u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * nthreads);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    for (; i < n16; i += nthreads) { const probe_u32x4 v = __builtin_nontemporal_load(p + i); acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
    if (acc == 0x12345678u) sink[0] = acc;
}
}  // namespace

extern "C" int msq_read_stream_probe(const void* buf, int64_t bytes, int blocks, int inflight, void* sink, void* stream) {
    if (bytes < 0 || blocks <= 0) { msq_set_error_("msq_read_stream_probe: bad size"); return MSQ_ERR_BAD_ARG; }
    if (bytes == 0) return MSQ_OK;
    if (!buf || !sink || (reinterpret_cast<uintptr_t>(buf) & 15)) { msq_set_error_("msq_read_stream_probe: null or unaligned buffer"); return MSQ_ERR_BAD_ARG; }
    const int64_t n16 = bytes / 16;
    if (inflight >= 8) hipLaunchKernelGGL(k_read_probe<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const probe_u32x4*)buf, (uint32_t*)sink, n16);
    else hipLaunchKernelGGL(k_read_probe<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const probe_u32x4*)buf, (uint32_t*)sink, n16);
    if (hipGetLastError() != hipSuccess) { msq_set_error_("msq_read_stream_probe: launch failed"); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}
