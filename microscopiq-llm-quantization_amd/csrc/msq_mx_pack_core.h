// msq_mx_pack_core.h -- the MX-FP8 (e4m3, block 32, 8-bit scale) activation pack of eight consecutive values per lane, shared by the
// stand-alone packer (msq_mx.hip k_mx_pack_a8_vec) and by the producers that pack what they compute (msq_vec.hip: RMSNorm, silu x up):
// one code path, so "fused" and "producer, then packer" give the same bytes by construction.
#pragma once
#include "msq_device.h"

namespace msq {

// shared scale byte of a block (cpp/shared_exp.cuh:14-53 with scale_bits 8): biased max exponent - elem emax,
// clamped to [0, 254]; 255 (NaN) when the block holds Inf / NaN
MSQ_D int mx_scale_byte(int max_biased_exp, int elem_emax, int& status) {
    if (max_biased_exp == 255) { status |= MSQ_STATUS_NAN; return 255; }
    int e = max_biased_exp - elem_emax;
    if (e - 127 > 127) { status |= MSQ_STATUS_NAN; return 255; }
    if (e - 127 < -127) e = 0;
    return e;
}

// Four neighbouring lanes (a quad) hold one block of 32 values, eight each: the largest magnitude crosses the quad by two quad
// permutes, every lane converts its own eight values (v_cvt_scalef32_pk_fp8_f32, sticky bit for half-away rounding, clamp first: the
// convert does not saturate).  cw = the lane's 8 code bytes, sb = the block's scale byte (the same in the four lanes).
// EVERY lane of the quad must call this (DPP reads the neighbours' registers).
MSQ_D void mx_pack8_e4m3_quad(const float (&a)[8], uint32_t (&cw)[2], int& sb, int flush, int& status) {
    typedef short v2s_t __attribute__((ext_vector_type(2)));
    uint32_t mag = 0u;
#pragma unroll
    for (int b = 0; b < 8; ++b) { const uint32_t t = f2u(a[b]) & 0x7FFFFFFFu; mag = t > mag ? t : mag; }   // whole magnitude: the Python-path exponent needs the significand
    {
        const uint32_t o1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mag, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
        mag = o1 > mag ? o1 : mag;
        const uint32_t o2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mag, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
        mag = o2 > mag ? o2 : mag;
    }
    const int se = biased_exp_py(mag);
    const bool fl = (se == 0) && flush;
    sb = mx_scale_byte(se, 8, status);
    const float s_op = u2f((uint32_t)sb << 23);                 // the converts read the exponent field only
    const float bound = __builtin_ldexpf(448.f, sb - 127);       // e4m3 max_norm x scale (exact)
    const uint32_t qb = half_away_quirk_bits(sb - 127 - 10);     // e4m3: smallest subnormal 2^-9, half of it 2^-10, x the block scale
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float x0 = fl ? 0.f : sticky_half_away(a[2 * p], qb), x1 = fl ? 0.f : sticky_half_away(a[2 * p + 1], qb);
        x0 = __builtin_amdgcn_fmed3f(x0, -bound, bound); x1 = __builtin_amdgcn_fmed3f(x1, -bound, bound);   // e4m3 does not saturate
        v2s_t cur = __builtin_bit_cast(v2s_t, (p & 1) ? cw[p >> 1] : 0u);
        if ((p & 1) == 0) cur = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(cur, x0, x1, s_op, false);
        else cur = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(cur, x0, x1, s_op, true);
        cw[p >> 1] = __builtin_bit_cast(uint32_t, cur);
    }
}

}  // namespace msq
