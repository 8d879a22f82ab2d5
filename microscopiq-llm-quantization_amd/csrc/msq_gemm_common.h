// msq_gemm_common.h -- device helpers shared by the fused dequant-GEMM kernels (msq_gemm.hip: k_qgemm3 / k_qgemv..., msq_gemm256.hip:
// k_qgemm256): operand typedefs, fragment dequantisation, packed-plane loads through buffer descriptors, the half-step convert,
// and the LDS-transposed epilogue of a wave tile.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"

using namespace msq;

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#ifndef MSQ_MX_PIN_READS
#define MSQ_MX_PIN_READS 0   /* 1: sched_barrier between the LDS prefetch of group mf + 1 and the MFMAs of group mf (hipcc sinks the reads below them); same-box A/B 99.7 vs 99.5 us: the second wave of the SIMD already covers the wait */
#endif
#ifndef MSQ_MX_XBUFS
#define MSQ_MX_XBUFS 3      /* 4 = four activation buffers staged three K-steps ahead: measured equal (101.0 vs 101.2 us), kept at 3 */
#endif
#ifndef MSQ_MXABL
#define MSQ_MXABL 0      /* ablation of k_mxgemm (scripts/experiments/build_mx_ablation.sh): 1 no LDS fragment reads, 2 no weight loads, 4 no LDS-DMA, 8 no barrier, 16 no stores */
#endif
#ifndef MSQ_ABL
#define MSQ_ABL 0
/* k_qgemm3 timing experiments (results are wrong by construction): 1 no LDS fragment reads, 2 no converts, 4 no packed loads,
   8 no activation staging, 16 no output stores.  Bits 32 ... 1024 emulate the instruction mix of 256-row block shapes inside the
   128-row kernel: 32 convert only fragments nf 0 / 1 (the MFMAs of nf 2 / 3 reuse them), 64 read every activation fragment
   twice, 128 issue every LDS-DMA piece twice (1024: the same bytes again instead of the neighbouring half tile), 256 load one
   packed slot per half-step instead of two, 512 four ds_write_b128 + twelve ds_read_b128 of "shared weight fragments" per K-step */
#endif
#ifndef MSQ_EXP_SKIPBAR
#define MSQ_EXP_SKIPBAR 0   /* timing experiment only (results are WRONG): no wait / barrier after the first K-step of every pair */
#endif
#ifndef MSQ_STAGGER
#define MSQ_STAGGER 1       /* eight-wave blocks, extension-bit layout: waves 4-7 run half a K-step behind waves 0-3 (k_qgemm3); 0 = off, 2 = also for MSQ-U1 without extension bits */
#endif
#define TILE_N 64
#define TILE_K 64
#ifndef MSQ_GV_NT
#define MSQ_GV_NT true     /* decode kernels: packed planes are read once -- non-temporal loads keep the activation rows in L2 */
#endif

// ---------------------------------------------------------------------------
// fragment dequant: 8 elements (one MFMA operand fragment) -> 4 dwords of bf16x2
// ---------------------------------------------------------------------------
template <int OUT_KIND>
MSQ_D u32x4_t dequant_frag(uint32_t inl, uint32_t o0, uint32_t o1, float s_in, float s_out) {
    u32x4_t r;
    r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(inl, s_in, 0));
    r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(inl, s_in, 1));
    r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(inl, s_in, 2));
    r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(inl, s_in, 3));
    if (OUT_KIND == MSQ_PLANE_FP8) {
        r[0] |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o0, s_out, false));
        r[1] |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o0, s_out, true));
        r[2] |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o1, s_out, false));
        r[3] |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o1, s_out, true));
    } else {
        r[0] |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(o0, s_out, false));
        r[1] |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(o0, s_out, true));
        r[2] |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(o1, s_out, false));
        r[3] |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(o1, s_out, true));
    }
    return r;
}

MSQ_D u32x4_t dequant_frag_in_only(uint32_t inl, float s_in) {
    u32x4_t r;
    r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(inl, s_in, 0));
    r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(inl, s_in, 1));
    r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(inl, s_in, 2));
    r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(inl, s_in, 3));
    return r;
}

// E8M0 byte `idx` (0..3) of a scale dword as the f32 operand of the scaled converts:
// the hardware reads only the exponent field (sign and mantissa are ignored; verified
// on MI355X), so one shift is enough.
MSQ_D float scale_operand(uint32_t d, int idx) {
    uint32_t s = (idx == 3) ? (d >> 1) : (d << (23 - 8 * idx));
    return __builtin_bit_cast(float, s);
}

// unified layout (MSQ-U1): one e4m3 code per weight, one scale per 32 k, optional extension bit
template <int OUT_KIND> struct IsUnified { static constexpr bool v = (OUT_KIND == MSQ_PLANE_U8 || OUT_KIND == MSQ_PLANE_U8X); };
// does this (in, out) kind pair carry a scale plane?
template <int IN_KIND, int OUT_KIND> struct HasScale { static constexpr bool v = (IN_KIND != MSQ_PLANE_NONE) || IsUnified<OUT_KIND>::v; };

// extension bit of the two elements of dword d of fragment nf -> bf16 mantissa bit 3 of both halves
MSQ_D uint32_t ext_or(uint32_t r, uint32_t ext, int nf, int d) {
    const int sh = nf * 4 + d;
    const uint32_t rot = sh ? __builtin_amdgcn_alignbit(ext, ext, sh) : ext;
    return (rot & 0x00080008u) | r;
}
template <int OUT_KIND>
MSQ_D u32x4_t dequant_frag_unified(uint32_t o0, uint32_t o1, float s, uint32_t ext, int nf) {
    u32x4_t r;
    r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o0, s, false));
    r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o0, s, true));
    r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o1, s, false));
    r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o1, s, true));
    if (OUT_KIND == MSQ_PLANE_U8X) {
        r[0] = ext_or(r[0], ext, nf, 0); r[1] = ext_or(r[1], ext, nf, 1);
        r[2] = ext_or(r[2], ext, nf, 2); r[3] = ext_or(r[3], ext, nf, 3);
    }
    return r;
}

// all 8 fragments of one 64x64 tile for this lane
struct TileRegs {
    u32x4_t inl[2];      // [kf]  dword nf
    u32x4_t out[8];      // 8-bit kinds use [0..3] = (kf*2 + nf/2); bf16 kind uses [kf*4 + nf]
    u32x4_t scl;         // dword nf: bytes kf*2 + io   (unified: dword kf: byte nf)
    uint32_t ext[2];     // unified U8X: extension bits of half kf
};

template <int IN_KIND, int OUT_KIND>
MSQ_D u32x4_t tile_frag(const TileRegs& t, int nf, int kf) {
    if (IsUnified<OUT_KIND>::v) {
        const u32x4_t o = t.out[kf * 2 + (nf >> 1)];
        return dequant_frag_unified<OUT_KIND>(o[(nf & 1) * 2], o[(nf & 1) * 2 + 1], scale_operand(t.scl[kf], nf), t.ext[kf], nf);
    }
    if (IN_KIND == MSQ_PLANE_NONE) return t.out[kf * 4 + nf];
    const uint32_t sd = t.scl[nf];
    const float s_in = scale_operand(sd, kf * 2);
    if (OUT_KIND == MSQ_PLANE_BF16) {
        u32x4_t r = dequant_frag_in_only(t.inl[kf][nf], s_in);
        const u32x4_t o = t.out[kf * 4 + nf];
        r[0] |= o[0]; r[1] |= o[1]; r[2] |= o[2]; r[3] |= o[3];
        return r;
    } else {
        const float s_out = scale_operand(sd, kf * 2 + 1);
        const u32x4_t o = t.out[kf * 2 + (nf >> 1)];
        return dequant_frag<OUT_KIND>(t.inl[kf][nf], o[(nf & 1) * 2], o[(nf & 1) * 2 + 1], s_in, s_out);
    }
}

// Lane / fragment -> (n, k) map of the tile layout (v_mfma_f32_16x16x32_bf16 operand order): per half tile
// (32 k) four fragments f = nf of 8 consecutive k per lane: n = 16 f + (l & 15), k = 32 kf + 8 (l >> 4) + j.
// (A 32x32x16 layout was built and measured 8-10 % slower end to end; it is gone.)
MSQ_D int frag_n(int lane, int f) { return f * 16 + (lane & 15); }
MSQ_D int frag_k(int lane, int kf) { return kf * 32 + (lane >> 4) * 8; }

template <int OUT_KIND> struct OutSlots { static constexpr int n = (OUT_KIND == MSQ_PLANE_BF16) ? 8 : 4; };
// bytes of the scale plane per tile and lane group
template <int OUT_KIND> struct SclBytes { static constexpr int n = IsUnified<OUT_KIND>::v ? 8 : 16; };

#define BN 256
#define BK 64

// ---------------------------------------------------------------------------
// fused unpack-dequant-GEMM, software-pipelined at half-K-step (one MFMA k-fragment = 32 k) granularity.
//   half-step h:  issue packed loads for h+2 | convert packed(h+1) -> wf[(h+1)&1] | 32 MFMAs on wf[h&1]
// so the scaled converts of the next fragment set fill the VALU slots between the MFMAs of the
// current one (sched_group_barrier pins the interleave), and only one half-step of packed data
// is in flight per buffer (register budget: 128 acc + 32 wf + <=48 packed + 12 xf).
// ---------------------------------------------------------------------------
template <int OUT_KIND> struct HalfSlots { static constexpr int n = (OUT_KIND == MSQ_PLANE_BF16) ? 4 : 2; };
// vector-memory loads one half-step issues for the packed operand (vmcnt bookkeeping)
template <int IN_KIND, int OUT_KIND> struct HalfLoads {
    static constexpr int n = (IN_KIND != MSQ_PLANE_NONE ? 1 : 0) + (((MSQ_ABL & 256) && IsUnified<OUT_KIND>::v) ? 1 : HalfSlots<OUT_KIND>::n) + (OUT_KIND == MSQ_PLANE_U8X ? 1 : 0);
};

template <int IN_KIND, int OUT_KIND>
struct HalfRegs {
    u32x4_t inl;
    u32x4_t out[HalfSlots<OUT_KIND>::n];
    uint32_t ext;        // U8X: extension bits of this half
};

// --- packed-plane loads through buffer descriptors: the per-lane offset (lane * 16) never changes and the
// slot offset is wave-uniform, so it rides in the SGPR soffset operand: zero VALU address arithmetic.
struct PlaneRsrc {
    __amdgpu_buffer_rsrc_t inl, out, scl;
};
MSQ_D __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int64_t bytes) {
    // make every descriptor input provably wave-uniform (otherwise hipcc wraps each buffer op in a
    // waterfall loop: guide T20)
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    uint32_t n = bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bytes;
    n = __builtin_amdgcn_readfirstlane(n);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}
// SGPR offsets of the buffer ops.  Every caller passes a value derived from blockIdx / the wave id (itself a readfirstlane)
// and loop counters, which hipcc already proves wave-uniform: wrapping them in v_readfirstlane (as round 1 did) FORCES a
// VGPR detour -- the index arithmetic (v_add, v_min, quarter-rate v_mul_lo_u32) moved to the VALU and every use paid a
// readfirstlane + hazard nops: 238 vector instructions of the U8X kernel, ~80 per K-step.  No waterfall loop appears
// without it (checked in the ISA: no s_and_saveexec in any GEMM kernel).  MSQ_UNI_RFL=1 restores the wrapper (A/B).
#ifndef MSQ_UNI_RFL
#define MSQ_UNI_RFL 0
#endif
MSQ_D uint32_t uni(uint32_t v) { return MSQ_UNI_RFL ? __builtin_amdgcn_readfirstlane(v) : v; }
// ... instead the few BASE indices (tile row, first / last K-step, the K-step counter once per step) go through one
// readfirstlane each: everything derived from them is then scalar arithmetic by construction, in every instantiation
// (with no anchor at all hipcc's uniformity analysis gave up in 11 of the 72 GEMM kernels and emitted waterfall loops).
MSQ_D int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <int IN_KIND, int OUT_KIND>
MSQ_D void load_half_buf(HalfRegs<IN_KIND, OUT_KIND>& h, const PlaneRsrc& r, int lane16, uint32_t tile2kf) {
    constexpr int HS = HalfSlots<OUT_KIND>::n;
    if (IN_KIND != MSQ_PLANE_NONE)
        h.inl = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r.inl, lane16, uni(tile2kf * 1024u), 0));
#pragma unroll
    for (int s = 0; s < HS; ++s) {
        if ((MSQ_ABL & 256) && IsUnified<OUT_KIND>::v && s > 0) { h.out[s] = h.out[0]; continue; }
        h.out[s] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r.out, lane16, uni((tile2kf * HS + s) * 1024u), 0));
    }
    if (OUT_KIND == MSQ_PLANE_U8X)     // extension plane rides in the inlier descriptor: 64 lanes x 4 B per half
        h.ext = __builtin_amdgcn_raw_buffer_load_b32(r.inl, lane16 >> 2, uni(tile2kf * 256u), 0);
}

// one quarter of a half-step's conversion work: fragment nf = q/2, dwords 2*(q%2) .. +1
template <int IN_KIND, int OUT_KIND>
MSQ_D void convert_quarter(u32x4_t (&wf)[4], const HalfRegs<IN_KIND, OUT_KIND>& h, const u32x4_t& scl, int kf, int q) {
    const int nf = q >> 1, hh = q & 1;
    if (IsUnified<OUT_KIND>::v) {      // scl[kf] byte nf; one convert per dword (+ rotate / and-or for the extension bit)
        const float s = scale_operand(scl[kf], nf);
        const uint32_t o = h.out[nf >> 1][(nf & 1) * 2 + hh];
        uint32_t r0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o, s, false));
        uint32_t r1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o, s, true));
        if (OUT_KIND == MSQ_PLANE_U8X) { r0 = ext_or(r0, h.ext, nf, 2 * hh); r1 = ext_or(r1, h.ext, nf, 2 * hh + 1); }
        wf[nf][2 * hh] = r0; wf[nf][2 * hh + 1] = r1;
        return;
    }
    if (IN_KIND == MSQ_PLANE_NONE) {
        wf[nf][2 * hh] = h.out[nf][2 * hh]; wf[nf][2 * hh + 1] = h.out[nf][2 * hh + 1];
        return;
    }
    const float s_in = scale_operand(scl[nf], kf * 2);
    uint32_t r0, r1;
    if (hh == 0) {
        r0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(h.inl[nf], s_in, 0));
        r1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(h.inl[nf], s_in, 1));
    } else {
        r0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(h.inl[nf], s_in, 2));
        r1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(h.inl[nf], s_in, 3));
    }
    if (OUT_KIND == MSQ_PLANE_BF16) {
        r0 |= h.out[nf][2 * hh]; r1 |= h.out[nf][2 * hh + 1];
    } else {
        const float s_out = scale_operand(scl[nf], kf * 2 + 1);
        const uint32_t o = h.out[nf >> 1][(nf & 1) * 2 + hh];
        if (OUT_KIND == MSQ_PLANE_FP8) {
            r0 |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o, s_out, false));
            r1 |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o, s_out, true));
        } else {
            r0 |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(o, s_out, false));
            r1 |= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(o, s_out, true));
        }
    }
    wf[nf][2 * hh] = r0; wf[nf][2 * hh + 1] = r1;
}

// force the compiler to have these registers loaded here (its s_waitcnt lands at this point)
MSQ_D void keep_live4(u32x4_t& v) { asm volatile("" : "+v"(v)); }
MSQ_D void keep_live1(uint32_t& v) { asm volatile("" : "+v"(v)); }
template <int IN_KIND, int OUT_KIND>
MSQ_D void keep_live(HalfRegs<IN_KIND, OUT_KIND>& h) {
    if (IN_KIND != MSQ_PLANE_NONE) keep_live4(h.inl);
    if (OUT_KIND == MSQ_PLANE_U8X) keep_live1(h.ext);
#pragma unroll
    for (int s = 0; s < HalfSlots<OUT_KIND>::n; ++s) keep_live4(h.out[s]);
}

// Between the last tied-accumulator MFMA of a K-loop and the first read of the accumulators.  The MFMAs are inline asm: hipcc's hazard
// recogniser does not know that they write a[...], so (1) the passes of the last ones are given by hand (s_nop), and (2) every
// accumulator is then REDEFINED by an empty asm statement -- volatile asm statements keep their order, so whatever hipcc itself does
// with an accumulator afterwards (v_accvgpr_read, but also a register-allocator SPILL: `scratch_store_dwordx4 off, a[200:203]` was
// placed right behind the last MFMA, in front of the s_nop, in the MX-FP6 kernels as soon as the epilogue's register needs changed,
// and stored a half-written accumulator) is tied to the value defined AFTER the wait states.
template <int MF>
MSQ_D void acc_fence(f32x4_t (&acc)[MF][4]) {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
}

// ---------------------------------------------------------------------------
// Epilogue of a 128(m) x 64(n) wave tile through LDS: the MFMA result layout gives every lane 4
// consecutive n of one row (8-byte pieces, 32-byte runs per row) -- stored directly they reach L2 as
// partial lines and the store tail is issue-bound (measured 7 % of the kernel).  Each wave instead
// transposes its tile through its own 8 KiB LDS slice (XOR-swizzled 16-byte chunks, no block barrier
// needed) and writes whole 128-byte (bf16) / 256-byte (f32) row segments with 16-byte stores.
// ---------------------------------------------------------------------------
// y16 (16-bit YT only): 0 = bf16, 1 = fp16 (IEEE half, round to nearest even) -- a wave-uniform run-time switch, not a third
// instantiation of every GEMM kernel: an fp16 model gets its dtype back without a cast pass over the output.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <typename YT, int NMF = 8>
MSQ_D void store_wave_tile_lds(const f32x4_t (&acc)[NMF][4], char* wsm, YT* __restrict__ Y, int m_base, int n_base,
                               int M, int N, const float* __restrict__ bias, int lane, int y16 = 0) {
    const int c = lane & 15, g = lane >> 4;
    constexpr int ROW_B = 64 * (int)sizeof(YT);             // bytes per tile row: 128 (bf16) / 256 (f32)
    constexpr int RP = 8192 / ROW_B;                         // rows per pass: 64 / 32
    constexpr int MF_PER_PASS = RP / 16;                     // 4 / 2
    constexpr int CHUNKS = ROW_B / 16;                       // 16-byte chunks per row: 8 / 16
    float bv[4][4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[nf][j] = bias ? bias[n_base + nf * 16 + g * 4 + j] : 0.f;
#pragma unroll
    for (int p = 0; p < NMF / MF_PER_PASS; ++p) {
#pragma unroll
        for (int i = 0; i < MF_PER_PASS; ++i) {
            const int mf = p * MF_PER_PASS + i;
            const int row = i * 16 + c;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                f32x4_t v = acc[mf][nf];
                v[0] += bv[nf][0]; v[1] += bv[nf][1]; v[2] += bv[nf][2]; v[3] += bv[nf][3];
                if (sizeof(YT) == 4) {
                    const int chunk = (nf * 4 + g) ^ (row & (CHUNKS - 1));
                    *reinterpret_cast<float4*>(wsm + row * ROW_B + chunk * 16) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint32_t plo, phi;
                    if (y16) {
                        f16x2_t lo, hi;
                        lo[0] = (_Float16)v[0]; lo[1] = (_Float16)v[1]; hi[0] = (_Float16)v[2]; hi[1] = (_Float16)v[3];
                        plo = __builtin_bit_cast(uint32_t, lo); phi = __builtin_bit_cast(uint32_t, hi);
                    } else {
                        bf16x2_t lo, hi;
                        lo[0] = (__bf16)v[0]; lo[1] = (__bf16)v[1]; hi[0] = (__bf16)v[2]; hi[1] = (__bf16)v[3];
                        plo = __builtin_bit_cast(uint32_t, lo); phi = __builtin_bit_cast(uint32_t, hi);
                    }
                    const int chunk = (nf * 2 + (g >> 1)) ^ (row & (CHUNKS - 1));
                    *reinterpret_cast<uint2*>(wsm + row * ROW_B + chunk * 16 + (g & 1) * 8) = make_uint2(plo, phi);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        constexpr int ROWS_PER_INSTR = 64 / CHUNKS;          // 8 / 4
#pragma unroll
        for (int t = 0; t < RP / ROWS_PER_INSTR; ++t) {
            const int row = t * ROWS_PER_INSTR + lane / CHUNKS;
            const int chunk = lane % CHUNKS;
            const u32x4_t d = *reinterpret_cast<const u32x4_t*>(wsm + row * ROW_B + ((chunk ^ (row & (CHUNKS - 1))) * 16));
            const int m = m_base + p * RP + row;
            if (m < M)
                *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(Y) + ((int64_t)m * N + n_base) * (int64_t)sizeof(YT) + chunk * 16) = d;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // reads done before the next pass overwrites
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------
// The same epilogue WITHOUT the trip through LDS (16-bit outputs): v_permlane16_swap_b32 exchanges, between two accumulator quads of
// one row fragment, the odd 16-lane rows of one with the even rows of the other -- afterwards every lane holds 8 consecutive n of
// its output row (16 bytes): lanes g = 0 / 2 the halves of quad a's 16 columns, lanes g = 1 / 3 those of quad b.  One 16-byte store
// per quad pair writes 16 rows x 64 contiguous bytes (the pair's two stores cover the row's 128-byte line back to back); per tile
// and wave 64 swaps replace 64 ds_write_b64 + 32 ds_read_b128 and their waits.  fp32 outputs already hold 16 bytes per lane.
// Results are the same bits as store_wave_tile_lds.  With `bias == nullptr` the 256 additions of zero are not issued.
// ---------------------------------------------------------------------------
#ifndef MSQ_EPI_DIRECT
#define MSQ_EPI_DIRECT 1     /* 0: the LDS-transposed epilogue (A / B) */
#endif
template <typename YT, int NMF, bool BIAS>
MSQ_D void store_wave_tile_direct_(const f32x4_t (&acc)[NMF][4], YT* __restrict__ Y, int m_base, int n_base, int M, int N,
                                   const float* __restrict__ bias, int lane, int y16) {
    const int c = lane & 15, g = lane >> 4;
    float bv[4][4];
    if (BIAS) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const float4 t = *reinterpret_cast<const float4*>(bias + n_base + nf * 16 + g * 4);
            bv[nf][0] = t.x; bv[nf][1] = t.y; bv[nf][2] = t.z; bv[nf][3] = t.w;
        }
    }
    // this lane's column within the wave's 64 for pair (a, b = a + 1): even g -> quad a, columns 4 g .. 4 g + 7; odd g -> quad b, 4 (g - 1) ..
    const int colp = (g & 1) ? 16 + (g - 1) * 4 : g * 4;
#pragma unroll
    for (int mf = 0; mf < NMF; ++mf) {
        const int m = m_base + mf * 16 + c;
        char* rowp = reinterpret_cast<char*>(Y) + ((int64_t)m * N + n_base) * (int64_t)sizeof(YT);
        if (sizeof(YT) == 4) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                f32x4_t v = acc[mf][nf];
                if (BIAS) { v[0] += bv[nf][0]; v[1] += bv[nf][1]; v[2] += bv[nf][2]; v[3] += bv[nf][3]; }
                if (m < M) *reinterpret_cast<float4*>(rowp + (nf * 16 + g * 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
            for (int a = 0; a < 4; a += 2) {
                uint32_t d[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4_t v = acc[mf][a + q];
                    if (BIAS) { v[0] += bv[a + q][0]; v[1] += bv[a + q][1]; v[2] += bv[a + q][2]; v[3] += bv[a + q][3]; }
                    if (y16) {
                        f16x2_t lo, hi;
                        lo[0] = (_Float16)v[0]; lo[1] = (_Float16)v[1]; hi[0] = (_Float16)v[2]; hi[1] = (_Float16)v[3];
                        d[q][0] = __builtin_bit_cast(uint32_t, lo); d[q][1] = __builtin_bit_cast(uint32_t, hi);
                    } else {
                        bf16x2_t lo, hi;
                        lo[0] = (__bf16)v[0]; lo[1] = (__bf16)v[1]; hi[0] = (__bf16)v[2]; hi[1] = (__bf16)v[3];
                        d[q][0] = __builtin_bit_cast(uint32_t, lo); d[q][1] = __builtin_bit_cast(uint32_t, hi);
                    }
                }
                // odd rows of quad a's dwords <-> even rows of quad b's
                const auto r0 = __builtin_amdgcn_permlane16_swap(d[0][0], d[1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(d[0][1], d[1][1], false, false);
                const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
                if (m < M) *reinterpret_cast<u32x4_t*>(rowp + (a * 16 + colp) * 2) = o;
            }
        }
    }
}
template <typename YT, int NMF = 8>
MSQ_D void store_wave_tile_direct(const f32x4_t (&acc)[NMF][4], YT* __restrict__ Y, int m_base, int n_base, int M, int N,
                                  const float* __restrict__ bias, int lane, int y16 = 0) {
    if (bias) store_wave_tile_direct_<YT, NMF, true>(acc, Y, m_base, n_base, M, N, bias, lane, y16);
    else store_wave_tile_direct_<YT, NMF, false>(acc, Y, m_base, n_base, M, N, bias, lane, y16);
}
