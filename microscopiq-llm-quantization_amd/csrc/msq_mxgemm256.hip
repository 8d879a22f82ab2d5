// msq_mxgemm256.hip -- k_mxgemm256: the MX-native W4A8 / W6A8 / W8A8 GEMM (BASELINE config 3 on the CDNA4 scaled MFMA,
// v_mfma_scale_f32_16x16x128_f8f6f4; operands as k_mxgemm, msq_gemm.hip: MX-FP8 activation codes + E8M0 scale bytes against MX-FP4 /
// MX-FP6 / exact e4m3 weight codes in operand order, mx_ops.py:332-457 semantics, block 32 along K) with 256-row wave tiles and
// hand-placed accumulators -- the recipe of k_qgemm256 (msq_gemm256.hip) applied to the matrix path that does no dequantisation at all:
//   * block 256(m) x 256(n), four waves 1 x 4, ONE wave per SIMD, the 64 accumulator quads pinned to a[0:255] by tied inline-asm MFMAs;
//   * K-step 128 = sixteen groups of four MFMAs (activation fragment mf against the four weight fragments); an MFMA with an fp8
//     operand occupies the matrix pipe for 32 cycles and the issue port for 8: every MFMA is followed by ONE filler, fenced by
//     sched_barriers: MFMA 0 | ds_read_b128 (low half of fragment mf + PF), MFMA 1 | ds_read_b128 (high half), MFMA 2 | ds_read_u8 (its
//     scale byte), MFMA 3 | one vector-memory op (LDS-DMA piece of the activation tile two K-steps ahead, or a weight load);
//   * four activation buffers (codes 32 KiB + scale bytes 1 KiB each), the block barrier PF groups before the end of the K-step;
//   * weight codes and scale dwords stream from global memory through buffer descriptors, three K-steps deep for every operand
//     width (k_mxgemm keeps the 24- / 32-byte operands two deep: 128 accumulators + 99 ring registers did not fit two waves per SIMD).
// Same operand bytes, same tile order, same accumulation order per output element as k_mxgemm: bit-identical results.
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

#include "msq_gemm_common.h"

#ifndef MSQ_MX256_PF
#define MSQ_MX256_PF 2
#endif
#ifndef MSQ_MX256_RT
#define MSQ_MX256_RT 4
#endif
#ifndef MSQ_MX256_ABL
#define MSQ_MX256_ABL 0    /* timing experiments (wrong results): 1 no fragment reads, 4 no weight loads, 8 no staging, 16 no barrier, 32 no stores, 64 no MFMAs */
#endif

namespace {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v6i_t __attribute__((ext_vector_type(6)));
typedef int v4i_t __attribute__((ext_vector_type(4)));

// D(a[..]) += A(weight codes, scale byte NF of `sa`) x B(activation codes, scale byte 0 of `sb`), accumulator tied in place.
// CBSZ = A-operand format of the instruction: 4 e2m1 (4 VGPRs), 0 e4m3 (8), 2 e2m3 / 3 e3m2 (6).
#define MX_MFMA_ASM(CB, LO, HI) \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[" #LO ",0,0] op_sel_hi:[" #HI ",0,0] cbsz:" #CB \
                 : "+a"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
template <int CBSZ, int NF, typename AT>
MSQ_D void mfma_mx(f32x4_t& acc, const AT& a, const v8i_t& b, uint32_t sa, uint32_t sb) {
    if constexpr (CBSZ == 4) { if constexpr (NF == 0) MX_MFMA_ASM(4, 0, 0); else if constexpr (NF == 1) MX_MFMA_ASM(4, 1, 0); else if constexpr (NF == 2) MX_MFMA_ASM(4, 0, 1); else MX_MFMA_ASM(4, 1, 1); }
    else if constexpr (CBSZ == 0) { if constexpr (NF == 0) MX_MFMA_ASM(0, 0, 0); else if constexpr (NF == 1) MX_MFMA_ASM(0, 1, 0); else if constexpr (NF == 2) MX_MFMA_ASM(0, 0, 1); else MX_MFMA_ASM(0, 1, 1); }
    else if constexpr (CBSZ == 2) { if constexpr (NF == 0) MX_MFMA_ASM(2, 0, 0); else if constexpr (NF == 1) MX_MFMA_ASM(2, 1, 0); else if constexpr (NF == 2) MX_MFMA_ASM(2, 0, 1); else MX_MFMA_ASM(2, 1, 1); }
    else { if constexpr (NF == 0) MX_MFMA_ASM(3, 0, 0); else if constexpr (NF == 1) MX_MFMA_ASM(3, 1, 0); else if constexpr (NF == 2) MX_MFMA_ASM(3, 0, 1); else MX_MFMA_ASM(3, 1, 1); }
}
#undef MX_MFMA_ASM

template <int WF> struct WOperand { typedef v4i_t type; static constexpr int loads = 1; static constexpr int frag_bytes = 1024; };
template <> struct WOperand<1> { typedef v8i_t type; static constexpr int loads = 2; static constexpr int frag_bytes = 2048; };
template <> struct WOperand<2> { typedef v6i_t type; static constexpr int loads = 2; static constexpr int frag_bytes = 1536; };
template <> struct WOperand<3> { typedef v6i_t type; static constexpr int loads = 2; static constexpr int frag_bytes = 1536; };

// The vector-memory op behind MFMA `slot` (2 or 3) of group mf: 0 .. PPW - 1 an LDS-DMA piece, 100 + part a weight load, 200 the scale-tile
// DMA, -1 nothing (see the kernel); count(): ops in groups < upto, `dma_only` for the tail steps whose weight loads hipcc deletes.
template <int MF, int NPART> struct MxSched {
    static constexpr int op(int mf, int slot) {
        if (MF == 16) {
            if (slot == 3) {
                if ((mf & 1) == 0) return mf >> 1;
                const int part = mf >> 1;
                return part < NPART ? 100 + part : (part == NPART ? 200 : -1);
            }
            if (NPART + 1 > 8) { if (mf == 1) return 100 + 8; if (mf == 3) return 200; }
            return -1;
        }
        if (slot == 3) return (mf & 1) == 0 ? (mf >> 1) : 100 + (mf >> 1);
        const int part = 4 + mf;
        return part < NPART ? 100 + part : (part == NPART ? 200 : -1);
    }
    static constexpr int count(int upto, bool dma_only) {
        int n = 0;
        for (int mf = 0; mf < upto; ++mf)
            for (int slot = 2; slot <= 3; ++slot) {
                const int o = op(mf, slot);
                if (o >= 0 && (!dma_only || o < 100 || o >= 200)) n += 1;
            }
        return n;
    }
};

// WF = weight operand format as k_mxgemm: 0 e2m1 (16 B per lane and 16 n), 1 e4m3 (32 B in two half-slots), 2 / 3 = fp6 e2m3 / e3m2
// (24 B per lane: a 16-byte and an 8-byte piece, 1.5 KiB per fragment slot)
// MF = activation fragments per wave tile: 16 = the 256-row block above (one block per CU); 8 = its 128-row form, two blocks per
// CU -- for grids that 256-row blocks do not fill (q/k/v 2048 x 12288: 1.5 rounds instead of 2; o and down: one round of 256 blocks).
// With 128 accumulators + 128 other registers per lane the weight ring is two K-steps deep there (three sets of the 32-byte operand
// do not fit); the second block of the CU covers what that exposes.
template <typename YT, int WF, int MF>
__global__ void __launch_bounds__(256, (MF == 16) ? 1 : 2)
k_mxgemm256(const uint8_t* __restrict__ Xc, const uint8_t* __restrict__ Xs, const uint8_t* __restrict__ Wc, const uint8_t* __restrict__ Ws,
            const float* __restrict__ bias, YT* __restrict__ Y, int M, int N, int K, int y16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename WOperand<WF>::type AT;
    constexpr int WL = WOperand<WF>::loads;                      // vector-memory loads per weight fragment
    constexpr int FB = WOperand<WF>::frag_bytes;
    constexpr int CBSZ = (WF >= 2) ? WF : (WF == 1 ? 0 : 4);
    constexpr int BM = 16 * MF;                                  // rows per block
    constexpr int PPW = MF / 2;                                  // 1 KiB staging pieces per wave and K-step
    constexpr int WD = (MF == 16) ? 3 : 2;                       // weight ring depth in K-steps
    constexpr int KS = 128, A_TILE = BM * KS;                    // 32 / 16 KiB of activation codes per buffer
    constexpr int XS_BASE = 4 * A_TILE;                          // four 1 KiB scale tiles behind the code buffers: [row][4 bytes]
    constexpr int PF = MSQ_MX256_PF;
    static_assert(PF >= 1 && PF <= 3, "fragment ring of four");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int MT = (M + BM - 1) / BM, NTB = N / 256;
    const int KT = K / KS;
    const int bid = (int)blockIdx.x;
    int bm, bn;
    if ((NTB & 7) == 0) {                                        // XCD-aware order, as k_qgemm256
        constexpr int RT = MSQ_MX256_RT * (16 / MF);
        const int xcd = bid & 7, i = bid >> 3;
        const int npx = NTB >> 3, per_group = RT * npx, full = MT / RT;
        int rg, j, R;
        if (i < full * per_group) { rg = i / per_group; j = i % per_group; R = RT; }
        else { rg = full; j = i - full * per_group; R = MT - full * RT; }
        bm = rg * RT + j % R;
        bn = (j / R) * 8 + xcd;
    } else { bm = bid % MT; bn = bid / MT; }
    const int m0 = bm * BM, n0 = bn * 256;
    const int64_t wtiles = (int64_t)(N / 64) * KT;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(Wc, wtiles * 4 * FB);
    const __amdgpu_buffer_rsrc_t wsr = make_rsrc(Ws, wtiles * 256);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(Xc, (int64_t)M * K);
    const __amdgpu_buffer_rsrc_t xsr = make_rsrc(Xs, (int64_t)M * (K / 32));
    const uint32_t tile_row32 = (uint32_t)sgpr((n0 / 64 + wid) * KT);
    const int lane16 = lane * 16, lane8 = lane * 8, lane4 = lane * 4;

    // activation staging: wave w copies rows (BM / 4) w .. as PPW 1 KiB pieces (8 rows x 128 B; lane l: row 8 p + l / 8, source chunk
    // (l & 7) ^ ((row >> 1) & 7)) and 64 scale dwords (one per lane; with 128-row blocks waves 2 / 3 write the rows of waves 0 / 1
    // again -- the same bytes to the same place: every wave issues the same number of vector-memory ops, which the wait counts rely on)
    int aoff[8];                                                 // (first PPW used; a dependent array bound captured by the staging lambda fails to instantiate on the host pass)
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int row = (wid * PPW + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < M ? gr : M - 1;
        aoff[p] = (int)((int64_t)gr * K + chunk * 16);
    }
    const int xs_row0 = (wid * 64) % BM;
    int xs_goff = m0 + xs_row0 + lane; xs_goff = (xs_goff < M ? xs_goff : M - 1) * (K / 32);
    auto stage_piece = [&](int kt, int buf, int p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (void __attribute__((address_space(3)))*)(smem + buf * A_TILE + (wid * PPW + p) * 1024),
                                                 16, aoff[p], (uint32_t)kt * KS, 0, 0);
    };
    auto stage_scales = [&](int kt, int buf) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsr, (void __attribute__((address_space(3)))*)(smem + XS_BASE + buf * 1024 + xs_row0 * 4),
                                                 4, xs_goff, (uint32_t)kt * 4u, 0, 0);
    };
    // fragment mf of this lane: row mf * 16 + c, chunks g and 4 + g (k = 16 g .. and 64 + 16 g ..); scale byte g of that row
    const int sw = (c >> 1) & 7;
    const int rdl = c * 128 + ((g ^ sw) << 4), rdh = c * 128 + (((4 + g) ^ sw) << 4);
    const int xs_rd = XS_BASE + c * 4 + g;

    f32x4_t acc[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    struct WSet { AT w[4]; uint32_t s; };
    WSet w0, w1, w2;
    // one vector-memory load of the weight set of K-step kt: part 0 .. 4 WL - 1 = (fragment nf, piece), part 4 WL = the scale dword
    auto load_w_part = [&](WSet& ws, int kt, int part) {
        const uint32_t base = (tile_row32 + (uint32_t)kt) * 4u;
        if (part == 4 * WL) { ws.s = __builtin_amdgcn_raw_buffer_load_b32(wsr, lane4, (tile_row32 + (uint32_t)kt) * 256u, 0); return; }
        const int nf = part / WL, h = part % WL;
        if constexpr (WF == 0) {
            ws.w[nf] = __builtin_bit_cast(v4i_t, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, (base + nf) * 1024u, 0));
        } else if constexpr (WF == 1) {
            const u32x4_t t = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, ((base + nf) * 2u + h) * 1024u, 0));
            ws.w[nf][4 * h + 0] = (int)t[0]; ws.w[nf][4 * h + 1] = (int)t[1]; ws.w[nf][4 * h + 2] = (int)t[2]; ws.w[nf][4 * h + 3] = (int)t[3];
        } else {
            if (h == 0) {
                const u32x4_t t = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, (base + nf) * 1536u, 0));
                ws.w[nf][0] = (int)t[0]; ws.w[nf][1] = (int)t[1]; ws.w[nf][2] = (int)t[2]; ws.w[nf][3] = (int)t[3];
            } else {
                const u32x2_t t = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(wr, lane8, (base + nf) * 1536u + 1024u, 0));
                ws.w[nf][4] = (int)t[0]; ws.w[nf][5] = (int)t[1];
            }
        }
    };
    auto load_w_all = [&](WSet& ws, int kt) {
#pragma unroll
        for (int part = 0; part <= 4 * WL; ++part) load_w_part(ws, kt, part);
    };

    const int kl = sgpr(KT - 1);
    const int k1 = (1 <= kl) ? 1 : kl;
#pragma unroll
    for (int p = 0; p < PPW; ++p) stage_piece(0, 0, p);
    stage_scales(0, 0);
#pragma unroll
    for (int p = 0; p < PPW; ++p) stage_piece(k1, 1, p);
    stage_scales(k1, 1);
    load_w_all(w0, 0);
    if constexpr (WD == 3) load_w_all(w1, k1);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    // fragment ring: four slots of {low half, high half, scale byte} (a power of two that divides the MF groups of a K-step: the
    // slot of fragment mf + PF of the NEXT K-step is the slot its group mf + PF - MF will read)
    u32x4_t xl[4], xh[4];
    uint32_t xsc[4];
#pragma unroll
    for (int f = 0; f < PF; ++f) {
        xl[f] = *reinterpret_cast<const u32x4_t*>(smem + rdl + f * 2048);
        xh[f] = *reinterpret_cast<const u32x4_t*>(smem + rdh + f * 2048);
        xsc[f] = *reinterpret_cast<const uint8_t*>(smem + xs_rd + f * 64);
    }

    // Vector-memory ops of a K-step: PPW LDS-DMA pieces of the activation tile two K-steps ahead, NPART weight loads of the K-step
    // WD - 1 ahead, the scale-tile DMA.  Two filler slots per group take one op each: slot 3 (behind MFMA 3) and slot 2 (behind MFMA 2,
    // next to the scale-byte read).  MxSched::op(mf, slot) names the op: 0 .. PPW - 1 a piece, 100 + part a weight load, 200 the scale
    // DMA, -1 nothing -- the schedule, the wait counts in front of the barrier (group BAR_G) and the tail-step counts all read it.
    //   MF = 16: slot 3 of even groups = piece mf / 2, of odd groups = weight part mf / 2 (then the scale DMA); a ninth weight load and
    //            the scale DMA of the 24- / 32-byte operands ride in slot 2 of groups 1 and 3;
    //   MF = 8 : slot 3 of even groups = piece mf / 2, of odd groups = weight part mf / 2 (0 .. 3); slot 2 of group mf = weight part
    //            4 + mf (4 .. NPART - 1), then the scale DMA.
    constexpr int NPART = 4 * WL + 1;                            // weight loads per K-step: 5 (fp4) / 9
    constexpr int BAR_G = MF - PF;
    typedef MxSched<MF, NPART> SCH;
    auto issue = [&](int op, WSet& wl, int ktw, int ktn, int buf2) {
        if (op < 0) return;
        if (op < 100) { if (!(MSQ_MX256_ABL & 8)) stage_piece(ktn, buf2, op); }
        else if (op < 200) { if (!(MSQ_MX256_ABL & 4)) load_w_part(wl, ktw, op - 100); }
        else { if (!(MSQ_MX256_ABL & 8)) stage_scales(ktn, buf2); }
    };
    // ops issued before the barrier in front of group BAR_G: exactly those may stay in flight
    constexpr int N_WAIT = SCH::count(BAR_G, false);
    // The steps after the loop load weights nobody will use: hipcc deletes those loads, so such a step issues only its LDS-DMA ops and
    // the loop's wait count would let ops of the PREVIOUS step -- the tile the next tail step reads -- stay in flight (seen as 25-28
    // differing launches of 300 with K = 640 / 1024 and the five-load fp4 operand, scripts/experiments/stress_round4.py; the same
    // trap as k_mxgemm's).  Tail steps therefore let only their own LDS-DMA ops stay in flight.
    constexpr int N_WAIT_TAIL = SCH::count(BAR_G, true);
#define MX256_SB() __builtin_amdgcn_sched_barrier(0)
#define MX256_STEP(KT_CUR, WCUR, WLOAD, NWAIT)                                                                       \
    {                                                                                                                \
        const int kt_ = sgpr(KT_CUR);                                                                                \
        const int buf = abuf, bufn = (abuf + 1) & 3, buf2 = (abuf + 2) & 3;                                          \
        abuf = bufn;                                                                                                 \
        const char* acur = smem + buf * A_TILE;                                                                      \
        const char* anxt = smem + bufn * A_TILE;                                                                     \
        const char* scur = smem + buf * 1024;                                                                        \
        const char* snxt = smem + bufn * 1024;                                                                       \
        const int ktn = (kt_ + 2 <= kl) ? kt_ + 2 : kl;            /* branch-free tail: re-stage / re-load the last tile */ \
        const int ktw = (kt_ + (WD - 1) <= kl) ? kt_ + (WD - 1) : kl;                                                \
        _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) {                                                          \
            if (mf == BAR_G) {                                                                                       \
                __builtin_amdgcn_s_waitcnt(0x0F70 | ((NWAIT) & 15) | (((NWAIT) >> 4) << 14));   /* vmcnt(NWAIT) only */ \
                if (!(MSQ_MX256_ABL & 16)) __builtin_amdgcn_s_barrier();                                             \
            }                                                                                                        \
            const u32x4_t lo_ = xl[mf & 3], hi_ = xh[mf & 3];                                                        \
            const v8i_t bfr = {(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
            const uint32_t sb_ = xsc[mf & 3];                                                                        \
            const int f_ = mf + PF;                                                                                  \
            MX256_SB();                                                                                              \
            if (!(MSQ_MX256_ABL & 64)) mfma_mx<CBSZ, 0>(acc[mf][0], WCUR.w[0], bfr, WCUR.s, sb_);                    \
            MX256_SB();                                                                                              \
            if (!(MSQ_MX256_ABL & 1)) xl[f_ & 3] = *reinterpret_cast<const u32x4_t*>((f_ < MF ? acur : anxt) + rdl + (f_ & (MF - 1)) * 2048); \
            MX256_SB();                                                                                              \
            if (!(MSQ_MX256_ABL & 64)) mfma_mx<CBSZ, 1>(acc[mf][1], WCUR.w[1], bfr, WCUR.s, sb_);                    \
            MX256_SB();                                                                                              \
            if (!(MSQ_MX256_ABL & 1)) xh[f_ & 3] = *reinterpret_cast<const u32x4_t*>((f_ < MF ? acur : anxt) + rdh + (f_ & (MF - 1)) * 2048); \
            MX256_SB();                                                                                              \
            if (!(MSQ_MX256_ABL & 64)) mfma_mx<CBSZ, 2>(acc[mf][2], WCUR.w[2], bfr, WCUR.s, sb_);                    \
            MX256_SB();                                                                                              \
            if (!(MSQ_MX256_ABL & 1)) xsc[f_ & 3] = *reinterpret_cast<const uint8_t*>((f_ < MF ? scur : snxt) + xs_rd + (f_ & (MF - 1)) * 64); \
            issue(SCH::op(mf, 2), WLOAD, ktw, ktn, buf2);                                                              \
            MX256_SB();                                                                                              \
            if (!(MSQ_MX256_ABL & 64)) mfma_mx<CBSZ, 3>(acc[mf][3], WCUR.w[3], bfr, WCUR.s, sb_);                    \
            MX256_SB();                                                                                              \
            issue(SCH::op(mf, 3), WLOAD, ktw, ktn, buf2);                                                              \
        }                                                                                                            \
    }

    int abuf = 0;
    {
        int kt = 0;
        if constexpr (WD == 3) {
            for (; kt + 2 < KT; kt += 3) { MX256_STEP(kt, w0, w2, N_WAIT) MX256_STEP(kt + 1, w1, w0, N_WAIT) MX256_STEP(kt + 2, w2, w1, N_WAIT) }
            if (kt < KT) { MX256_STEP(kt, w0, w2, N_WAIT_TAIL) ++kt; }
            if (kt < KT) { MX256_STEP(kt, w1, w0, N_WAIT_TAIL) ++kt; }
        } else {
            for (; kt + 1 < KT; kt += 2) { MX256_STEP(kt, w0, w1, N_WAIT) MX256_STEP(kt + 1, w1, w0, N_WAIT) }
            if (kt < KT) { MX256_STEP(kt, w0, w1, N_WAIT_TAIL) ++kt; }
        }
    }
#undef MX256_STEP
#undef MX256_SB

    acc_fence<MF>(acc);                                          // the MFMAs are opaque to hipcc's hazard recogniser (msq_gemm_common.h)
    __builtin_amdgcn_s_waitcnt(0x0070);                          // drain the re-staged tail tiles before the epilogue reuses LDS
    __builtin_amdgcn_s_barrier();
    if (MSQ_MX256_ABL & 32) { float t = 0.f; _Pragma("unroll") for (int i = 0; i < MF; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; if (t == 1.2345f) reinterpret_cast<float*>(Y)[0] = t; return; }
#pragma unroll
    for (int h = 0; h < MF / 8; ++h) {
        const f32x4_t (&acch)[8][4] = *reinterpret_cast<const f32x4_t (*)[8][4]>(&acc[h * 8]);
        if (MSQ_EPI_DIRECT) store_wave_tile_direct<YT>(acch, Y, m0 + h * 128, n0 + wid * 64, M, N, bias, lane, y16);
        else store_wave_tile_lds<YT>(acch, smem + wid * 8192, Y, m0 + h * 128, n0 + wid * 64, M, N, bias, lane, y16);
    }
}

struct DevOnceMx { std::atomic<uint64_t> mask{0}; };
inline bool attr_needed_mx(const DevOnceMx& o) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    return !(o.mask.load(std::memory_order_acquire) & (1ull << d));
}
inline void attr_done_mx(DevOnceMx& o) {
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) o.mask.fetch_or(1ull << d, std::memory_order_release);
}

}  // namespace

// Launcher (called by mx_linear, msq_gemm.hip).  Preconditions checked by the caller: N % 256 == 0, K % 128 == 0, offsets below 4 GiB.
// mf = 16: 256-row blocks; 8: 128-row blocks, two per CU.
extern "C" int msq_launch_mxgemm256(int wf, const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales, const float* bias,
                                    void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int mf, void* stream) {
    const int bmr = 16 * mf;
    const int MT = (int)((M + bmr - 1) / bmr), NTB = (int)(N / 256);
    const dim3 grid((unsigned)(MT * NTB)), blk(256);
    const size_t lds = 4 * ((size_t)bmr * 128 + 1024);
    const int y16 = (y_dtype == 1) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
#define MX256_LAUNCH(YT, WFV, MFV)                                                                                     \
    do { static DevOnceMx once_;                                                                                       \
         if (attr_needed_mx(once_)) { (void)hipFuncSetAttribute((const void*)k_mxgemm256<YT, WFV, MFV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done_mx(once_); } \
         hipLaunchKernelGGL((k_mxgemm256<YT, WFV, MFV>), grid, blk, lds, st, (const uint8_t*)x_codes, (const uint8_t*)x_scales, (const uint8_t*)w_codes, \
                            (const uint8_t*)w_scales, bias, (YT*)Y, (int)M, (int)N, (int)K, y16); } while (0)
#define MX256_MF(YT, WFV) do { if (mf == 16) MX256_LAUNCH(YT, WFV, 16); else MX256_LAUNCH(YT, WFV, 8); } while (0)
#define MX256_DISPATCH(YT) do { if (wf == 0) MX256_MF(YT, 0); else if (wf == 1) MX256_MF(YT, 1); else if (wf == 2) MX256_MF(YT, 2); else MX256_MF(YT, 3); } while (0)
    if (y_dtype == 0) MX256_DISPATCH(float); else MX256_DISPATCH(uint16_t);
#undef MX256_DISPATCH
#undef MX256_MF
#undef MX256_LAUNCH
    return (int)hipGetLastError();
}
