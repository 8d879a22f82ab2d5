// msq_gemm_stream.hip -- k_qgemm_sk: the fused unpack-dequant-GEMM of the unified layouts (MSQ-U1 / U1X) for the regime between decode
// and prefill, 32 < M <= 512 (msq_qlinear_bf16; replaces number_system/mx/linear.py:91 `F.linear` at the batch sizes `llm/opt.py:332-376`
// (`benchmark`) and chunked prefill produce).
//
// Why a third GEMM kernel.  Up to M ~ 300 the Linear is bound by the packed-weight stream (77.6 MB at N16384 K4096: 9.7 us at 8 TB/s,
// against 3.4 / 6.9 / 13.7 us of MFMA time at M = 64 / 128 / 256), yet the prefill kernels' 128 x 256 / 256 x 256 blocks give such a grid
// 64-128 blocks for 256 CUs; they filled the chip by cutting K over BLOCKS (split-K) and paid for it with fp32 partial planes written,
// read back and summed by a second launch (M = 128: 4 x 8.4 MB out and in again).  The decode kernels stream every packed byte once but
// re-read the activation rows per tile through the vector L1 and hold them in registers: beyond 32 rows they spill.
//
// Here the cut of K is made INSIDE the block: one block = WN strips of 64 columns x all of K, its KG groups of WN waves take the K-tiles
// kg, kg + KG, ... (one moving window of KG tiles per block: a quarter as many separate DRAM streams as contiguous runs), every wave keeps a
// (16 MF) x 64 accumulator tile over its tiles, and the KG partial tiles meet in LDS at the end -- summed in the fixed order of the
// groups (bit-identical run to run), every wave reducing and storing its own share of the block's output.  No partial plane, one launch,
// every packed byte read once, the grid is N / (64 WN) x ceil(M / 16 MF) blocks >= the CU count for the wide projections.
//   * packed planes: the same planes and tile order as every other kernel (nothing is re-packed), through buffer descriptors with SGPR
//     offsets, three tiles (six half-step sets) in flight per wave;
//   * activation tile of a K-tile (16 MF rows x 64 k, bf16): LDS-DMA into the group's own ring (XOR-swizzled 16-byte chunks, conflict-free
//     fragment reads), staged two tiles ahead.  WN = 1: the ring is private to the wave (two buffers, restaged behind the wave's own last read);
//     WN = 2: the two waves of a group share the tile (three buffers).  One counted vmcnt + block barrier per tile in every form: nothing else
//     orders LDS-DMA data for a ds_read, not even inside one wave;
//   * the per-wave stream is k_qgemm3's half-step (4 MFMAs per activation fragment, one quarter of the NEXT half-step's weight fragments
//     converted behind them: v_cvt_scalef32_pk_bf16_fp8 + the extension-bit rotate / and-or).
// Forms (launcher; the dispatcher's sk_rule picks 1 and 2): 1 = MF 4, WN 1, KG 8 (64 rows, eight waves, one strip); 2 = MF 8, WN 1, KG 4 (128 rows,
// four waves); measured and left to the MSQ_GEMM_SK switch: 3 = MF 8, WN 2, KG 2 (128 x 128 blocks), 4 = form 2 with eight waves and one activation
// buffer each, 5 = MF 4, WN 2, KG 4 (64 x 128 blocks) -- none of them beats 1 / 2 or the split-K GEMM where those do not apply.
// What bounds the kernel: the ~70 GB/s at which one CU pulls bytes out of L2 -- (2 M + 74) K bytes per strip and block (MI355X_MICROARCH.md, "Indexed rows").  Results are the fp32 sums of KG partial sums, each accumulated k ascending: within fp32 rounding
// of the single-pass kernels' (<= 2e-5 max|y| in the tests), identical from run to run.
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

#include "msq_gemm_common.h"

#ifndef MSQ_SK_ABL
#define MSQ_SK_ABL 0   /* timing experiments (results are wrong by construction; scripts/experiments/build_q256.sh -s msq_gemm_stream): 1 no activation
                          staging, 2 no MFMAs / converts, 8 no LDS reduction */
#endif
#ifndef MSQ_SK_ROT
#define MSQ_SK_ROT 1   /* 1: every block walks its K-tiles from its own starting tile (a fixed function of the block index): at any moment the blocks of an
                          XCD read DIFFERENT activation tiles and DRAM pages instead of all the same ones.  0: every block starts at tile 0 (A / B) */
#endif
#ifndef MSQ_SK_RING
#define MSQ_SK_RING 0  /* 3 / 4 force the depth of the packed-plane ring (0: 4 for the 64-row form, 3 for the 128-row forms) */
#endif
#ifndef MSQ_SK_NT
#define MSQ_SK_NT 0    /* 1: non-temporal packed loads in the WN = 1 forms (every packed byte is read by ONE block, once) */
#endif

namespace {

// one half-step of a wave: MFMAs of the current weight fragments (wf_use) against the MF activation fragments at `rd` (+ 2048 per
// fragment), the next half-step's fragments (wf_make) converted a quarter per group behind them
template <int OUT_KIND, int MF>
MSQ_D void sk_half_step(f32x4_t (&acc)[MF][4], const u32x4_t (&wf_use)[4], u32x4_t (&wf_make)[4],
                        const HalfRegs<MSQ_PLANE_NONE, OUT_KIND>& pk, const u32x4_t& sc, int kf_make, const char* rd) {
    constexpr int QPG = (MF >= 8) ? 1 : 8 / MF;                 // quarters converted per group
    constexpr int GPQ = (MF >= 8) ? MF / 8 : 1;                 // groups per quarter
    bf16x8_t xf[3];
    xf[0] = *reinterpret_cast<const bf16x8_t*>(rd);
    xf[1] = *reinterpret_cast<const bf16x8_t*>(rd + 2048);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        if (mf + 2 < MF) xf[(mf + 2) % 3] = *reinterpret_cast<const bf16x8_t*>(rd + (mf + 2) * 2048);
        if (MSQ_SK_ABL & 2) { asm volatile("" :: "v"(xf[mf % 3])); continue; }
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            if (MF >= 8)     // 128 accumulator registers: pinned to AGPRs by a tied inline-asm operand (hipcc left to itself parks part of them in
                             // VGPRs and shuttles 400+ v_accvgpr copies through the loop); an accumulator is touched once per half-step: no hazard
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mf][nf]) : "v"(wf_use[nf]), "v"(xf[mf % 3]));
            else
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf_use[nf]), xf[mf % 3], acc[mf][nf], 0, 0, 0);
        }
        if ((mf % GPQ) == 0) {
#pragma unroll
            for (int q = 0; q < QPG; ++q) convert_quarter<MSQ_PLANE_NONE, OUT_KIND>(wf_make, pk, sc, kf_make, (mf / GPQ) * QPG + q);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int OUT_KIND> struct TileSet {
    HalfRegs<MSQ_PLANE_NONE, OUT_KIND> h0, h1;
    u32x4_t sc;
};

template <int OUT_KIND, typename YT, int MF, int WN, int KG, int NB>
__global__ void __launch_bounds__(64 * WN * KG, 1)
k_qgemm_sk(const uint16_t* __restrict__ X, const uint8_t* __restrict__ ext_plane, const uint8_t* __restrict__ code_plane,
           const uint8_t* __restrict__ scl_plane, const float* __restrict__ bias, YT* __restrict__ Y, int M, int N, int K,
           int scl_groups, int y16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IN_KIND = MSQ_PLANE_NONE;
    // NB = activation buffers per k-group: 3 with WN > 1 (shared tile, one barrier per tile); WN = 1 (private ring): 2, or 1 where eight
    // waves of 128 rows leave no room for more (the tile is re-staged behind its last read and awaited at once: the SIMD's other wave covers it)
    static_assert((WN == 1 && (NB == 1 || NB == 2)) || (WN > 1 && (NB == 3 || NB == 2)), "activation ring");
    // WN > 1 with TWO buffers (form 6: eight waves on 128 x 128 blocks -- three buffers of 16 KiB for four k-groups do not fit): the tile's ONE
    // barrier does double duty -- every wave of the group has finished reading buffer b AND tile i + 1 (staged right behind the previous barrier,
    // awaited in front of this one) is visible -- and tile i + 2 is staged into b right behind it.
    constexpr bool LATE = (WN > 1 && NB == 2);
    constexpr int BM = 16 * MF;                                  // block rows
    constexpr int A_TILE = BM * BK * 2;                          // bytes of one activation tile
    constexpr int PPW = BM / 8 / WN;                             // 1 KiB staging pieces (8 rows) per wave and tile
    static_assert(PPW <= 16, "staging pieces per wave");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wid / WN, wn = wid % WN;
    const int c = lane & 15, g = lane >> 4;
    const int MT = (M + BM - 1) / BM, NTB = N / (64 * WN);
    const int bid = (int)blockIdx.x;
    int bm, bn;
    if ((NTB & 7) == 0) {                                        // the row blocks of a column panel on ONE XCD, back to back: its packed bytes come from HBM once
        const int xcd = bid & 7, i = bid >> 3;
        bm = i % MT; bn = (i / MT) * 8 + xcd;
    } else { bm = bid % MT; bn = bid / MT; }
    const int m0 = bm * BM;
    const int strip = bn * WN + wn;
    const int KT = K / BK;
    const int nt = sgpr((kg < KT) ? (KT - kg + KG - 1) / KG : 0);   // tiles of this wave: kg, kg + KG, ...  (WN > 1: the launcher guarantees KT % KG == 0)
    char* const smem_g = smem + kg * (NB * A_TILE);

    const int64_t ntiles = (int64_t)(N / TILE_N) * KT;
    PlaneRsrc pr;
    pr.inl = make_rsrc(ext_plane, ntiles * 2 * (OUT_KIND == MSQ_PLANE_U8X ? 256 : 1024));
    pr.out = make_rsrc(code_plane, ntiles * 2 * HalfSlots<OUT_KIND>::n * 1024);
    constexpr int SCLB = SclBytes<OUT_KIND>::n;
    pr.scl = make_rsrc(scl_plane, ntiles * scl_groups * SCLB);
    const int lane16 = lane * 16;
    const int scl_lane_off = (lane & (scl_groups - 1)) * SCLB;
    const uint32_t scl_tile_bytes = (uint32_t)scl_groups * (uint32_t)SCLB;
    const uint32_t tile_row32 = (uint32_t)sgpr(strip * KT);
    // tile index of this wave's i-th tile, clamped to its last one (branch-free tail: re-load / re-stage the last tile)
    const int i_last = sgpr(nt > 0 ? nt - 1 : 0);
    // ... rotated by the block's own offset: the sum over K of a block starts at tile `rot` and wraps -- a fixed order per block (the same from
    // run to run), another one than the single-pass kernels' (fp32 rounding apart, the same sum)
    const int rot = MSQ_SK_ROT ? sgpr((int)(((uint32_t)(bid >> 3) * 40503u + (uint32_t)(bid & 7) * 5u) % (uint32_t)KT)) : 0;
    auto kt_of = [&](int i) -> int {
        const int ic = i < i_last ? i : i_last;
        int kt = kg + ic * KG; kt = kt < KT ? kt : KT - 1;       // (kg >= KT: a wave without tiles re-reads tile KT - 1 and multiplies nothing)
        kt += rot;
        return kt >= KT ? kt - KT : kt;
    };
    auto load_set = [&](TileSet<OUT_KIND>& t, int i) {
        const uint32_t tile = tile_row32 + (uint32_t)kt_of(i);
        constexpr int AUX = (MSQ_SK_NT && WN == 1) ? 2 : 0;     // 2 = nt
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) {
            HalfRegs<IN_KIND, OUT_KIND>& h = kf ? t.h1 : t.h0;
            const uint32_t t2 = tile * 2u + (uint32_t)kf;
            h.out[0] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(pr.out, lane16, (t2 * 2u + 0u) * 1024u, AUX));
            h.out[1] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(pr.out, lane16, (t2 * 2u + 1u) * 1024u, AUX));
            if (OUT_KIND == MSQ_PLANE_U8X) h.ext = __builtin_amdgcn_raw_buffer_load_b32(pr.inl, lane16 >> 2, t2 * 256u, AUX);
        }
        const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(pr.scl, scl_lane_off, tile * scl_tile_bytes, AUX));
        t.sc = u32x4_t{v[0], v[1], 0u, 0u};
    };
    constexpr int SET_LOADS = 2 * HalfLoads<IN_KIND, OUT_KIND>::n + 1;   // vector-memory ops of one load_set: 5 (U8) / 7 (U8X)

    // activation staging: the group's WN waves copy the tile's BM rows as 1 KiB pieces (8 rows); lane l of piece pp fetches row
    // 8 pp + l / 8, source chunk (l & 7) ^ ((row >> 1) & 7): the LDS image stays lane-linear, fragment reads are conflict-free
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(X, (int64_t)M * K * 2);
    int aoff[16];
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int row = (wn * PPW + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < M ? gr : M - 1;
        aoff[p] = (int)(((int64_t)gr * K + chunk * 8) * 2);
    }
    auto stage = [&](int i, int buf) {
        if (MSQ_SK_ABL & 1) return;
        const uint32_t koff = (uint32_t)kt_of(i) * (BK * 2);
#pragma unroll
        for (int p = 0; p < PPW; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (void __attribute__((address_space(3)))*)(smem_g + buf * A_TILE + (wn * PPW + p) * 1024),
                                                     16, aoff[p], koff, 0, 0);
    };
    // REG: the private two-buffer ring is filled through REGISTERS (buffer_load_b128 -> ds_write_b128 into the same lane-linear image): LDS
    // writes and reads of one wave execute in order, the compiler's own vmcnt / lgkmcnt waits order everything, no barrier and no lockstep in
    // the K-loop -- with LDS-DMA every tile needs a block barrier (see below), and eight waves that wait for each other's loads at every tile
    // lose what the cut of K bought (64 rows, 16384 x 4096: 18.9 -> 21.9 us; 128 rows, 12288 x 4096: 21.6 -> 27.5).  The loads of tile i + 2 are
    // issued at the end of tile i IN FRONT of the packed loads (vmcnt is in-order: behind them they would wait for the whole ring) and written
    // to LDS at the end of tile i + 1.
    constexpr bool REG = (WN == 1 && NB == 2 && !(MSQ_SK_ABL & 16));
    u32x4_t xs[REG ? PPW : 1];
    auto xload = [&](int i) {
        const uint32_t koff = (uint32_t)kt_of(i) * (BK * 2);
#pragma unroll
        for (int p = 0; p < (REG ? PPW : 0); ++p) xs[p] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xr, aoff[p], koff, 0));
    };
    auto xwrite = [&](int buf) {
#pragma unroll
        for (int p = 0; p < (REG ? PPW : 0); ++p) *reinterpret_cast<u32x4_t*>(smem_g + buf * A_TILE + p * 1024 + lane16) = xs[p];
    };
    const int sw = (c >> 1) & 7;
    const int rd0 = c * 128 + (((0 + g) ^ sw) << 4);
    const int rd1 = c * 128 + (((4 + g) ^ sw) << 4);

    f32x4_t acc[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // packed-plane ring: RING tile sets in flight per wave (4 where the registers allow: two waves per SIMD at 64 accumulator registers; the
    // chip-wide stream is bound by bytes in flight -- 256 CUs x 8 waves x 4 tiles x 4.6 KB)
    constexpr int RING = (MSQ_SK_RING > 0) ? MSQ_SK_RING : ((MF <= 4) ? 4 : 3);
    TileSet<OUT_KIND> sA, sB, sC, sD;
    u32x4_t wfA[4], wfB[4];
    if (REG) xload(0);
    else { stage(0, 0); if (NB > 1) stage(1, 1); }
    load_set(sA, 0);
    load_set(sB, 1);
    load_set(sC, 2);
    if (RING == 4) load_set(sD, 3);
    __builtin_amdgcn_s_waitcnt(0);
    if (REG) { xwrite(0); xload(1); }
    else __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) convert_quarter<IN_KIND, OUT_KIND>(wfA, sA.h0, sA.sc, 0, q);
    // LDS-DMA data is ordered for a ds_read ONLY by the issuing wave's counted vmcnt followed by a barrier the reader has passed
    // (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait that retires it") -- also when issuer and reader are the
    // same wave: the first build of the private-ring forms read right behind its own vmcnt and returned rare wrong tiles once the ring was one
    // buffer deep (round 6, tests/test_gpu_n2_gemm_midm.py form 4).  So every tile ends with the wait AND a block barrier, and every wave runs
    // the same number of tiles (the block's maximum): a wave whose tiles have run out keeps staging / loading (clamped to its last tile: the
    // vmcnt arithmetic stays the same) and skips the arithmetic.
    const int nt_all = REG ? nt : sgpr((KT + KG - 1) / KG);

    // One tile of the wave.  In flight at its end, oldest first (vmcnt counts in issue order): [the loads of tile i + 2 (previous tile),]
    // the DMA pieces of tile i + 2, the loads of tile i + 3 -- everything older (the DMA pieces of tile i + 1) has landed.
    //   WN = 1 (private ring of two): tile i + 2 is staged into the buffer tile i was just read from, behind the last fragment read;
    //   WN = 2 (ring of three, one barrier per tile): tile i + 2 goes into the buffer tile i - 1 was read from, at the top of the tile.
    //   WN = 1, ONE buffer: tile i + 1 is staged into the only buffer behind the last read of tile i and awaited right away (only the loads of
    //   tile i + RING are younger)
    constexpr int N_WAIT = (NB == 1 || LATE) ? SET_LOADS : SET_LOADS + ((MSQ_SK_ABL & 1) ? 0 : PPW) + SET_LOADS;
    int buf = 0;
#define SK_TILE(I, CUR, NXT)                                                                                       \
    {                                                                                                              \
        const char* abase = smem_g + buf * A_TILE;                                                                 \
        const int buf2 = (NB <= 2) ? buf : ((buf == 0) ? 2 : buf - 1);          /* (buf + 2) % 3 */                \
        if (NB == 3) { stage((I) + 2, buf2); __builtin_amdgcn_sched_barrier(0); }                                  \
        if (REG || (I) < nt) {                                                                                     \
            sk_half_step<OUT_KIND, MF>(acc, wfA, wfB, CUR.h1, CUR.sc, 1, abase + rd0);                             \
            keep_live(NXT.h0); keep_live4(NXT.sc);     /* hipcc's vmcnt wait for tile i + 1's planes lands here */  \
            sk_half_step<OUT_KIND, MF>(acc, wfB, wfA, NXT.h0, NXT.sc, 0, abase + rd1);                             \
        }                                                                                                          \
        if (REG) {                                                                                                 \
            xwrite(buf ^ 1);                       /* tile i + 1 (requested a tile ago) into the buffer tile i - 1 was read from */ \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            xload((I) + 2);                        /* in front of the packed loads: in-order vmcnt */              \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            load_set(CUR, (I) + RING);                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        } else if (LATE) {                                                                                         \
            __builtin_amdgcn_s_waitcnt(0x0070 | (N_WAIT & 15) | ((N_WAIT >> 4) << 14));     /* vmcnt(N_WAIT) lgkmcnt(0): tile i + 1 has landed, this wave's reads are done */ \
            __builtin_amdgcn_s_barrier();                                                                          \
            stage((I) + 2, buf2);                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            load_set(CUR, (I) + RING);                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        } else {                                                                                                   \
            if (NB <= 2) { __builtin_amdgcn_s_waitcnt(0xC07F); stage((I) + NB, buf2); __builtin_amdgcn_sched_barrier(0); }   /* lgkmcnt(0): this wave's reads of the buffer are done */ \
            load_set(CUR, (I) + RING);                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            __builtin_amdgcn_s_waitcnt(0x0F70 | (N_WAIT & 15) | ((N_WAIT >> 4) << 14));                            \
            __builtin_amdgcn_s_barrier();                                                                          \
        }                                                                                                          \
        buf = (buf + 1 == NB) ? 0 : buf + 1;                                                                       \
    }
    if constexpr (RING == 3) {
        int i = 0;
        for (; i + 2 < nt_all; i += 3) {
            SK_TILE(i, sA, sB)
            SK_TILE(i + 1, sB, sC)
            SK_TILE(i + 2, sC, sA)
        }
        if (i < nt_all) {
            SK_TILE(i, sA, sB)
            if (i + 1 < nt_all) SK_TILE(i + 1, sB, sC)
        }
    } else {
        int i = 0;
        for (; i + 3 < nt_all; i += 4) {
            SK_TILE(i, sA, sB)
            SK_TILE(i + 1, sB, sC)
            SK_TILE(i + 2, sC, sD)
            SK_TILE(i + 3, sD, sA)
        }
        if (i < nt_all) {
            SK_TILE(i, sA, sB)
            if (i + 1 < nt_all) {
                SK_TILE(i + 1, sB, sC)
                if (i + 2 < nt_all) SK_TILE(i + 2, sC, sD)
            }
        }
    }
#undef SK_TILE

    // ---- the KG partial tiles meet in LDS: red[wave][quad (mf, nf)][lane] float4 (conflict-free), summed in the order of the groups
    __builtin_amdgcn_s_waitcnt(0x0070);                          // vmcnt(0) lgkmcnt(0): the re-staged tail tiles have landed, nothing reads the ring any more
    __builtin_amdgcn_s_barrier();
    if (MF >= 8) acc_fence<MF>(acc);                             // the asm MFMAs are opaque to hipcc's hazard recogniser (msq_gemm_common.h)
    // (eight waves of 128 rows: 256 KiB of partial tiles -- they meet in two phases of MF / 2 fragment rows each)
    constexpr int PH = (WN * KG * MF * 4 > 128) ? 2 : 1;
    constexpr int MFP = MF / PH;                                 // fragment rows per phase
    constexpr int UPP = MFP * 2 / KG;                            // units per wave and phase
    static_assert(MFP * 2 % KG == 0 && UPP >= 1, "every wave reduces and stores at least one unit per phase");
    float4* red = reinterpret_cast<float4*>(smem);
    const int n_base = strip * TILE_N;
    const int colp = (g & 1) ? 16 + (g - 1) * 4 : g * 4;         // 16-bit outputs: this lane's 8 columns inside the unit's 32 after the swap (see store_wave_tile_direct_)
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
        if (ph > 0) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }     // the previous phase has been read
#pragma unroll
        for (int ml = 0; ml < MFP; ++ml)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const f32x4_t& a4 = acc[ph * MFP + ml][nf];
                red[(wid * (MFP * 4) + ml * 4 + nf) * 64 + lane] = make_float4(a4[0], a4[1], a4[2], a4[3]);
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int uu = 0; uu < UPP; ++uu) {
            if (MSQ_SK_ABL & 8) break;
            const int u = kg * UPP + uu;                         // unit of this phase: fragment row ml, quad pair (a, a + 1)
            const int ml = u >> 1, a = (u & 1) * 2, mf = ph * MFP + ml;
            f32x4_t s[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 t = red[((0 * WN + wn) * (MFP * 4) + ml * 4 + a + q) * 64 + lane];
#pragma unroll
                for (int k2 = 1; k2 < KG; ++k2) {
                    const float4 p = red[((k2 * WN + wn) * (MFP * 4) + ml * 4 + a + q) * 64 + lane];
                    t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
                }
                s[q] = f32x4_t{t.x, t.y, t.z, t.w};
                if (bias) {
                    const float4 b = *reinterpret_cast<const float4*>(bias + n_base + (a + q) * 16 + g * 4);
                    s[q][0] += b.x; s[q][1] += b.y; s[q][2] += b.z; s[q][3] += b.w;
                }
            }
            const int m = m0 + mf * 16 + c;
            char* rowp = reinterpret_cast<char*>(Y) + ((int64_t)m * N + n_base) * (int64_t)sizeof(YT);
            if (sizeof(YT) == 4) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (m < M) *reinterpret_cast<float4*>(rowp + ((a + q) * 16 + g * 4) * 4) = make_float4(s[q][0], s[q][1], s[q][2], s[q][3]);
            } else {
                uint32_t d[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (y16) {
                        f16x2_t lo, hi;
                        lo[0] = (_Float16)s[q][0]; lo[1] = (_Float16)s[q][1]; hi[0] = (_Float16)s[q][2]; hi[1] = (_Float16)s[q][3];
                        d[q][0] = __builtin_bit_cast(uint32_t, lo); d[q][1] = __builtin_bit_cast(uint32_t, hi);
                    } else {
                        bf16x2_t lo, hi;
                        lo[0] = (__bf16)s[q][0]; lo[1] = (__bf16)s[q][1]; hi[0] = (__bf16)s[q][2]; hi[1] = (__bf16)s[q][3];
                        d[q][0] = __builtin_bit_cast(uint32_t, lo); d[q][1] = __builtin_bit_cast(uint32_t, hi);
                    }
                }
                const auto r0 = __builtin_amdgcn_permlane16_swap(d[0][0], d[1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(d[0][1], d[1][1], false, false);
                const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
                if (m < M) *reinterpret_cast<u32x4_t*>(rowp + (a * 16 + colp) * 2) = o;
            }
        }
    }
}

struct DevOnceS { std::atomic<uint64_t> mask{0}; };
inline bool attr_needed_s(const DevOnceS& o) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    return !(o.mask.load(std::memory_order_acquire) & (1ull << d));
}
inline void attr_done_s(DevOnceS& o) {
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) o.mask.fetch_or(1ull << d, std::memory_order_release);
}

}  // namespace

// Shape of the block for M rows (0 = the kernel does not apply): form 1 = MF 4, WN 1, KG 8; 2 = MF 8, WN 1, KG 4; 3 = MF 8, WN 2, KG 2;
// 4 = MF 8, WN 1, KG 8 with ONE activation buffer per wave (eight waves of 128 rows: two per SIMD); 5 = MF 4, WN 2, KG 4 (64 x 128 blocks);
// 6 = MF 8, WN 2, KG 4 (128 x 128 blocks, EIGHT waves, two activation buffers per k-group, two-phase reduction).
// `form` > 0 forces (tests, A / B).  Preconditions checked by the caller: unified layout, bf16 activations, K % 64 == 0, N % 256 == 0.
extern "C" int msq_qgemm_sk_form(int64_t M, int64_t N, int64_t K, int form) {
    if (form <= 0) form = (M <= 64) ? 1 : ((M <= 128) ? 2 : 3);
    const int64_t KT = K / 64;
    if (form == 3 && ((KT & 1) || (N % 128))) return 0;
    if ((form == 5 || form == 6) && ((KT & 3) || (N % 128))) return 0;
    if (form < 1 || form > 6 || KT < 1) return 0;
    return form;
}

extern "C" int msq_launch_qgemm_sk(const void* X, const void* ext_plane, const void* code_plane, const void* scale_plane, const float* bias, void* Y,
                                   int y_dtype, int64_t M, int64_t N, int64_t K, int out_kind, int scl_groups, int form, void* stream) {
    form = msq_qgemm_sk_form(M, N, K, form);
    if (!form) return (int)hipErrorInvalidValue;
    const int mf = (form == 1 || form == 5) ? 4 : 8, wn = (form == 3 || form == 5 || form == 6) ? 2 : 1,
              kg = (form == 1 || form == 4) ? 8 : ((form == 2 || form == 5 || form == 6) ? 4 : 2);
    const int bm = 16 * mf, nb = (form == 4) ? 1 : ((wn == 1 || form == 6) ? 2 : 3);
    const int MT = (int)((M + bm - 1) / bm), NTB = (int)(N / (64 * wn));
    const dim3 grid((unsigned)(MT * NTB)), blk((unsigned)(64 * wn * kg));
    size_t ring = (size_t)kg * nb * bm * 128, redb = (size_t)wn * kg * mf * 4 * 1024;
    if (redb > 131072) redb /= 2;                                // two phases (see the kernel)
    const size_t lds = ring > redb ? ring : redb;
    const int y16 = (y_dtype == 1) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
#define SK_LAUNCH(OK, YT, MFV, WNV, KGV, NBV)                                                                              \
    do { static DevOnceS once_;                                                                                        \
         if (attr_needed_s(once_)) { (void)hipFuncSetAttribute((const void*)k_qgemm_sk<OK, YT, MFV, WNV, KGV, NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done_s(once_); } \
         hipLaunchKernelGGL((k_qgemm_sk<OK, YT, MFV, WNV, KGV, NBV>), grid, blk, lds, st, (const uint16_t*)X, (const uint8_t*)ext_plane, (const uint8_t*)code_plane, \
                            (const uint8_t*)scale_plane, bias, (YT*)Y, (int)M, (int)N, (int)K, scl_groups, y16); } while (0)
#define SK_FORM(OK, YT) do { if (form == 1) SK_LAUNCH(OK, YT, 4, 1, 8, 2); else if (form == 2) SK_LAUNCH(OK, YT, 8, 1, 4, 2); \
                             else if (form == 3) SK_LAUNCH(OK, YT, 8, 2, 2, 3); else if (form == 4) SK_LAUNCH(OK, YT, 8, 1, 8, 1); \
                             else if (form == 5) SK_LAUNCH(OK, YT, 4, 2, 4, 3); else SK_LAUNCH(OK, YT, 8, 2, 4, 2); } while (0)
    if (out_kind == MSQ_PLANE_U8) { if (y_dtype == 0) SK_FORM(MSQ_PLANE_U8, float); else SK_FORM(MSQ_PLANE_U8, uint16_t); }
    else { if (y_dtype == 0) SK_FORM(MSQ_PLANE_U8X, float); else SK_FORM(MSQ_PLANE_U8X, uint16_t); }
#undef SK_FORM
#undef SK_LAUNCH
    return (int)hipGetLastError();
}
