// msq_gemm256p.hip -- k_qgemm256p: the PERSISTENT form of k_qgemm256 (msq_gemm256.hip): one resident block per CU walks a list of
// work segments, so that (a) a grid that is not a whole number of rounds of 256 x 256 tiles over the 256 CUs loses nothing -- the
// tiles of the last, part-filled round are cut along K over ALL blocks (stream-K) and summed through fp32 partial tiles in the
// workspace -- and (b) the stores of one tile and the first loads of the next overlap: the K-loop runs on across the tile
// boundary (the loads of the next tile's first two K-steps are issued during the last two K-steps of this one), the epilogue's
// global stores are issued and NOT waited for.  (msq_qlinear_bf16 at prefill sizes; replaces number_system/mx/linear.py:91
// `F.linear` on weights whose values are those of utils/quant.py:147-266; true Llama-2-7B shapes llm/llama.py:226-256.)
//
// The K-step itself is k_qgemm256's, instruction for instruction (tied-accumulator MFMAs in a[0:255], one filler per MFMA shadow,
// LDS-DMA activation tiles in four buffers, packed planes through buffer descriptors): see the header of msq_gemm256.hip.
// What is new:
//   * SCHEDULE.  T = MT x NTB tiles, P blocks (one per CU).  `full` = T / P rounds of whole tiles are dealt as before (block b takes
//     tile ids b, b + P, ...: the XCD-aware order of k_qgemm256).  The R = T - full P remaining tiles form a flat stream of R x KT
//     K-steps that is cut into P equal runs of q K-steps (q even): block b owns [b q, (b + 1) q).  A run covers the END of one tile
//     (from K-step a > 0: a "tail piece") and / or the BEGINNING of the next (a "head piece").  Every block runs its stream-K run
//     first (tail piece, then head piece), then its whole tiles.
//   * FIX-UP.  A tail piece leaves its 256 x 256 fp32 accumulators in workspace slot b (write-through stores, then ONE flag word per
//     block).  The block that holds the head piece of a tile (K-step 0 onward) finishes it: it polls the flags of the blocks that hold
//     the rest of the tile, adds their slots in block order (a fixed order: results repeat bit for bit) and writes Y.  A head piece is
//     the LAST thing a block does in its run and a tail piece the FIRST thing its neighbour does, so the data are there when the owner
//     looks (no block ever waits for a block that waits: tail pieces wait for nothing).  Hand-off as MI355X guide G16, R1: sc1 stores,
//     every storing wave drains vmcnt, barrier, one relaxed agent-scope flag store; consumer: one lane polls (bounded), barrier,
//     sc1 loads.  The flags are zeroed by a memset node in front of every launch (msq_launch_qgemm256p).
//   * TILE SWITCH.  First K-step of a segment: MFMAs with C = 0 (no accumulator clearing); its barrier lets the epilogue's stores stay
//     in flight (vmcnt counts in order: N_WAIT + the stores).  The first two K-steps of a segment are peeled so that hipcc's own
//     waits for the packed loads see the stores in front of them.  Activation rows and output rows beyond M are handled by the
//     range check of per-tile buffer descriptors (base = first row of the tile, records = valid rows): every store instruction is
//     always issued, which is what the vmcnt arithmetic relies on.
// Sums: a tile that is not cut accumulates exactly as in k_qgemm256 (bit-identical); a cut tile is the fp32 sum of its pieces'
// accumulators in K order -- within fp32 rounding of the uncut sum, identical from run to run.
#include <stdio.h>
#include <stdlib.h>
#include <atomic>

#include "msq_gemm_common.h"

#ifndef MSQ_QP_RT
#define MSQ_QP_RT 4            /* row tiles per XCD super-tile */
#endif
#ifndef MSQ_QP_PF
#define MSQ_QP_PF 2            /* activation-fragment reads in flight ahead of their MFMA group */
#endif
#ifndef MSQ_QP_NT
#define MSQ_QP_NT 0            /* aux bits of the Y stores (2 = non-temporal) */
#endif
#ifndef MSQ_QP_ABL
#define MSQ_QP_ABL 0           /* timing experiments (wrong results): 1 no output stores */
#endif

namespace {

typedef __attribute__((address_space(1))) unsigned int gu32;

// D(a[..]) += A(weight fragment, v) x B(activation fragment, v): accumulator tied in place in an AGPR quad
MSQ_D void mfma_acc(f32x4_t& acc, const u32x4_t& w, const bf16x8_t& x) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
}
// D(a[..]) = A x B: the first MFMA of a tile on this quad ("+a": same registers, the old value is ignored by the instruction)
MSQ_D void mfma_new(f32x4_t& acc, const u32x4_t& w, const bf16x8_t& x) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "+a"(acc) : "v"(w), "v"(x));
}

template <int OUT_KIND>
MSQ_D uint32_t cvt_dword(const HalfRegs<MSQ_PLANE_NONE, OUT_KIND>& h, float sop, int d) {
    const int nf = d >> 2, j = d & 3;
    const uint32_t o = h.out[nf >> 1][(nf & 1) * 2 + (j >> 1)];
    return __builtin_bit_cast(uint32_t, (j & 1) ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o, sop, true)
                                                : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(o, sop, false));
}
template <int OUT_KIND>
MSQ_D uint32_t ext_dword(uint32_t r, const HalfRegs<MSQ_PLANE_NONE, OUT_KIND>& h, int d) {
    return (OUT_KIND == MSQ_PLANE_U8X) ? ext_or(r, h.ext, d >> 2, d & 3) : r;
}

// 16 bytes of the segment table, by hand: for a ds_read of its own hipcc first waits for every LDS-DMA piece in flight (vmcnt(0): the
// pieces write LDS) -- that would drain the next tile's prefetch at every tile switch.  The table is written once, before any DMA.
MSQ_D int4 lds_read16(int byte_addr) {
    int4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(byte_addr) : "memory");
    return v;
}
MSQ_D __amdgpu_buffer_rsrc_t rsrc_of(uint64_t base, uint32_t records) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, records, 0x00020000);
}

// tile id (dispatch order of k_qgemm256) -> tile coordinates
MSQ_D void tile_coords(int bid, int MT, int NTB, int& bm, int& bn) {
    if ((NTB & 7) == 0) {
        constexpr int RT = MSQ_QP_RT;
        const int xcd = bid & 7, i = bid >> 3;
        const int npx = NTB >> 3, per_group = RT * npx, full = MT / RT;
        int rg, j, R;
        if (i < full * per_group) { rg = i / per_group; j = i % per_group; R = RT; }
        else { rg = full; j = i - full * per_group; R = MT - full * RT; }
        bm = rg * RT + j % R;
        bn = (j / R) * 8 + xcd;
    } else { bm = bid % MT; bn = bid / MT; }
}

// Epilogue of a 128(m) x 64(n) half wave tile through the wave's 8 KiB LDS slice (store_wave_tile_lds of msq_gemm_common.h) with buffer
// stores: `yr` = descriptor of the tile's rows (range check drops rows >= M), every store instruction is issued whatever M is.
// Head pieces add the partial tiles of blocks peer0 .. peer1 (workspace slots, lane-linear quads, sc1 loads) while the accumulators
// are read: the accumulators themselves are never written outside the tied MFMAs (any other definition makes hipcc shuffle them
// between register files at every join of the control flow).  `qbase` = first quad of this half (h * 32).
template <typename YT>
MSQ_D void store_half_tile(const f32x4_t (&acc)[8][4], char* wsm, __amdgpu_buffer_rsrc_t yr, int row0, int N,
                           const float (&bv)[4][4], int lane, int y16, uint64_t slots, int peer0, int peer1, int wid, int qbase, u32x4_t (&pf)[8]) {
    const int c = lane & 15, g = lane >> 4;
    constexpr int ROW_B = 64 * (int)sizeof(YT);
    constexpr int RP = 8192 / ROW_B;
    constexpr int MF_PER_PASS = RP / 16;
    constexpr int CHUNKS = ROW_B / 16;
    constexpr int QP = MF_PER_PASS * 4;                      // quads per pass: 16 / 8
#pragma unroll
    for (int p = 0; p < 8 / MF_PER_PASS; ++p) {
#pragma unroll
        for (int sp = 0; sp < QP / 8; ++sp) {                // eight quads (two fragments mf) at a time: the K-loop's registers stay live across the epilogue
            __builtin_amdgcn_sched_barrier(0);
            f32x4_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = acc[p * MF_PER_PASS + sp * 2 + (k >> 2)][k & 3];
            if (peer1 >= peer0) {                            // head piece: add the peers' partial tiles, in block order
                const int sidx = (qbase >> 3) + p * (QP / 8) + sp;       // this batch of eight quads within the wave's 64
                for (int pb = peer0; pb <= peer1; ++pb) {
                    // the batch in `pf` was loaded one step ago; fetch the next one (next peer of this batch, else the first peer of the
                    // next batch) before this one is added: sixteen 1 KiB loads in flight per wave instead of eight
                    const bool more = pb < peer1;
                    const int pbn = more ? pb + 1 : peer0, sn = more ? sidx : sidx + 1;
                    const __amdgpu_buffer_rsrc_t srn = __builtin_amdgcn_make_buffer_rsrc((void*)(slots + (uint64_t)pbn * 262144ull), 0, 262144u, 0x00020000);
                    u32x4_t nx[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        nx[k] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(srn, lane * 16, (wid * 64 + sn * 8 + k) * 1024, 16));
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += __builtin_bit_cast(f32x4_t, pf[k]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) pf[k] = nx[k];
                }
            }
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const int i = sp * 2 + i2;
                const int row = i * 16 + c;
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    f32x4_t w = v[i2 * 4 + nf];
                    w[0] += bv[nf][0]; w[1] += bv[nf][1]; w[2] += bv[nf][2]; w[3] += bv[nf][3];
                    if (sizeof(YT) == 4) {
                        const int chunk = (nf * 4 + g) ^ (row & (CHUNKS - 1));
                        *reinterpret_cast<float4*>(wsm + row * ROW_B + chunk * 16) = make_float4(w[0], w[1], w[2], w[3]);
                    } else {
                        uint32_t plo, phi;
                        if (y16) {
                            f16x2_t lo, hi;
                            lo[0] = (_Float16)w[0]; lo[1] = (_Float16)w[1]; hi[0] = (_Float16)w[2]; hi[1] = (_Float16)w[3];
                            plo = __builtin_bit_cast(uint32_t, lo); phi = __builtin_bit_cast(uint32_t, hi);
                        } else {
                            bf16x2_t lo, hi;
                            lo[0] = (__bf16)w[0]; lo[1] = (__bf16)w[1]; hi[0] = (__bf16)w[2]; hi[1] = (__bf16)w[3];
                            plo = __builtin_bit_cast(uint32_t, lo); phi = __builtin_bit_cast(uint32_t, hi);
                        }
                        const int chunk = (nf * 2 + (g >> 1)) ^ (row & (CHUNKS - 1));
                        *reinterpret_cast<uint2*>(wsm + row * ROW_B + chunk * 16 + (g & 1) * 8) = make_uint2(plo, phi);
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        constexpr int ROWS_PER_INSTR = 64 / CHUNKS;
#pragma unroll
        for (int t = 0; t < RP / ROWS_PER_INSTR; ++t) {
            const int row = t * ROWS_PER_INSTR + lane / CHUNKS;
            const int chunk = lane % CHUNKS;
            const u32x4_t d = *reinterpret_cast<const u32x4_t*>(wsm + row * ROW_B + ((chunk ^ (row & (CHUNKS - 1))) * 16));
            const int voff = (row0 + p * RP + row) * N * (int)sizeof(YT) + chunk * 16;
            if (!(MSQ_QP_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(d, yr, voff, 0, MSQ_QP_NT);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // reads done before the next pass overwrites
        __builtin_amdgcn_wave_barrier();
    }
}

// one work segment of a block: K-steps [kt0, kt1) of tile `tile`; role 0 = writes Y, 1 = producer piece (leaves a partial tile in slot
// `blockIdx`), 2 = owner's piece (adds the slots of blocks peer0 .. peer1 -- all below its own index --, then writes Y)
struct Seg { int tile, kt0, kt1, role, peer0, peer1; };

// Host and device share this arithmetic (msq_qgemm256p_segments below lists a block's segments for the tests).  P blocks, `full` whole rounds, R stream-K tiles of KT
// K-steps cut into runs of q.
__host__ __device__ inline int seg_count(int b, int P, int full, int R, int KT, int q) {
    const int U = R * KT, u0 = b * q;
    if (u0 >= U) return full;
    const int u1 = (u0 + q < U) ? u0 + q : U;
    const int ka0 = u0 % KT;
    const int lenA = (KT - ka0 < u1 - u0) ? KT - ka0 : u1 - u0;
    return full + 1 + ((u0 + lenA < u1) ? 1 : 0);
}
__host__ __device__ inline Seg seg_get(int i, int b, int P, int full, int R, int KT, int q) {
    Seg s;
    const int U = R * KT, u0 = b * q;
    int nsk = 0, sA = 0, ka0 = 0, lenA = 0, lenB = 0;
    if (u0 < U) {
        const int u1 = (u0 + q < U) ? u0 + q : U;
        sA = u0 / KT; ka0 = u0 - sA * KT;
        lenA = (KT - ka0 < u1 - u0) ? KT - ka0 : u1 - u0;
        lenB = u1 - u0 - lenA;
        nsk = 1 + (lenB > 0 ? 1 : 0);
    }
    // A run covers the END of one stream-K tile (piece A) and / or the BEGINNING of the next (piece B).  The block that holds a tile's LAST
    // K-steps owns it: it waits for the partial tiles of the blocks that hold the tile's earlier K-steps -- LOWER block indices only, so the
    // wait never depends on a block that has not been dispatched yet (workgroups start in index order): no co-residency assumption.  A
    // producer piece is the FIRST thing its block does (nothing it waits for), the whole tiles follow, the owner's piece comes LAST: the
    // partial tiles it needs were written a whole round of tiles earlier.
    const bool hasB = lenB > 0;
    const bool Aends = (nsk > 0) && (ka0 + lenA == KT);
    const bool Aprod = (nsk > 0) && !Aends;                      // A stops inside its tile: a producer (then there is no piece B)
    const int nfirst = (hasB || Aprod) ? 1 : 0;
    if (i >= nfirst && i < nfirst + full) { s.tile = b + (i - nfirst) * P; s.kt0 = 0; s.kt1 = KT; s.role = 0; s.peer0 = 0; s.peer1 = -1; return s; }
    s.peer0 = 0; s.peer1 = -1;
    if (i < nfirst && hasB) {                                    // the beginning of the next tile: producer (a whole tile when q >= KT)
        s.tile = full * P + sA + 1; s.kt0 = 0; s.kt1 = lenB; s.role = (lenB == KT) ? 0 : 1;
        return s;
    }
    s.tile = full * P + sA; s.kt0 = ka0; s.kt1 = ka0 + lenA;
    if (Aprod) s.role = 1;
    else if (ka0 > 0) { s.role = 2; s.peer0 = (sA * KT) / q; s.peer1 = b - 1; }     // owner: the blocks whose runs hold K-steps [0, ka0) of this tile
    else s.role = 0;
    return s;
}

template <int OUT_KIND, typename YT>
__global__ void __launch_bounds__(256, 1)
k_qgemm256p(const uint16_t* __restrict__ X, const uint8_t* __restrict__ ext_plane, const uint8_t* __restrict__ code_plane,
            const uint8_t* __restrict__ scl_plane, const float* __restrict__ bias, YT* __restrict__ Y, int M, int N, int K,
            int scl_groups, int y16, int P, int full, int R, int q, char* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IN_KIND = MSQ_PLANE_NONE;
    constexpr int MF = 16;
    constexpr int BM = 256;
    constexpr int A_TILE = BM * BK * 2;                          // 32 KiB per activation buffer
    constexpr int PPW = 8;                                       // 1 KiB staging pieces (8 rows) per wave and K-step
    constexpr int PF = MSQ_QP_PF;
    constexpr int TAB_OFF = 4 * A_TILE;                          // the block's segment table lies behind the four activation buffers
    constexpr int TAB_MAX = 256;                                 // segments per block (32 bytes each); msq_qgemm256p_plan refuses longer lists
    static_assert(PF >= 1 && PF <= 3, "fragment ring of four");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int MT = (M + BM - 1) / BM, NTB = N / 256;
    const int KT = K / BK;
    const int b = (int)blockIdx.x;
    const int nseg = sgpr(seg_count(b, P, full, R, KT, q));
    if (nseg == 0) return;

    const int64_t ntiles = (int64_t)(N / TILE_N) * KT;
    PlaneRsrc pr;
    pr.inl = make_rsrc(ext_plane, ntiles * 2 * (OUT_KIND == MSQ_PLANE_U8X ? 256 : 1024));
    pr.out = make_rsrc(code_plane, ntiles * 2 * HalfSlots<OUT_KIND>::n * 1024);
    constexpr int SCLB = SclBytes<OUT_KIND>::n;
    pr.scl = make_rsrc(scl_plane, ntiles * scl_groups * SCLB);
    const int lane16 = lane * 16;
    const int scl_lane_off = (lane & (scl_groups - 1)) * SCLB;
    const uint32_t scl_tile_bytes = (uint32_t)scl_groups * (uint32_t)SCLB;
    auto load_scales = [&](uint32_t tile) -> u32x4_t {
        const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(pr.scl, scl_lane_off, tile * scl_tile_bytes, 0));
        return u32x4_t{v[0], v[1], 0u, 0u};
    };

    // activation staging: wave w copies rows 64 w .. 64 w + 63 of the tile as eight 1 KiB pieces (8 rows each); lane l of piece p fetches
    // row 8 p + l / 8, source chunk (l & 7) ^ ((row >> 1) & 7).  The per-lane offset is the same for every tile: the tile's first row sits
    // in the descriptor's base and the rows the tile does not have are beyond its records.
    int aoff[8];
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int row = (wid * PPW + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        aoff[p] = (row * K + chunk * 8) * 2;
    }
    auto stage_piece = [&](__amdgpu_buffer_rsrc_t xr_, uint32_t xs_, int buf, int p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_, (void __attribute__((address_space(3)))*)(smem + buf * A_TILE + (wid * PPW + p) * 1024),
                                                 16, aoff[p], xs_, 0, 0);
    };
    auto load_part = [&](HalfRegs<IN_KIND, OUT_KIND>& h, uint32_t tile2kf, int part) {
        if (part < 2) h.out[part] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(pr.out, lane16, (tile2kf * 2u + (uint32_t)part) * 1024u, 0));
        else if (OUT_KIND == MSQ_PLANE_U8X) h.ext = __builtin_amdgcn_raw_buffer_load_b32(pr.inl, lane16 >> 2, tile2kf * 256u, 0);
    };
    const int sw = (c >> 1) & 7;
    const int rd0 = c * 128 + (((0 + g) ^ sw) << 4);
    const int rd1 = c * 128 + (((4 + g) ^ sw) << 4);

    f32x4_t acc[MF][4];                                          // a[0:255]; every segment's first half-step writes all of them (mfma_new)
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "=a"(acc[i][j]));

    HalfRegs<IN_KIND, OUT_KIND> pk0, pk1, pk2, pk3;
    u32x4_t wfA[4], wfB[4];
    u32x4_t sc_cur = {0, 0, 0, 0}, sc_nxt = {0, 0, 0, 0}, sc_nn = {0, 0, 0, 0};
    bf16x8_t xf[4];

    // per-tile scalars: X descriptor (base = first row of the tile), first packed tile of this wave's 64-column strip
    const uint64_t x_addr = (uint64_t)X;
    auto x_base_of = [&](int m0) -> uint64_t { return x_addr + (uint64_t)m0 * (uint64_t)K * 2ull; };
    auto x_rec_of = [&](int m0) -> uint32_t { const int rows = (M - m0 < BM) ? M - m0 : BM; return (uint32_t)rows * (uint32_t)K * 2u; };

    // The block's segment list, once, into LDS: {m0, n0, kt0, kt1, role, peer0, peer1, -}; one thread per segment (the integer divisions
    // of the tile order run in parallel lanes, and no per-segment scalar is carried through the K-loop).
    int* tab = reinterpret_cast<int*>(smem + TAB_OFF);
    for (int i = tid; i < nseg; i += 256) {
        const Seg sgi = seg_get(i, b, P, full, R, KT, q);
        int bmi, bni;
        tile_coords(sgi.tile, MT, NTB, bmi, bni);
        int4 lo = make_int4(bmi * BM, bni * 256, sgi.kt0, sgi.kt1), hi = make_int4(sgi.role, sgi.peer0, sgi.peer1, 0);
        *reinterpret_cast<int4*>(tab + i * 8) = lo;
        *reinterpret_cast<int4*>(tab + i * 8 + 4) = hi;
    }
    __syncthreads();
    const int tab_addr = (int)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + TAB_OFF);
    int kt0 = sgpr(tab[2]), kt1 = sgpr(tab[3]);
    uint64_t xb = x_base_of(sgpr(tab[0]));
    uint32_t xn = x_rec_of(sgpr(tab[0]));
    int trow = sgpr((sgpr(tab[1]) / TILE_N + wid) * KT);

    // prologue of the block: the first two K-steps of its first segment
    {
        const __amdgpu_buffer_rsrc_t xr0 = rsrc_of(xb, xn);
#pragma unroll
        for (int p = 0; p < PPW; ++p) stage_piece(xr0, (uint32_t)kt0 * 128u, 0, p);
#pragma unroll
        for (int p = 0; p < PPW; ++p) stage_piece(xr0, (uint32_t)(kt0 + 1) * 128u, 1, p);
        const uint32_t t0 = (uint32_t)(trow + kt0);
        load_half_buf<IN_KIND, OUT_KIND>(pk0, pr, lane16, t0 * 2u + 0u);
        load_half_buf<IN_KIND, OUT_KIND>(pk1, pr, lane16, t0 * 2u + 1u);
        load_half_buf<IN_KIND, OUT_KIND>(pk2, pr, lane16, (t0 + 1u) * 2u + 0u);
        load_half_buf<IN_KIND, OUT_KIND>(pk3, pr, lane16, (t0 + 1u) * 2u + 1u);
        sc_cur = load_scales(t0);
        sc_nxt = load_scales(t0 + 1u);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) wfA[d >> 2][d & 3] = ext_dword<OUT_KIND>(cvt_dword<OUT_KIND>(pk0, scale_operand(sc_cur[0], d >> 2), d), pk0, d);
    float sop = scale_operand(sc_cur[1], 0);
#pragma unroll
    for (int f = 0; f < PF; ++f) xf[f] = *reinterpret_cast<const bf16x8_t*>(smem + rd0 + f * 2048);

    constexpr int HL = HalfLoads<IN_KIND, OUT_KIND>::n;          // 2 (U8) / 3 (U8X)
    constexpr int BAR_G = MF - PF;
    constexpr int EXT_G = 10, SCL_G = 14;
    constexpr int N_WAIT = [] {                                  // vector-memory ops a K-step has issued in front of its barrier
        int n = 0;
        for (int hs = 0; hs < 2; ++hs)
            for (int mf = 0; mf < MF; ++mf) {
                if (hs == 1 && mf >= BAR_G) break;
                if ((mf & 3) == 1) n += 1;
                if (mf == 2 || mf == 6) n += 1;
                if (mf == EXT_G && HL == 3) n += 1;
                if (hs == 0 && mf == SCL_G) n += 1;
            }
        return n;
    }();
    // global stores of one tile's epilogue per wave (always issued: buffer stores, range-checked) -- the first barrier of the next segment
    // lets them stay in flight.  vmcnt is a 6-bit count: the fp32 form waits for its oldest stores.
    constexpr int N_STORE = (int)sizeof(YT) * 16;
    constexpr int N_WAIT_FIRST = (N_WAIT + N_STORE > 63) ? 63 : N_WAIT + N_STORE;

#define QP_SB() __builtin_amdgcn_sched_barrier(0)
#define QP_HALF(WF_USE, WF_MAKE, PK_SRC, SC_SRC, KF_MAKE, SC_NEXT, KF_NEXT, ACUR, RDC, ANXT, RDN, HS1, FIRST, NWAIT_, XR_ST, XS_ST, BUF_ST, LOADSET, LOADTILE, SCLTILE) \
    {                                                                                                              \
        _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) {                                                        \
            if ((HS1) && mf == BAR_G) {                                                                            \
                __builtin_amdgcn_s_waitcnt(0x0F70 | ((NWAIT_) & 15) | (((NWAIT_) >> 4) << 14));   /* vmcnt(NWAIT_) only */ \
                __builtin_amdgcn_s_barrier();                                                                      \
            }                                                                                                      \
            uint32_t cv_ = 0;                                                                                      \
            QP_SB();                                                                                               \
            if (FIRST) mfma_new(acc[mf][0], WF_USE[0], xf[mf & 3]); else mfma_acc(acc[mf][0], WF_USE[0], xf[mf & 3]); \
            QP_SB();                                                                                               \
            if (mf + PF < MF) xf[(mf + PF) & 3] = *reinterpret_cast<const bf16x8_t*>((ACUR) + (RDC) + (mf + PF) * 2048);   \
            else xf[(mf + PF) & 3] = *reinterpret_cast<const bf16x8_t*>((ANXT) + (RDN) + (mf + PF - MF) * 2048);   \
            QP_SB();                                                                                               \
            if (FIRST) mfma_new(acc[mf][1], WF_USE[1], xf[mf & 3]); else mfma_acc(acc[mf][1], WF_USE[1], xf[mf & 3]); \
            QP_SB();                                                                                               \
            cv_ = cvt_dword<OUT_KIND>(PK_SRC, sop, mf);                                                            \
            QP_SB();                                                                                               \
            if (FIRST) mfma_new(acc[mf][2], WF_USE[2], xf[mf & 3]); else mfma_acc(acc[mf][2], WF_USE[2], xf[mf & 3]); \
            QP_SB();                                                                                               \
            WF_MAKE[mf >> 2][mf & 3] = ext_dword<OUT_KIND>(cv_, PK_SRC, mf);                                       \
            QP_SB();                                                                                               \
            if (FIRST) mfma_new(acc[mf][3], WF_USE[3], xf[mf & 3]); else mfma_acc(acc[mf][3], WF_USE[3], xf[mf & 3]); \
            QP_SB();                                                                                               \
            if ((mf & 3) == 1) stage_piece(XR_ST, XS_ST, BUF_ST, ((HS1) ? PPW / 2 : 0) + (mf >> 2));               \
            if (mf == 2) load_part(LOADSET, LOADTILE, 0);                                                          \
            if (mf == 6) load_part(LOADSET, LOADTILE, 1);                                                          \
            if (mf == EXT_G) load_part(LOADSET, LOADTILE, 2);                                                      \
            if (!(HS1) && mf == SCL_G) sc_nn = load_scales(SCLTILE);                                               \
            if (((mf + 1) & 3) == 0)                                                                               \
                sop = (mf + 1 < MF) ? scale_operand(SC_SRC[KF_MAKE], (mf + 1) >> 2) : scale_operand(SC_NEXT[KF_NEXT], 0); \
        }                                                                                                          \
    }
    // One K-step (flat position: K-step KT_CUR of the current segment).  Its loads fetch the K-step two positions on: K-step KT_CUR + 2 of
    // this segment, or -- in the last two K-steps -- the first two of the next segment (after the last segment: its own tiles again).
#define QP_KSTEP(KT_CUR, FIRST, NWAIT_, CONV1, LOAD1, CONV2, LOAD2)                                                \
    {                                                                                                              \
        const int kt_ = (KT_CUR);                                                                                  \
        const int buf = abuf, bufn = (abuf + 1) & 3, buf2 = (abuf + 2) & 3;                                        \
        abuf = bufn;                                                                                               \
        const char* acur = smem + buf * A_TILE;                                                                    \
        const char* anxt = smem + bufn * A_TILE;                                                                   \
        const int k2_ = kt_ + 2;                                                                                   \
        const bool inc_ = k2_ < kt1;                                                                               \
        const int kst_ = inc_ ? k2_ : ktn0 + (k2_ - kt1);                                                          \
        const uint32_t wt_ = (uint32_t)((inc_ ? trow : trow_n) + kst_);                                            \
        const uint32_t xs_ = (uint32_t)kst_ * 128u;                                                                \
        const __amdgpu_buffer_rsrc_t xr_ = rsrc_of(inc_ ? xb : xb_n, inc_ ? xn : xn_n);                            \
        QP_HALF(wfA, wfB, CONV1, sc_cur, 1, sc_nxt, 0, acur, rd0, acur, rd1, false, FIRST, NWAIT_, xr_, xs_, buf2, LOAD1, wt_ * 2u + 0u, wt_) \
        QP_HALF(wfB, wfA, CONV2, sc_nxt, 0, sc_nxt, 1, acur, rd1, anxt, rd0, true, false, NWAIT_, xr_, xs_, buf2, LOAD2, wt_ * 2u + 1u, wt_) \
        sc_cur = sc_nxt; sc_nxt = sc_nn;                                                                           \
    }

    gu32* flags = (gu32*)ws;                                     // P flag words (zeroed by the launcher), then the partial-tile slots
    char* slots = ws + 4096;
    int abuf = 0, pend = 0;
    for (int si = 0; si < nseg; ++si) {
        // the segment after this one (its first two K-steps are fetched by this segment's last two)
        uint64_t xb_n = xb; uint32_t xn_n = xn; int trow_n = trow, ktn0 = kt1 - 2;
        int kt1n = kt1;
        if (si + 1 < nseg) {
            const int4 t = lds_read16(tab_addr + (si + 1) * 32);
            const int m0n = sgpr(t.x), n0n = sgpr(t.y);
            ktn0 = sgpr(t.z); kt1n = sgpr(t.w);
            xb_n = x_base_of(m0n); xn_n = x_rec_of(m0n);
            trow_n = sgpr((n0n / TILE_N + wid) * KT);
        }
        {
            int kt = kt0;
            QP_KSTEP(kt, true, N_WAIT_FIRST, pk1, pk0, pk2, pk1)
            QP_KSTEP(kt + 1, false, N_WAIT, pk3, pk2, pk0, pk3)
            if (pend) {                                          // the previous segment's partial tile: every wave's stores are behind a vmcnt(N_WAIT) + barrier now
                if (tid == 0) __hip_atomic_store(flags + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pend = 0;
            }
            for (kt += 2; kt < kt1; kt += 2) {
                QP_KSTEP(kt, false, N_WAIT, pk1, pk0, pk2, pk1)
                QP_KSTEP(kt + 1, false, N_WAIT, pk3, pk2, pk0, pk3)
            }
        }
        // the MFMAs are opaque to hipcc's hazard recogniser: give the last of them their passes before the accumulators are read
        acc_fence<MF>(acc);
        // Epilogue LDS: the activation buffer of the K-step just finished (the next K-step's tile sits in buffer `abuf`, the one after is
        // landing in abuf + 1, abuf + 2 is the target of the next K-step's DMA; abuf + 3 is rewritten only after the next K-step's barrier).
        // Every wave's last fragment reads of it must be back first: one more wait + barrier per tile.
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        char* epi = smem + ((abuf + 3) & 3) * A_TILE + wid * 8192;
        const int4 tl = lds_read16(tab_addr + si * 32), th = lds_read16(tab_addr + si * 32 + 16);
        const int m0 = sgpr(tl.x), n0 = sgpr(tl.y), role = sgpr(th.x), peer0 = sgpr(th.y), peer1 = sgpr(th.z);
        if (role == 1) {
            // tail piece: the accumulators go to this block's slot, lane-linear (1 KiB per wave instruction), write-through (sc1).  The
            // stores are NOT waited for here: the flag that publishes them is stored behind the second barrier of the next segment,
            // whose vmcnt(N_WAIT) retires every older operation of every wave -- unless this is the block's last segment.
            const __amdgpu_buffer_rsrc_t sr = rsrc_of((uint64_t)slots + (uint64_t)b * 262144ull, 262144u);
#pragma unroll
            for (int i = 0; i < MF; i += 2) {
                __builtin_amdgcn_sched_barrier(0);               // eight quads through VGPRs at a time (the K-loop's registers stay live)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i + (k >> 2)][k & 3]), sr, lane16, (wid * 64 + i * 4 + k) * 1024, 16);
            }
            pend = 1;
            if (si + 1 == nseg) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains its stores ...
                __syncthreads();                                 // ... before ONE lane signals for the block
                if (tid == 0) __hip_atomic_store(flags + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pend = 0;
            }
        } else {
            if (role == 2) {
                if (tid == 0) {                                  // one lane polls one word per peer (bounded: a missing producer shows as a wrong tile, not a hang)
                    for (int pb = peer0; pb <= peer1; ++pb) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(flags + pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u && spins < (1u << 22)) { __builtin_amdgcn_s_sleep(8); ++spins; }
                        if (spins >= (1u << 22)) __hip_atomic_store(flags + P, 1u + (unsigned)pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                __syncthreads();
            }
            // Y: base = (m0, n0 + 64 wid), records = the tile's valid rows
            const int rows = (M - m0 < BM) ? M - m0 : BM;
            const uint64_t yb = (uint64_t)Y + ((uint64_t)m0 * (uint64_t)N + (uint64_t)(n0 + wid * 64)) * sizeof(YT);
            const __amdgpu_buffer_rsrc_t yr = rsrc_of(yb, (uint32_t)rows * (uint32_t)N * (uint32_t)sizeof(YT));
            u32x4_t pf[8];                                       // head piece: the batch of partial quads fetched ahead (store_half_tile)
#pragma unroll
            for (int k = 0; k < 8; ++k) pf[k] = u32x4_t{0u, 0u, 0u, 0u};
            if (role == 2) {
                const __amdgpu_buffer_rsrc_t sr0 = rsrc_of((uint64_t)slots + (uint64_t)peer0 * 262144ull, 262144u);
#pragma unroll
                for (int k = 0; k < 8; ++k) pf[k] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(sr0, lane16, (wid * 64 + k) * 1024, 16));
            }
            float bv[4][4];
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[nf][j] = 0.f;
            if (bias) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    const float4 t = *reinterpret_cast<const float4*>(bias + n0 + wid * 64 + nf * 16 + g * 4);
                    bv[nf][0] = t.x; bv[nf][1] = t.y; bv[nf][2] = t.z; bv[nf][3] = t.w;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4_t (&acch)[8][4] = *reinterpret_cast<const f32x4_t (*)[8][4]>(&acc[h * 8]);
                store_half_tile<YT>(acch, epi, yr, h * 128, N, bv, lane, y16, (uint64_t)slots, peer0, peer1, wid, h * 32, pf);
            }
        }
        // next segment
        xb = xb_n; xn = xn_n; trow = trow_n; kt0 = ktn0; kt1 = kt1n;
    }
#undef QP_KSTEP
#undef QP_HALF
#undef QP_SB
    __builtin_amdgcn_s_waitcnt(0x0070);                          // the re-staged last tiles may still be landing in LDS
}

struct DevOnceP { std::atomic<uint64_t> mask{0}; };
inline bool attr_neededP(const DevOnceP& o) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    return !(o.mask.load(std::memory_order_acquire) & (1ull << d));
}
inline void attr_doneP(DevOnceP& o) {
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) o.mask.fetch_or(1ull << d, std::memory_order_release);
}
int cu_count() {
    static std::atomic<int> cached[64];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 256;
    int v = cached[d].load(std::memory_order_relaxed);
    if (v > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
    cached[d].store(v, std::memory_order_relaxed);
    return v;
}

}  // namespace

// The schedule of a launch (host logic, no device needed): P blocks, `full` whole rounds of tiles, R stream-K tiles cut into runs of q
// K-steps.  cus = CU count of the device (0: 256).  Returns 0 and fills the plan when the persistent kernel applies, else -1:
// it needs an even number of K-steps per tile (runs and pieces are whole pairs of K-steps) and at least two K-steps.
// ws_bytes = 4096 (flag words) + one 256 KiB slot per block that can hold a tail piece; 0 when no tile is cut.
extern "C" int msq_qgemm256p_plan(int64_t M, int64_t N, int64_t K, int cus, int* P_, int* full_, int* R_, int* q_, int64_t* ws_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || (N % 256) || (K % 128)) return -1;
    const int64_t MT = (M + 255) / 256, NTB = N / 256, KT = K / 64, T = MT * NTB;
    if (cus <= 0) cus = 256;
    if (T * KT >= (1ll << 30) || KT < 2) return -1;
    int P = cus;
    int64_t full = T / P, R = T - full * P, q = 0;
    if (R > 0) {
        q = (R * KT + P - 1) / P;
        q += (q & 1);
        if (q < 8) q = (KT < 8) ? KT : 8;                        // short runs: fewer blocks take part in the stream-K round
        if (q > KT) q = KT;
    }
    if (full == 0 && R > 0) { const int64_t nb = (R * KT + q - 1) / q; if (nb < P) P = (int)nb; }
    if (full + 2 > 256) return -1;                               // the block's segment table in LDS holds 256 entries
    if (P_) *P_ = P;
    if (full_) *full_ = (int)full;
    if (R_) *R_ = (int)R;
    if (q_) *q_ = (int)q;
    if (ws_bytes) *ws_bytes = (R > 0 && q < KT) ? 4096 + (int64_t)((R * KT + q - 1) / q) * 262144 : 0;
    return 0;
}

// The work list of block b under a plan: up to `cap` segments as 6 ints each {tile id, first K-step, end K-step, role, first peer, last peer}
// (role 0 writes Y, 1 leaves a partial tile in slot b, 2 adds the slots of its peers and writes Y).  Returns the segment count.  Host only.
extern "C" int msq_qgemm256p_segments(int b, int P, int full, int R, int KT, int q, int* out, int cap) {
    const int n = seg_count(b, P, full, R, KT, q);
    for (int i = 0; i < n && i < cap; ++i) {
        const Seg s = seg_get(i, b, P, full, R, KT, q);
        out[i * 6 + 0] = s.tile; out[i * 6 + 1] = s.kt0; out[i * 6 + 2] = s.kt1; out[i * 6 + 3] = s.role; out[i * 6 + 4] = s.peer0; out[i * 6 + 5] = s.peer1;
    }
    return n;
}

// Launcher (called by qlinear_bf16_impl, msq_gemm.hip).  Preconditions checked by the caller: unified layout, N % 256 == 0, every
// buffer offset below 4 GiB, msq_qgemm256p_plan(...) == 0 and workspace_bytes >= its ws_bytes.  Returns hipGetLastError().
extern "C" int msq_launch_qgemm256p(const void* X, const void* ext_plane, const void* code_plane, const void* scale_plane, const float* bias, void* Y,
                                    int y_dtype, int64_t M, int64_t N, int64_t K, int out_kind, int scl_groups, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
    int P = 0, full = 0, R = 0, q = 0;
    int64_t wsb = 0;
    // The dispatcher's rule sized the workspace for 256 CUs; this device (a CPX / DPX partition, a CU mask, another part) may have another count
    // and another plan: when that plan does not apply or needs more slots than the caller gave, return MSQ_QP_FALLBACK -- the dispatcher then
    // takes k_qgemm256 instead of failing the call or writing past the workspace (advisor, round 5).
    if (msq_qgemm256p_plan(M, N, K, cu_count(), &P, &full, &R, &q, &wsb) || wsb > workspace_bytes || (wsb > 0 && !workspace)) return -12345;
    const dim3 grid((unsigned)P), blk(256);
    const size_t lds = (size_t)4 * 256 * 128 + 256 * 32;          // four activation buffers + the segment table
    const int y16 = (y_dtype == 1) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (wsb > 0) {
        if (!workspace) return (int)hipErrorInvalidValue;
        const hipError_t e = hipMemsetAsync(workspace, 0, 4096, st);   // flags + the time-out word (a memset node under graph capture)
        if (e != hipSuccess) return (int)e;
    }
#define QP_LAUNCH(OK, YT)                                                                                              \
    do { static DevOnceP once_;                                                                                        \
         if (attr_neededP(once_)) { (void)hipFuncSetAttribute((const void*)k_qgemm256p<OK, YT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_doneP(once_); } \
         hipLaunchKernelGGL((k_qgemm256p<OK, YT>), grid, blk, lds, st, (const uint16_t*)X, (const uint8_t*)ext_plane, (const uint8_t*)code_plane, \
                            (const uint8_t*)scale_plane, bias, (YT*)Y, (int)M, (int)N, (int)K, scl_groups, y16, P, full, R, q, (char*)workspace); } while (0)
    if (out_kind == MSQ_PLANE_U8) { if (y_dtype == 0) QP_LAUNCH(MSQ_PLANE_U8, float); else QP_LAUNCH(MSQ_PLANE_U8, uint16_t); }
    else { if (y_dtype == 0) QP_LAUNCH(MSQ_PLANE_U8X, float); else QP_LAUNCH(MSQ_PLANE_U8X, uint16_t); }
#undef QP_LAUNCH
    return (int)hipGetLastError();
}
