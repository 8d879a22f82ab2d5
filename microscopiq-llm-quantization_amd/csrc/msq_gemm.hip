// msq_gemm.hip -- packed ("GEMM-ready", tile-major) weight formats MSQ-T1 / MSQ-U1, their repack / unpack kernels and
// the GEMMs of the hot path for gfx950:
//   k_qgemm3      fused unpack-dequant-GEMM, bf16 activations (msq_qlinear_bf16)
//   k_mxgemm      MX-native GEMM on v_mfma_scale_f32_16x16x128_f8f6f4: MX-FP8 activations x MX-FP4 codes or an exact
//                 e4m3 weight operand (msq_qlinear_mx_w4a8 / _w8a8)
//   k_qgemv, k_mxgemv   weight-streaming decode kernels of both paths (M <= 32 / 64), k_splitk_reduce
//
// Design: the weight operand never touches LDS.  Every wavefront streams
// its own 64(n) x 64(k) packed tiles straight from global memory (L2-resident across
// the row-tiles of an XCD) in the exact fragment order of v_mfma_f32_16x16x32_bf16,
// converts them in-register with the CDNA4 scaled converts
// (v_cvt_scalef32_pk_bf16_fp4 / _fp8 / _bf8: two elements + E8M0 block scale per
// instruction) and ORs inlier and outlier parts (one of them is always +0).  Only the
// activation tile is staged through LDS (global_load_lds, XOR-swizzled on the source
// side so the LDS image stays lane-linear and ds_read_b128 is conflict-free).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"

using namespace msq;

#include "msq_gemm_common.h"


// ---------------------------------------------------------------------------
// repack: codes[N][K] (u32: bits 0-7 inlier code, 8-23 outlier code) + per-block
// exponents -> tile-major planes.  One wave per tile.
// ---------------------------------------------------------------------------
template <int IN_KIND, int OUT_KIND>
__global__ void __launch_bounds__(256)
k_repack(const uint32_t* __restrict__ codes, const float* __restrict__ e_in, const float* __restrict__ e_out,
         uint8_t* __restrict__ inl_plane, uint8_t* __restrict__ out_plane, uint8_t* __restrict__ scl_plane,
         int64_t N, int64_t K, int block, int* status) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t KT = K / TILE_K, NT = N / TILE_N;
    if (tile >= KT * NT) return;
    const int64_t nt = tile / KT, kt = tile % KT;
    const int64_t nblk = K / block;
    constexpr int OS = OutSlots<OUT_KIND>::n;
    uint32_t sc[4] = {0, 0, 0, 0};
    int st = 0;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
        u32x4_t inl4;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int64_t n = nt * TILE_N + frag_n(lane, nf);
            const int64_t k = kt * TILE_K + frag_k(lane, kf);
            const u32x4_t c0 = *reinterpret_cast<const u32x4_t*>(codes + n * K + k);
            const u32x4_t c1 = *reinterpret_cast<const u32x4_t*>(codes + n * K + k + 4);
            const uint32_t cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            uint32_t iw = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) iw |= (cc[j] & 0xFu) << (4 * j);
            inl4[nf] = iw;
            if (OUT_KIND == MSQ_PLANE_BF16) {
                u32x4_t o;
#pragma unroll
                for (int w = 0; w < 4; ++w) o[w] = ((cc[2 * w] >> 8) & 0xFFFFu) | (((cc[2 * w + 1] >> 8) & 0xFFFFu) << 16);
                *reinterpret_cast<u32x4_t*>(out_plane + ((tile * OS + kf * 4 + nf) * 64 + lane) * 16) = o;
            } else {
                uint32_t o0 = 0, o1 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o0 |= ((cc[j] >> 8) & 0xFFu) << (8 * j); o1 |= ((cc[4 + j] >> 8) & 0xFFu) << (8 * j); }
                // slot kf*2 + nf/2, dwords (nf&1)*2 .. +1 of the lane's 16 bytes
                uint8_t* dst = out_plane + ((tile * OS + kf * 2 + (nf >> 1)) * 64 + lane) * 16 + (nf & 1) * 8;
                *reinterpret_cast<uint2*>(dst) = make_uint2(o0, o1);
            }
            // scales of the block holding this fragment
            const int64_t kb = k / block;
            const float ei = e_in[n * nblk + kb], eo = e_out[n * nblk + kb];
            uint32_t bi, bo;
            if (ei != ei) { bi = 255; st |= MSQ_STATUS_NAN; } else { const float b = ei + 127.f; bi = (b < 0.f || b > 254.f) ? 255u : (uint32_t)b; if (b < 0.f || b > 254.f) st |= MSQ_STATUS_INEXACT; }
            const float ef = eo - ei;
            if (ef != ef) { bo = 255; st |= MSQ_STATUS_NAN; } else { const float b = ef + 127.f; bo = (b < 0.f) ? 0u : ((b > 254.f) ? 254u : (uint32_t)b); }
            sc[nf] |= (bi << (16 * kf)) | (bo << (16 * kf + 8));
        }
        if (IN_KIND != MSQ_PLANE_NONE)
            *reinterpret_cast<u32x4_t*>(inl_plane + ((tile * 2 + kf) * 64 + lane) * 16) = inl4;
    }
    if (IN_KIND != MSQ_PLANE_NONE) {
        const bool per_lane = block < 32;
        const int groups = per_lane ? 64 : 16;
        if (per_lane || (lane >> 4) == 0) {
            u32x4_t s4; s4[0] = sc[0]; s4[1] = sc[1]; s4[2] = sc[2]; s4[3] = sc[3];
            *reinterpret_cast<u32x4_t*>(scl_plane + (tile * groups + (per_lane ? lane : (lane & 15))) * 16) = s4;
        }
    }
    if (st && status) atomicOr(status, st);
}

// load all packed data of a tile for this lane (coalesced: every slot is 64 lanes x 16 B)
template <int IN_KIND, int OUT_KIND, bool NT = false>
MSQ_D void load_tile(TileRegs& t, const uint8_t* inl_plane, const uint8_t* out_plane, const uint8_t* scl_plane,
                     int64_t tile, int lane, int scl_groups) {
    constexpr int OS = OutSlots<OUT_KIND>::n;
    if (IsUnified<OUT_KIND>::v) {
        const uint2 sc = *reinterpret_cast<const uint2*>(scl_plane + (tile * 16 + (lane & 15)) * 8);
        t.scl[0] = sc.x; t.scl[1] = sc.y;
        if (OUT_KIND == MSQ_PLANE_U8X) {
            t.ext[0] = *reinterpret_cast<const uint32_t*>(inl_plane + ((tile * 2 + 0) * 64 + lane) * 4);
            t.ext[1] = *reinterpret_cast<const uint32_t*>(inl_plane + ((tile * 2 + 1) * 64 + lane) * 4);
        } else { t.ext[0] = 0; t.ext[1] = 0; }
    } else if (IN_KIND != MSQ_PLANE_NONE) {
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
            t.inl[kf] = *reinterpret_cast<const u32x4_t*>(inl_plane + ((tile * 2 + kf) * 64 + lane) * 16);
        t.scl = *reinterpret_cast<const u32x4_t*>(scl_plane + (tile * scl_groups + (lane & (scl_groups - 1))) * 16);
    }
#pragma unroll
    for (int s = 0; s < OS; ++s)
        t.out[s] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(out_plane + ((tile * OS + s) * 64 + lane) * 16))
                      : *reinterpret_cast<const u32x4_t*>(out_plane + ((tile * OS + s) * 64 + lane) * 16);
}

// ---------------------------------------------------------------------------
// unpack: planes -> dense W[N][K] (f32 or bf16), same converts as the GEMM.
// ---------------------------------------------------------------------------
template <int IN_KIND, int OUT_KIND, typename OT>
__global__ void __launch_bounds__(256)
k_unpack(const uint8_t* __restrict__ inl_plane, const uint8_t* __restrict__ out_plane,
         const uint8_t* __restrict__ scl_plane, OT* __restrict__ W, int64_t N, int64_t K, int scl_groups) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t KT = K / TILE_K, NT = N / TILE_N;
    if (tile >= KT * NT) return;
    const int64_t nt = tile / KT, kt = tile % KT;
    TileRegs t;
    load_tile<IN_KIND, OUT_KIND>(t, inl_plane, out_plane, scl_plane, tile, lane, scl_groups);
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const u32x4_t f = tile_frag<IN_KIND, OUT_KIND>(t, nf, kf);
            const int64_t n = nt * TILE_N + frag_n(lane, nf);
            const int64_t k = kt * TILE_K + frag_k(lane, kf);
            if (sizeof(OT) == 2) {
                *reinterpret_cast<u32x4_t*>(reinterpret_cast<uint16_t*>(W) + n * K + k) = f;
            } else {
                float* d = reinterpret_cast<float*>(W) + n * K + k;
                *reinterpret_cast<float4*>(d) = make_float4(u2f(f[0] << 16), u2f(f[0] & 0xFFFF0000u), u2f(f[1] << 16), u2f(f[1] & 0xFFFF0000u));
                *reinterpret_cast<float4*>(d + 4) = make_float4(u2f(f[2] << 16), u2f(f[2] & 0xFFFF0000u), u2f(f[3] << 16), u2f(f[3] & 0xFFFF0000u));
            }
        }
}


// MF = 16-row MFMA fragments per wave along m: 8 (wave tile 128 x 64, two blocks per CU, accumulators in VGPRs) or 16 (wave
// tile 256 x 64, block 256 x 256, ONE block of four waves per CU: the 256 accumulator registers live in AGPRs and every
// converted weight fragment feeds 16 MFMAs instead of 8: half the converts, rotates and and-ors per MFMA).
// WN = waves along n (4: block 256 columns; 8: one block of eight waves per CU covering 512 columns of ONE activation tile --
// half the LDS-DMA issue per wave and half the activation bytes pulled from L2 per CU and K-step).
// KG = k-groups per block (1, or 2 with WN = 4, WM = 1, MF = 8): two groups of four waves share the block's output tile, each
// runs the unchanged K-loop over one half of the block's K range out of its own three activation buffers, and group 1 hands
// its accumulators to group 0 through LDS at the end (fixed order: first half + second half).  For grids of at most one block
// per CU this puts two waves on every SIMD -- the same occupancy two resident blocks give the large grids -- without any
// partial plane in memory; under split-K it halves the K-steps per block at the same number of partial planes.
template <int IN_KIND, int OUT_KIND, typename YT, int WM, int MF = 8, int WN = 4, int KG = 1>
__global__ void __launch_bounds__(64 * WN * WM * KG, (MF == 16) ? 1 : ((WN == 8 || KG == 2) ? 1 : 2))
k_qgemm3(const uint16_t* __restrict__ X, const uint8_t* __restrict__ inl_plane, const uint8_t* __restrict__ out_plane,
         const uint8_t* __restrict__ scl_plane, const float* __restrict__ bias, YT* __restrict__ Y, int M, int N, int K,
         int scl_groups, int ksplit, float* __restrict__ partial, int y16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    static_assert(KG == 1 || (KG == 2 && WN == 4 && WM == 1 && MF == 8), "k-groups: four-wave 128 x 256 blocks only");
    const int wid_blk = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (SGPR offsets below)
    const int kgid = (KG == 2) ? (wid_blk >> 2) : 0;            // k-group of this wave
    const int wid = (KG == 2) ? (wid_blk & 3) : wid_blk;        // wave inside its group
    constexpr int WROWS = 16 * MF;                              // rows of a wave tile
    constexpr int BMT = WROWS * WM;                             // block rows: WM wave rows
    constexpr int PPW = BMT / 8 / (WN * WM);                     // 1 KiB staging pieces (8 rows) per wave and K-step
    constexpr int A_TILE = BMT * BK * 2;
    const int wm = (WM == 2) ? wid / WN : 0, wn = wid % WN;
    constexpr int BNW = 64 * WN;                                // block columns
    const int c = lane & 15, g = lane >> 4;
    const int MT = (M + BMT - 1) / BMT, NTB = N / BNW;
    const int ks = (int)(blockIdx.x % (unsigned)ksplit);
    const int bid = (int)(blockIdx.x / (unsigned)ksplit);
    int bm, bn;
    if ((NTB & 7) == 0 && ksplit == 1) {
        // XCD-aware order (blocks are dealt round-robin over the 8 XCDs): XCD x owns the column panels
        // bn = 8 cp + x and walks them in super-tiles of (up to) 8 row tiles x 4 panels, so the 32 blocks
        // resident on an XCD share 4 packed W panels and 8 activation tiles out of its own L2.
        const int xcd = bid & 7, i = bid >> 3;
        const int npx = NTB >> 3, per_group = 8 * npx, full = MT >> 3;
        int rg, j, R;
        if (i < full * per_group) { rg = i / per_group; j = i % per_group; R = 8; }
        else { rg = full; j = i - full * per_group; R = MT - full * 8; }
        bm = rg * 8 + j % R;
        bn = (j / R) * 8 + xcd;
    } else { bm = bid % MT; bn = bid / MT; }
    const int m0 = bm * BMT, n0 = bn * BNW;
    const int KT = K / BK;
    // split-K: this block covers K-steps [kt_lo, kt_hi) and writes an fp32 partial tile
    const int kchunk = (KT + ksplit - 1) / ksplit;
    int kt_lo = ks * kchunk;
    int kt_hi = (kt_lo + kchunk < KT) ? kt_lo + kchunk : KT;
    if (KG == 2) {                                              // the launcher guarantees an even number of K-steps per block
        const int half = (kt_hi - kt_lo) >> 1;
        kt_lo += kgid * half; kt_hi = kt_lo + half;
    }
    char* const smem_g = smem + kgid * (3 * A_TILE);            // this group's activation buffers
    // Stagger (eight-wave blocks): waves 4-7 ("followers", the SIMD partners of waves 0-3) run half a K-step behind.  Between two
    // block barriers a leader does {kf 0, kf 1} of K-step j, a follower {kf 1 of j - 1, kf 0 of j}: the partner of a wave that sits
    // in its step head (packed-load waits, LDS-DMA issue, first LDS reads) is in the middle of its MFMA stream, not in the same
    // head (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  Nothing the compiler sees depends on the role: every wave stages
    // the first half of its pieces of tile kt + 2 at the top of K-step kt and the second half between the two half-steps (a
    // follower's second half belongs to tile kt + 3: only the K offset and the buffer index differ), and the block barrier with its
    // vmcnt wait is an inline-asm block that skips itself by a scalar branch -- after kf 0 for followers, after kf 1 for leaders.
    // (With compiler-visible branches on the role hipcc spills into the loop; a second barrier per K-step instead of the
    // branch measured no gain.)  A tile is read during two barrier intervals, so the ring has four activation buffers.
    // Measured, M2048 N16384 K4096, three boxes: posit layout 202.9 -> 199.0, 199.2 -> 196.5, 200.9 -> 198.2 us; the layout without
    // extension bits loses 0.5-1 % (182.1 -> 183.9), so it keeps the plain schedule.
    constexpr bool STG = WN == 8 && KG == 1 && WM == 1 && MF == 8 &&
                         ((MSQ_STAGGER >= 1 && OUT_KIND == MSQ_PLANE_U8X) || (MSQ_STAGGER >= 2 && OUT_KIND == MSQ_PLANE_U8));
    constexpr int PA = STG ? (PPW + 1) / 2 : PPW;               // pieces staged at the top of a K-step; the rest between the half-steps
    const int fol = STG ? sgpr(wid_blk >> 2) : 0;               // 1 = follower
    const int64_t tile_row = (int64_t)(n0 / TILE_N + wn) * KT;

    // packed planes and activations through buffer descriptors (SGPR slot / K-step offsets)
    const int64_t ntiles = (int64_t)(N / TILE_N) * KT;
    PlaneRsrc pr;
    pr.inl = make_rsrc(inl_plane, ntiles * 2 * (OUT_KIND == MSQ_PLANE_U8X ? 256 : 1024));
    pr.out = make_rsrc(out_plane, ntiles * 2 * HalfSlots<OUT_KIND>::n * 1024);
    constexpr bool HAS_SCALE = HasScale<IN_KIND, OUT_KIND>::v;
    constexpr int SCLB = SclBytes<OUT_KIND>::n;               // scale bytes per lane group and tile
    pr.scl = make_rsrc(scl_plane, ntiles * scl_groups * SCLB);
    const int lane16 = lane * 16;
    const int scl_lane_off = (lane & (scl_groups - 1)) * SCLB;
    const uint32_t scl_tile_bytes = (uint32_t)scl_groups * (uint32_t)SCLB;
    auto load_scales = [&](uint32_t tile) -> u32x4_t {
        if (SCLB == 16) return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(pr.scl, scl_lane_off, uni(tile * scl_tile_bytes), 0));
        const u32x2_t v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(pr.scl, scl_lane_off, uni(tile * scl_tile_bytes), 0));
        return u32x4_t{v[0], v[1], 0u, 0u};
    };
    const uint32_t tile_row32 = (uint32_t)sgpr((int)tile_row);

    // A staging sources: piece = 4*wid + p, row = 8*piece + lane/8, swizzled source chunk
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(X, (int64_t)M * K * 2);
    int aoff[8];                                                // PPW <= 8 (a dependent array bound captured by the lambda below makes hipcc drop the host stub)
    static_assert(PPW <= 8, "staging pieces per wave");
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int piece = wid * PPW + p;
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < M ? gr : M - 1;
        aoff[p] = (int)(((int64_t)gr * K + chunk * 8) * 2);
    }
    auto stage_A_pieces = [&](int kt, int buf, int p_lo, int p_hi) {
#pragma unroll
        for (int p = 0; p < PPW; ++p)
            if (p >= p_lo && p < p_hi) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (void __attribute__((address_space(3)))*)(smem_g + buf * A_TILE + (wid * PPW + p) * 1024),
                                                         16, aoff[p], uni((uint32_t)kt * (BK * 2)), 0, 0);
                if (MSQ_ABL & 128)      // a second copy of the piece into the spare area behind the three buffers (WN = 8: 64 KiB of LDS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (void __attribute__((address_space(3)))*)(smem_g + 3 * A_TILE + (wid * PPW + p) * 1024),
                                                             16, aoff[p], uni((uint32_t)kt * (BK * 2) + ((MSQ_ABL & 1024) ? 0u : 64u)), 0, 0);
            }
    };
    auto stage_A = [&](int kt, int buf) { stage_A_pieces(kt, buf, 0, PPW); };
    // LDS read base of this lane for kf = 0 / 1 (row term (row>>1)&7 == (c>>1)&7 for every mf)
    const int sw = (c >> 1) & 7;
    const int rd0 = (wm * WROWS + c) * 128 + (((0 + g) ^ sw) << 4);
    const int rd1 = (wm * WROWS + c) * 128 + (((4 + g) ^ sw) << 4);

    f32x4_t acc[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Packed-operand register ring: 4 half-step sets.  During half-step i the wave multiplies with the
    // fragments of i, converts set (i + 1) and issues the loads of half-step i + 4 into the set it converted
    // one half-step ago, so a packed load has three half-steps (~1 us) to return: long enough for an L2 miss
    // served by the Infinity Cache (vmcnt is in-order: every wait also waits for all older loads, so the
    // distance has to cover the slowest of them).  The ring returns to its start every 2 K-steps.
    // (fp4 + bf16-plane operands are 20 VGPRs per set: they keep the two-set ring, distance one half-step.)
    constexpr bool DEEP = !(IN_KIND == MSQ_PLANE_FP4 && OUT_KIND == MSQ_PLANE_BF16);
    HalfRegs<IN_KIND, OUT_KIND> pk0, pk1, pk2, pk3;
    u32x4_t wfA[4], wfB[4];
    u32x4_t sc_cur = {0, 0, 0, 0}, sc_nxt = {0, 0, 0, 0}, sc_nn = {0, 0, 0, 0};

    const int kt_last = sgpr((kt_hi > kt_lo) ? kt_hi - 1 : ((kt_lo < KT) ? kt_lo : KT - 1));
    const int kt0 = sgpr((kt_lo < KT) ? kt_lo : KT - 1);       // an empty split still runs a harmless prologue
    const int kt1 = (kt0 + 1 <= kt_last) ? kt0 + 1 : kt_last;
    // activation tiles: three LDS buffers, staged TWO K-steps ahead (an L2 miss of the activation stream is as
    // long as a K-step; with one step of distance the wait before the barrier exposed it)
    stage_A(kt0, 0);
    if (kt0 + 1 <= kt_last) stage_A(kt0 + 1, 1);
    if (STG && fol) stage_A_pieces((kt0 + 2 <= kt_last) ? kt0 + 2 : kt_last, 2, PA, PPW);   // what a follower's K-step kt0 - 1 would have staged
    load_half_buf<IN_KIND, OUT_KIND>(pk0, pr, lane16, (tile_row32 + kt0) * 2u + 0u);
    load_half_buf<IN_KIND, OUT_KIND>(pk1, pr, lane16, (tile_row32 + kt0) * 2u + 1u);
    if (DEEP) {
        load_half_buf<IN_KIND, OUT_KIND>(pk2, pr, lane16, (tile_row32 + kt1) * 2u + 0u);
        load_half_buf<IN_KIND, OUT_KIND>(pk3, pr, lane16, (tile_row32 + kt1) * 2u + 1u);
    }
    if (HAS_SCALE) { sc_cur = load_scales(tile_row32 + kt0); if (DEEP) sc_nxt = load_scales(tile_row32 + kt1); }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) convert_quarter<IN_KIND, OUT_KIND>(wfA, pk0, sc_cur, 0, q);

    // One half-step, hand-interleaved: group mf = { LDS read of A fragment mf+2, 4 MFMAs on fragment mf,
    // one quarter (2 dwords) of the NEXT half-step's weight fragments converted }, groups fenced with
    // sched_barrier so the compiler keeps the interleave.
    constexpr int GPQ = (MF >= 8) ? MF / 8 : 1;                 // MFMA groups per converted quarter
    constexpr int QPG = (MF >= 8) ? 1 : 8 / MF;                 // quarters converted per group (MF = 4: two)
#define MSQ_HALF_STEP(WF_USE, WF_MAKE, PK_SRC, SC_SRC, KF_MAKE, RD)                                         \
    {                                                                                                        \
        bf16x8_t xf[3];                                                                                      \
        xf[0] = *reinterpret_cast<const bf16x8_t*>(abase + (RD));                                            \
        if (MF > 1) xf[1] = *reinterpret_cast<const bf16x8_t*>(abase + (RD) + 2048);                         \
        _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) {                                                  \
            if (mf + 2 < MF && !(MSQ_ABL & 1)) xf[(mf + 2) % 3] = *reinterpret_cast<const bf16x8_t*>(abase + (RD) + (mf + 2) * 2048); \
            if (MSQ_ABL & 64) { bf16x8_t xd_ = *reinterpret_cast<const bf16x8_t*>(abase + (RD) + ((mf + 2) & 7) * 2048 + 1024); asm volatile("" :: "v"(xd_)); } \
            if ((MSQ_ABL & 512) && (mf & 1) == 0) {                                                          \
                if ((RD) == rd0 || mf < 4) { u32x4_t wd_ = *reinterpret_cast<const u32x4_t*>(smem + 49152 + (mf >> 1) * 1024 + lane * 16); asm volatile("" :: "v"(wd_)); } \
                if (mf < 4) { u32x4_t wd2_ = *reinterpret_cast<const u32x4_t*>(smem + 49152 + 4096 + (mf >> 1) * 1024 + lane * 16); asm volatile("" :: "v"(wd2_)); } \
                if (mf >= 4) *reinterpret_cast<u32x4_t*>(smem + 49152 + 8192 + wid * 4096 + (mf >> 1) * 1024 + lane * 16) = WF_MAKE[(mf >> 1) & 1]; \
            }                                                                                                \
            _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                                 \
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, WF_USE[(MSQ_ABL & 32) ? (nf & 1) : nf]), xf[mf % 3], acc[mf][nf], 0, 0, 0); \
            if (!(MSQ_ABL & 2) && (mf % GPQ) == 0) {                                                         \
                _Pragma("unroll") for (int q_ = 0; q_ < QPG; ++q_) if (!(MSQ_ABL & 32) || (mf / GPQ) * QPG + q_ < 4) convert_quarter<IN_KIND, OUT_KIND>(WF_MAKE, PK_SRC, SC_SRC, KF_MAKE, (mf / GPQ) * QPG + q_); } \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
    }
    // One K-step at ring position (CONV1 = set of (kt, kf 1), CONV2 = set of (kt + 1, kf 0)); the sets converted
    // one half-step earlier (LOAD1, LOAD2) receive (kt + 2, kf 0) and (kt + 2, kf 1).
    constexpr int N_INFLIGHT = HalfLoads<IN_KIND, OUT_KIND>::n * (DEEP ? 2 : 1);
    constexpr int N_WAIT_ST = N_INFLIGHT + PPW * ((MSQ_ABL & 128) ? 2 : 1);   // + the LDS-DMA ops of the tile staged in this K-step
    // follower, at its barrier between the half-steps: the pieces it staged after the previous such barrier, the second packed
    // set of the previous K-step, this K-step's scale load, first pieces and first packed set stay in flight (issue order,
    // youngest last); everything older -- its pieces of tile kt + 1 -- has landed
    constexpr int N_WAIT_FOL = N_INFLIGHT + PPW + (HAS_SCALE ? 1 : 0);
    int abuf = 0;
    /* wait + block barrier executed only by the waves whose role COND_ (an SGPR) equals WANT_: a scalar branch inside the asm
       block, invisible to hipcc (its own vmcnt bookkeeping stays conservative: it never assumes this wait happened) */
#define MSQ_ROLE_BARRIER(COND_, WANT_, N_)                                                                        \
    asm volatile("s_cmp_lg_u32 %0, " #WANT_ "\n\ts_cbranch_scc1 .Lmsq_nb_%=\n\ts_waitcnt vmcnt(%1) lgkmcnt(0)\n\ts_barrier\n.Lmsq_nb_%=:" \
                 :: "s"(COND_), "n"(N_) : "memory", "scc")
#define MSQ_K_STEP(KT_CUR, CONV1, LOAD1, CONV2, LOAD2, BAR)                                                       \
    {                                                                                                        \
        const int kt_ = sgpr(KT_CUR);                                                                        \
        const int buf = abuf;                                                                                \
        const int buf2 = STG ? ((abuf + 2) & 3) : ((abuf == 0) ? 2 : abuf - 1);   /* (abuf + 2) % (4 or 3) */   \
        abuf = STG ? ((abuf + 1) & 3) : ((abuf == 2) ? 0 : abuf + 1);                                        \
        const char* abase = smem_g + buf * A_TILE;                                                           \
        const int ktn = (kt_ + 1 <= kt_last) ? kt_ + 1 : kt_last;   /* branch-free tail: re-load the last tile */ \
        const int ktnn2 = (kt_ + 2 <= kt_last) ? kt_ + 2 : kt_last;                                          \
        const int ktnn = DEEP ? ktnn2 : ktn;                                                                 \
        /* ---- half-step kf = 0: MFMAs on wfA, make wfB from CONV1 (loaded three half-steps ago) */        \
        keep_live(CONV1);                    /* take the vmcnt wait BEFORE new loads are issued */            \
        /* branch-free tail: the last two K-steps re-stage the last tile into a buffer nobody reads; every   \
           LDS-DMA is drained (vmcnt(0) + barrier) before the epilogue reuses the LDS */                     \
        if (HAS_SCALE) { if (DEEP) sc_nn = load_scales(tile_row32 + ktnn); else sc_nxt = load_scales(tile_row32 + ktn); } \
        __builtin_amdgcn_sched_barrier(0);   /* issue order (vmcnt is in-order): scales, LDS-DMA, packed loads */ \
        if (!(MSQ_ABL & 8)) stage_A_pieces(ktnn2, buf2, 0, PA);                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (!(MSQ_ABL & 4)) load_half_buf<IN_KIND, OUT_KIND>(LOAD1, pr, lane16, (tile_row32 + ktnn) * 2u + 0u); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        MSQ_HALF_STEP(wfA, wfB, CONV1, sc_cur, 1, rd0)                                                       \
        if (STG) {                                                                                           \
            MSQ_ROLE_BARRIER(fol, 1, N_WAIT_FOL);                                                            \
            const int ktb_ = (kt_ + 2 + fol <= kt_last) ? kt_ + 2 + fol : kt_last;                           \
            if (!(MSQ_ABL & 8)) stage_A_pieces(ktb_, (buf + 2 + fol) & 3, PA, PPW);                          \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
        /* ---- half-step kf = 1: MFMAs on wfB, make next wfA from CONV2 */                                  \
        keep_live(CONV2);                                                                                    \
        if (!(MSQ_ABL & 4)) load_half_buf<IN_KIND, OUT_KIND>(LOAD2, pr, lane16, (tile_row32 + ktnn) * 2u + 1u); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        MSQ_HALF_STEP(wfB, wfA, CONV2, sc_nxt, 0, rd1)                                                       \
        /* A(kt+1) and the scales must have landed before the barrier: wait explicitly for everything older \
           than the two packed load groups of this K-step (vmcnt is in-order), which stay in flight.  hipcc \
           derives its own waits from register uses only and does not cover the LDS-DMA (stale 8-row         \
           activation pieces were seen without this once two blocks shared a CU). */                         \
        /* the tile staged in THIS K-step (4 LDS-DMA ops, older than the packed loads and younger than the    \
           scale load thanks to the sched_barriers above) may stay in flight too: it is needed two barriers   \
           from now */                                                                                       \
        if (STG) MSQ_ROLE_BARRIER(fol, 0, N_WAIT_ST);                                                        \
        else if (!(MSQ_EXP_SKIPBAR && (BAR) == 0)) __builtin_amdgcn_s_waitcnt(0x0070 | (N_WAIT_ST & 15) | ((N_WAIT_ST >> 4) << 14)); \
        sc_cur = sc_nxt; if (DEEP) sc_nxt = sc_nn;                                                           \
        if (!STG && !(MSQ_EXP_SKIPBAR && (BAR) == 0)) __builtin_amdgcn_s_barrier();                          \
    }

    {
        int kt = kt_lo;
        if (DEEP) {
            for (; kt + 1 < kt_hi; kt += 2) {
                MSQ_K_STEP(kt, pk1, pk0, pk2, pk1, 0)
                MSQ_K_STEP(kt + 1, pk3, pk2, pk0, pk3, 1)
            }
            if (kt < kt_hi) MSQ_K_STEP(kt, pk1, pk0, pk2, pk1, 1)
        } else {
            for (; kt < kt_hi; ++kt) MSQ_K_STEP(kt, pk1, pk0, pk0, pk1, 1)
        }
    }
#undef MSQ_ROLE_BARRIER
#undef MSQ_K_STEP
#undef MSQ_HALF_STEP

    // the re-staged tail tiles (and nothing else) may still be landing in LDS: drain before the epilogue reuses it
    __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (MSQ_ABL & 16) { float t = 0.f; _Pragma("unroll") for (int i = 0; i < MF; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; if (t == 1.2345f) reinterpret_cast<float*>(Y)[0] = t; return; }
    // all waves are past the last K-step barrier: the A buffers are dead, every wave owns 8 KiB
    if constexpr (KG == 2) {
        // group 1 -> LDS (32 KiB per wave, [fragment][lane] float4: conflict-free), group 0 adds: first half + second half
        float4* red = reinterpret_cast<float4*>(smem) + wid * 2048;
        if (kgid == 1) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[(i * 4 + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (kgid == 0) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = red[(i * 4 + j) * 64 + lane];
                    acc[i][j][0] += t.x; acc[i][j][1] += t.y; acc[i][j][2] += t.z; acc[i][j][3] += t.w;
                }
        }
        // the epilogue slices (8 KiB per wave from the start of LDS) overlap the reduction area of wave 0: every wave of
        // group 0 must have read its partner's data before any of them writes
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (kgid == 1) return;
    }
    if constexpr (MF < 8) {                                    // 64-row wave tile: one pass (bf16) / two (f32)
        if (ksplit > 1)
            store_wave_tile_lds<float, MF>(acc, smem + wid * 8192, partial + (int64_t)ks * M * N, m0 + wm * WROWS, n0 + wn * 64, M, N, nullptr, lane);
        else
            store_wave_tile_lds<YT, MF>(acc, smem + wid * 8192, Y, m0 + wm * WROWS, n0 + wn * 64, M, N, bias, lane, y16);
    } else {
#pragma unroll
        for (int h = 0; h < MF / 8; ++h) {                      // 128 rows at a time through the wave's 8 KiB slice
            const f32x4_t (&acch)[8][4] = *reinterpret_cast<const f32x4_t (*)[8][4]>(&acc[h * 8]);
            if (ksplit > 1)
                store_wave_tile_lds<float>(acch, smem + wid * 8192, partial + (int64_t)ks * M * N, m0 + wm * WROWS + h * 128, n0 + wn * 64, M, N,
                                           nullptr, lane);
            else
                store_wave_tile_lds<YT>(acch, smem + wid * 8192, Y, m0 + wm * WROWS + h * 128, n0 + wn * 64, M, N, bias, lane, y16);
        }
    }
}

// ---------------------------------------------------------------------------
// MX-native W4A8 GEMM (BASELINE config 3, "CDNA4 fp8 MFMA path"): plain OCP-MX operands (mx_ops.py:332-457, block 32
// along K): e2m1 weight codes x e4m3 activation codes, both with E8M0 block scales, multiplied by
// v_mfma_scale_f32_16x16x128_f8f6f4 -- no dequantisation at all: the codes and the scale bytes ARE the operands.
// Same skeleton as k_qgemm3: block 128(m) x 256(n), 4 waves 1 x 4, wave tile 128 x 64, K-step 128 (one MFMA k),
// two blocks per CU; activation codes global -> LDS by LDS-DMA (128 rows x 128 B per K-step: the same geometry and
// swizzle as the bf16 tile), three buffers staged two K-steps ahead; weight codes (16 B per lane and 16 n) and scale
// bytes stream from global memory in operand order, three K-steps deep.
// Operand layout measured on MI355X (scripts/experiments/mx_mfma_probe.hip, mx_mfma_layout.hip): lane (r = l & 15,
// kg = l >> 4); fp4: 32 consecutive k = 32 kg .. of row r in VGPR 0-3; fp8: k = 16 kg .. +15 in VGPR 0-3 and
// k = 64 + 16 kg .. +15 in VGPR 4-7; the scale VGPR of lane (r, kg) carries the E8M0 byte of block kg of row r;
// result D[lane, e] = C[a-row 4 (l >> 4) + e][b-col l & 15].
// The MFMA accumulates the 128 products of one instruction with ~15 bits relative to the largest term (measured
// against exact arithmetic), looser than the fp32 accumulation of the bf16 path: tolerance 1e-4 * max|y|.
// ---------------------------------------------------------------------------
typedef int v8i_t __attribute__((ext_vector_type(8)));

// W8: the weight operand holds e4m3 codes (fake-quant values of any quantiser packed exactly by msq_mx_pack_w8,
// 8.25 bits/weight: MicroScopiQ inliers + outliers) instead of e2m1: 32 B per lane and 16 n in two half-slots, the
// same MFMA rate (an fp8 operand on either side sets it).  The three-deep weight ring would need 99 VGPRs next
// to the 128 accumulators, so the fp8 ring is two deep: the weights of K-step kt + 1 are requested at the top of
// K-step kt and awaited at its end (vmcnt(5): the LDS-DMA of this step, issued after them, stays in flight).  The
// kernel sits at the register limit; hipcc used to park loop-invariant LDS addresses in scratch and a reload behind
// the weight loads made every K-step wait for the loads it had just issued (in-order vmcnt): those addresses are
// re-derived from the lane id inside the loop instead (no spills).
// WF = weight operand format: 0 e2m1 (16 B per lane and 16 n), 1 e4m3 (32 B in two half-slots), 2 / 3 = fp6 e2m3 / e3m2
// (the MFMA's own format codes; 24 B per lane: a 16-byte and an 8-byte piece, 1.5 KiB per fragment slot = 6 bits per
// weight; lane (n % 16, kg) holds k = 32 kg .. +31, six bits each, little endian: scripts/experiments/mx_mfma_layout_fp6.hip).
// The fp6 operand uses the two-deep ring and the register budget of the e4m3 operand.
// KG = 2: two k-groups of four waves per block, as k_qgemm3 (single-pass grids of at most one block per CU).
// MFM = 16-row fragments per wave along m: 8 (block 128 x 256) or 4 (block 64 x 256: grids between one and two 128-row blocks
// per CU and half-chip grids with a short K, as k_qgemm3's 64-row tiles).
template <typename YT, int WF, int KG = 1, int MFM = 8>
__global__ void __launch_bounds__(256 * KG, KG == 2 ? 1 : 2)
k_mxgemm(const uint8_t* __restrict__ Xc, const uint8_t* __restrict__ Xs, const uint8_t* __restrict__ Wc,
         const uint8_t* __restrict__ Ws, const float* __restrict__ bias, YT* __restrict__ Y, int M, int N, int K,
         int ksplit, float* __restrict__ partial, int y16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool W8 = (WF != 0);                                // wide operand: two pieces per fragment, two-deep ring
    constexpr bool W6 = (WF >= 2);
    static_assert(MFM == 8 || MFM == 4, "block height");
    constexpr int BMX = 16 * MFM, KS = 128, A_TILE = BMX * KS;    // 16 / 8 KiB per activation buffer
    constexpr int PPW = BMX / 32;                                 // 1 KiB staging pieces (8 rows) per wave and K-step: 4 / 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid_blk = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgid = (KG == 2) ? (wid_blk >> 2) : 0;            // k-group of this wave
    const int wid = (KG == 2) ? (wid_blk & 3) : wid_blk;        // wave inside its group
    const int wn = wid;
    const int c = lane & 15, g = lane >> 4;
    const int MT = (M + BMX - 1) / BMX, NTB = N / BN;
    const int KT = K / KS;
    const int ks = (int)(blockIdx.x % (unsigned)ksplit);           // split-K (small M): K-steps [kt_lo, kt_hi), fp32 partial tile
    const int bid = (int)(blockIdx.x / (unsigned)ksplit);
    const int kchunk = (KT + ksplit - 1) / ksplit;
    int kt_lo = ks * kchunk;
    int kt_hi = (kt_lo + kchunk < KT) ? kt_lo + kchunk : KT;
    if (KG == 2) {                                              // the launcher guarantees an even number of K-steps per block
        const int half = (kt_hi - kt_lo) >> 1;
        kt_lo += kgid * half; kt_hi = kt_lo + half;
    }
    int bm, bn;
    if ((NTB & 7) == 0 && ksplit == 1) {                           // XCD-aware order, as k_qgemm3
        const int xcd = bid & 7, i = bid >> 3;
        const int npx = NTB >> 3, per_group = 8 * npx, full = MT >> 3;
        int rg, j, R;
        if (i < full * per_group) { rg = i / per_group; j = i % per_group; R = 8; }
        else { rg = full; j = i - full * per_group; R = MT - full * 8; }
        bm = rg * 8 + j % R;
        bn = (j / R) * 8 + xcd;
    } else { bm = bid % MT; bn = bid / MT; }
    const int m0 = bm * BMX, n0 = bn * BN;
    const int64_t wtiles = (int64_t)(N / 64) * KT;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(Wc, wtiles * (W6 ? 6144 : (W8 ? 8192 : 4096)));
    const __amdgpu_buffer_rsrc_t wsr = make_rsrc(Ws, wtiles * 256);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(Xc, (int64_t)M * K);
    const __amdgpu_buffer_rsrc_t xsr = make_rsrc(Xs, (int64_t)M * (K / 32));
    const uint32_t tile_row32 = (uint32_t)sgpr((n0 / 64 + wn) * KT);
    const int lane16 = lane * 16;
    // activation staging: piece = 4 wid + p covers rows 8 piece .. +7, 16-byte chunk (lane & 7) ^ ((row >> 1) & 7)
    // Two lane offsets only (pieces 0 and 1; piece p + 2 lies 16 rows = 16 K bytes further, the chunk swizzle repeats
    // every 16 rows): rows >= M are NOT clamped, their offset is >= M K = the descriptor's size and the hardware
    // range check makes the load a no-op / zero fill; those rows of the result are never stored.
    int aoff[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (wid * PPW + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        aoff[p] = (int)((int64_t)(m0 + row) * K + chunk * 16);
    }
    // the 4 scale bytes of (row, K-step) travel with the tile: every wave copies one dword per row for its 32 rows
    // (lanes 32-63 repeat lanes 0-31 into the upper half of the wave's 256 bytes: no wave-dependent branch, the
    // same number of vector-memory ops in every wave)
    // activation buffers: 3 staged two K-steps ahead (MSQ_MX_XBUFS = 4: the fp4 kernel stages three ahead; no gain)
    constexpr int XBUFS = W8 ? 3 : MSQ_MX_XBUFS;
    constexpr int XS_BASE = XBUFS * A_TILE;
    char* const smem_g = smem + kgid * (XBUFS * (A_TILE + 1024));   // this group's code and scale tiles
    // (64-row blocks: waves 2 and 3 repeat the copies of waves 0 and 1 -- same bytes to the same place)
    constexpr int XSW = (MFM == 8) ? 3 : 1;                      // wave -> 32-row scale group: wid & XSW
    int xs_goff = m0 + (wid & XSW) * 32 + (lane & 31); xs_goff = (xs_goff < M ? xs_goff : M - 1) * (K / 32);
    auto stage_A = [&](int kt, int buf) {
        int k16 = 16 * K;
        asm volatile("" : "+s"(k16));                             // keeps the two derived offsets out of registers across K-steps
#pragma unroll
        for (int p = 0; p < PPW; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (void __attribute__((address_space(3)))*)(smem_g + buf * A_TILE + (wid * PPW + p) * 1024),
                                                     16, p < 2 ? aoff[p] : aoff[p - 2] + k16, uni((uint32_t)kt * KS), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsr, (void __attribute__((address_space(3)))*)(smem_g + XS_BASE + buf * 1024 + (wid & XSW) * 256),
                                                 4, xs_goff, uni((uint32_t)kt * 4u), 0, 0);
    };
    // LDS reads of one B fragment (row mf * 16 + c): chunks g and 4 + g (k = 16 g .. and 64 + 16 g ..)
    const int sw = (c >> 1) & 7;
    const int rdl = c * 128 + ((g ^ sw) << 4), rdh = c * 128 + (((4 + g) ^ sw) << 4);
    // scale byte of (row mf * 16 + c, block g): row r lives at wave r / 32, slot r % 32: + (mf / 2) * 256 + (mf % 2) * 64
    const int xs_rd = XS_BASE + c * 4 + g;

    f32x4_t acc[MFM][4];
#pragma unroll
    for (int i = 0; i < MFM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int WV = W8 ? 2 : 1;                               // 16-byte pieces per lane and 16 n
    struct WSet { u32x4_t w[4][WV]; uint32_t s; };
    WSet w0, w1, w2;
    auto load_w = [&](WSet& ws, int kt) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int h = 0; h < WV; ++h) {
                if (W6 && h == 1) {
                    const u32x2_t t2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(wr, lane * 8, uni(((tile_row32 + (uint32_t)kt) * 4u + nf) * 1536u + 1024u), 0));
                    ws.w[nf][1] = u32x4_t{t2[0], t2[1], 0u, 0u};
                } else if (W6) {
                    ws.w[nf][0] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, uni(((tile_row32 + (uint32_t)kt) * 4u + nf) * 1536u), 0));
                } else {
                    ws.w[nf][h] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, uni((((tile_row32 + (uint32_t)kt) * 4u + nf) * WV + h) * 1024u), 0));
                }
            }
        ws.s = __builtin_amdgcn_raw_buffer_load_b32(wsr, lane * 4, uni((tile_row32 + (uint32_t)kt) * 256u), 0);
    };
    constexpr int CBSZ = W6 ? WF : (W8 ? 0 : 4);                 // A-operand format: e2m3 / e3m2 / e4m3 / e2m1
    constexpr int N_WAIT_MX = (W8 ? 0 : 5) + (PPW + 1) * (XBUFS == 4 ? 2 : 1);   // 128-row blocks: 5 / 10 (15 with four buffers)
    const int kl = sgpr((kt_hi > kt_lo) ? kt_hi - 1 : ((kt_lo < KT) ? kt_lo : KT - 1));   // an empty split runs a harmless prologue
    const int kf0 = sgpr((kt_lo < KT) ? kt_lo : KT - 1), kf1 = (kf0 + 1 <= kl) ? kf0 + 1 : kl;
    stage_A(kf0, 0);
    stage_A(kf1, 1);
    if (XBUFS == 4) stage_A((kf1 + 1 <= kl) ? kf1 + 1 : kl, 2);
    load_w(w0, kf0);
    if (!W8) load_w(w1, kf1);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

#define MSQ_MX_STEP(KT_CUR, WCUR, WLOAD, NWAIT)                                                               \
    {                                                                                                        \
        const int kt_ = sgpr(KT_CUR);                                                                        \
        const int buf = abuf, buf2 = (XBUFS == 4) ? ((abuf + 3) & 3) : ((abuf == 0) ? 2 : abuf - 1);         \
        abuf = (XBUFS == 4) ? ((abuf + 1) & 3) : ((abuf == 2) ? 0 : abuf + 1);                               \
        const char* abase = smem_g + buf * A_TILE;                                                           \
        const int k1 = (kt_ + 1 <= kl) ? kt_ + 1 : kl, k2 = (kt_ + 2 <= kl) ? kt_ + 2 : kl;   /* branch-free tail */ \
        if (!(MSQ_MXABL & 2)) load_w(WLOAD, W8 ? k1 : k2);   /* issue order (vmcnt is in-order): weights, then LDS-DMA */ \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (!(MSQ_MXABL & 4)) stage_A((XBUFS == 4) ? ((kt_ + 3 <= kl) ? kt_ + 3 : kl) : k2, buf2);           \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        /* fp8 ring: the kernel sits at the register limit and hipcc parks loop-invariant LDS addresses in scratch; a  \
           reload after the weight loads would wait for them (in-order vmcnt).  The three addresses are re-derived from \
           the lane id instead (volatile asm: not hoisted), 10 VALU per K-step */                                      \
        int rdl_ = rdl, rdh_ = rdh, xs_rd_ = xs_rd;                                                           \
        if (W8) { int l_; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_)); \
                  const int c_ = l_ & 15, g_ = l_ >> 4, sw_ = (c_ >> 1) & 7;                                  \
                  rdl_ = c_ * 128 + ((g_ ^ sw_) << 4); rdh_ = rdl_ ^ 64; xs_rd_ = XS_BASE + c_ * 4 + g_; }    \
        u32x4_t xl[2], xh[2];                                                                                \
        xl[0] = *reinterpret_cast<const u32x4_t*>(abase + rdl_); xh[0] = *reinterpret_cast<const u32x4_t*>(abase + rdh_);           \
        uint32_t xsc[2];                                                                                     \
        xsc[0] = *reinterpret_cast<const uint8_t*>(smem_g + xs_rd_ + buf * 1024);                               \
        _Pragma("unroll") for (int mf = 0; mf < MFM; ++mf) {                                                 \
            if (mf + 1 < MFM && !(MSQ_MXABL & 1)) { xl[(mf + 1) & 1] = *reinterpret_cast<const u32x4_t*>(abase + rdl_ + (mf + 1) * 2048);                 \
                              xh[(mf + 1) & 1] = *reinterpret_cast<const u32x4_t*>(abase + rdh_ + (mf + 1) * 2048);                 \
                              xsc[(mf + 1) & 1] = *reinterpret_cast<const uint8_t*>(smem_g + xs_rd_ + buf * 1024 + ((mf + 1) >> 1) * 256 + ((mf + 1) & 1) * 64); }  \
            if (!W8 && MSQ_MX_PIN_READS) __builtin_amdgcn_sched_barrier(0);   /* hipcc otherwise sinks these reads below the MFMAs of this group and waits for them at once (fp8 ring: the pin costs registers -> a scratch reload in the loop, slower) */ \
            const u32x4_t lo = xl[(MSQ_MXABL & 1) ? 0 : (mf & 1)], hi = xh[(MSQ_MXABL & 1) ? 0 : (mf & 1)];  \
            const v8i_t bfr = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]}; \
            const int sb_ = (int)xsc[(MSQ_MXABL & 1) ? 0 : (mf & 1)];                                        \
            _Pragma("unroll") for (int nf = 0; nf < 4; ++nf) {                                               \
                const u32x4_t wl_ = WCUR.w[nf][0], wh_ = WCUR.w[nf][WV - 1];                                  \
                const v8i_t afr = {(int)wl_[0], (int)wl_[1], (int)wl_[2], (int)wl_[3], W8 ? (int)wh_[0] : 0, W8 ? (int)wh_[1] : 0, W8 ? (int)wh_[2] : 0, W8 ? (int)wh_[3] : 0}; \
                if (nf == 0) acc[mf][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afr, bfr, acc[mf][0], CBSZ, 0, 0, (int)WCUR.s, 0, sb_); \
                else if (nf == 1) acc[mf][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afr, bfr, acc[mf][1], CBSZ, 0, 1, (int)WCUR.s, 0, sb_); \
                else if (nf == 2) acc[mf][2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afr, bfr, acc[mf][2], CBSZ, 0, 2, (int)WCUR.s, 0, sb_); \
                else acc[mf][3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afr, bfr, acc[mf][3], CBSZ, 0, 3, (int)WCUR.s, 0, sb_); \
            }                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
        /* this K-step's 5 weight loads and 5 LDS-DMA ops (all for K-step kt + 2) stay in flight; everything \
           issued in the previous K-step -- the tile and the weights of K-step kt + 1 -- has landed */       \
        __builtin_amdgcn_s_waitcnt(0x0070 | (NWAIT));                                                        \
        if (!(MSQ_MXABL & 8)) __builtin_amdgcn_s_barrier();                                                  \
    }

    int abuf = 0;
    {
        int kt = kt_lo;
        // The steps after the loop load weights nobody will use: hipcc deletes those loads, so such a step issues only its
        // LDS-DMA ops and the loop's wait count (which lets a whole step's loads stay in flight) would no longer cover the
        // tile staged one step earlier -- the next tail step would read it unawaited (seen as run-to-run differences with
        // the short K-steps of the 64-row blocks).  Tail steps therefore let only their own LDS-DMA ops stay in flight.
        constexpr int N_WAIT_TAIL = (PPW + 1) * (XBUFS == 4 ? 2 : 1);
        if constexpr (W8) {
            for (; kt + 1 < kt_hi; kt += 2) { MSQ_MX_STEP(kt, w0, w1, N_WAIT_MX) MSQ_MX_STEP(kt + 1, w1, w0, N_WAIT_MX) }
            if (kt < kt_hi) { MSQ_MX_STEP(kt, w0, w1, N_WAIT_TAIL) ++kt; }
        } else {
            for (; kt + 2 < kt_hi; kt += 3) { MSQ_MX_STEP(kt, w0, w2, N_WAIT_MX) MSQ_MX_STEP(kt + 1, w1, w0, N_WAIT_MX) MSQ_MX_STEP(kt + 2, w2, w1, N_WAIT_MX) }
            if (kt < kt_hi) { MSQ_MX_STEP(kt, w0, w2, N_WAIT_TAIL) ++kt; }
            if (kt < kt_hi) { MSQ_MX_STEP(kt, w1, w0, N_WAIT_TAIL) ++kt; }
        }
    }
#undef MSQ_MX_STEP
    __builtin_amdgcn_s_waitcnt(0x0070);                        // drain the re-staged tail tiles before the epilogue reuses LDS
    __builtin_amdgcn_s_barrier();
    if (MSQ_MXABL & 16) { float t = 0.f; _Pragma("unroll") for (int i = 0; i < MFM; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; if (t == 1.2345f) reinterpret_cast<float*>(Y)[0] = t; return; }
    if constexpr (KG == 2) {                                    // group 1 -> LDS, group 0 adds (first half of K + second half), as k_qgemm3
        float4* red = reinterpret_cast<float4*>(smem) + wid * (MFM * 4 * 64);
        if (kgid == 1) {
#pragma unroll
            for (int i = 0; i < MFM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[(i * 4 + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (kgid == 0) {
#pragma unroll
            for (int i = 0; i < MFM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = red[(i * 4 + j) * 64 + lane];
                    acc[i][j][0] += t.x; acc[i][j][1] += t.y; acc[i][j][2] += t.z; acc[i][j][3] += t.w;
                }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                           // the epilogue slices overlap wave 0's hand-over area
        if (kgid == 1) return;
    }
    if (ksplit > 1) store_wave_tile_lds<float, MFM>(acc, smem + wid * 8192, partial + (int64_t)ks * M * N, m0, n0 + wn * 64, M, N, nullptr, lane);
    else store_wave_tile_lds<YT, MFM>(acc, smem + wid * 8192, Y, m0, n0 + wn * 64, M, N, bias, lane, y16);
}

// ---------------------------------------------------------------------------
// MX-native decode kernel (M <= 16): streams the 4.25-bit weight once.  One wave = one task = KC consecutive
// 64(n) x 128(k) packed tiles of one 64-column strip: per tile 4 weight fragments (16 B per lane each) + one scale
// dword, 4 scaled MFMAs against the activation fragment, which is read straight from global memory (X codes are
// <= 16 rows: L2-resident).  The next tile's weights are in flight while the current one is consumed; fp32 partial
// tiles per k-chunk, summed by k_splitk_reduce.
// ---------------------------------------------------------------------------
// WAVES (4 or 16) waves per block take consecutive k-chunks of one strip and meet in LDS.  When one block covers all
// of K (nkb == 1, `direct`) wave 0 adds the bias and writes Y itself: one launch, no partial planes, no reduce kernel.
// NFB = 16-column fragments per block: 4 (the whole 64-column strip) or 2 (half strips: twice the blocks for the
// narrow projections, where 64 blocks leave three quarters of the CUs without work)
template <int WF, int MG, int WAVES, int NFB = 4>
__global__ void __launch_bounds__(64 * WAVES)
k_mxgemv(const uint8_t* __restrict__ Xc, const uint8_t* __restrict__ Xs, const uint8_t* __restrict__ Wc,
         const uint8_t* __restrict__ Ws, float* __restrict__ partial, int M, int N, int K, int kc,
         int direct, const float* __restrict__ bias, void* __restrict__ Y, int y_bf16) {
    extern __shared__ __attribute__((aligned(16))) char smem_v[];
    float (*red)[4 * NFB * MG][64] = reinterpret_cast<float (*)[4 * NFB * MG][64]>(smem_v);      // [WAVES - 1][4 NFB MG][64]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int KT = K / 128;
    const int nks = (KT + kc - 1) / kc;
    const int nkb = (nks + WAVES - 1) / WAVES;                 // the block's waves take WAVES consecutive k-chunks
    const int sidx = blockIdx.x / nkb, kb = blockIdx.x % nkb;
    const int strip = (NFB == 4) ? sidx : (sidx >> 1);
    const int nf0 = (NFB == 4) ? 0 : (sidx & 1) * 2;           // first fragment of this block inside the strip
    const int ks = kb * WAVES + wid;
    const int kt_lo = ks * kc < KT ? ks * kc : KT;
    const int kt_hi = (kt_lo + kc < KT) ? kt_lo + kc : KT;
    const int64_t tile_row = (int64_t)strip * KT;
    // MG groups of 16 activation rows; rows >= M are clamped, their results are never stored
    const uint8_t* xrow[MG];
    const uint8_t* xsrow[MG];
#pragma unroll
    for (int j = 0; j < MG; ++j) {
        int m = j * 16 + c; m = m < M ? m : M - 1;
        xrow[j] = Xc + (int64_t)m * K + g * 16;
        xsrow[j] = Xs + (int64_t)m * (K / 32) + g;
    }
    f32x4_t acc[NFB][MG];
#pragma unroll
    for (int i = 0; i < NFB; ++i)
#pragma unroll
        for (int j = 0; j < MG; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr bool W8 = (WF != 0), W6 = (WF >= 2);               // as k_mxgemm
    constexpr int WV = W8 ? 2 : 1, CBSZ = W6 ? WF : (W8 ? 0 : 4);
    struct WT { u32x4_t w[NFB][WV]; uint32_t s; };
    auto load_w = [&](WT& t, int64_t tile) {
#pragma unroll
        for (int nf = 0; nf < NFB; ++nf)
#pragma unroll
            for (int h = 0; h < WV; ++h) {
                if (W6 && h == 1) {
                    const u32x2_t t2 = MSQ_GV_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(Wc + (tile * 4 + nf0 + nf) * 1536 + 1024 + lane * 8)) : *reinterpret_cast<const u32x2_t*>(Wc + (tile * 4 + nf0 + nf) * 1536 + 1024 + lane * 8);
                    t.w[nf][1] = u32x4_t{t2[0], t2[1], 0u, 0u};
                } else if (W6) {
                    t.w[nf][0] = MSQ_GV_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(Wc + (tile * 4 + nf0 + nf) * 1536 + lane * 16)) : *reinterpret_cast<const u32x4_t*>(Wc + (tile * 4 + nf0 + nf) * 1536 + lane * 16);
                } else {
                    t.w[nf][h] = MSQ_GV_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(Wc + (((tile * 4 + nf0 + nf) * WV + h) * 64 + lane) * 16)) : *reinterpret_cast<const u32x4_t*>(Wc + (((tile * 4 + nf0 + nf) * WV + h) * 64 + lane) * 16);
                }
            }
        t.s = *reinterpret_cast<const uint32_t*>(Ws + (tile * 64 + lane) * 4) >> (8 * nf0);       // byte nf of the dword = fragment nf0 + nf
    };
    // tiles in flight ahead of the one being multiplied: 1 with four waves per block (~3000 waves in the grid), 2 with
    // sixteen (fewer, longer waves)
    constexpr bool DEEP2 = (WAVES >= 8);
    WT cur, nxt, nx2;
    if (kt_lo < kt_hi) load_w(cur, tile_row + kt_lo);
    if (DEEP2 && kt_lo < kt_hi) load_w(nxt, tile_row + ((kt_lo + 1 < kt_hi) ? kt_lo + 1 : kt_lo));
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        if (DEEP2) { const int ktn = (kt + 2 < kt_hi) ? kt + 2 : kt_hi - 1; load_w(nx2, tile_row + ktn); }
        else { const int ktn = (kt + 1 < kt_hi) ? kt + 1 : kt; load_w(nxt, tile_row + ktn); }
        v8i_t bfr[MG];
        int sb[MG];
#pragma unroll
        for (int j = 0; j < MG; ++j) {
            const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(xrow[j] + (int64_t)kt * 128);        // k = 16 g ..
            const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(xrow[j] + (int64_t)kt * 128 + 64);   // k = 64 + 16 g ..
            sb[j] = (int)xsrow[j][kt * 4];
            bfr[j] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int nf = 0; nf < NFB; ++nf) {
            const u32x4_t wl = cur.w[nf][0], wh = cur.w[nf][WV - 1];
            const v8i_t afr = {(int)wl[0], (int)wl[1], (int)wl[2], (int)wl[3], W8 ? (int)wh[0] : 0, W8 ? (int)wh[1] : 0, W8 ? (int)wh[2] : 0, W8 ? (int)wh[3] : 0};
#pragma unroll
            for (int j = 0; j < MG; ++j) {
                if (nf == 0) acc[0][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afr, bfr[j], acc[0][j], CBSZ, 0, 0, (int)cur.s, 0, sb[j]);
                else if (nf == 1) acc[1][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afr, bfr[j], acc[1][j], CBSZ, 0, 1, (int)cur.s, 0, sb[j]);
                else if (nf == 2) acc[2 % NFB][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afr, bfr[j], acc[2 % NFB][j], CBSZ, 0, 2, (int)cur.s, 0, sb[j]);
                else acc[3 % NFB][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afr, bfr[j], acc[3 % NFB][j], CBSZ, 0, 3, (int)cur.s, 0, sb[j]);
            }
        }
        cur = nxt;
        if (DEEP2) nxt = nx2;
    }
    // the four k-chunks meet in LDS (fixed order: wave 0 + 1 + 2 + 3), one partial plane per block
    if (wid > 0) {
#pragma unroll
        for (int j = 0; j < MG; ++j)
#pragma unroll
            for (int nf = 0; nf < NFB; ++nf)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wid - 1][(j * NFB + nf) * 4 + e][lane] = acc[nf][j][e];
    }
    __syncthreads();
    if (wid != 0) return;
    // D[lane, e] = C[n = 4 g + e][m = 16 j + c]
#pragma unroll
    for (int j = 0; j < MG; ++j) {
        const int m = j * 16 + c;
        if (m >= M) continue;
        const int n0 = strip * 64 + nf0 * 16 + g * 4;
        float* pbase = partial + ((int64_t)kb * M + m) * N + n0;
#pragma unroll
        for (int nf = 0; nf < NFB; ++nf) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = (j * NFB + nf) * 4 + e;
                float t = acc[nf][j][e];
#pragma unroll
                for (int w = 0; w < WAVES - 1; ++w) t += red[w][r][lane];           // fixed order: k-chunk 0 + 1 + 2 + ...
                v[e] = t;
            }
            if (!direct) { *reinterpret_cast<float4*>(pbase + nf * 16) = make_float4(v[0], v[1], v[2], v[3]); continue; }
            const int n = n0 + nf * 16;
            if (bias) { v[0] += bias[n]; v[1] += bias[n + 1]; v[2] += bias[n + 2]; v[3] += bias[n + 3]; }
            if (!y_bf16) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Y) + (int64_t)m * N + n) = make_float4(v[0], v[1], v[2], v[3]);
            else if (y_bf16 == 2) {
                f16x2_t lo, hi;
                lo[0] = (_Float16)v[0]; lo[1] = (_Float16)v[1]; hi[0] = (_Float16)v[2]; hi[1] = (_Float16)v[3];
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(Y) + (int64_t)m * N + n) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
            } else {
                bf16x2_t lo, hi;
                lo[0] = (__bf16)v[0]; lo[1] = (__bf16)v[1]; hi[0] = (__bf16)v[2]; hi[1] = (__bf16)v[3];
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(Y) + (int64_t)m * N + n) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Small-M (decode) kernel: M <= 16 (MG = 1; the template still takes MG 16-row groups).  HBM-bound: the job is to stream the packed weight once at full
// rate.  One wave = one task = KC consecutive 64x64 packed tiles of one 64-column strip; per tile it
// converts the 8 fragments and issues 8*MG MFMAs against the activation fragments, which it reads
// straight from global memory (the whole X is <= 512 KiB and stays in L2).  Next tile's packed data
// is in flight while the current one is consumed; 12 waves per CU keep ~80 KiB of loads in flight.
// The four waves of a block sum their k-chunks in LDS and write one fp32 partial tile, k_splitk_reduce sums those.
// ---------------------------------------------------------------------------
// WAVES = 16 (unified layouts, `direct`): one block covers all of K, wave 0 adds the bias and writes Y itself -- one
// launch, no partial planes, no reduce kernel (as k_mxgemv).
template <int IN_KIND, int OUT_KIND, int MG, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
k_qgemv(const uint16_t* __restrict__ X, const uint8_t* __restrict__ inl_plane, const uint8_t* __restrict__ out_plane,
        const uint8_t* __restrict__ scl_plane, float* __restrict__ partial, int M, int N, int K, int scl_groups, int kc,
        int direct, const float* __restrict__ bias, void* __restrict__ Y, int y_bf16, int x_f16) {
    extern __shared__ __attribute__((aligned(16))) char smem_v[];
    float (*red)[16 * MG][64] = reinterpret_cast<float (*)[16 * MG][64]>(smem_v);      // [WAVES - 1][16 MG][64]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int KT = K / TILE_K;
    const int nks = (KT + kc - 1) / kc;                        // k-chunks per strip
    const int nkb = (nks + WAVES - 1) / WAVES;                 // the block's waves take WAVES consecutive k-chunks
    const int strip = blockIdx.x / nkb, kb = blockIdx.x % nkb; //   of one strip and share its X rows in L1
    const int ks = kb * WAVES + wid;
    const int kt_lo = ks * kc < KT ? ks * kc : KT;             // chunks past the end are empty
    const int kt_hi = (kt_lo + kc < KT) ? kt_lo + kc : KT;
    const int64_t tile_row = (int64_t)strip * KT;

    f32x4_t acc[4][MG];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MG; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // activation fragment rows of this lane (rows >= M are clamped; their results are never stored)
    const uint16_t* xrow[MG];
#pragma unroll
    for (int j = 0; j < MG; ++j) { int m = j * 16 + c; m = m < M ? m : M - 1; xrow[j] = X + (int64_t)m * K + g * 8; }

    constexpr bool DEEP2 = (WAVES == 16);                      // tiles in flight ahead: 2 with few long waves, else 1
    TileRegs cur, nxt, nx2;
    load_tile<IN_KIND, OUT_KIND, MSQ_GV_NT>(cur, inl_plane, out_plane, scl_plane, tile_row + (kt_lo < KT ? kt_lo : KT - 1), lane, scl_groups);
    if (DEEP2) load_tile<IN_KIND, OUT_KIND, MSQ_GV_NT>(nxt, inl_plane, out_plane, scl_plane, tile_row + ((kt_lo + 1 < kt_hi) ? kt_lo + 1 : (kt_lo < KT ? kt_lo : KT - 1)), lane, scl_groups);
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        if (DEEP2) { const int ktn = (kt + 2 < kt_hi) ? kt + 2 : kt_hi - 1; load_tile<IN_KIND, OUT_KIND, MSQ_GV_NT>(nx2, inl_plane, out_plane, scl_plane, tile_row + ktn, lane, scl_groups); }
        else { const int ktn = (kt + 1 < kt_hi) ? kt + 1 : kt; load_tile<IN_KIND, OUT_KIND, MSQ_GV_NT>(nxt, inl_plane, out_plane, scl_plane, tile_row + ktn, lane, scl_groups); }
        bf16x8_t xf[2][MG];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int j = 0; j < MG; ++j)
                if (x_f16) {     // fp16 activations (an fp16 model at decode sizes): converted here, half -> float (exact) -> bf16 (RNE) = x.to(bfloat16), no cast launch
                    union { u32x4_t u; _Float16 h[8]; } r;
                    r.u = *reinterpret_cast<const u32x4_t*>(xrow[j] + (int64_t)kt * TILE_K + kf * 32);
                    bf16x8_t t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (__bf16)(float)r.h[e];
                    xf[kf][j] = t;
                } else
                xf[kf][j] = *reinterpret_cast<const bf16x8_t*>(xrow[j] + (int64_t)kt * TILE_K + kf * 32);
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, tile_frag<IN_KIND, OUT_KIND>(cur, nf, kf));
#pragma unroll
                for (int j = 0; j < MG; ++j)
                    acc[nf][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[kf][j], acc[nf][j], 0, 0, 0);
            }
        cur = nxt;
        if (DEEP2) nxt = nx2;
    }
    // the four k-chunks meet in LDS (fixed order: wave 0 + 1 + 2 + 3): one partial plane per block
    if (wid > 0) {
#pragma unroll
        for (int j = 0; j < MG; ++j)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wid - 1][(j * 4 + nf) * 4 + e][lane] = acc[nf][j][e];
    }
    __syncthreads();
    if (wid != 0) return;
    // D[n = 4 g + r][m = c]: 4 consecutive n of row m
    float* pbase = partial + (int64_t)kb * M * N;
#pragma unroll
    for (int j = 0; j < MG; ++j) {
        const int m = j * 16 + c;
        if (m >= M) continue;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n = strip * TILE_N + nf * 16 + g * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = (j * 4 + nf) * 4 + e;
                float t = acc[nf][j][e];
#pragma unroll
                for (int w = 0; w < WAVES - 1; ++w) t += red[w][r][lane];           // fixed order: k-chunk 0 + 1 + 2 + ...
                v[e] = t;
            }
            if (!direct) { *reinterpret_cast<float4*>(pbase + (int64_t)m * N + n) = make_float4(v[0], v[1], v[2], v[3]); continue; }
            if (bias) { v[0] += bias[n]; v[1] += bias[n + 1]; v[2] += bias[n + 2]; v[3] += bias[n + 3]; }
            if (!y_bf16) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Y) + (int64_t)m * N + n) = make_float4(v[0], v[1], v[2], v[3]);
            else if (y_bf16 == 2) {
                f16x2_t lo, hi;
                lo[0] = (_Float16)v[0]; lo[1] = (_Float16)v[1]; hi[0] = (_Float16)v[2]; hi[1] = (_Float16)v[3];
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(Y) + (int64_t)m * N + n) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
            } else {
                bf16x2_t lo, hi;
                lo[0] = (__bf16)v[0]; lo[1] = (__bf16)v[1]; hi[0] = (__bf16)v[2]; hi[1] = (__bf16)v[3];
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(Y) + (int64_t)m * N + n) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Decode kernel of the unified layouts (MSQ-U1 / U1X) for the wide projections (more than 128 strips of 64 columns), single
// launch: one block = one strip over ALL of K, its WAVES waves take consecutive runs of kc tiles and meet in LDS.  What changed
// against k_qgemv<..., 16> (measured with COLD weights, scripts/experiments/decode_cold.py: a graph that replays ONE weight
// streams it from the Infinity Cache and hides all of this):
//  * WAVES = 8 / 4 / 2 blocks run two or more per CU: the 344 strips of the fused gate / up projection are resident at once instead
//    of taking a second, quarter-full round of sixteen-wave blocks (31.2 -> 27.0 us at M = 1 with eight waves, 23.7 with four);
//  * any K (wave w takes tiles w, w + WAVES, ... to the end);
//  * the hand-over is summed by all waves (wave w owns results w, w + WAVES, ...; fixed order over the k-runs: bit-identical
//    run to run), not by wave 0 alone;
//  * XPF: the activation fragments of the next tile are requested one tile ahead (M > 1: sixteen different rows, an exposed L2
//    round trip per tile otherwise).
// Per tile a wave loads four code slots (16 B per lane), the scale pair and, for U1X, two extension dwords; two tiles in flight
// (DEEP3: three -- slower with cold weights: more bytes in flight than the DRAM pages like, see the launcher).
// Half strips over all of K for the 4096-wide projections (128 blocks) were built and measured slower than the split-K planes
// of k_qgemv (down projection 14.7 -> 17.3 us): a CU keeps only so many bytes in flight, 128 CUs cannot pull what 256 can.
// ---------------------------------------------------------------------------
template <int OUT_KIND, int MG, int WAVES, bool XPF, bool DEEP3 = false>
__global__ void __launch_bounds__(64 * WAVES)
k_qgemv_u(const uint16_t* __restrict__ X, const uint8_t* __restrict__ ext_plane, const uint8_t* __restrict__ code_plane,
          const uint8_t* __restrict__ scl_plane, int M, int N, int K, int kc, const float* __restrict__ bias, void* __restrict__ Y,
          int y_kind, int x_f16) {
    constexpr bool EXT = (OUT_KIND == MSQ_PLANE_U8X);
    constexpr int NFB = 4;                                       // 16-column fragments per block
    constexpr int R = MG * NFB * 4;                              // results per lane
    extern __shared__ __attribute__((aligned(16))) char smem_u[];
    float (*red)[R][64] = reinterpret_cast<float (*)[R][64]>(smem_u);          // [WAVES][R][64]
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int KT = K / TILE_K;
    const int strip = blockIdx.x;
    // wave w takes tiles w, w + WAVES, ...: the block reads one moving window of WAVES tiles (1-2 % faster with cold weights than a
    // contiguous run per wave: a quarter as many separate streams for the DRAM pages)
    (void)kc;
    constexpr int stepk = WAVES;
    const int kt_lo = wid < KT ? wid : KT;
    const int kt_hi = KT;
    const int64_t tile_row = (int64_t)strip * KT;

    f32x4_t acc[NFB][MG];
#pragma unroll
    for (int i = 0; i < NFB; ++i)
#pragma unroll
        for (int j = 0; j < MG; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint16_t* xrow[MG];
#pragma unroll
    for (int j = 0; j < MG; ++j) { int m = j * 16 + c; m = m < M ? m : M - 1; xrow[j] = X + (int64_t)m * K + g * 8; }

    struct UT { u32x4_t code[2][2]; uint32_t scl[2]; uint32_t ext[2]; };
    struct XF { u32x4_t v[2][MG]; };
    auto load_u = [&](UT& t, int kt) {
        const int64_t tile = tile_row + kt;
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                t.code[kf][p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(code_plane + ((tile * 4 + kf * 2 + p) * 64 + lane) * 16));   // streamed once: keep the activation rows in L2
        const uint2 sc = *reinterpret_cast<const uint2*>(scl_plane + (tile * 16 + c) * 8);
        t.scl[0] = sc.x; t.scl[1] = sc.y;
        if (EXT) {
            t.ext[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(ext_plane + ((tile * 2 + 0) * 64 + lane) * 4));
            t.ext[1] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(ext_plane + ((tile * 2 + 1) * 64 + lane) * 4));
        }
    };
    auto load_x = [&](XF& x, int kt) {
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int j = 0; j < MG; ++j) x.v[kf][j] = *reinterpret_cast<const u32x4_t*>(xrow[j] + (int64_t)kt * TILE_K + kf * 32);
    };
    UT cur, nxt, nx2;
    XF xc, xn;
    const int kt_safe = kt_lo < KT ? kt_lo : KT - 1;
    load_u(cur, kt_safe);
    if (XPF) load_x(xc, kt_safe);
    if (DEEP3) load_u(nxt, (kt_lo + stepk < kt_hi) ? kt_lo + stepk : kt_safe);
    for (int kt = kt_lo; kt < kt_hi; kt += stepk) {
        if (DEEP3) load_u(nx2, (kt + 2 * stepk < kt_hi) ? kt + 2 * stepk : kt);
        else load_u(nxt, (kt + stepk < kt_hi) ? kt + stepk : kt);
        if (XPF) load_x(xn, (kt + stepk < kt_hi) ? kt + stepk : kt);
        else load_x(xc, kt);
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) {
            bf16x8_t xf[MG];
#pragma unroll
            for (int j = 0; j < MG; ++j) {
                if (x_f16) {     // fp16 activations: half -> float (exact) -> bf16 (RNE) = x.to(bfloat16), no cast launch
                    union { u32x4_t u; _Float16 h[8]; } r;
                    r.u = xc.v[kf][j];
                    bf16x8_t t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (__bf16)(float)r.h[e];
                    xf[j] = t;
                } else xf[j] = __builtin_bit_cast(bf16x8_t, xc.v[kf][j]);
            }
#pragma unroll
            for (int nf = 0; nf < NFB; ++nf) {
                const u32x4_t o = cur.code[kf][nf >> 1];
                const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, dequant_frag_unified<OUT_KIND>(o[(nf & 1) * 2], o[(nf & 1) * 2 + 1],
                                                                                               scale_operand(cur.scl[kf], nf), cur.ext[kf], nf));
#pragma unroll
                for (int j = 0; j < MG; ++j)
                    acc[nf][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[j], acc[nf][j], 0, 0, 0);
            }
        }
        cur = nxt;
        if (DEEP3) nxt = nx2;
        if (XPF) xc = xn;
    }
    // hand-over: every wave leaves its partial results, wave w sums results w, w + WAVES, ... over the k-runs in order
#pragma unroll
    for (int j = 0; j < MG; ++j)
#pragma unroll
        for (int nf = 0; nf < NFB; ++nf)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wid][(j * NFB + nf) * 4 + e][lane] = acc[nf][j][e];
    __syncthreads();
    for (int r = wid; r < R; r += WAVES) {
        float t = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t += red[w][r][lane];
        const int e = r & 3, nf = (r >> 2) % NFB, j = (r >> 2) / NFB;
        const int m = j * 16 + c;
        if (m >= M) continue;
        const int n = strip * TILE_N + nf * 16 + g * 4 + e;      // D[n = 4 g + e][m = c]
        if (bias) t += bias[n];
        if (y_kind == 0) reinterpret_cast<float*>(Y)[(int64_t)m * N + n] = t;
        else if (y_kind == 2) reinterpret_cast<uint16_t*>(Y)[(int64_t)m * N + n] = __builtin_bit_cast(uint16_t, (_Float16)t);
        else reinterpret_cast<uint16_t*>(Y)[(int64_t)m * N + n] = __builtin_bit_cast(uint16_t, (__bf16)t);
    }
}

// sum of the split-K partial tiles (+ bias) -> Y.  The grids are tiny at decode sizes (M N / 1024 workgroups), so
// the kernel is one load round trip long only if all planes are requested before the first add: the loads are
// issued 8 at a time, the sum keeps the k order.
template <typename YT>
__global__ void __launch_bounds__(256)
k_splitk_reduce(const float* __restrict__ partial, const float* __restrict__ bias, YT* __restrict__ Y, int64_t MN, int N,
                int ksplit, int y16) {
    const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= MN) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= ksplit; k += 8) {
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = *reinterpret_cast<const float4*>(partial + (int64_t)(k + u) * MN + i4);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (k + u == 0) s = p[0];
            else { s.x += p[u].x; s.y += p[u].y; s.z += p[u].z; s.w += p[u].w; }
        }
    }
    for (; k < ksplit; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(partial + (int64_t)k * MN + i4);
        if (k == 0) s = p;
        else { s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
    }
    if (bias) { const int n = (int)(i4 % N); s.x += bias[n]; s.y += bias[n + 1]; s.z += bias[n + 2]; s.w += bias[n + 3]; }
    if (sizeof(YT) == 4) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Y) + i4) = s;
    else if (y16) {
        f16x2_t lo, hi;
        lo[0] = (_Float16)s.x; lo[1] = (_Float16)s.y; hi[0] = (_Float16)s.z; hi[1] = (_Float16)s.w;
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(Y) + i4) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
    } else {
        bf16x2_t lo, hi;
        lo[0] = (__bf16)s.x; lo[1] = (__bf16)s.y; hi[0] = (__bf16)s.z; hi[1] = (__bf16)s.w;
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(Y) + i4) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
    }
}

// ===========================================================================
// C ABI
// ===========================================================================
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: remember it per device (one bit per
// device ordinal, atomically; ordinals >= 64 simply set the attribute on every launch), not once per process.
#include <atomic>
struct DevOnce { std::atomic<uint64_t> mask{0}; };
static inline bool attr_needed(const DevOnce& o) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    return !(o.mask.load(std::memory_order_acquire) & (1ull << d));
}
static inline void attr_done(DevOnce& o) {
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) o.mask.fetch_or(1ull << d, std::memory_order_release);
}

static thread_local char g_err2[256] = "";
extern "C" const char* msq_last_error(void);
static int fail2(int code, const char* msg);
static int check_launch2(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char b[200]; snprintf(b, sizeof(b), "%s: %s", what, hipGetErrorString(e)); return fail2(MSQ_ERR_LAUNCH, b); }
    return MSQ_OK;
}
// MSQ_PACK_TWO_PASS (tests / A-B): msq_outlier_pack through the two-pass kernels even where the single-pass one applies; msq_set_tuning or environment
#include <atomic>
#include <limits.h>
static std::atomic<int> g_tune_two_pass{INT_MIN};
static bool pack_two_pass_forced() {
    const int t = g_tune_two_pass.load(std::memory_order_relaxed);
    if (t != INT_MIN) return t != 0;
    return getenv("MSQ_PACK_TWO_PASS") != nullptr;
}
extern "C" void msq_set_error_(const char* msg);
// tile layout of the packed planes: 1 = 16x16x32 fragments.  (Layout 2 = 32x32x16 fragments was built
// and measured 8-10 % slower end to end -- the chip holds a lower clock on that MFMA shape -- and removed;
// the frag_n / frag_k maps keep the seam.)
static int fail2(int code, const char* msg) { msq_set_error_(msg); (void)g_err2; return code; }

// bf16-plane packing of given values (in_kind NONE, out_kind BF16): exact iff every value is a bf16
__global__ void __launch_bounds__(256)
k_pack_values_bf16(const float* __restrict__ W, uint8_t* __restrict__ out_plane, int64_t N, int64_t K, int* status) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t KT = K / TILE_K, NT = N / TILE_N;
    if (tile >= KT * NT) return;
    const int64_t nt = tile / KT, kt = tile % KT;
    int st = 0;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int64_t n = nt * TILE_N + frag_n(lane, nf);
            const int64_t k = kt * TILE_K + frag_k(lane, kf);
            const float4 v0 = *reinterpret_cast<const float4*>(W + n * K + k), v1 = *reinterpret_cast<const float4*>(W + n * K + k + 4);
            const float v[8] = {v0.x + 0.f, v0.y + 0.f, v0.z + 0.f, v0.w + 0.f, v1.x + 0.f, v1.y + 0.f, v1.z + 0.f, v1.w + 0.f};
            u32x4_t o;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                o[w] = (f2u(v[2 * w]) >> 16) | (f2u(v[2 * w + 1]) & 0xFFFF0000u);
                if (((f2u(v[2 * w]) | f2u(v[2 * w + 1])) & 0xFFFFu) != 0u) st |= (v[2 * w] != v[2 * w] || v[2 * w + 1] != v[2 * w + 1]) ? MSQ_STATUS_NAN : MSQ_STATUS_INEXACT;
            }
            *reinterpret_cast<u32x4_t*>(out_plane + ((tile * 8 + kf * 4 + nf) * 64 + lane) * 16) = o;
        }
    if (st && status) atomicOr(status, st);
}

extern "C" {

int msq_packed_kinds(int inlier_fmt, int outlier_fmt, int* in_kind, int* out_kind) {
    msq_host::FmtInfo fi, fo;
    if (!msq_host::format_info(inlier_fmt, &fi) || !msq_host::format_info(outlier_fmt, &fo))
        return fail2(MSQ_ERR_BAD_ARG, "msq_packed_kinds: unknown element format");
    int ik = MSQ_PLANE_NONE, ok = MSQ_PLANE_BF16;
    if (inlier_fmt == MSQ_FMT_FP4_E2M1 || inlier_fmt == MSQ_FMT_INT2) {
        ik = MSQ_PLANE_FP4;
        switch (outlier_fmt) {
            case MSQ_FMT_FP8_E4M3: case MSQ_FMT_FP4_E2M1: case MSQ_FMT_FP6_E3M2: case MSQ_FMT_FP6_E2M3:
            case MSQ_FMT_INT4: case MSQ_FMT_INT2: ok = MSQ_PLANE_FP8; break;
            case MSQ_FMT_FP8_E5M2: ok = MSQ_PLANE_BF8; break;
            default: ok = MSQ_PLANE_BF16; break;      // posit<n,es>, int8, fp16, bf16
        }
    }
    if (in_kind) *in_kind = ik;
    if (out_kind) *out_kind = ok;
    return MSQ_OK;
}

// significant bits (implicit 1 + fraction) a format's values can carry
static int sig_bits(const msq_host::FmtInfo& f, int fmt_id) {
    if (f.kind == 1) {                                         // posit<n,es>: n - 1 (sign) - 2 (shortest regime) - es fraction bits
        const int n = (fmt_id >> 2) & 0x3F, es = fmt_id & 3;
        const int fr = n - 3 - es;
        return 1 + (fr > 0 ? fr : 0);
    }
    return f.mbits - 1;                                        // mbits counts sign + implicit one (formats.py:65-129)
}

int msq_packed_kinds_layout(int inlier_fmt, int outlier_fmt, int layout, int* in_kind, int* out_kind) {
    if (layout == MSQ_LAYOUT_PLANES) return msq_packed_kinds(inlier_fmt, outlier_fmt, in_kind, out_kind);
    if (layout != MSQ_LAYOUT_UNIFIED) return fail2(MSQ_ERR_BAD_ARG, "msq_packed_kinds_layout: bad layout");
    msq_host::FmtInfo fi, fo;
    if (!msq_host::format_info(inlier_fmt, &fi) || !msq_host::format_info(outlier_fmt, &fo))
        return fail2(MSQ_ERR_BAD_ARG, "msq_packed_kinds_layout: unknown element format");
    const int si = sig_bits(fi, inlier_fmt), so = sig_bits(fo, outlier_fmt);
    const int sg = si > so ? si : so;
    if (fi.kind != 0 || sg > 5)
        return fail2(MSQ_ERR_UNSUPPORTED, "msq_packed_kinds_layout: the unified layout holds float/int inliers and values of at most 5 significant bits");
    if (in_kind) *in_kind = MSQ_PLANE_NONE;
    if (out_kind) *out_kind = (sg <= 4) ? MSQ_PLANE_U8 : MSQ_PLANE_U8X;
    return MSQ_OK;
}

int msq_packed_sizes(int64_t N, int64_t K, int block, int in_kind, int out_kind, int64_t* inl_bytes,
                     int64_t* out_bytes, int64_t* scale_bytes, int64_t* workspace_bytes) {
    if (N <= 0 || K <= 0 || (N % TILE_N) || (K % TILE_K))
        return fail2(MSQ_ERR_UNSUPPORTED, "msq_packed_sizes: N and K must be positive multiples of 64");
    if (!(block == 8 || block == 16 || block == 32 || block == 64 || block == 128) || (K % block))
        return fail2(MSQ_ERR_UNSUPPORTED, "msq_packed_sizes: block must be 8/16/32/64/128 and divide K");
    const int64_t tiles = (N / TILE_N) * (K / TILE_K);
    const int groups = block < 32 ? 64 : 16;
    if (out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X) {
        if (in_kind != MSQ_PLANE_NONE) return fail2(MSQ_ERR_BAD_ARG, "msq_packed_sizes: unified planes have no inlier plane");
        if (inl_bytes) *inl_bytes = (out_kind == MSQ_PLANE_U8X) ? tiles * 2 * 256 : 0;      // extension bits
        if (out_bytes) *out_bytes = tiles * 4 * 1024;
        if (scale_bytes) *scale_bytes = tiles * 16 * 8;
        if (workspace_bytes) *workspace_bytes = 0;
        return MSQ_OK;
    }
    if (inl_bytes) *inl_bytes = (in_kind == MSQ_PLANE_NONE) ? 0 : tiles * 2 * 1024;
    if (out_bytes) *out_bytes = tiles * ((out_kind == MSQ_PLANE_BF16) ? 8 : 4) * 1024;
    if (scale_bytes) *scale_bytes = (in_kind == MSQ_PLANE_NONE) ? 0 : tiles * groups * 16;
    // two-pass path: u32 codes + block exponents (+ the [N, block] mean/std tables of variant 1)
    if (workspace_bytes) *workspace_bytes = N * K * 4 + 2 * N * (K / block) * 4 + 2 * N * block * 4;
    return MSQ_OK;
}

// implemented in msq_pack_emit.hip (heavy template instantiations, own TU)
int msq_pack_fused_(const float* W, void* inl_plane, void* out_plane, void* scale_plane, int* status, int64_t N,
                    int64_t K, int block, int inlier_fmt, int outlier_fmt, int in_sb, int out_sb, float std_dev, int rmode,
                    int flush, int in_kind, int out_kind, void* stream);
int msq_pack_emit_(const float* W, uint32_t* codes, float* e_in, float* e_out, int* status, int64_t N, int64_t K,
                   int block, int inlier_fmt, int outlier_fmt, int in_sb, int out_sb, float std_dev, int rmode,
                   int flush, int in_kind, int out_kind, int variant, const float* vmean, const float* vstd, void* stream);
// implemented in msq_quant.hip: mean / unbiased std over the block-count axis (mx_ops.py:62-66,248)
int msq_mxops_stats_(const float* in, float* vmean, float* vstd, int64_t pre, int64_t axis_len, int64_t post, int block,
                     int* status, void* stream);

int msq_outlier_pack(const float* W, void* inl_plane, void* out_plane, void* scale_plane, int* status_flag,
                     void* workspace, int64_t workspace_bytes, int64_t N, int64_t K, int block, int inlier_fmt,
                     int outlier_fmt, int inlier_scale_bits, int outlier_scale_bits, float std_dev, int rmode,
                     int flush_fp32_subnorms, int variant, int layout, void* stream) {
    if (variant != MSQ_VARIANT_QUANT && variant != MSQ_VARIANT_MXOPS) return fail2(MSQ_ERR_BAD_ARG, "msq_outlier_pack: bad variant");
    int ik, ok;
    int rc = msq_packed_kinds_layout(inlier_fmt, outlier_fmt, layout, &ik, &ok);
    if (rc) return rc;
    int64_t ib, ob, sb, wb;
    rc = msq_packed_sizes(N, K, block, ik, ok, &ib, &ob, &sb, &wb);
    if (rc) return rc;
    if (!W || !out_plane || (ik != MSQ_PLANE_NONE && (!inl_plane || !scale_plane)))
        return fail2(MSQ_ERR_BAD_ARG, "msq_outlier_pack: null buffer");
    if (layout == MSQ_LAYOUT_UNIFIED) {
        if (!scale_plane || (ok == MSQ_PLANE_U8X && !inl_plane)) return fail2(MSQ_ERR_BAD_ARG, "msq_outlier_pack: null buffer");
        if (variant != MSQ_VARIANT_QUANT) return fail2(MSQ_ERR_UNSUPPORTED, "msq_outlier_pack: the unified layout packs the canonical quantiser only");
        rc = msq_pack_fused_(W, inl_plane, out_plane, scale_plane, status_flag, N, K, block, inlier_fmt, outlier_fmt,
                             inlier_scale_bits, outlier_scale_bits, std_dev, rmode, flush_fp32_subnorms, ik, ok, stream);
        if (rc == MSQ_ERR_UNSUPPORTED) return fail2(rc, "msq_outlier_pack: the unified layout needs nearest rounding and block <= 64");
        return rc;
    }
    // single-pass kernel for the common configurations (float/int inliers, nearest rounding, block <= 64) ...
    if (variant == MSQ_VARIANT_QUANT && !pack_two_pass_forced()) {
        rc = msq_pack_fused_(W, inl_plane, out_plane, scale_plane, status_flag, N, K, block, inlier_fmt, outlier_fmt,
                             inlier_scale_bits, outlier_scale_bits, std_dev, rmode, flush_fp32_subnorms, ik, ok, stream);
        if (rc != MSQ_ERR_UNSUPPORTED) return rc;
    }
    // ... everything else: emit u32 codes + exponents into the workspace, then repack
    if (!workspace || workspace_bytes < wb) return fail2(MSQ_ERR_BAD_ARG, "msq_outlier_pack: workspace too small (msq_packed_sizes)");
    uint32_t* codes = (uint32_t*)workspace;
    float* e_in = (float*)(codes + N * K);
    float* e_out = e_in + N * (K / block);
    float *vmean = nullptr, *vstd = nullptr;
    if (variant == MSQ_VARIANT_MXOPS) {
        vmean = e_out + N * (K / block); vstd = vmean + N * block;
        rc = msq_mxops_stats_(W, vmean, vstd, N, K, 1, block, status_flag, stream);
        if (rc) return rc;
    }
    rc = msq_pack_emit_(W, codes, e_in, e_out, status_flag, N, K, block, inlier_fmt, outlier_fmt, inlier_scale_bits,
                        outlier_scale_bits, std_dev, rmode, flush_fp32_subnorms, ik, ok, variant, vmean, vstd, stream);
    if (rc) return rc;
    const int64_t tiles = (N / TILE_N) * (K / TILE_K);
    const dim3 grid((unsigned)((tiles + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define MSQ_RP(IK, OK) hipLaunchKernelGGL((k_repack<IK, OK>), grid, blk, 0, st, codes, e_in, e_out, (uint8_t*)inl_plane, \
                                          (uint8_t*)out_plane, (uint8_t*)scale_plane, N, K, block, status_flag)
    if (ik == MSQ_PLANE_NONE) MSQ_RP(MSQ_PLANE_NONE, MSQ_PLANE_BF16);
    else if (ok == MSQ_PLANE_FP8) MSQ_RP(MSQ_PLANE_FP4, MSQ_PLANE_FP8);
    else if (ok == MSQ_PLANE_BF8) MSQ_RP(MSQ_PLANE_FP4, MSQ_PLANE_BF8);
    else MSQ_RP(MSQ_PLANE_FP4, MSQ_PLANE_BF16);
#undef MSQ_RP
    return check_launch2("msq_outlier_pack(repack)");
}

// implemented in msq_pack_emit.hip
int msq_pack_values_u_(const float* W, void* ext_plane, void* code_plane, void* scale_plane, int* status, int64_t N,
                       int64_t K, int out_kind, void* stream);

int msq_pack_values(const float* Wq, void* inl_plane, void* out_plane, void* scale_plane, int* status_flag, int64_t N,
                    int64_t K, int in_kind, int out_kind, void* stream) {
    int rc = msq_packed_sizes(N, K, 32, in_kind, out_kind, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (in_kind != MSQ_PLANE_NONE) return fail2(MSQ_ERR_UNSUPPORTED, "msq_pack_values: only single-plane kinds (U8, U8X, BF16 with in_kind NONE)");
    if (!Wq || !out_plane) return fail2(MSQ_ERR_BAD_ARG, "msq_pack_values: null buffer");
    if (out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X) {
        if (!scale_plane || (out_kind == MSQ_PLANE_U8X && !inl_plane)) return fail2(MSQ_ERR_BAD_ARG, "msq_pack_values: null buffer");
        return msq_pack_values_u_(Wq, inl_plane, out_plane, scale_plane, status_flag, N, K, out_kind, stream);
    }
    if (out_kind != MSQ_PLANE_BF16) return fail2(MSQ_ERR_UNSUPPORTED, "msq_pack_values: unsupported plane kind");
    const int64_t tiles = (N / TILE_N) * (K / TILE_K);
    const dim3 grid((unsigned)((tiles + 3) / 4)), blk(256);
    hipLaunchKernelGGL(k_pack_values_bf16, grid, blk, 0, (hipStream_t)stream, Wq, (uint8_t*)out_plane, N, K, status_flag);
    return check_launch2("msq_pack_values");
}

int msq_outlier_unpack(const void* inl_plane, const void* out_plane, const void* scale_plane, void* W_out,
                       int out_dtype, int64_t N, int64_t K, int block, int in_kind, int out_kind, void* stream) {
    int rc = msq_packed_sizes(N, K, block, in_kind, out_kind, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (!out_plane || !W_out || (in_kind != MSQ_PLANE_NONE && (!inl_plane || !scale_plane)))
        return fail2(MSQ_ERR_BAD_ARG, "msq_outlier_unpack: null buffer");
    if ((out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X) && (!scale_plane || (out_kind == MSQ_PLANE_U8X && !inl_plane)))
        return fail2(MSQ_ERR_BAD_ARG, "msq_outlier_unpack: null buffer");
    if (out_dtype != 0 && out_dtype != 2) return fail2(MSQ_ERR_UNSUPPORTED, "msq_outlier_unpack: out_dtype must be 0 (f32) or 2 (bf16)");
    const int64_t tiles = (N / TILE_N) * (K / TILE_K);
    const dim3 grid((unsigned)((tiles + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    const int groups = block < 32 ? 64 : 16;
#define MSQ_UP(IK, OK)                                                                                               \
    do { if (out_dtype == 0) hipLaunchKernelGGL((k_unpack<IK, OK, float>), grid, blk, 0, st, (const uint8_t*)inl_plane,  \
                    (const uint8_t*)out_plane, (const uint8_t*)scale_plane, (float*)W_out, N, K, groups);              \
         else hipLaunchKernelGGL((k_unpack<IK, OK, uint16_t>), grid, blk, 0, st, (const uint8_t*)inl_plane,             \
                    (const uint8_t*)out_plane, (const uint8_t*)scale_plane, (uint16_t*)W_out, N, K, groups); } while (0)
    if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_BF16) MSQ_UP(MSQ_PLANE_NONE, MSQ_PLANE_BF16);
    else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_FP8) MSQ_UP(MSQ_PLANE_FP4, MSQ_PLANE_FP8);
    else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_BF8) MSQ_UP(MSQ_PLANE_FP4, MSQ_PLANE_BF8);
    else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_BF16) MSQ_UP(MSQ_PLANE_FP4, MSQ_PLANE_BF16);
    else if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_U8) MSQ_UP(MSQ_PLANE_NONE, MSQ_PLANE_U8);
    else if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_U8X) MSQ_UP(MSQ_PLANE_NONE, MSQ_PLANE_U8X);
    else return fail2(MSQ_ERR_UNSUPPORTED, "msq_outlier_unpack: unsupported plane kinds");
#undef MSQ_UP
    return check_launch2("msq_outlier_unpack");
}

// Block shape: 128 x 256 (4 waves, two blocks per CU) or 256 x 256 (8 waves, one block per CU).  Both keep
// the same per-wave tile, so the instruction mix is identical.  Measured over the Llama shapes
// (scripts/experiments/wm_sweep.py) the small block wins or ties nearly everywhere (finer tail, half the wasted
// activation rows, two independent barriers per CU), so it is the default; MSQ_GEMM_WM=2 forces the
// large one for experiments.
static int pick_wm(int64_t M, int64_t N) {
    static const int forced = [] { const char* e = getenv("MSQ_GEMM_WM"); return e ? atoi(e) : 0; }();
    (void)M; (void)N;
    return forced == 2 ? 2 : 1;
}

// split-K factor: enough blocks to fill the 256 CUs when M is small (decode / short prefill)
// Split-K for grids that do not fill the chip (MX GEMM): cost model fitted to HIP-graph replays at M = 65 ... 512 on
// the four Llama-7B shapes (scripts/experiments/mx_ks_sweep.py).  A block spends `tau` us per K-step, all
// blocks run concurrently up to 512 (two per CU), the partial planes cost ks M N 8 bytes (fp32 written + read back) at
// ~8 TB/s while they stay cache-resident (<= 32 MB), ~6.5 TB/s up to 128 MB, ~3.6 TB/s beyond; one split writes Y
// directly.  Power-of-two splits with at least `min_steps` K-steps per chunk.
static int splitk_by_cost(int64_t blocks, int64_t KT, int64_t M, int64_t N, double tau, int max_ks, int min_steps) {
    int best = 1;
    double best_t = 1e30;
    for (int ks = 1; ks <= max_ks && (int64_t)ks * min_steps <= KT; ks *= 2) {
        const double rounds = (double)((blocks * ks + 511) / 512);
        const double pbytes = (double)ks * (double)M * (double)N * 4.0;
        const double bw = pbytes <= 32e6 ? 8.0e6 : (pbytes <= 128e6 ? 6.5e6 : 3.6e6);       // bytes per us
        const double t = tau * (double)((KT + ks - 1) / ks) * rounds + (ks > 1 ? 2.0 * pbytes / bw : 0.0);
        if (t < best_t) { best_t = t; best = ks; }
    }
    return best;
}
// wave-tile height (MF x 16 rows): 16 = block 256 x 256 with the accumulators in AGPRs, one block per CU; taken when
// the grid still has at least two such blocks per CU pair... (measured: see DESIGN.md); MSQ_GEMM_MF forces 8 / 16.
static int pick_mf(int64_t M, int64_t N, int out_kind) {
    static const int forced = [] { const char* e = getenv("MSQ_GEMM_MF"); return e ? atoi(e) : 0; }();
    if (forced == 8 || forced == 16) return (out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X) ? forced : 8;
    (void)M; (void)N;
    return 8;
}
static int pick_ksplit(int64_t M, int64_t N, int64_t K, int mf = 8) {
    static const int forced = [] { const char* e = getenv("MSQ_GEMM_KS"); return e ? atoi(e) : 0; }();
    const int wm = pick_wm(M, N);
    if (mf == 16) return 1;
    const int64_t blocks = ((M + 128 * wm - 1) / (128 * wm)) * (N / BN);
    const int64_t KT = K / BK;
    if (forced > 0) return forced < KT ? forced : (int)KT;
    if (blocks >= 192 || KT < 8) return 1;
    // cost model fitted to HIP-graph replays of the four Llama-7B shapes at M = 65 ... 512 (scripts/experiments/
    // ks_sweep_graph.py, round 2): a block spends ~0.62 us per K-step (the 128 x 256 tile is MFMA-bound per step
    // whatever M is), 256 blocks run at a time, and the fp32 partial planes cost ~0.4 us per MB (written by the GEMM's
    // epilogue, read back by k_splitk_reduce).  It picks the measured best split in all 16 sweep cells; the earlier
    // "fill 256 blocks" rule over-split short K (4096 x 4096 M 65: 18.9 -> 17.1 us, M 256: 24.9 -> 23.1) and
    // under-used one pass for wide N (11008 x 4096 M 512: 58.2 -> 50.2 us).
    int best = 1;
    double best_t = 1e30;
    for (int ks = 1; ks <= 32 && ks * 2 <= KT; ks *= 2) {
        const double rounds = (double)((blocks * ks + 255) / 256);
        const double mb = (ks > 1) ? (double)ks * (double)M * (double)N * 4.0e-6 : 0.0;
        const double t = 0.62 * (double)((KT + ks - 1) / ks) * rounds + 0.4 * mb;
        if (t < best_t) { best_t = t; best = ks; }
    }
    return best;
}

// Decode kernel or (split-K) GEMM.  Measured from HIP graphs (scripts/experiments/mx_midm_check.py): with the LDS
// reduction of four k-chunks the decode kernel wins for every Llama-7B shape up to M = 32 (two 16-row groups per wave);
// with four groups (M <= 64) every wave re-reads 64 activation rows per tile from L2 and the partial planes grow with
// M N: it still wins for the 4096 x 4096 projections (14.6 vs 19.3 us), not for N = 16384 or K = 11008.
// MSQ_GEMV_MAX_M (tuning only) replaces the rule by M <= value.
static bool use_gemv(int64_t M, int64_t N, int64_t K) {
    static int v = -2;
    if (v == -2) { const char* e = getenv("MSQ_GEMV_MAX_M"); v = e ? atoi(e) : -1; if (v > 64) v = 64; }
    if (v >= 0) return M <= v;
    // four row groups through the wide-projection kernel (k_qgemv_u) while re-reading the activation rows per tile still costs less
    // than the GEMM's 128-row tiles waste (cold weights, q/k/v 12288 x 4096: M = 33 / 48 / 64 21.0 / 23.2 / 26.1 us against 26.8 / 27.4 /
    // 27.7; gate / up 22016 x 4096: 38.9 / 40.9 / 45.5 against 42.9 / 40.5 / 41.5)
    const int64_t strips = N / TILE_N;
    return M <= 32 || (M <= 64 && N <= 4096 && K <= 4096) || (M <= 48 && strips > 128 && strips <= 256) || (M <= 36 && strips > 256);
}
// single-launch decode (unified layouts): as mx_direct_kc; 64-k tiles: K <= 4096 = at most 4 tiles per wave.
// MSQ_GEMV_DIRECT=0 (tuning only) disables it.
static int direct_kc(int64_t M, int64_t N, int64_t K) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MSQ_GEMV_DIRECT"); on = e ? atoi(e) : 1; }
    if (!on || M > 32 || N < 8192) return 0;
    const int kc = (int)((K / BK + 15) / 16);
    return kc <= 4 ? kc : 0;
}
static int pick_kc(int64_t N, int64_t K) {
    const int64_t KT = K / BK, strips = N / TILE_N;
    static const int tw = [] { const char* e = getenv("MSQ_GEMV_TARGET_WAVES"); return e ? atoi(e) : 3072; }();
    int64_t kc = (strips * KT + tw - 1) / tw;
    if (kc < 1) kc = 1;
    while ((KT + kc - 1) / kc > 32) ++kc;
    return (int)kc;
}

static bool qp_rule(int64_t M, int64_t N, int64_t K, int64_t* ws_bytes);
int64_t msq_qlinear_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || (N % BN) || (K % BK)) return 0;
    if (use_gemv(M, N, K)) { const int kc = pick_kc(N, K); return (((K / BK + kc - 1) / kc + 3) / 4) * M * N * 4; }   // (k_qgemm_sk needs none; sized for the rule's fall-backs all the same)
    const int ks = pick_ksplit(M, N, K);
    int64_t w = ks > 1 ? (int64_t)ks * M * N * 4 : 0, wp = 0;
    if (qp_rule(M, N, K, &wp) && wp > w) w = wp;                 // flag words + partial-tile slots of the persistent kernel's stream-K round
    return w;
}

#ifndef MSQ_MX128_DEFAULT
#define MSQ_MX128_DEFAULT 1    /* 1: the MF = 8 form of k_mxgemm256 where the cost rule prefers it */
#endif
#ifndef MSQ_MX256_DEFAULT
#define MSQ_MX256_DEFAULT 1    /* 1: k_mxgemm256 is the default for full grids of 256 x 256 blocks (set once measured faster) */
#endif
int msq_launch_mxgemm256(int wf, const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales, const float* bias,
                         void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int mf, void* stream);   // msq_mxgemm256.hip
#ifndef MSQ_Q128_DEFAULT
#define MSQ_Q128_DEFAULT 1     /* 1: the MF = 8 form of k_qgemm256 (128-row blocks, two per CU) where the cost rule below prefers it */
#endif
#ifndef MSQ_Q256_DEFAULT
#define MSQ_Q256_DEFAULT 1     /* 1: k_qgemm256 is the default for full grids of 256 x 256 blocks (set once measured faster) */
#endif
int msq_launch_qgemm256(const void* X, const void* ext_plane, const void* code_plane, const void* scale_plane, const float* bias, void* Y,
                        int y_dtype, int64_t M, int64_t N, int64_t K, int out_kind, int scl_groups, int mf, void* stream);   // msq_gemm256.hip
// the persistent stream-K form (msq_gemm256p.hip): plan = host arithmetic of the schedule (0 = applies) and the workspace it needs
int msq_qgemm256p_plan(int64_t M, int64_t N, int64_t K, int cus, int* P, int* full, int* R, int* q, int64_t* ws_bytes);
int msq_launch_qgemm256p(const void* X, const void* ext_plane, const void* code_plane, const void* scale_plane, const float* bias, void* Y,
                         int y_dtype, int64_t M, int64_t N, int64_t K, int out_kind, int scl_groups, void* workspace, int64_t workspace_bytes, void* stream);   // -12345: plan of THIS device does not fit: fall back
#ifndef MSQ_QP_DEFAULT
#define MSQ_QP_DEFAULT 0       /* 1: k_qgemm256p where qp_rule() prefers it */
#endif
// MSQ_GEMM_256 (tuning and A / B, read per call): 0 = k_qgemm3 only, 1 / 2 = force the 256- / 128-row form of k_qgemm256, 3 = force the
// persistent kernel wherever its plan applies, unset = the rules
// Both switches have a thread-safe form (advisor, round 4: getenv beside another thread's setenv is undefined behaviour): a value set
// through msq_set_tuning() is held in an atomic and wins; only while none is set is the environment consulted, per call (single-threaded
// tests and A / B scripts flip it inside one process).
#include <atomic>
#include <limits.h>
static std::atomic<int> g_tune_gemm256{INT_MIN}, g_tune_mx256{INT_MIN}, g_tune_sk{INT_MIN};
// MSQ_GEMM_SK (tuning and A / B, read per call like MSQ_GEMM_256): 0 = never k_qgemm_sk, 1 / 2 / 3 = force its form (msq_gemm_stream.hip:
// 64-row / 128-row strips, 128 x 128 blocks) wherever the kernel applies, unset = sk_rule()
static int sk_forced_env() {
    const int t = g_tune_sk.load(std::memory_order_relaxed);
    if (t != INT_MIN) return t;
    const char* e = getenv("MSQ_GEMM_SK");
    return e ? atoi(e) : -1;
}
static int q256_forced_env() {
    const int t = g_tune_gemm256.load(std::memory_order_relaxed);
    if (t != INT_MIN) return t;
    const char* e = getenv("MSQ_GEMM_256");
    return e ? atoi(e) : -1;
}
extern "C" void msq_set_tuning_lowp_(const char* key, int value);
extern "C" int msq_set_tuning_act_(const char* key, int value);
extern "C" int msq_set_tuning_mx_(const char* key, int value);
extern "C" int msq_set_tuning_vec_(const char* key, int value);
extern "C" int msq_set_tuning(const char* key, int value) {
    if (!key) return MSQ_ERR_BAD_ARG;
    if (!strcmp(key, "MSQ_GEMM_256")) { g_tune_gemm256.store(value, std::memory_order_relaxed); return MSQ_OK; }
    if (!strcmp(key, "MSQ_MX_256")) { g_tune_mx256.store(value, std::memory_order_relaxed); return MSQ_OK; }
    if (!strcmp(key, "MSQ_GEMM_SK")) { g_tune_sk.store(value, std::memory_order_relaxed); return MSQ_OK; }
    if (!strcmp(key, "MSQ_MX_LOWP_PAIR4")) { msq_set_tuning_lowp_("mx_lowp_pair4", value); return MSQ_OK; }   // 0: one lane per block pair (k_mx_lowp_pair)
    if (!strcmp(key, "MSQ_OUTLIER_LOWP_PK")) { msq_set_tuning_lowp_("outlier_lowp_pk", value); return MSQ_OK; }   // 0: the op-by-op in-dtype fake-quant kernel only
    if (!strcmp(key, "MSQ_PACK_TWO_PASS")) { g_tune_two_pass.store(value, std::memory_order_relaxed); return MSQ_OK; }
    if (msq_set_tuning_act_(key, value) || msq_set_tuning_mx_(key, value) || msq_set_tuning_vec_(key, value)) return MSQ_OK;   // MSQ_ACT_ROWS, MSQ_MX_PACK_BLOCK, MSQ_VEC_GENERIC
    return MSQ_ERR_UNSUPPORTED;
}
// k_qgemm_sk (msq_gemm_stream.hip): K cut over the waves INSIDE a block, partial tiles summed in LDS -- form 1 / 2 / 3, 0 = not for this shape
extern "C" int msq_qgemm_sk_form(int64_t M, int64_t N, int64_t K, int form);
extern "C" int msq_launch_qgemm_sk(const void* X, const void* ext_plane, const void* code_plane, const void* scale_plane, const float* bias, void* Y,
                                   int y_dtype, int64_t M, int64_t N, int64_t K, int out_kind, int scl_groups, int form, void* stream);
// The rule (also behind msq_qlinear_kernel_choice), from profiles/r06_midm_forms.txt (HIP-graph device times, posit / fp8 outliers, the Llama-2-7B
// projections, 16384 x 4096 and 8192 x 8192).  What bounds the kernel is the ~70 GB/s at which ONE CU pulls bytes out of L2 (MI355X_MICROARCH.md,
// "Indexed rows" / "ring-gemm"): a block re-reads its activation rows over all of K next to its strip of packed weights -- (2 rows + 74) K bytes --
// so it wins where that is less than what split-K planes cost: 64-row blocks (form 1) up to M = 64 on every projection with K <= 8192 (48-64 rows:
// 12288 x 4096 24.2 -> 17.8 us, 16384 x 4096 28.9 -> 22.0, 22016 x 4096 36.6 -> 31.9, 4096 x 4096 16.5 -> 14.2, 8192 x 8192 26.8 -> 25.2; 4096 x 11008
// loses: 22.1 -> 27.9) and, as two row blocks per strip, up to M = 128 on the 4096 x 4096 projections (21.5 -> 14.4).  The 128-row forms (2, 4)
// and the 128 / 64 x 128 blocks (3, 5) tie with or lose to the split-K GEMM once their activation ring is filled in an architecturally
// ordered way: forced forms only.
// Round 6, late: form 6 (128 x 128 blocks, eight waves) on one-round grids of 129-256 rows with posit outliers: 16384 x 4096 44.2 -> 40.1 us,
// 12288 x 4096 40.2 -> 36.7 (fp8 outliers tie: the split-K GEMM keeps them); it loses on two-round grids (22016, 8192 x 8192) and short N.
static int sk_rule(int64_t M, int64_t N, int64_t K, bool unified_bf16x, int out_kind = MSQ_PLANE_U8) {
    if (!unified_bf16x) return 0;
    const int forced = sk_forced_env();
    if (forced == 0) return 0;
    if (forced > 0) return msq_qgemm_sk_form(M, N, K, forced);
    if (M <= 32 || M > 256) return 0;
    const int64_t strips = N / TILE_N, KT = K / BK;
    int form = 0;
    if (M <= 64) form = (strips >= 64 && strips <= 352 && KT <= 128) ? 1 : 0;
    else if (M <= 128) form = (strips >= 64 && strips <= 128 && KT <= 64) ? 1 : 0;
    else if (out_kind == MSQ_PLANE_U8X && KT <= 64 && strips >= 192 && strips <= 288) form = 6;     // 2 row blocks x strips / 2 = 192 ... 288 blocks
    return form ? msq_qgemm_sk_form(M, N, K, form) : 0;
}
// persistent kernel for this shape?  (M > 64: the decode kernels come first)
static bool qp_rule(int64_t M, int64_t N, int64_t K, int64_t* ws_bytes) {
    int64_t wsb = 0;
    if (ws_bytes) *ws_bytes = 0;
    const int forced = q256_forced_env();
    if (forced >= 0 && forced != 3) return false;
    if (forced < 0 && !MSQ_QP_DEFAULT) return false;
    if (msq_qgemm256p_plan(M, N, K, 0, nullptr, nullptr, nullptr, nullptr, &wsb)) return false;
    const int64_t b256 = ((M + 255) / 256) * (N / 256);
    if (forced < 0 && b256 < 96) return false;                  // small grids: k_qgemm3's 64-row blocks and split-K
    if (ws_bytes) *ws_bytes = wsb;
    return true;
}
// The rule itself (also behind msq_qlinear_kernel_choice): wave-tile height of the hand-allocated kernel for a prefill-size grid, 0 = k_qgemm3.
static int q256_rule(int64_t M, int64_t N, int out_kind) {
    const int64_t b256 = ((M + 255) / 256) * (N / 256), b128 = ((M + 127) / 128) * (N / 256);
    const int64_t r16 = (b256 + 255) / 256, r8x2 = (b128 + 255) / 256;
    const bool posit_out = out_kind == MSQ_PLANE_U8X;
    if (MSQ_Q128_DEFAULT && b128 >= 144 && r8x2 * (posit_out ? 107 : 100) < 2 * r16 * 100) return 8;
    if (MSQ_Q256_DEFAULT && (b256 >= 224 || (b256 > 128 && (posit_out || M <= 512)))) return 16;
    return 0;
}
static int mx256_rule(int64_t M, int64_t N, int wf) {
    const int64_t b256 = ((M + 255) / 256) * (N / 256), b128 = ((M + 127) / 128) * (N / 256);
    const int64_t r16 = (b256 + 255) / 256, r8x2 = (b128 + 255) / 256;
    if (MSQ_MX128_DEFAULT && wf == 0 && b128 >= 144 && r16 >= 2 && r8x2 < 2 * r16) return 8;
    if (MSQ_MX256_DEFAULT && b256 > 128) return 16;
    return 0;
}
// The decision of the prefill-size dispatch in ONE place (dispatcher, msq_qlinear_kernel_choice, msq_qlinear_kernel_name): mf = 16 / 8 =
// k_qgemm256 in its 256- / 128-row form, persistent = k_qgemm256p, neither = k_qgemm3.  Default rule: a cost in rounds of the grid over
// the 256 CUs -- 256-row blocks run one per CU (r16 rounds); 128-row blocks two per CU, each pair about as long as one 256-row block
// (r8x2 half-rounds, times 1.07 with posit outliers: their longer convert chain hides less well behind half the MFMAs); the 128-row form
// wins where its finer granularity saves at least that (q/k/v 2048 x 12288: 1.5 against 2 rounds) and on the one-round grids of 144 ...
// 256 blocks (o, down); k_qgemm3 keeps the small grids (64-row blocks, split-K).  `ws_bytes` = bytes of workspace the call may use.
struct QFamily { int mf; bool persistent; };
static QFamily q_family(int64_t M, int64_t N, int64_t K, bool unified_bf16x, int out_kind, int64_t ws_bytes) {
    QFamily f = {0, false};
    if (!unified_bf16x) return f;
    const int forced = q256_forced_env();
    int64_t qp_ws = 0;
    if (qp_rule(M, N, K, &qp_ws) && ws_bytes >= qp_ws) { f.persistent = true; return f; }
    const int mf_rule = q256_rule(M, N, out_kind);
    if (forced == 2 || (forced < 0 && mf_rule == 8)) f.mf = 8;
    else if (forced == 1 || forced == 3 || (forced < 0 && mf_rule == 16)) f.mf = 16;   // (3 without the persistent kernel's workspace: the 256-row form)
    return f;
}
// k_mxgemm256 form for a shape (16 / 8 / 0 = k_mxgemm): mx256_rule and the MSQ_MX_256 switch (1 / 2 force the 256- / 128-row form, 0
// disables both; read per call -- tests and A / B scripts flip it inside one process; not for concurrent use with setenv)
static int mx256_family(int64_t M, int64_t N, int wf) {
    const int t = g_tune_mx256.load(std::memory_order_relaxed);
    const char* e256 = (t == INT_MIN) ? getenv("MSQ_MX_256") : nullptr;
    const int forced = (t != INT_MIN) ? t : (e256 ? atoi(e256) : -1);
    const int mf_rule = mx256_rule(M, N, wf);
    return (forced == 2 || (forced < 0 && mf_rule == 8)) ? 8 : ((forced == 1 || (forced != 0 && mf_rule == 16)) ? 16 : 0);
}
static int qlinear_bf16_impl(const void* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                     const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int block,
                     int in_kind, int out_kind, void* workspace, int64_t workspace_bytes, void* stream, int x_f16) {
    if (M <= 0) return (M == 0) ? MSQ_OK : fail2(MSQ_ERR_BAD_ARG, "msq_qlinear_bf16: negative M");
    int rc = msq_packed_sizes(N, K, block, in_kind, out_kind, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (N % BN) return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_bf16: N must be a multiple of 256");
    if (!X || !Y || !out_plane || (in_kind != MSQ_PLANE_NONE && (!inl_plane || !scale_plane)))
        return fail2(MSQ_ERR_BAD_ARG, "msq_qlinear_bf16: null buffer");
    const bool unified = (out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X);
    if (unified && (!scale_plane || (out_kind == MSQ_PLANE_U8X && !inl_plane)))
        return fail2(MSQ_ERR_BAD_ARG, "msq_qlinear_bf16: null buffer");
    if (y_dtype != 0 && y_dtype != 1 && y_dtype != 2) return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_bf16: y_dtype must be 0 (f32), 1 (fp16) or 2 (bf16)");
    if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_bf16: dimension too large");
    {   // activations and packed planes are addressed with 32-bit buffer offsets (make_rsrc clamps num_records)
        int64_t ib = 0, ob = 0, sb = 0;
        msq_packed_sizes(N, K, block, in_kind, out_kind, &ib, &ob, &sb, nullptr);
        if (M * K * 2 > 0xFFFFFFFFll || ib > 0xFFFFFFFFll || ob > 0xFFFFFFFFll || sb > 0xFFFFFFFFll)
            return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_bf16: activations (M*K*2 bytes) and every packed plane must stay below 4 GiB; split M (or N) on the host");
    }
    hipStream_t st0 = (hipStream_t)stream;
    const int y16 = (y_dtype == 1) ? 1 : 0;                       // fp16 output: the 16-bit kernels with the half conversion
    const int groups0 = unified ? 16 : (block < 32 ? 64 : 16);
    // fp16 activations are converted inside the decode kernels only: where the bf16 call takes k_qgemm_sk the caller casts (one kernel, one
    // summation order for both activation dtypes: tests/test_gpu_n2_gemm.py::test_decode_kernels_take_fp16_activations)
    if (x_f16 && sk_rule(M, N, K, unified, out_kind))
        return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_f16x: this shape runs k_qgemm_sk, which reads bf16 activations: cast them to bf16");
    if (const int skf = sk_rule(M, N, K, unified && !x_f16, out_kind)) {
        const int e = msq_launch_qgemm_sk(X, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, out_kind, groups0, skf, stream);
        if (e) { char b[200]; snprintf(b, sizeof(b), "msq_qlinear_bf16(K cut inside the block, k_qgemm_sk form %d): %s", skf, hipGetErrorString((hipError_t)e)); return fail2(MSQ_ERR_LAUNCH, b); }
        return MSQ_OK;
    }
    if (use_gemv(M, N, K)) {
        const int mg = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
        const int kcd = unified ? direct_kc(M, N, K) : 0;                // > 0: one block of 16 waves covers all of K
        const int kc = kcd ? kcd : pick_kc(N, K);
        const int nks = kcd ? 1 : (int)(((K / BK + kc - 1) / kc + 3) / 4);    // partial planes: one per four k-chunks
        // unified layouts, M <= 32, more than 128 strips: the single-launch kernel (MSQ_GEMV_U=0: the earlier kernels, tuning only)
        static const int gvu = [] { const char* e = getenv("MSQ_GEMV_U"); return e ? atoi(e) : 1; }();
        if (gvu && N / TILE_N > 128 && in_kind == MSQ_PLANE_NONE && (out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X)) {
            const int64_t strips = N / TILE_N, KTv = K / TILE_K;
            // Waves per block (= k-runs per strip) so that the grid has ~1000-2000 waves, two tiles in flight each: a COLD read stream
            // is fastest with a few MB in flight (scripts/experiments/hbm_read.hip: 1024 waves x 4 KB reach 6 TB/s, 8192 x 8 KB only
            // 4.5 -- DRAM pages thrash); measured on the fused projections (waves x tiles in flight, us at M = 1): q/k/v 16 x 3
            // 15.8, 8 x 3 14.8, 8 x 2 13.6, 4 x 2 18.3; gate / up 8 x 3 27.5, 4 x 3 24.5, 4 x 2 23.7, 2 x 2 36.3.
            // MSQ_GEMV_U_WAVES=2 / 4 / 8 / 16 forces.
            static const int fw = [] { const char* e = getenv("MSQ_GEMV_U_WAVES"); return e ? atoi(e) : 0; }();
            const int wv = (fw == 2 || fw == 4 || fw == 8 || fw == 16) ? fw : (strips <= 256 ? 8 : (strips <= 640 ? 4 : 2));
            const int kcu = (int)((KTv + wv - 1) / wv);
            const dim3 ugrid((unsigned)strips);
            const int yk = y_dtype == 2 ? 1 : (y_dtype == 1 ? 2 : 0);
#define MSQ_GVU1(OK, MGV, WV, XP)                                                                                         \
            do { static DevOnce once_;                                                                                   \
                 const size_t l_ = (size_t)WV * MGV * 16 * 64 * 4;                                                        \
                 if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_qgemv_u<OK, MGV, WV, XP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l_); attr_done(once_); } \
                 hipLaunchKernelGGL((k_qgemv_u<OK, MGV, WV, XP, false>), ugrid, dim3(64 * WV), l_, st0, (const uint16_t*)X, (const uint8_t*)inl_plane, (const uint8_t*)out_plane, \
                                    (const uint8_t*)scale_plane, (int)M, (int)N, (int)K, kcu, bias, Y, yk, x_f16); } while (0)
            // activation prefetch: from two rows on (one row: every lane reads the same line, an L1 hit), where the registers allow
#define MSQ_GVUW(OK, WV) do { if (mg == 1) { if (M > 1) MSQ_GVU1(OK, 1, WV, true); else MSQ_GVU1(OK, 1, WV, false); } else if (mg == 2) MSQ_GVU1(OK, 2, WV, false); else MSQ_GVU1(OK, 4, WV, false); } while (0)
#define MSQ_GVU(OK) do { if (wv == 2) MSQ_GVUW(OK, 2); else if (wv == 4) MSQ_GVUW(OK, 4); else if (wv == 8) MSQ_GVUW(OK, 8); else MSQ_GVUW(OK, 16); } while (0)
            if (out_kind == MSQ_PLANE_U8) MSQ_GVU(MSQ_PLANE_U8); else MSQ_GVU(MSQ_PLANE_U8X);
#undef MSQ_GVU
#undef MSQ_GVUW
#undef MSQ_GVU1
            return check_launch2("msq_qlinear_bf16(decode, unified layouts)");
        }
        if (kcd || (workspace && workspace_bytes >= (int64_t)nks * M * N * 4)) {
            const dim3 vgrid((unsigned)((N / TILE_N) * nks));
            const size_t ldsv = (size_t)((kcd ? 16 : 4) - 1) * 16 * mg * 64 * 4;
#define MSQ_GV1(IK, OK, MGV, WV)                                                                                        \
            do { static DevOnce once_;                                                                         \
                 if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_qgemv<IK, OK, MGV, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (WV - 1) * 16 * MGV * 64 * 4); attr_done(once_); } \
                 hipLaunchKernelGGL((k_qgemv<IK, OK, MGV, WV>), vgrid, dim3(64 * WV), ldsv, st0, (const uint16_t*)X, (const uint8_t*)inl_plane, (const uint8_t*)out_plane, (const uint8_t*)scale_plane, \
                                    (float*)workspace, (int)M, (int)N, (int)K, groups0, kc, kcd ? 1 : 0, bias, Y, y_dtype == 2 ? 1 : (y_dtype == 1 ? 2 : 0), x_f16); } while (0)
#define MSQ_GV(IK, OK) do { if (mg == 1) MSQ_GV1(IK, OK, 1, 4); else if (mg == 2) MSQ_GV1(IK, OK, 2, 4); else MSQ_GV1(IK, OK, 4, 4); } while (0)
            if (kcd) {
                if (out_kind == MSQ_PLANE_U8) { if (mg == 1) MSQ_GV1(MSQ_PLANE_NONE, MSQ_PLANE_U8, 1, 16); else MSQ_GV1(MSQ_PLANE_NONE, MSQ_PLANE_U8, 2, 16); }
                else { if (mg == 1) MSQ_GV1(MSQ_PLANE_NONE, MSQ_PLANE_U8X, 1, 16); else MSQ_GV1(MSQ_PLANE_NONE, MSQ_PLANE_U8X, 2, 16); }
                return check_launch2("msq_qlinear_bf16(decode, single launch)");
            }
            if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_BF16) MSQ_GV(MSQ_PLANE_NONE, MSQ_PLANE_BF16);
            else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_FP8) MSQ_GV(MSQ_PLANE_FP4, MSQ_PLANE_FP8);
            else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_BF8) MSQ_GV(MSQ_PLANE_FP4, MSQ_PLANE_BF8);
            else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_BF16) MSQ_GV(MSQ_PLANE_FP4, MSQ_PLANE_BF16);
            else if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_U8) MSQ_GV(MSQ_PLANE_NONE, MSQ_PLANE_U8);
            else if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_U8X) MSQ_GV(MSQ_PLANE_NONE, MSQ_PLANE_U8X);
            else return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_bf16: unsupported plane kinds");
#undef MSQ_GV
#undef MSQ_GV1
            rc = check_launch2("msq_qlinear_bf16(gemv)");
            if (rc) return rc;
            const int64_t MN0 = M * N;
            const dim3 rg((unsigned)((MN0 / 4 + 255) / 256));
            if (y_dtype == 0) hipLaunchKernelGGL(k_splitk_reduce<float>, rg, dim3(256), 0, st0, (const float*)workspace, bias, (float*)Y, MN0, (int)N, nks, y_dtype == 1 ? 1 : 0);
            else hipLaunchKernelGGL(k_splitk_reduce<uint16_t>, rg, dim3(256), 0, st0, (const float*)workspace, bias, (uint16_t*)Y, MN0, (int)N, nks, y_dtype == 1 ? 1 : 0);
            return check_launch2("msq_qlinear_bf16(gemv reduce)");
        }
    }
    if (x_f16) return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_f16x: fp16 activations are converted inside the decode kernels only (M <= 32; <= 64 for the 4096 x 4096 class): cast them to bf16 for this shape");
    // 256-row wave tiles with hand-placed AGPR accumulators (k_qgemm256, msq_gemm256.hip): one wave per SIMD, half the converts and
    // packed loads per MFMA; and its 128-row form (MF = 8).
    {
        // Which hand-allocated kernel (msq_gemm256.hip / msq_gemm256p.hip), if any: q_family() -- the rule in rounds of the grid over the
        // 256 CUs (measured: profiles/r04_q128_sweep.txt, 13 values of M x the four Llama-2-7B projections x both outlier formats), the
        // MSQ_GEMM_256 switch (tuning and A / B) and the workspace the persistent kernel needs.  msq_qlinear_kernel_choice / _name report
        // the same function's answer.
        const QFamily qf = q_family(M, N, K, unified && !x_f16, out_kind, workspace ? workspace_bytes : 0);
        if (qf.persistent) {
            const int e = msq_launch_qgemm256p(X, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, out_kind, groups0, workspace, workspace ? workspace_bytes : 0, stream);
            if (e == -12345) {                                       // this device's CU count gives another plan than the rule's 256: the 256-row kernel
                const int e2 = msq_launch_qgemm256(X, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, out_kind, groups0, 16, stream);
                if (e2) { char b[200]; snprintf(b, sizeof(b), "msq_qlinear_bf16(k_qgemm256 after the persistent plan did not fit): %s", hipGetErrorString((hipError_t)e2)); return fail2(MSQ_ERR_LAUNCH, b); }
                return MSQ_OK;
            }
            if (e) { char b[200]; snprintf(b, sizeof(b), "msq_qlinear_bf16(persistent 256-row tiles, k_qgemm256p): %s", hipGetErrorString((hipError_t)e)); return fail2(MSQ_ERR_LAUNCH, b); }
            return MSQ_OK;
        }
        if (qf.mf) {
            const int e = msq_launch_qgemm256(X, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, out_kind, groups0, qf.mf, stream);
            if (e) { char b[200]; snprintf(b, sizeof(b), "msq_qlinear_bf16(%d-row wave tiles, k_qgemm256<MF = %d>): %s", 16 * qf.mf, qf.mf, hipGetErrorString((hipError_t)e)); return fail2(MSQ_ERR_LAUNCH, b); }
            return MSQ_OK;
        }
    }
    const int mf_sel = pick_mf(M, N, out_kind);
    const int wm_sel = (mf_sel == 16) ? 1 : pick_wm(M, N);
    const int brow = 16 * mf_sel * wm_sel;
    const int MT = (int)((M + brow - 1) / brow), NTB = (int)(N / BN);
    int ksplit = pick_ksplit(M, N, K, mf_sel);
    if (ksplit > 1 && (!workspace || workspace_bytes < (int64_t)ksplit * M * N * 4)) ksplit = 1;   // no scratch: one pass
    const dim3 grid((unsigned)(MT * NTB * ksplit)), blk(256 * wm_sel);
    const size_t lds = (size_t)3 * brow * BK * 2;             // three activation buffers
    hipStream_t st = (hipStream_t)stream;
    const int groups = groups0;
    float* partial = (float*)workspace;
#define MSQ_LAUNCH1(KERN, IK, OK, YT, WMV)                                                                             \
    do { static DevOnce once_;                                                                                 \
         if (attr_needed(once_)) { hipFuncSetAttribute((const void*)KERN<IK, OK, YT, WMV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done(once_); } \
         hipLaunchKernelGGL((KERN<IK, OK, YT, WMV>), grid, blk, lds, st, (const uint16_t*)X, (const uint8_t*)inl_plane,  \
                (const uint8_t*)out_plane, (const uint8_t*)scale_plane, bias, (YT*)Y, (int)M, (int)N, (int)K, groups, ksplit, partial, y16); } while (0)
#define MSQ_LAUNCH(KERN, IK, OK)                                                                                       \
    do { if (y_dtype == 0) { if (wm_sel == 2) MSQ_LAUNCH1(KERN, IK, OK, float, 2); else MSQ_LAUNCH1(KERN, IK, OK, float, 1); } \
         else { if (wm_sel == 2) MSQ_LAUNCH1(KERN, IK, OK, uint16_t, 2); else MSQ_LAUNCH1(KERN, IK, OK, uint16_t, 1); } } while (0)
#define MSQ_DISPATCH(KERN)                                                                                             \
    do { if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_BF16) MSQ_LAUNCH(KERN, MSQ_PLANE_NONE, MSQ_PLANE_BF16); \
         else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_FP8) MSQ_LAUNCH(KERN, MSQ_PLANE_FP4, MSQ_PLANE_FP8); \
         else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_BF8) MSQ_LAUNCH(KERN, MSQ_PLANE_FP4, MSQ_PLANE_BF8); \
         else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_BF16) MSQ_LAUNCH(KERN, MSQ_PLANE_FP4, MSQ_PLANE_BF16); \
         else if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_U8) MSQ_LAUNCH(KERN, MSQ_PLANE_NONE, MSQ_PLANE_U8); \
         else if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_U8X) MSQ_LAUNCH(KERN, MSQ_PLANE_NONE, MSQ_PLANE_U8X); \
         else return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_bf16: unsupported plane kinds"); } while (0)
    // Eight waves along n (one block of 128 x 512 per CU instead of two of 128 x 256): the eight waves share ONE activation
    // tile, so every wave issues half the LDS-DMA pieces and the CU pulls half the activation bytes from L2 per K-step.
    // Measured (scripts/experiments/shape_ab.py, posit / fp8 outliers): +3.5...7 % / +1...3 % whenever the grid still has
    // one block per CU (M2048 N16384: 214 -> 206 us; N8192 K28672: 755 -> 719 us), -30 % with half-empty grids.
    // MSQ_GEMM_WN=4 (tuning only) keeps the four-wave blocks.
    static const int wn_forced = [] { const char* e = getenv("MSQ_GEMM_WN"); return e ? atoi(e) : 0; }();
    // ... and only when its last round of 256 blocks is at least 85 % full: a half-empty last round of eight-wave blocks
    // leaves half of the CUs idle, while the four-wave blocks of the same tail run one per CU on all of them (fused QKV
    // N = 12288 at M = 2048, 1.5 rounds: 181.7 -> 161.0 us posit, 163.4 -> 142.7 fp8; N = 10240: 181.6 -> 148.6 us).
    const int64_t b8 = ((M + 127) / 128) * (N / 512);
    const bool wn8 = wn_forced == 8 || (wn_forced == 0 && b8 >= 256 && b8 * 100 >= 85 * 256 * ((b8 + 255) / 256));
    if (wn8 && mf_sel == 8 && unified && (N % 512) == 0 && ksplit == 1) {
        const dim3 grid8((unsigned)(((M + 127) / 128) * (N / 512))), blk8(512);
        const size_t lds8 = 65536;                                  // max(3 x 16 KiB activation buffers, 8 x 8 KiB epilogue slices)
#define MSQ_LAUNCH8(OK, YT)                                                                                            \
        do { static DevOnce once_;                                                                                     \
             if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_qgemm3<MSQ_PLANE_NONE, OK, YT, 1, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8); attr_done(once_); } \
             hipLaunchKernelGGL((k_qgemm3<MSQ_PLANE_NONE, OK, YT, 1, 8, 8>), grid8, blk8, lds8, st, (const uint16_t*)X, (const uint8_t*)inl_plane, \
                    (const uint8_t*)out_plane, (const uint8_t*)scale_plane, bias, (YT*)Y, (int)M, (int)N, (int)K, groups, 1, partial, y16); } while (0)
        if (out_kind == MSQ_PLANE_U8) { if (y_dtype == 0) MSQ_LAUNCH8(MSQ_PLANE_U8, float); else MSQ_LAUNCH8(MSQ_PLANE_U8, uint16_t); }
        else { if (y_dtype == 0) MSQ_LAUNCH8(MSQ_PLANE_U8X, float); else MSQ_LAUNCH8(MSQ_PLANE_U8X, uint16_t); }
#undef MSQ_LAUNCH8
        return check_launch2("msq_qlinear_bf16(eight waves along n)");
    }
    // 64-row wave tiles (64 x 256 blocks instead of 128 x 256): for grids a little larger than the chip -- between one and
    // two 128-row blocks per CU, e.g. 5120 x 5120 at M = 2048 (320 blocks) or M = 3072 on 4096 x 4096 (384) -- the second
    // round is mostly empty; half-height blocks fill it: 113.7 -> 108.2 us posit / 99.0 -> 88.3 us fp8 at 320 blocks,
    // 83.1 -> 78.8 us fp8 at 384 (scripts/experiments/shape_ab.py).  At exactly 256 blocks (Llama-2-7B's 4096 x 4096 and
    // down_proj at M = 2048) they LOSE 5-14 %: a converted weight fragment then feeds 4 MFMAs instead of 8 and the loop is
    // vector-issue bound, so the rule starts above 256.  MSQ_GEMM_MF=4 / 8 forces (tuning only).
    static const int mf_forced = [] { const char* e = getenv("MSQ_GEMM_MF"); return e ? atoi(e) : 0; }();
    const int64_t blocks128 = ((M + 127) / 128) * (N / BN);
    // ... and for grids of about half a chip of 128-row blocks with a short K (112 ... 128 blocks, K <= 4096: M = 1024 on a
    // 4096 x 4096 projection, M = 512 on N = 8192): the cost model would split K in two and pay 33 MB of partial planes;
    // half-height blocks fill the chip in ONE pass instead (fp8 outliers 39.3 -> 33.9 / 41.0 -> 33.6 us, posit 42.7 -> 41.5 /
    // 43.7 -> 40.6 us; with fp8 outliers also from 96 blocks: M = 768 35.8 -> 31.6 us).  Longer K (11008), smaller grids and larger ones (144+ blocks: more than 256 half-height blocks, 50 -> 60 us) lose, so the window is narrow.
    static const int ks_forced = [] { const char* e = getenv("MSQ_GEMM_KS"); return e ? atoi(e) : 0; }();
    if (unified && wm_sel == 1 && mf_sel == 8 && mf_forced == 0 && ks_forced == 0 && blocks128 >= (out_kind == MSQ_PLANE_U8 ? 96 : 112) && blocks128 <= 128 && K / BK <= 64) ksplit = 1;
    const bool mf4 = unified && ksplit == 1 && wm_sel == 1 && mf_sel == 8 &&
                     (mf_forced == 4 || (mf_forced == 0 && ((blocks128 > 256 && blocks128 < 448) || (blocks128 >= (out_kind == MSQ_PLANE_U8 ? 96 : 112) && blocks128 <= 128 && K / BK <= 64))));
    if (mf4) {
        const dim3 grid4((unsigned)(((M + 63) / 64) * (N / BN))), blk4(256);
        const size_t lds4 = 4 * 8192;                               // max(3 x 8 KiB activation buffers, 4 x 8 KiB epilogue slices)
#define MSQ_LAUNCH4(OK, YT)                                                                                            \
        do { static DevOnce once_;                                                                                     \
             if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_qgemm3<MSQ_PLANE_NONE, OK, YT, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); attr_done(once_); } \
             hipLaunchKernelGGL((k_qgemm3<MSQ_PLANE_NONE, OK, YT, 1, 4>), grid4, blk4, lds4, st, (const uint16_t*)X, (const uint8_t*)inl_plane, \
                    (const uint8_t*)out_plane, (const uint8_t*)scale_plane, bias, (YT*)Y, (int)M, (int)N, (int)K, groups, 1, partial, y16); } while (0)
        if (out_kind == MSQ_PLANE_U8) { if (y_dtype == 0) MSQ_LAUNCH4(MSQ_PLANE_U8, float); else MSQ_LAUNCH4(MSQ_PLANE_U8, uint16_t); }
        else { if (y_dtype == 0) MSQ_LAUNCH4(MSQ_PLANE_U8X, float); else MSQ_LAUNCH4(MSQ_PLANE_U8X, uint16_t); }
#undef MSQ_LAUNCH4
        return check_launch2("msq_qlinear_bf16(64-row wave tiles)");
    }
    // Two k-groups per block (eight waves, KG = 2) for single-pass grids of at most one block per CU (192 ... 256 blocks:
    // Llama-2-7B's 4096 x 4096 projections and down_proj at M = 2048 are exactly 256): two waves per SIMD without partial
    // planes.  Measured (scripts/experiments/shape_ab.py, posit / fp8): M2048 4096 x 4096 62.0 -> 58.0 / 55.8 -> 53.6 us,
    // K = 11008 154.1 -> 147.5 / 138.2 -> 131.0, M1536 57.0 -> 51.0 / 49.8 -> 46.6, 8192 x 8192 M1024 116.0 -> 107.3 us.
    // Under split-K it buys nothing (the groups share the CU's MFMA pipes: a full pass over K = 4096 takes 42.6 us instead
    // of 47.4; more blocks on more CUs is what shortens it), so the rule is single-pass only.  MSQ_GEMM_KG=1 / 2 forces.
    static const int kg_forced = [] { const char* e = getenv("MSQ_GEMM_KG"); return e ? atoi(e) : 0; }();
    const int64_t KTq = K / BK;
    const bool kg2 = unified && mf_sel == 8 && wm_sel == 1 && (KTq % (2 * ksplit)) == 0 && KTq / (2 * ksplit) >= 2 &&
                     (kg_forced == 2 || (kg_forced == 0 && ksplit == 1 && blocks128 <= 256));
    if (kg2) {
        const dim3 blk2(512);
        const size_t lds2 = 4 * 32768;                              // max(2 x 3 x 16 KiB activation buffers, 4 x 32 KiB accumulator hand-over)
#define MSQ_LAUNCHKG(OK, YT)                                                                                           \
        do { static DevOnce once_;                                                                                     \
             if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_qgemm3<MSQ_PLANE_NONE, OK, YT, 1, 8, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); attr_done(once_); } \
             hipLaunchKernelGGL((k_qgemm3<MSQ_PLANE_NONE, OK, YT, 1, 8, 4, 2>), grid, blk2, lds2, st, (const uint16_t*)X, (const uint8_t*)inl_plane, \
                    (const uint8_t*)out_plane, (const uint8_t*)scale_plane, bias, (YT*)Y, (int)M, (int)N, (int)K, groups, ksplit, partial, y16); } while (0)
        if (out_kind == MSQ_PLANE_U8) { if (y_dtype == 0) MSQ_LAUNCHKG(MSQ_PLANE_U8, float); else MSQ_LAUNCHKG(MSQ_PLANE_U8, uint16_t); }
        else { if (y_dtype == 0) MSQ_LAUNCHKG(MSQ_PLANE_U8X, float); else MSQ_LAUNCHKG(MSQ_PLANE_U8X, uint16_t); }
#undef MSQ_LAUNCHKG
        rc = check_launch2("msq_qlinear_bf16(two k-groups per block)");
        if (rc || ksplit == 1) return rc;
        const int64_t MNk = M * N;
        const dim3 rgk((unsigned)((MNk / 4 + 255) / 256));
        if (y_dtype == 0) hipLaunchKernelGGL(k_splitk_reduce<float>, rgk, dim3(256), 0, st, partial, bias, (float*)Y, MNk, (int)N, ksplit, y_dtype == 1 ? 1 : 0);
        else hipLaunchKernelGGL(k_splitk_reduce<uint16_t>, rgk, dim3(256), 0, st, partial, bias, (uint16_t*)Y, MNk, (int)N, ksplit, y_dtype == 1 ? 1 : 0);
        return check_launch2("msq_qlinear_bf16(split-K reduce)");
    }
    if (mf_sel == 16) {
#define MSQ_LAUNCH16(OK, YT)                                                                                           \
        do { static DevOnce once_;                                                                                     \
             if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_qgemm3<MSQ_PLANE_NONE, OK, YT, 1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done(once_); } \
             hipLaunchKernelGGL((k_qgemm3<MSQ_PLANE_NONE, OK, YT, 1, 16>), grid, blk, lds, st, (const uint16_t*)X, (const uint8_t*)inl_plane, \
                    (const uint8_t*)out_plane, (const uint8_t*)scale_plane, bias, (YT*)Y, (int)M, (int)N, (int)K, groups, ksplit, partial, y16); } while (0)
        if (out_kind == MSQ_PLANE_U8) { if (y_dtype == 0) MSQ_LAUNCH16(MSQ_PLANE_U8, float); else MSQ_LAUNCH16(MSQ_PLANE_U8, uint16_t); }
        else { if (y_dtype == 0) MSQ_LAUNCH16(MSQ_PLANE_U8X, float); else MSQ_LAUNCH16(MSQ_PLANE_U8X, uint16_t); }
#undef MSQ_LAUNCH16
        return check_launch2("msq_qlinear_bf16(256-row wave tiles)");
    }
    MSQ_DISPATCH(k_qgemm3);
#undef MSQ_DISPATCH
#undef MSQ_LAUNCH
#undef MSQ_LAUNCH1
    rc = check_launch2("msq_qlinear_bf16");
    if (rc || ksplit == 1) return rc;
    const int64_t MN = M * N;
    const dim3 rgrid((unsigned)((MN / 4 + 255) / 256));
    if (y_dtype == 0) hipLaunchKernelGGL(k_splitk_reduce<float>, rgrid, dim3(256), 0, st, partial, bias, (float*)Y, MN, (int)N, ksplit, y_dtype == 1 ? 1 : 0);
    else hipLaunchKernelGGL(k_splitk_reduce<uint16_t>, rgrid, dim3(256), 0, st, partial, bias, (uint16_t*)Y, MN, (int)N, ksplit, y_dtype == 1 ? 1 : 0);
    return check_launch2("msq_qlinear_bf16(split-K reduce)");
}

int msq_qlinear_bf16(const void* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                     const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int block,
                     int in_kind, int out_kind, void* workspace, int64_t workspace_bytes, void* stream) {
    return qlinear_bf16_impl(X, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, block, in_kind, out_kind, workspace,
                             workspace_bytes, stream, 0);
}
// the same Linear on fp16 activations at the decode sizes: the weight-streaming kernels convert them (half -> bf16, round to
// nearest even: what x.to(bfloat16) gives) while they load them
int msq_qlinear_f16x(const void* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                     const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int block,
                     int in_kind, int out_kind, void* workspace, int64_t workspace_bytes, void* stream) {
    return qlinear_bf16_impl(X, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, block, in_kind, out_kind, workspace,
                             workspace_bytes, stream, 1);
}

// ---------------------------------------------------------------------------
// W4A8 Linear (BASELINE config 3; number_system/mx/linear.py:66-91 with a_elem_format = an 8-bit
// format): quantise the activations along K (one HBM-bound pass, bf16 out, exact) and run the fused
// dequant-GEMM on them.  The reference multiplies the two fake-quantised fp32 tensors with F.linear;
// both operands are exact in bf16, so the MFMA bf16 path computes the same products (fp32 accumulate).
// A scaled-fp8 MFMA formulation needs one product per (inlier, outlier) x (inlier, outlier) scale pair
// = 4 v_mfma_scale_f32_16x16x128_f8f6f4 at 32 cycles each per 128 k, twice the matrix time of bf16.
// ---------------------------------------------------------------------------
int64_t msq_act_quant_workspace_bytes(int64_t M, int64_t K, int block, int variant);
int msq_act_quant_bf16(const float* X, void* Xq, int* status_flag, void* workspace, int64_t workspace_bytes, int64_t M,
                       int64_t K, int block, int inlier_fmt, int outlier_fmt, int inlier_scale_bits, int outlier_scale_bits,
                       float std_dev, int rmode, int flush_fp32_subnorms, int variant, void* stream);

static int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }

int64_t msq_qlinear_w4a8_workspace_bytes(int64_t M, int64_t N, int64_t K, int a_block, int a_variant) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return align256(M * K * 2) + align256(msq_act_quant_workspace_bytes(M, K, a_block, a_variant)) +
           msq_qlinear_workspace_bytes(M, N, K);
}

static int qlinear_w4a8_impl(const void* X, int x_bf16, const void* inl_plane, const void* out_plane, const void* scale_plane,
                             const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int w_block, int in_kind,
                             int out_kind, int a_block, int a_inlier_fmt, int a_outlier_fmt, int a_inlier_scale_bits,
                             int a_outlier_scale_bits, float a_std_dev, int a_rmode, int a_flush_fp32_subnorms, int a_variant,
                             int* status_flag, void* workspace, int64_t workspace_bytes, void* stream) {
    if (M <= 0) return (M == 0) ? MSQ_OK : fail2(MSQ_ERR_BAD_ARG, "msq_qlinear_w4a8: negative M");
    if (!workspace || workspace_bytes < msq_qlinear_w4a8_workspace_bytes(M, N, K, a_block, a_variant))
        return fail2(MSQ_ERR_BAD_ARG, "msq_qlinear_w4a8: workspace too small (msq_qlinear_w4a8_workspace_bytes)");
    char* xq = (char*)workspace;
    char* aws = xq + align256(M * K * 2);
    const int64_t aws_bytes = align256(msq_act_quant_workspace_bytes(M, K, a_block, a_variant));
    char* gws = aws + aws_bytes;
    int rc = x_bf16 ? msq_act_quant_bf16_x16(X, xq, status_flag, aws, aws_bytes, M, K, a_block, a_inlier_fmt, a_outlier_fmt,
                                             a_inlier_scale_bits, a_outlier_scale_bits, a_std_dev, a_rmode, a_flush_fp32_subnorms,
                                             a_variant, stream)
                    : msq_act_quant_bf16((const float*)X, xq, status_flag, aws, aws_bytes, M, K, a_block, a_inlier_fmt, a_outlier_fmt,
                                         a_inlier_scale_bits, a_outlier_scale_bits, a_std_dev, a_rmode, a_flush_fp32_subnorms,
                                         a_variant, stream);
    if (rc) return rc;
    return msq_qlinear_bf16(xq, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, w_block, in_kind, out_kind,
                            gws, workspace_bytes - (gws - (char*)workspace), stream);
}
int msq_qlinear_w4a8(const float* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                     const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int w_block, int in_kind,
                     int out_kind, int a_block, int a_inlier_fmt, int a_outlier_fmt, int a_inlier_scale_bits,
                     int a_outlier_scale_bits, float a_std_dev, int a_rmode, int a_flush_fp32_subnorms, int a_variant,
                     int* status_flag, void* workspace, int64_t workspace_bytes, void* stream) {
    return qlinear_w4a8_impl(X, 0, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, w_block, in_kind, out_kind, a_block,
                             a_inlier_fmt, a_outlier_fmt, a_inlier_scale_bits, a_outlier_scale_bits, a_std_dev, a_rmode,
                             a_flush_fp32_subnorms, a_variant, status_flag, workspace, workspace_bytes, stream);
}
int msq_qlinear_w4a8_x16(const void* X, const void* inl_plane, const void* out_plane, const void* scale_plane,
                         const float* bias, void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int w_block, int in_kind,
                         int out_kind, int a_block, int a_inlier_fmt, int a_outlier_fmt, int a_inlier_scale_bits,
                         int a_outlier_scale_bits, float a_std_dev, int a_rmode, int a_flush_fp32_subnorms, int a_variant,
                         int* status_flag, void* workspace, int64_t workspace_bytes, void* stream) {
    return qlinear_w4a8_impl(X, 1, inl_plane, out_plane, scale_plane, bias, Y, y_dtype, M, N, K, w_block, in_kind, out_kind, a_block,
                             a_inlier_fmt, a_outlier_fmt, a_inlier_scale_bits, a_outlier_scale_bits, a_std_dev, a_rmode,
                             a_flush_fp32_subnorms, a_variant, status_flag, workspace, workspace_bytes, stream);
}

// MX-native W4A8 GEMM on pre-packed operands (msq_mx_pack_a8 / msq_mx_pack_w4).  Few row tiles (small M): K is
// split over power-of-two many work items so that about one block per CU streams the weight; fp32 partial tiles
// go through `workspace` (msq_qlinear_mx_w4a8_workspace_bytes; NULL = single pass).
// Split-K of the MX GEMM (splitk_by_cost): a 128-k step takes ~0.77 us per block.  The earlier rule (fill 256 blocks,
// chunks of >= 2 K-steps) over-split short K: 4096 x 4096 M128 22.9 -> 17.1-18.9 us, N11008 M512 40.3 -> 30.2 us.
static int pick_mx_ksplit(int64_t M, int64_t N, int64_t K) {
    static const int forced = [] { const char* e = getenv("MSQ_MX_GEMM_KS"); return e ? atoi(e) : 0; }();   // tuning only
    const int64_t blocks = ((M + 127) / 128) * (N / BN), KT = K / 128;
    if (forced > 0) return forced < KT ? forced : (int)KT;
    if (blocks >= 192 || KT < 4) return 1;
    return splitk_by_cost(blocks, KT, M, N, 0.77, 16, 2);
}
// decode kernel or GEMM, as use_gemv: up to M = 32 always, up to 64 for N <= 4096 (13.1 vs 17.6 us at 4096 x 4096,
// 16.3 vs 17.4 at K = 11008; 22.1 vs 18.1 at N = 16384).  MSQ_MX_GEMV_MAX_M (tuning only): M <= value.
static bool use_mx_gemv(int64_t M, int64_t N, int64_t K) {
    static int v = -2;
    if (v == -2) { const char* e = getenv("MSQ_MX_GEMV_MAX_M"); v = e ? atoi(e) : -1; if (v > 64) v = 64; }
    if (v >= 0) return M <= v;
    (void)K;
    return M <= 32 || (M <= 64 && N <= 4096);
}
// single-launch decode: 16 waves per block = 16 k-chunks of one strip, summed in LDS, output written by wave 0.
// Needs M <= 32 (LDS: 15 x 16 MG x 64 floats); used where it measured faster than decode + reduce
// (scripts/experiments/mx_decode_check.py: N >= 8192 and K <= 4096: 6.0 vs 8.1 us at N16384 M1, 4.8 vs 6.6 at N11008;
// e4m3 operand 12.8 vs 14.6); returns the K-steps per wave or 0.
// MSQ_MX_GEMV_DIRECT=0 (tuning only) disables it.
static int mx_direct_kc(int64_t M, int64_t N, int64_t K) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MSQ_MX_GEMV_DIRECT"); on = e ? atoi(e) : 1; }
    static int half = -1;
    if (half < 0) { const char* e = getenv("MSQ_MX_GEMV_HALF"); half = e ? atoi(e) : 1; }      // tuning only
    // fewer than 128 blocks are per-CU bandwidth bound (measured): below N = 8192 the blocks take half strips (N / 32)
    if (!on || M > 32 || N < (half ? 4096 : 8192)) return 0;
    const int kc = (int)((K / 128 + 15) / 16);
    return kc <= 2 ? kc : 0;                                     // long K: few long waves lose to ~3000 short ones (measured)
}
// decode path: K-steps per wave so that there are ~3000 waves (4 per block) and at most 32 k-chunks
static int pick_mx_kc(int64_t N, int64_t K) {
    const int64_t KT = K / 128, strips = N / 64;
    int64_t kc = (strips * KT + 3071) / 3072;
    if (kc < 1) kc = 1;
    while ((KT + kc - 1) / kc > 32) ++kc;
    return (int)kc;
}
int64_t msq_qlinear_mx_w4a8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || (N % BN) || (K % 128)) return 0;
    if (use_mx_gemv(M, N, K)) { if (mx_direct_kc(M, N, K)) return 0; const int kc = pick_mx_kc(N, K); return (((K / 128 + kc - 1) / kc + 3) / 4) * M * N * 4; }
    const int ks = pick_mx_ksplit(M, N, K);
    return ks > 1 ? (int64_t)ks * M * N * 4 : 0;
}
// wf: weight operand format of k_mxgemm (0 e2m1, 1 e4m3, 2 fp6 e2m3, 3 fp6 e3m2)
static int mx_linear(int wf, const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales, const float* bias,
                     void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes,
                     void* stream) {
    if (M <= 0) return (M == 0) ? MSQ_OK : fail2(MSQ_ERR_BAD_ARG, "msq_qlinear_mx_w4a8: negative M");
    if (N <= 0 || K <= 0 || (N % BN) || (K % 128)) return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_mx_w4a8: N must be a multiple of 256 and K of 128");
    if (!x_codes || !x_scales || !w_codes || !w_scales || !Y) return fail2(MSQ_ERR_BAD_ARG, "msq_qlinear_mx_w4a8: null buffer");
    if (y_dtype != 0 && y_dtype != 1 && y_dtype != 2) return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_mx_w4a8: y_dtype must be 0 (f32), 1 (fp16) or 2 (bf16)");
    const int y16 = (y_dtype == 1) ? 1 : 0;
    if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30) || M * K > 0xFFFFFFFFll) return fail2(MSQ_ERR_UNSUPPORTED, "msq_qlinear_mx_w4a8: dimension too large");
    hipStream_t st = (hipStream_t)stream;
    if (use_mx_gemv(M, N, K)) {
        const int mg = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
        const int kcd = mx_direct_kc(M, N, K);                           // > 0: one block of 16 waves covers all of K
        const int kc = kcd ? kcd : pick_mx_kc(N, K);
        const int waves = kcd ? 16 : 4;
        const int nks = kcd ? 1 : (int)(((K / 128 + kc - 1) / kc + 3) / 4);   // partial planes: one per four k-chunks
        if (kcd || (workspace && workspace_bytes >= (int64_t)nks * M * N * 4)) {
            const size_t ldsv = (size_t)(waves - 1) * 16 * mg * 64 * 4;
#define MSQ_MXV(W8V, MGV, WV)                                                                                          \
            do { static DevOnce once_;                                                                         \
                 if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_mxgemv<W8V, MGV, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (WV - 1) * 16 * MGV * 64 * 4); attr_done(once_); } \
                 hipLaunchKernelGGL((k_mxgemv<W8V, MGV, WV>), dim3((unsigned)((N / 64) * nks)), dim3(64 * WV), ldsv, st, (const uint8_t*)x_codes, (const uint8_t*)x_scales, \
                                    (const uint8_t*)w_codes, (const uint8_t*)w_scales, (float*)workspace, (int)M, (int)N, (int)K, kc, kcd ? 1 : 0, bias, Y, y_dtype == 2 ? 1 : (y_dtype == 1 ? 2 : 0)); } while (0)
            if (kcd && N < 8192) {                                       // half strips: N / 32 blocks
                const size_t ldsh = (size_t)15 * 8 * mg * 64 * 4;
#define MSQ_MXH(W8V, MGV)                                                                                               \
                do { static DevOnce once_;                                                                     \
                     if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_mxgemv<W8V, MGV, 16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 15 * 8 * MGV * 64 * 4); attr_done(once_); } \
                     hipLaunchKernelGGL((k_mxgemv<W8V, MGV, 16, 2>), dim3((unsigned)(N / 32)), dim3(1024), ldsh, st, (const uint8_t*)x_codes, (const uint8_t*)x_scales, \
                                        (const uint8_t*)w_codes, (const uint8_t*)w_scales, (float*)workspace, (int)M, (int)N, (int)K, kc, 1, bias, Y, y_dtype == 2 ? 1 : (y_dtype == 1 ? 2 : 0)); } while (0)
                if (mg == 1) { if (wf == 0) MSQ_MXH(0, 1); else if (wf == 1) MSQ_MXH(1, 1); else if (wf == 2) MSQ_MXH(2, 1); else MSQ_MXH(3, 1); }
                else { if (wf == 0) MSQ_MXH(0, 2); else if (wf == 1) MSQ_MXH(1, 2); else if (wf == 2) MSQ_MXH(2, 2); else MSQ_MXH(3, 2); }
#undef MSQ_MXH
                return check_launch2("msq_qlinear_mx_w4a8(decode, single launch, half strips)");
            }
            if (kcd && mg == 1 && (N / 64 > 256 || (wf != 0 && N / 64 > 128))) {
                // Waves per block as for k_qgemv_u: with cold weights a grid of ~1000-2000 waves streams fastest, and more than 256
                // strips of sixteen-wave blocks (the 24- / 32-byte operands need 92-94 registers: one block per CU) take a second,
                // part-filled round.  Measured (scripts/experiments/decode_cold.py, M = 1, us; 16 / 8 / 4 waves): fused gate / up
                // (344 strips) e4m3 operand 25.0 / 24.6 / 22.7, fp4 15.3 / 13.9 / 13.5; q/k/v (192 strips) e4m3 15.1 / 14.1 / 16.4,
                // fp4 10.2 / 10.8 / 11.6 (stays at sixteen).
                const bool w4 = N / 64 > 256;
                const int wvx = w4 ? 4 : 8;
                const int kcx = (int)((K / 128 + wvx - 1) / wvx);
                const size_t ldsx = (size_t)(wvx - 1) * 16 * 64 * 4;
#define MSQ_MXVX(W8V, WV)                                                                                               \
                do { static DevOnce once_;                                                                              \
                     if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_mxgemv<W8V, 1, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (WV - 1) * 16 * 64 * 4); attr_done(once_); } \
                     hipLaunchKernelGGL((k_mxgemv<W8V, 1, WV>), dim3((unsigned)(N / 64)), dim3(64 * WV), ldsx, st, (const uint8_t*)x_codes, (const uint8_t*)x_scales, \
                                        (const uint8_t*)w_codes, (const uint8_t*)w_scales, (float*)workspace, (int)M, (int)N, (int)K, kcx, 1, bias, Y, y_dtype == 2 ? 1 : (y_dtype == 1 ? 2 : 0)); } while (0)
                if (w4) { if (wf == 0) MSQ_MXVX(0, 4); else if (wf == 1) MSQ_MXVX(1, 4); else if (wf == 2) MSQ_MXVX(2, 4); else MSQ_MXVX(3, 4); }
                else { if (wf == 1) MSQ_MXVX(1, 8); else if (wf == 2) MSQ_MXVX(2, 8); else MSQ_MXVX(3, 8); }
#undef MSQ_MXVX
                return check_launch2("msq_qlinear_mx_w4a8(decode, single launch, four- / eight-wave blocks)");
            }
            if (kcd) {
                if (mg == 1) { if (wf == 0) MSQ_MXV(0, 1, 16); else if (wf == 1) MSQ_MXV(1, 1, 16); else if (wf == 2) MSQ_MXV(2, 1, 16); else MSQ_MXV(3, 1, 16); }
                else { if (wf == 0) MSQ_MXV(0, 2, 16); else if (wf == 1) MSQ_MXV(1, 2, 16); else if (wf == 2) MSQ_MXV(2, 2, 16); else MSQ_MXV(3, 2, 16); }
                return check_launch2("msq_qlinear_mx_w4a8(decode, single launch)");
            }
            if (mg == 1) { if (wf == 0) MSQ_MXV(0, 1, 4); else if (wf == 1) MSQ_MXV(1, 1, 4); else if (wf == 2) MSQ_MXV(2, 1, 4); else MSQ_MXV(3, 1, 4); }
            else if (mg == 2) { if (wf == 0) MSQ_MXV(0, 2, 4); else if (wf == 1) MSQ_MXV(1, 2, 4); else if (wf == 2) MSQ_MXV(2, 2, 4); else MSQ_MXV(3, 2, 4); }
            else { if (wf == 0) MSQ_MXV(0, 4, 4); else if (wf == 1) MSQ_MXV(1, 4, 4); else if (wf == 2) MSQ_MXV(2, 4, 4); else MSQ_MXV(3, 4, 4); }
#undef MSQ_MXV
            int rc0 = check_launch2("msq_qlinear_mx_w4a8(decode)");
            if (rc0) return rc0;
            const int64_t MN0 = M * N;
            const dim3 rg((unsigned)((MN0 / 4 + 255) / 256));
            if (y_dtype == 0) hipLaunchKernelGGL(k_splitk_reduce<float>, rg, dim3(256), 0, st, (const float*)workspace, bias, (float*)Y, MN0, (int)N, nks, y_dtype == 1 ? 1 : 0);
            else hipLaunchKernelGGL(k_splitk_reduce<uint16_t>, rg, dim3(256), 0, st, (const float*)workspace, bias, (uint16_t*)Y, MN0, (int)N, nks, y_dtype == 1 ? 1 : 0);
            return check_launch2("msq_qlinear_mx_w4a8(decode reduce)");
        }
    }
    // 256-row wave tiles, one wave per SIMD, accumulators placed by hand (k_mxgemm256, msq_mxgemm256.hip), and its 128-row form (MF = 8,
    // two blocks per CU) for the grids 256-row blocks do not fill.  MSQ_MX_256=1 / 2 force the 256- / 128-row form, 0 disables both
    // (read per call: tests and A / B scripts flip it inside one process).  Default rule: the cost in rounds of msq_qlinear_bf16.
    {
        // measured (profiles/r04_mx128_sweep.txt): 256-row blocks win from 144 blocks on -- also on part-filled rounds, where k_mxgemm's
        // 128-row blocks run two per CU on part of the chip (M512 N22016: 43.6 against 59.7 us; e4m3 operand M1024 N22016: 95.3
        // against 114.8).  The 128-row form pays with the 16-byte MX-FP4 operand only, where a second round would be less than half
        // full (q/k/v 2048 x 12288, 384 blocks: 74.7 us against 88.4, k_mxgemm 79.5); with the 24- / 32-byte operands its two-deep
        // weight ring costs more than the saved half round (116.7 against 95.3), so those never take it.  One-round grids of <= 128
        // blocks of 256 rows (o, down at M = 2048) stay on k_mxgemm.
        const int mfsel = mx256_family(M, N, wf);                   // the rule + the MSQ_MX_256 switch (also behind msq_qlinear_kernel_choice)
        if (mfsel) {
            const int e = msq_launch_mxgemm256(wf, x_codes, x_scales, w_codes, w_scales, bias, Y, y_dtype, M, N, K, mfsel, stream);
            if (e) { char b[200]; snprintf(b, sizeof(b), "msq_qlinear_mx_w4a8(k_mxgemm256, MF = %d): %s", mfsel, hipGetErrorString((hipError_t)e)); return fail2(MSQ_ERR_LAUNCH, b); }
            return MSQ_OK;
        }
    }
    const int MT = (int)((M + 127) / 128), NTB = (int)(N / BN);
    int ksplit = pick_mx_ksplit(M, N, K);
    if (ksplit > 1 && (!workspace || workspace_bytes < (int64_t)ksplit * M * N * 4)) ksplit = 1;
    // two k-groups per block for single-pass grids of at most one block per CU (rule and switch as msq_qlinear_bf16)
    static const int kg_forced = [] { const char* e = getenv("MSQ_MX_KG"); return e ? atoi(e) : 0; }();
    const int64_t KTm = K / 128;
    const bool kg2 = ksplit == 1 && (KTm % 2) == 0 && KTm >= 4 && (kg_forced == 2 || (kg_forced == 0 && (int64_t)MT * NTB <= 256));   // (ksplit == 1 here means >= 192 blocks: disjoint from the 64-row windows below)
    // 64-row blocks (MSQ_MX_MF=4 / 8 forces, tuning only).  Measured from HIP graphs (scripts/experiments/mx_mf4_graph.py):
    //  * ONE pass of 64-row blocks instead of a split-K launch whenever they fit the chip (<= 256 blocks): the fp4 kernel then
    //    takes a flat 17.4-19.5 us for K = 4096 at any M (512 x 4096 x 4096: 23.2 -> 17.9 us, 768: 26.4 -> 18.6, M384 N8192:
    //    26.7 -> 18.6, M1024: 30.4 -> 21.4 from Python); with K = 11008 only from 192 blocks on (M768: 44.5 -> 41.3 us, M1024:
    //    54.1 -> 42.8).  The 24- / 32-byte operands take ~25 us for the pass: a gain from 192 blocks on (768 x 4096 x 4096:
    //    29.2 -> 25.1 us, M384 N8192: 31.2 -> 25.3), a loss below.
    //  * between one and two 128-row blocks per CU only the fp4 operand gains (5120 x 5120 54.1 -> 51.2 us); the wider operands
    //    lose up to 20 % there (half as many MFMAs per weight load).
    static const int mf_forced = [] { const char* e = getenv("MSQ_MX_MF"); return e ? atoi(e) : 0; }();
    static const int ks_forced = [] { const char* e = getenv("MSQ_MX_GEMM_KS"); return e ? atoi(e) : 0; }();
    const int64_t blocks128 = (int64_t)MT * NTB, blocks64 = ((M + 63) / 64) * NTB;
    const bool one_pass64 = blocks64 <= 256 && M > 64 &&
                            (KTm <= 32 ? (wf == 0 || blocks64 >= 192) : (wf == 0 && blocks64 >= 192));
    if (mf_forced == 0 && ks_forced == 0 && one_pass64) ksplit = 1;
    const bool mf4 = ksplit == 1 && (mf_forced == 4 || (mf_forced == 0 && ((wf == 0 && blocks128 > 256 && blocks128 < 448) || one_pass64)));
    if (mf4) {
        const dim3 grid4((unsigned)(((M + 63) / 64) * NTB)), blk4(256);
        size_t lds4 = (size_t)(wf ? 3 : MSQ_MX_XBUFS) * (64 * 128 + 1024);
        if (lds4 < 4 * 8192) lds4 = 4 * 8192;                          // the epilogue stages 8 KiB per wave
        // ... with two k-groups when the blocks leave every CU at most one (the K-step of a 64-row block is latency, not MFMA:
        // 0.45 us for 16 MFMAs per wave): MSQ_MX_KG=1 keeps one group
        const bool kg4 = (KTm % 2) == 0 && KTm >= 4 && blocks64 <= 256 && kg_forced != 1;
        if (kg4) { lds4 *= 2; if (lds4 < 4 * 16384) lds4 = 4 * 16384; }   // two groups; 4 x 16 KiB accumulator hand-over
        const dim3 blk4k(kg4 ? 512 : 256);
#define MSQ_MXL4(YT, W8V)                                                                                             \
        do { if (kg4) { static DevOnce once_;                                                                         \
             if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_mxgemm<YT, W8V, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); attr_done(once_); } \
             hipLaunchKernelGGL((k_mxgemm<YT, W8V, 2, 4>), grid4, blk4k, lds4, st, (const uint8_t*)x_codes, (const uint8_t*)x_scales, (const uint8_t*)w_codes, \
                                (const uint8_t*)w_scales, bias, (YT*)Y, (int)M, (int)N, (int)K, 1, (float*)nullptr, y16); }         \
             else { static DevOnce once_;                                                                             \
             if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_mxgemm<YT, W8V, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); attr_done(once_); } \
             hipLaunchKernelGGL((k_mxgemm<YT, W8V, 1, 4>), grid4, blk4, lds4, st, (const uint8_t*)x_codes, (const uint8_t*)x_scales, (const uint8_t*)w_codes, \
                                (const uint8_t*)w_scales, bias, (YT*)Y, (int)M, (int)N, (int)K, 1, (float*)nullptr, y16); } } while (0)
        if (y_dtype == 0) { if (wf == 0) MSQ_MXL4(float, 0); else if (wf == 1) MSQ_MXL4(float, 1); else if (wf == 2) MSQ_MXL4(float, 2); else MSQ_MXL4(float, 3); }
        else { if (wf == 0) MSQ_MXL4(uint16_t, 0); else if (wf == 1) MSQ_MXL4(uint16_t, 1); else if (wf == 2) MSQ_MXL4(uint16_t, 2); else MSQ_MXL4(uint16_t, 3); }
#undef MSQ_MXL4
        return check_launch2("msq_qlinear_mx_w4a8(64-row blocks)");
    }
    const dim3 grid((unsigned)(MT * NTB * ksplit)), blk(kg2 ? 512 : 256);
    size_t lds = (size_t)(wf ? 3 : MSQ_MX_XBUFS) * (128 * 128 + 1024);   // code tiles + scale tiles
    if (kg2) { lds *= 2; if (lds < 4 * 32768) lds = 4 * 32768; }        // two groups; 4 x 32 KiB accumulator hand-over
    float* partial = (float*)workspace;
#define MSQ_MXL(YT, W8V)                                                                                              \
    do { if (kg2) { static DevOnce once_;                                                                      \
         if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_mxgemm<YT, W8V, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done(once_); } \
         hipLaunchKernelGGL((k_mxgemm<YT, W8V, 2>), grid, blk, lds, st, (const uint8_t*)x_codes, (const uint8_t*)x_scales, (const uint8_t*)w_codes, \
                            (const uint8_t*)w_scales, bias, (YT*)Y, (int)M, (int)N, (int)K, ksplit, partial, y16); }        \
         else { static DevOnce once_;                                                                          \
         if (attr_needed(once_)) { hipFuncSetAttribute((const void*)k_mxgemm<YT, W8V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done(once_); } \
         hipLaunchKernelGGL((k_mxgemm<YT, W8V>), grid, blk, lds, st, (const uint8_t*)x_codes, (const uint8_t*)x_scales, (const uint8_t*)w_codes, \
                            (const uint8_t*)w_scales, bias, (YT*)Y, (int)M, (int)N, (int)K, ksplit, partial, y16); } } while (0)
    if (y_dtype == 0) { if (wf == 0) MSQ_MXL(float, 0); else if (wf == 1) MSQ_MXL(float, 1); else if (wf == 2) MSQ_MXL(float, 2); else MSQ_MXL(float, 3); }
    else { if (wf == 0) MSQ_MXL(uint16_t, 0); else if (wf == 1) MSQ_MXL(uint16_t, 1); else if (wf == 2) MSQ_MXL(uint16_t, 2); else MSQ_MXL(uint16_t, 3); }
#undef MSQ_MXL
    int rc = check_launch2("msq_qlinear_mx_w4a8");
    if (rc || ksplit == 1) return rc;
    const int64_t MN = M * N;
    const dim3 rgrid((unsigned)((MN / 4 + 255) / 256));
    if (y_dtype == 0) hipLaunchKernelGGL(k_splitk_reduce<float>, rgrid, dim3(256), 0, st, partial, bias, (float*)Y, MN, (int)N, ksplit, y_dtype == 1 ? 1 : 0);
    else hipLaunchKernelGGL(k_splitk_reduce<uint16_t>, rgrid, dim3(256), 0, st, partial, bias, (uint16_t*)Y, MN, (int)N, ksplit, y_dtype == 1 ? 1 : 0);
    return check_launch2("msq_qlinear_mx_w4a8(split-K reduce)");
}
int msq_qlinear_mx_w4a8(const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales, const float* bias,
                        void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes,
                        void* stream) {
    return mx_linear(0, x_codes, x_scales, w_codes, w_scales, bias, Y, y_dtype, M, N, K, workspace, workspace_bytes, stream);
}
int msq_qlinear_mx_w8a8(const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales, const float* bias,
                        void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes,
                        void* stream) {
    return mx_linear(1, x_codes, x_scales, w_codes, w_scales, bias, Y, y_dtype, M, N, K, workspace, workspace_bytes, stream);
}
int msq_qlinear_mx_w6a8(const void* x_codes, const void* x_scales, const void* w_codes, const void* w_scales, const float* bias,
                        void* Y, int y_dtype, int64_t M, int64_t N, int64_t K, int w_format, void* workspace, int64_t workspace_bytes,
                        void* stream) {
    if (w_format != MSQ_FMT_FP6_E3M2 && w_format != MSQ_FMT_FP6_E2M3)
        return fail2(MSQ_ERR_BAD_ARG, "msq_qlinear_mx_w6a8: w_format must be MSQ_FMT_FP6_E3M2 or MSQ_FMT_FP6_E2M3");
    return mx_linear(w_format == MSQ_FMT_FP6_E3M2 ? 3 : 2, x_codes, x_scales, w_codes, w_scales, bias, Y, y_dtype, M, N, K, workspace,
                     workspace_bytes, stream);
}


// Which kernel family the dispatch rules pick for a shape (host logic only: no launch, no device needed; the environment switches
// of the A / B scripts are not consulted).  mx_wf < 0: msq_qlinear_bf16 with plane kind `out_kind`; mx_wf = 0 .. 3: msq_qlinear_mx_w4a8
// with that weight operand format.
int msq_qlinear_kernel_choice(int64_t M, int64_t N, int64_t K, int out_kind, int mx_wf) {
    if (M <= 0 || N <= 0 || K <= 0 || (N % BN)) return -1;
    if (mx_wf >= 0) {
        if (mx_wf > 3 || (K % 128)) return -1;
        if (use_mx_gemv(M, N, K)) return MSQ_KERNEL_DECODE;
        const int mf = mx256_family(M, N, mx_wf);
        return mf == 16 ? MSQ_KERNEL_T256 : (mf == 8 ? MSQ_KERNEL_T128 : MSQ_KERNEL_GEMM128);
    }
    if (K % BK) return -1;
    const bool unified = out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X;
    if (sk_rule(M, N, K, unified, out_kind)) return MSQ_KERNEL_STREAMK;
    if (use_gemv(M, N, K)) return MSQ_KERNEL_DECODE;
    const QFamily f = q_family(M, N, K, unified, out_kind, msq_qlinear_workspace_bytes(M, N, K));
    return f.persistent ? MSQ_KERNEL_PERSISTENT : (f.mf == 16 ? MSQ_KERNEL_T256 : (f.mf == 8 ? MSQ_KERNEL_T128 : MSQ_KERNEL_GEMM128));
}
// The kernel instantiation the same call launches for the GEMM itself, as text (bench.py's roofline.kernel): y_dtype 0 = float32 else 16-bit.
int msq_qlinear_kernel_name(int64_t M, int64_t N, int64_t K, int out_kind, int mx_wf, int y_dtype, char* buf, int cap) {
    if (!buf || cap < 1) return MSQ_ERR_BAD_ARG;
    const int fam = msq_qlinear_kernel_choice(M, N, K, out_kind, mx_wf);
    if (fam < 0) { buf[0] = 0; return MSQ_ERR_UNSUPPORTED; }
    const char* yt = y_dtype == 0 ? "float" : "uint16_t";
    if (mx_wf >= 0) {
        if (fam == MSQ_KERNEL_DECODE) snprintf(buf, cap, "k_mxgemv<WF = %d>", mx_wf);
        else if (fam == MSQ_KERNEL_GEMM128) snprintf(buf, cap, "k_mxgemm<%s, WF = %d>", yt, mx_wf);
        else snprintf(buf, cap, "k_mxgemm256<%s, %d, %d>", yt, mx_wf, fam == MSQ_KERNEL_T256 ? 16 : 8);
        return MSQ_OK;
    }
    if (fam == MSQ_KERNEL_STREAMK) {
        const int f = sk_rule(M, N, K, true, out_kind);
        snprintf(buf, cap, "k_qgemm_sk<%d, %s, %d, %d, %d, %d>", out_kind, yt, (f == 1 || f == 5) ? 4 : 8, (f == 3 || f == 5 || f == 6) ? 2 : 1,
                 (f == 1 || f == 4) ? 8 : ((f == 2 || f == 5 || f == 6) ? 4 : 2), f == 4 ? 1 : ((f == 3 || f == 5) ? 3 : 2));
    }
    else if (fam == MSQ_KERNEL_DECODE) snprintf(buf, cap, "k_qgemv_u / k_qgemv<out kind %d>", out_kind);
    else if (fam == MSQ_KERNEL_GEMM128) snprintf(buf, cap, "k_qgemm3<out kind %d, %s>", out_kind, yt);
    else if (fam == MSQ_KERNEL_PERSISTENT) snprintf(buf, cap, "k_qgemm256p<%d, %s>", out_kind, yt);
    else snprintf(buf, cap, "k_qgemm256<%d, %s, %d>", out_kind, yt, fam == MSQ_KERNEL_T256 ? 16 : 8);
    return MSQ_OK;
}

}  // extern "C"
