// msq_pack_emit.hip -- first half of msq_outlier_pack: the MicroScopiQ quantiser
// (utils/quant.py:147-266, blocks along K) emitting per-element plane codes and the
// per-block exponents.  One block per lane, own translation unit (heavy templates).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"
#include "msq_outlier_core.h"

extern "C" void msq_set_error_(const char* msg);

// A wave owns 64 consecutive blocks = one contiguous run of 64*BS floats (K % BS == 0): it streams the
// run with 16-byte coalesced loads, transposes through LDS (row stride BS+4 words: conflict-free
// ds_read_b128) so that every lane holds its own block, and sends the 32-bit codes back the same way.
template <int BS>
__global__ void __launch_bounds__(256)
k_pack_emit(const float* __restrict__ W, uint32_t* __restrict__ codes, OutlierArgs A, int in_kind, int out_kind) {
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
    if (full) {
        const float4* src = reinterpret_cast<const float4*>(W + g0 * BS);
#pragma unroll
        for (int t = 0; t < BS / 4; ++t) {
            const int f = lane + 64 * t;
            const int row = f / (BS / 4), c4 = f % (BS / 4);
            *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c4 * 4) = src[f];
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
        for (int b = 0; b < BS; ++b) a[b] = (gidx < nblocks) ? W[gidx * BS + b] : 0.f;
    }
    uint32_t mkw[(BS + 31) / 32];
    uint32_t cd[BS];
    float se_in = 0.f, se_out = 0.f;
    int status = 0;
    if (gidx < nblocks) {
        // variant 1 (mx_ops.py:210-330) statistics are per (row, position-in-block): [N, BS]
        const float* vm = A.vmean ? A.vmean + (gidx / A.nblk) * BS : nullptr;
        const float* vs = A.vstd ? A.vstd + (gidx / A.nblk) * BS : nullptr;
        if (A.fi.kind == 0 && A.rmode == 0)
            status = outlier_block_fast<BS, 0, true>(a, mkw, se_in, se_out, A, /*inner order*/ 1, vm, vs, 1, cd, in_kind, out_kind);
        else
            status = outlier_block<BS, true>(a, mkw, se_in, se_out, A, 1, vm, vs, 1, cd, in_kind, out_kind);
        A.e_in[gidx] = se_in;
        A.e_out[gidx] = se_out;
    }
    if (full) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < BS / 4; ++c)
            *reinterpret_cast<uint4*>(tl + lane * LDS_STRIDE + c * 4) = make_uint4(cd[c * 4], cd[c * 4 + 1], cd[c * 4 + 2], cd[c * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        uint4* dst = reinterpret_cast<uint4*>(codes + g0 * BS);
#pragma unroll
        for (int t = 0; t < BS / 4; ++t) {
            const int f = lane + 64 * t;
            const int row = f / (BS / 4), c4 = f % (BS / 4);
            dst[f] = *reinterpret_cast<const uint4*>(tl + row * LDS_STRIDE + c4 * 4);
        }
    } else if (gidx < nblocks) {
#pragma unroll
        for (int b = 0; b < BS; ++b) codes[gidx * BS + b] = cd[b];
    }
    if (status && A.status) atomicOr(A.status, status);
}

extern "C" int msq_pack_emit_(const float* W, uint32_t* codes, float* e_in, float* e_out, int* status, int64_t N,
                              int64_t K, int block, int inlier_fmt, int outlier_fmt, int in_sb, int out_sb,
                              float std_dev, int rmode, int flush, int in_kind, int out_kind, int variant,
                              const float* vmean, const float* vstd, void* stream) {
    msq_host::FmtInfo fi, fo;
    if (!msq_host::format_info(inlier_fmt, &fi) || !msq_host::format_info(outlier_fmt, &fo)) {
        msq_set_error_("msq_outlier_pack: unknown element format"); return MSQ_ERR_BAD_ARG; }
    if (in_sb <= 0 || out_sb <= 0 || in_sb > 8 || out_sb > 8 || rmode < 0 || rmode > 2) {
        msq_set_error_("msq_outlier_pack: bad scale bits / rounding mode"); return MSQ_ERR_BAD_ARG; }
    OutlierArgs A;
    A.fi = Fmt{fi.kind, fi.ebits, fi.mbits, fi.emax, fi.max_norm};
    A.fo = Fmt{fo.kind, fo.ebits, fo.mbits, fo.emax, fo.max_norm};
    A.in_sb = in_sb; A.out_sb = out_sb; A.k = std_dev; A.rmode = rmode; A.flush = flush; A.variant = variant;
    A.pre = N; A.axis_len = K; A.post = 1; A.nblk = K / block;
    A.mask = nullptr; A.e_in = e_in; A.e_out = e_out; A.n_out = nullptr; A.status = status;
    A.vmean = vmean; A.vstd = vstd;
    const int64_t nblocks = N * A.nblk;
    const dim3 grid((unsigned)((nblocks + 255) / 256)), blk(256);   // 4 waves x 64 blocks per workgroup
    hipStream_t st = (hipStream_t)stream;
#define MSQ_PE(BS) case BS: hipLaunchKernelGGL(k_pack_emit<BS>, grid, blk, 0, st, W, codes, A, in_kind, out_kind); break;
    switch (block) { MSQ_PE(8) MSQ_PE(16) MSQ_PE(32) MSQ_PE(64) MSQ_PE(128)
        default: msq_set_error_("msq_outlier_pack: block must be 8/16/32/64/128"); return MSQ_ERR_UNSUPPORTED; }
#undef MSQ_PE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

// ===========================================================================
// Fused pack: W[N,K] fp32 -> tile-major planes in ONE pass (4 B read + packed bytes written per
// weight; the two-kernel path above moves 8 more bytes per weight through the u32 code buffer).
// One wave per 64(n) x 64(k) tile:
//   1. the tile is streamed in with coalesced 16-byte loads (4 rows x 256 B per wave instruction) and
//      transposed through LDS (row stride 68 words: conflict-free ds_read_b128) so that lane r holds row r;
//   2. lane r quantises its 64 / BS blocks with the fast block maths (one codec trip per element) and
//      writes packed nibbles / bytes / halves and the two E8M0 bytes per 8-k group back to LDS
//      (overlaying the fp32 tile);
//   3. every lane gathers its MFMA fragments (layout 1: n = 16 nf + (l & 15), k = 32 kf + 8 (l >> 4) + j)
//      and stores the slots: 64 lanes x 16 B = one coalesced 1 KiB store per slot.
// ===========================================================================
template <int BS, int IN_KIND, int OUT_KIND, int HW>
__global__ void __launch_bounds__(256)
k_pack_tile(const float* __restrict__ W, uint8_t* __restrict__ inl_plane, uint8_t* __restrict__ out_plane,
            uint8_t* __restrict__ scl_plane, OutlierArgs A, int64_t N, int64_t K) {
    constexpr int WAVE_LDS = 64 * 68 * 4;                       // 17408 B per wave
    constexpr int INL_STRIDE = 9;                                // dwords per row (8 used)
    constexpr int OUT_STRIDE = (OUT_KIND == MSQ_PLANE_BF16) ? 36 : 18;   // dwords per row (32 / 16 used)
    constexpr int INL_OFF = 0, SCL_OFF = 64 * INL_STRIDE * 4, OUT_OFF = SCL_OFF + 64 * 8 * 2;
    static_assert(OUT_OFF % 16 == 0 && OUT_OFF + 64 * OUT_STRIDE * 4 <= WAVE_LDS, "LDS overlay does not fit");
    __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t KT = K / 64, NT = N / 64;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    if (tile >= KT * NT) return;
    const int64_t nt = tile / KT, kt = tile % KT;
    char* wl = lds + wv * WAVE_LDS;
    float* ft = reinterpret_cast<float*>(wl);
    // 1. coalesced load + transpose
    {
        const float* src = W + (nt * 64) * K + kt * 64;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 4 + (lane >> 4), c4 = lane & 15;
            *reinterpret_cast<float4*>(ft + row * 68 + c4 * 4) = *reinterpret_cast<const float4*>(src + (int64_t)row * K + c4 * 4);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    float all[64];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(ft + lane * 68 + c * 4);
        all[c * 4 + 0] = v.x; all[c * 4 + 1] = v.y; all[c * 4 + 2] = v.z; all[c * 4 + 3] = v.w;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();                             // every lane has its row: the fp32 image is dead
    uint32_t* inlT = reinterpret_cast<uint32_t*>(wl + INL_OFF);
    uint16_t* sclT = reinterpret_cast<uint16_t*>(wl + SCL_OFF);
    uint32_t* outT = reinterpret_cast<uint32_t*>(wl + OUT_OFF);
    int status = 0;
    // 2. quantise the blocks of this row; codes come from the hardware converts (v_cvt_scalef32_pk_fp4_f32 /
    //    _fp8_f32: the quantised values are already on the grid, so the conversion is exact) and are verified
    //    by converting them back exactly as the GEMM will (v_cvt_scalef32_pk_bf16_*): value == code * 2^scale.
#pragma unroll
    for (int j = 0; j < 64 / BS; ++j) {
        float a[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) a[b] = all[j * BS + b];
        uint32_t mkw[(BS + 31) / 32];
        float se_in, se_out;
        status |= outlier_block_fast<BS, 0, false, HW>(a, mkw, se_in, se_out, A, /*inner order*/ 1, nullptr, nullptr, 1);
        uint32_t bi, bo;
        if (se_in != se_in) { bi = 255; status |= MSQ_STATUS_NAN; }
        else { const float t = se_in + 127.f; bi = (t < 1.f || t > 254.f) ? 255u : (uint32_t)t; if (t < 1.f || t > 254.f) status |= MSQ_STATUS_INEXACT; }
        const float ef = se_out - se_in;
        if (ef != ef) { bo = 255; status |= MSQ_STATUS_NAN; }
        else { const float t = ef + 127.f; bo = (t < 1.f) ? 1u : ((t > 254.f) ? 254u : (uint32_t)t); }
        const float sc_in = exp2f_int(se_in), rc_in = exp2f_int(-se_in), rc_out = exp2f_int(-se_out);
        const float s_in_op = u2f(bi << 23), s_out_op = u2f(bo << 23);
#pragma unroll
        for (int t8 = 0; t8 < BS / 8; ++t8) {
            const int k8 = j * (BS / 8) + t8;                    // 8-k group index inside the tile (0..7)
            uint32_t iw = 0, ow[4] = {0, 0, 0, 0};
            bool ok = true;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int e0 = t8 * 8 + 2 * p, e1 = e0 + 1;
                const bool m0 = (mkw[e0 >> 5] >> (e0 & 31)) & 1u, m1 = (mkw[e1 >> 5] >> (e1 & 31)) & 1u;
                const float v0 = a[e0], v1 = a[e1];
                uint32_t d = 0;
                if (IN_KIND != MSQ_PLANE_NONE) {
                    const float qi0 = (m0 ? 0.f : v0 * rc_in) + 0.0f, qi1 = (m1 ? 0.f : v1 * rc_in) + 0.0f;
                    if (p == 0) iw = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(iw, qi0, qi1, 1.0f, 0);
                    else if (p == 1) iw = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(iw, qi0, qi1, 1.0f, 1);
                    else if (p == 2) iw = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(iw, qi0, qi1, 1.0f, 2);
                    else iw = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(iw, qi0, qi1, 1.0f, 3);
                }
                if (IN_KIND == MSQ_PLANE_NONE) {
                    ow[p] = (f2u(v0) >> 16) | (f2u(v1) & 0xFFFF0000u);
                } else if (OUT_KIND == MSQ_PLANE_BF16) {
                    ow[p] = (m0 ? (f2u(v0) >> 16) : 0u) | (m1 ? (f2u(v1) & 0xFFFF0000u) : 0u);
                } else {
                    typedef short v2s_t __attribute__((ext_vector_type(2)));
                    const float qo0 = (m0 ? (v0 * sc_in) * rc_out : 0.f) + 0.0f, qo1 = (m1 ? (v1 * sc_in) * rc_out : 0.f) + 0.0f;
                    v2s_t cur = __builtin_bit_cast(v2s_t, ow[p >> 1]);
                    if (OUT_KIND == MSQ_PLANE_FP8) {
                        if ((p & 1) == 0) cur = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(cur, qo0, qo1, 1.0f, false);
                        else cur = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(cur, qo0, qo1, 1.0f, true);
                    } else {
                        if ((p & 1) == 0) cur = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(cur, qo0, qo1, 1.0f, false);
                        else cur = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(cur, qo0, qo1, 1.0f, true);
                    }
                    ow[p >> 1] = __builtin_bit_cast(uint32_t, cur);
                }
                // verify with the GEMM's own dequant
                if (IN_KIND == MSQ_PLANE_NONE) d = ow[p];
                else {
                    if (p == 0) d = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(iw, s_in_op, 0));
                    else if (p == 1) d = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(iw, s_in_op, 1));
                    else if (p == 2) d = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(iw, s_in_op, 2));
                    else d = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(iw, s_in_op, 3));
                    if (OUT_KIND == MSQ_PLANE_BF16) d |= ow[p];
                    else if (OUT_KIND == MSQ_PLANE_FP8)
                        d |= ((p & 1) == 0) ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(ow[p >> 1], s_out_op, false))
                                            : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(ow[p >> 1], s_out_op, true));
                    else
                        d |= ((p & 1) == 0) ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(ow[p >> 1], s_out_op, false))
                                            : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(ow[p >> 1], s_out_op, true));
                }
                const uint32_t expect = (f2u(v0) >> 16) | (f2u(v1) & 0xFFFF0000u);
                ok = ok && (d == expect) && (((f2u(v0) | f2u(v1)) & 0xFFFFu) == 0u);
            }
            if (!ok && !(status & MSQ_STATUS_NAN)) status |= MSQ_STATUS_INEXACT;
            if (IN_KIND != MSQ_PLANE_NONE) inlT[lane * INL_STRIDE + k8] = iw;
            sclT[lane * 8 + k8] = (uint16_t)(bi | (bo << 8));
            if (OUT_KIND == MSQ_PLANE_BF16) {
#pragma unroll
                for (int w = 0; w < 4; ++w) outT[lane * OUT_STRIDE + k8 * 4 + w] = ow[w];
            } else {
                outT[lane * OUT_STRIDE + k8 * 2] = ow[0];
                outT[lane * OUT_STRIDE + k8 * 2 + 1] = ow[1];
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // 3. fragment gather + slot stores
    const int c = lane & 15, g = lane >> 4;
    constexpr int OS = (OUT_KIND == MSQ_PLANE_BF16) ? 8 : 4;
    uint32_t sc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
        uint32_t inl4[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n = nf * 16 + c, k8 = kf * 4 + g;
            if (IN_KIND != MSQ_PLANE_NONE) inl4[nf] = inlT[n * INL_STRIDE + k8];
            const uint32_t s = sclT[n * 8 + k8];
            sc[nf] |= ((s & 0xFFu) << (16 * kf)) | ((s >> 8) << (16 * kf + 8));
            if (OUT_KIND == MSQ_PLANE_BF16) {
                const uint4 o = *reinterpret_cast<const uint4*>(outT + n * OUT_STRIDE + k8 * 4);
                *reinterpret_cast<uint4*>(out_plane + ((tile * OS + kf * 4 + nf) * 64 + lane) * 16) = o;
            } else {
                const uint2 o = *reinterpret_cast<const uint2*>(outT + n * OUT_STRIDE + k8 * 2);
                *reinterpret_cast<uint2*>(out_plane + ((tile * OS + kf * 2 + (nf >> 1)) * 64 + lane) * 16 + (nf & 1) * 8) = o;
            }
        }
        if (IN_KIND != MSQ_PLANE_NONE)
            *reinterpret_cast<uint4*>(inl_plane + ((tile * 2 + kf) * 64 + lane) * 16) = make_uint4(inl4[0], inl4[1], inl4[2], inl4[3]);
    }
    if (IN_KIND != MSQ_PLANE_NONE) {
        const bool per_lane = BS < 32;
        const int groups = per_lane ? 64 : 16;
        if (per_lane || g == 0)
            *reinterpret_cast<uint4*>(scl_plane + (tile * groups + (per_lane ? lane : c)) * 16) = make_uint4(sc[0], sc[1], sc[2], sc[3]);
    }
    if (status && A.status) atomicOr(A.status, status);
}

// ===========================================================================
// Fused pack, unified layout (MSQ-U1): as k_pack_tile, but every 32-k half row gets ONE scale and every
// weight one e4m3 code (+ one extension bit when EXT: the 4th fraction bit of posit<8,1> outliers).
//   scale s = floor(log2(max|v|)) - 8 (+1 when max|v| 2^-s > 448): when an e4m3 outlier holds the maximum
//   this is the outlier's own scale, so its codes are unchanged; inlier e2m1 values widen exactly.
// Every code is decoded back with the GEMM's own instruction and compared with the fake-quant value.
// ===========================================================================
// BS == 0: the input already holds fake-quant VALUES (any block direction, GPTQ output, ...): no quantiser runs,
// the values are only encoded and checked (msq_pack_values).
template <int BS, bool EXT, int HW>
__global__ void __launch_bounds__(256, 2)
k_pack_tile_u(const float* __restrict__ W, uint8_t* __restrict__ ext_plane, uint8_t* __restrict__ code_plane,
              uint8_t* __restrict__ scl_plane, OutlierArgs A, int64_t N, int64_t K) {
    // Lane r owns row r of the tile in LDS (68 words = 272 B per row) and works on it IN PLACE, one block /
    // one 32-k half at a time, so that no 64-element register array is needed: the fake-quant values
    // overwrite the inputs; the codes of half h then go to bytes 32 h .. 32 h + 31 of the row (already
    // consumed), the two scale bytes to 64..65 and the extension bytes to 68..75 once both halves are encoded.
    constexpr int ROW_W = 68;                                    // words per row
    constexpr int WAVE_LDS = 64 * ROW_W * 4;
    __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t KT = K / 64, NT = N / 64;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    if (tile >= KT * NT) return;
    const int64_t nt = tile / KT, kt = tile % KT;
    float* ft = reinterpret_cast<float*>(lds + wv * WAVE_LDS);
    {
        const float* src = W + (nt * 64) * K + kt * 64;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 4 + (lane >> 4), c4 = lane & 15;
            *reinterpret_cast<float4*>(ft + row * ROW_W + c4 * 4) = *reinterpret_cast<const float4*>(src + (int64_t)row * K + c4 * 4);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    float* myrow = ft + lane * ROW_W;
    int status = 0;
    // 1. fake-quant of the row, block by block, in place
    if constexpr (BS > 0) {
        constexpr int B = BS > 0 ? BS : 64;
#pragma nounroll
        for (int j = 0; j < 64 / B; ++j) {
            float a[B];
#pragma unroll
            for (int c = 0; c < B / 4; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(myrow + j * B + c * 4);
                a[c * 4 + 0] = v.x; a[c * 4 + 1] = v.y; a[c * 4 + 2] = v.z; a[c * 4 + 3] = v.w;
            }
            uint32_t mkw[(B + 31) / 32];
            float se_in, se_out;
            status |= outlier_block_fast<B, 0, false, HW>(a, mkw, se_in, se_out, A, /*inner order*/ 1, nullptr, nullptr, 1);
#pragma unroll
            for (int c = 0; c < B / 4; ++c)
                *reinterpret_cast<float4*>(myrow + j * B + c * 4) = make_float4(a[c * 4 + 0], a[c * 4 + 1], a[c * 4 + 2], a[c * 4 + 3]);
        }
    }
    // 2. one scale + 32 codes (+ 32 extension bits) per half
    uint32_t sbytes = 0, eb_lo = 0, eb_hi = 0;
#pragma nounroll
    for (int h = 0; h < 2; ++h) {
        float v[32];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 x = *reinterpret_cast<const float4*>(myrow + h * 32 + c * 4);
            v[c * 4 + 0] = x.x + 0.0f; v[c * 4 + 1] = x.y + 0.0f; v[c * 4 + 2] = x.z + 0.0f; v[c * 4 + 3] = x.w + 0.0f;   // -0 -> +0
        }
        float mx = 0.f;
        bool bad = false;
#pragma unroll
        for (int b = 0; b < 32; ++b) { const float t = __builtin_fabsf(v[b]); mx = t > mx ? t : mx; bad |= !(t == t) || t > 3.0e38f; }
        int su = 0;
        if (mx > 0.f) {
            su = ilog2f(mx) - 8;
            if (__builtin_ldexpf(mx, -su) > 448.f) su += 1;      // mx * 2^-su lies in [256, 512): exact
        }
        su = su < -126 ? -126 : su;
        if (su > 127) { su = 127; status |= MSQ_STATUS_INEXACT; }
        if (bad) status |= MSQ_STATUS_NAN;
        const uint32_t sb = (uint32_t)(su + 127);
        const float s_op = u2f(sb << 23);
        sbytes |= sb << (8 * h);
        uint32_t cw[8];
        uint32_t eb = 0;                                         // bit j of byte t8: element 8 t8 + j
        bool ok = true;
#pragma unroll
        for (int p = 0; p < 16; ++p) {                           // pairs of elements
            typedef short v2s_t __attribute__((ext_vector_type(2)));
            uint32_t u0 = f2u(v[2 * p]), u1 = f2u(v[2 * p + 1]);
            const uint32_t expect = (u0 >> 16) | (u1 & 0xFFFF0000u);
            ok = ok && (((u0 | u1) & 0xFFFFu) == 0u);
            if (EXT) {                                           // split off bf16 mantissa bit 3 (f32 bit 19)
                eb |= ((u0 >> 19) & 1u) << (2 * p);
                eb |= ((u1 >> 19) & 1u) << (2 * p + 1);
                u0 &= ~(1u << 19); u1 &= ~(1u << 19);
            }
            // the convert divides by 2^su itself (only the exponent field of the scale operand is read)
            v2s_t cur = __builtin_bit_cast(v2s_t, (p & 1) ? cw[p >> 1] : 0u);
            if ((p & 1) == 0) cur = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(cur, u2f(u0), u2f(u1), s_op, false);
            else cur = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(cur, u2f(u0), u2f(u1), s_op, true);
            cw[p >> 1] = __builtin_bit_cast(uint32_t, cur);
            // decode exactly as the GEMM will and compare with the fake-quant value
            uint32_t d = ((p & 1) == 0) ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(cw[p >> 1], s_op, false))
                                        : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(cw[p >> 1], s_op, true));
            if (EXT) d |= (((eb >> (2 * p)) & 1u) << 3) | (((eb >> (2 * p + 1)) & 1u) << 19);
            ok = ok && (d == expect);
        }
        if (!ok && !bad) status |= MSQ_STATUS_INEXACT;
        uint32_t* crow = reinterpret_cast<uint32_t*>(myrow) + h * 8;     // bytes 32 h ..
#pragma unroll
        for (int c = 0; c < 2; ++c)
            *reinterpret_cast<uint4*>(crow + c * 4) = make_uint4(cw[c * 4], cw[c * 4 + 1], cw[c * 4 + 2], cw[c * 4 + 3]);
        if (h == 0) eb_lo = eb; else eb_hi = eb;
    }
    reinterpret_cast<uint32_t*>(myrow)[16] = sbytes;             // bytes 64..65: the two scales
    if (EXT) { reinterpret_cast<uint32_t*>(myrow)[17] = eb_lo; reinterpret_cast<uint32_t*>(myrow)[18] = eb_hi; }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // 3. fragment gather + slot stores
    const int c = lane & 15, g = lane >> 4;
    const uint32_t* rows = reinterpret_cast<const uint32_t*>(ft);
    uint32_t sc[2] = {0, 0};
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
        uint32_t ew = 0;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n = nf * 16 + c, k8 = kf * 4 + g;
            const uint32_t* r = rows + n * ROW_W;
            sc[kf] |= ((r[16] >> (8 * kf)) & 0xFFu) << (8 * nf);
            const uint2 o = *reinterpret_cast<const uint2*>(r + k8 * 2);
            *reinterpret_cast<uint2*>(code_plane + ((tile * 4 + kf * 2 + (nf >> 1)) * 64 + lane) * 16 + (nf & 1) * 8) = o;
            if (EXT) {
                const uint32_t eb = (r[17 + kf] >> (8 * g)) & 0xFFu;      // elements 8 g .. 8 g + 7 of half kf
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    ew |= ((eb >> j) & 1u) << ((3 + 16 * (j & 1) + 4 * nf + (j >> 1)) & 31);
            }
        }
        if (EXT) *reinterpret_cast<uint32_t*>(ext_plane + ((tile * 2 + kf) * 64 + lane) * 4) = ew;
    }
    if (g == 0) *reinterpret_cast<uint2*>(scl_plane + (tile * 16 + c) * 8) = make_uint2(sc[0], sc[1]);
    if (status && A.status) atomicOr(A.status, status);
}

// returns MSQ_ERR_UNSUPPORTED when the configuration needs the generic two-kernel path
extern "C" int msq_pack_fused_(const float* W, void* inl_plane, void* out_plane, void* scale_plane, int* status,
                               int64_t N, int64_t K, int block, int inlier_fmt, int outlier_fmt, int in_sb, int out_sb,
                               float std_dev, int rmode, int flush, int in_kind, int out_kind, void* stream) {
    msq_host::FmtInfo fi, fo;
    if (!msq_host::format_info(inlier_fmt, &fi) || !msq_host::format_info(outlier_fmt, &fo)) return MSQ_ERR_UNSUPPORTED;
    if (fi.kind != 0 || rmode != 0 || !(block == 8 || block == 16 || block == 32 || block == 64)) return MSQ_ERR_UNSUPPORTED;
    if (in_sb <= 0 || out_sb <= 0 || in_sb > 8 || out_sb > 8) return MSQ_ERR_UNSUPPORTED;
    OutlierArgs A;
    A.fi = Fmt{fi.kind, fi.ebits, fi.mbits, fi.emax, fi.max_norm};
    A.fo = Fmt{fo.kind, fo.ebits, fo.mbits, fo.emax, fo.max_norm};
    A.in_sb = in_sb; A.out_sb = out_sb; A.k = std_dev; A.rmode = rmode; A.flush = flush; A.variant = 0;
    A.pre = N; A.axis_len = K; A.post = 1; A.nblk = K / block;
    A.mask = nullptr; A.e_in = nullptr; A.e_out = nullptr; A.n_out = nullptr; A.status = status;
    A.vmean = nullptr; A.vstd = nullptr;
    const int64_t tiles = (N / 64) * (K / 64);
    const dim3 grid((unsigned)((tiles + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    // quantiser codec: 1 = hardware converts for inliers and outliers, 2 = for the inliers (posit outliers), 0 = arithmetic
    const int ih = hw_codec_kind(A.fi), oh = hw_codec_kind(A.fo);
    const int hw = (ih && oh) ? 1 : ((ih && fo.kind == 1) ? 2 : 0);
#define MSQ_PT(BS, IK, OK, HWV) hipLaunchKernelGGL((k_pack_tile<BS, IK, OK, HWV>), grid, blk, 0, st, W, (uint8_t*)inl_plane, \
                                                   (uint8_t*)out_plane, (uint8_t*)scale_plane, A, N, K)
#define MSQ_PTB(IK, OK, HWV) do { switch (block) { case 8: MSQ_PT(8, IK, OK, HWV); break; case 16: MSQ_PT(16, IK, OK, HWV); break; \
                                                   case 32: MSQ_PT(32, IK, OK, HWV); break; default: MSQ_PT(64, IK, OK, HWV); break; } } while (0)
    if (in_kind == MSQ_PLANE_NONE && out_kind == MSQ_PLANE_BF16) MSQ_PTB(MSQ_PLANE_NONE, MSQ_PLANE_BF16, 0);
    else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_FP8) { if (hw == 1) MSQ_PTB(MSQ_PLANE_FP4, MSQ_PLANE_FP8, 1); else MSQ_PTB(MSQ_PLANE_FP4, MSQ_PLANE_FP8, 0); }
    else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_BF8) { if (hw == 1) MSQ_PTB(MSQ_PLANE_FP4, MSQ_PLANE_BF8, 1); else MSQ_PTB(MSQ_PLANE_FP4, MSQ_PLANE_BF8, 0); }
    else if (in_kind == MSQ_PLANE_FP4 && out_kind == MSQ_PLANE_BF16) { if (hw == 2) MSQ_PTB(MSQ_PLANE_FP4, MSQ_PLANE_BF16, 2); else MSQ_PTB(MSQ_PLANE_FP4, MSQ_PLANE_BF16, 0); }
    else if (in_kind == MSQ_PLANE_NONE && (out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X)) {
#define MSQ_PU(BS, EXTV, HWV) hipLaunchKernelGGL((k_pack_tile_u<BS, EXTV, HWV>), grid, blk, 0, st, W, (uint8_t*)inl_plane, \
                                                 (uint8_t*)out_plane, (uint8_t*)scale_plane, A, N, K)
#define MSQ_PUB(EXTV, HWV) do { switch (block) { case 8: MSQ_PU(8, EXTV, HWV); break; case 16: MSQ_PU(16, EXTV, HWV); break; \
                                                 case 32: MSQ_PU(32, EXTV, HWV); break; default: MSQ_PU(64, EXTV, HWV); break; } } while (0)
        if (out_kind == MSQ_PLANE_U8) { if (hw == 1) MSQ_PUB(false, 1); else MSQ_PUB(false, 0); }
        else { if (hw == 2) MSQ_PUB(true, 2); else MSQ_PUB(true, 0); }
#undef MSQ_PUB
#undef MSQ_PU
    }
    else return MSQ_ERR_UNSUPPORTED;
#undef MSQ_PTB
#undef MSQ_PT
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

// values -> unified planes (no quantiser): out_kind MSQ_PLANE_U8 or MSQ_PLANE_U8X
extern "C" int msq_pack_values_u_(const float* W, void* ext_plane, void* code_plane, void* scale_plane, int* status,
                                  int64_t N, int64_t K, int out_kind, void* stream) {
    OutlierArgs A = {};
    A.status = status;
    const int64_t tiles = (N / 64) * (K / 64);
    const dim3 grid((unsigned)((tiles + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (out_kind == MSQ_PLANE_U8) hipLaunchKernelGGL((k_pack_tile_u<0, false, 0>), grid, blk, 0, st, W, (uint8_t*)ext_plane, (uint8_t*)code_plane, (uint8_t*)scale_plane, A, N, K);
    else hipLaunchKernelGGL((k_pack_tile_u<0, true, 0>), grid, blk, 0, st, W, (uint8_t*)ext_plane, (uint8_t*)code_plane, (uint8_t*)scale_plane, A, N, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

