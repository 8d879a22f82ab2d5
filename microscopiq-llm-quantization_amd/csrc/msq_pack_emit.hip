// msq_pack_emit.hip -- msq_outlier_pack, two-plane layout MSQ-T1, single pass: the MicroScopiQ quantiser
// (utils/quant.py:147-266, blocks along K) fused with the plane emission.  Own translation unit (heavy templates).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"
#include "msq_outlier_core.h"

extern "C" void msq_set_error_(const char* msg);

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

// implemented in msq_pack_unified.hip
extern "C" int msq_pack_unified_(const float* W, void* ext_plane, void* code_plane, void* scale_plane, const OutlierArgs* A,
                                 int64_t N, int64_t K, int block, int out_kind, int hw, void* stream);

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
    else if (in_kind == MSQ_PLANE_NONE && (out_kind == MSQ_PLANE_U8 || out_kind == MSQ_PLANE_U8X))
        return msq_pack_unified_(W, inl_plane, out_plane, scale_plane, &A, N, K, block, out_kind, hw, stream);
    else return MSQ_ERR_UNSUPPORTED;
#undef MSQ_PTB
#undef MSQ_PT
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}
