// msq_act.hip -- activation side of the W4A8 Linear (number_system/mx/linear.py:66-73): MicroScopiQ
// outlier-aware MX quantisation of X [M, K] along K, result written as bf16 for the fused GEMM.
// Every quantised value is code * 2^scale with an <= 8-bit code, so bf16 holds it exactly (checked:
// MSQ_STATUS_INEXACT otherwise).  HBM-bound: 4 B read + 2 B written per element.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"
#include "msq_outlier_core.h"

extern "C" void msq_set_error_(const char* msg);
extern "C" int msq_mxops_stats_(const float* in, float* vmean, float* vstd, int64_t pre, int64_t axis_len, int64_t post,
                                int block, int* status, void* stream);

namespace {

MSQ_D uint32_t bf16_bits_exact(float v, int& status) {
    const uint32_t u = f2u(v);
    if ((u & 0xFFFFu) && v == v) status |= MSQ_STATUS_INEXACT;
    return u >> 16;
}

// A wave owns 64 consecutive blocks = one contiguous run of 64*BS floats: coalesced 16-byte loads,
// transpose through LDS (row stride BS+4 words), one block per lane, and back the same way as bf16.
template <int BS, int RM, int HW>
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
    if (full) {
        const float4* src = reinterpret_cast<const float4*>(X + g0 * BS);
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
        for (int b = 0; b < BS; ++b) a[b] = (gidx < nblocks) ? X[gidx * BS + b] : 0.f;
    }
    uint32_t mkw[(BS + 31) / 32];
    float se_in = 0.f, se_out = 0.f;
    int status = 0;
    if (gidx < nblocks) {
        if (A.vmean && full && (A.nblk % 64 == 0)) {
            // the 64 blocks of this wave lie in one row: the statistics tables are wave-uniform, read them
            // through scalar loads (the mean / std of position b are the same for every lane)
            const int64_t prow = g0 / A.nblk;
            const uint64_t pm = (uint64_t)(A.vmean + prow * BS), ps = (uint64_t)(A.vstd + prow * BS);
            auto uni64 = [](uint64_t v) -> uint64_t {            // readfirstlane returns a signed int: widen through uint32_t
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
                return ((uint64_t)hi << 32) | (uint64_t)lo;
            };
            const float* vm = (const float*)uni64(pm);
            const float* vs = (const float*)uni64(ps);
            status = outlier_block_fast<BS, RM, false, HW>(a, mkw, se_in, se_out, A, /*inner order*/ 1, vm, vs, 1);
        } else {
            const float* vm = A.vmean ? A.vmean + (gidx / A.nblk) * BS : nullptr;
            const float* vs = A.vstd ? A.vstd + (gidx / A.nblk) * BS : nullptr;
            status = outlier_block_fast<BS, RM, false, HW>(a, mkw, se_in, se_out, A, /*inner order*/ 1, vm, vs, 1);
        }
    }
    uint32_t h[BS / 2];
#pragma unroll
    for (int b = 0; b < BS / 2; ++b) h[b] = bf16_bits_exact(a[2 * b], status) | (bf16_bits_exact(a[2 * b + 1], status) << 16);
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

}  // namespace

extern "C" int64_t msq_act_quant_workspace_bytes(int64_t M, int64_t K, int block, int variant) {
    if (variant != MSQ_VARIANT_MXOPS || M <= 0 || K <= 0 || block <= 0) return 0;
    return 2 * (int64_t)sizeof(float) * M * block;
}

extern "C" int msq_act_quant_bf16(const float* X, void* Xq, int* status_flag, void* workspace, int64_t workspace_bytes,
                                  int64_t M, int64_t K, int block, int inlier_fmt, int outlier_fmt, int inlier_scale_bits,
                                  int outlier_scale_bits, float std_dev, int rmode, int flush_fp32_subnorms, int variant,
                                  void* stream) {
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
        const int rc = msq_mxops_stats_(X, vmean, vstd, M, K, 1, block, status_flag, stream);
        if (rc) return rc;
        A.vmean = vmean; A.vstd = vstd;
    }
    const int64_t nblocks = M * A.nblk;
    const dim3 grid((unsigned)((nblocks + 255) / 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    // RM = 0: round-to-nearest (half away) specialised, through the hardware converts when both formats have
    // one (e4m3 / e5m2 / e2m1); -1: any rounding mode, arithmetic codec
    const bool hw = (rmode == 0) && hw_codec_kind(A.fi) && hw_codec_kind(A.fo);
#define MSQ_AQ(BS) do { if (hw) hipLaunchKernelGGL((k_act_quant<BS, 0, 1>), grid, blk, 0, st, X, (uint16_t*)Xq, A); \
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
