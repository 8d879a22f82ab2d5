// msq_pack_twopass.hip -- first half of the generic two-kernel msq_outlier_pack path (posit inliers, non-nearest
// rounding, block 128, the mx_ops variant): the MicroScopiQ quantiser (utils/quant.py:147-266, blocks along K)
// emitting per-element plane codes and the per-block exponents; k_repack (msq_gemm.hip) then builds the planes.
// One block per lane, own translation unit (heavy templates).
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

