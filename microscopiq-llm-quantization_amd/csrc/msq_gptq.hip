// msq_gptq.hip -- one launch per GPTQ column block (reference: llm/gptq.py:106-165).
//
// The reference walks the columns of a 128-column block one by one: quantise the column with
// quantize_mx_outlier_hessian (utils/quant.py:23-146: blocks of 16 consecutive OUTPUT rows), zero the `num_outliers`
// least important entries (importance q^2 / d^2, torch.topk(..., largest=False), llm/gptq.py:144-150), and feed the
// error (w - q) / d to every column to the right (W1[:, i:] -= err (x) Hinv1[i, i:], :155-157) -- ~25 eager launches per
// column.  Here the whole block is ONE launch:
//   * one lane per output row, 256 rows per workgroup, ceil(O / 256) workgroups (one per CU);
//   * the error columns of the block stay in LDS ([128][256] floats); column j is rebuilt lazily as
//     w = W0[:, j] - sum_{k<j} err_k * U[k, j], one rounded product and one rounded subtraction per k in the order
//     k = 0, 1, ... -- exactly the arithmetic of the reference's running rank-1 updates, without ever writing the
//     updated block back;
//   * the column entries of a quantisation block (BS consecutive rows = BS adjacent lanes) are gathered with lane
//     shuffles and quantised by the same register-resident block maths as msq_outlier_fakequant (bit-identical);
//   * the two column-wide quantities -- the outlier count (with the reference's every-BS-th-block quirk, utils/quant.py:66)
//     and the pruning choice -- cross workgroups through 8-byte agent-scope atomics and one counter barrier per column.
//     Pruning "the n smallest importances" is a no-op whenever at least n entries are already zero (the usual case for
//     int2 / fp4 inliers); otherwise an exact radix select over the importance bits runs (4 digit passes + an index-ordered
//     tie pass).  Ties are broken by the LOWEST ROW INDEX (torch.topk leaves the choice among equal values unspecified;
//     tests/golden/gptq_exact.npz holds the reference solver's output under this rule).
// Rows are independent except for those two reductions, so nothing else is exchanged between workgroups.
// Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"
#include "msq_outlier_core.h"

using namespace msq;

namespace {

constexpr int GPTQ_WG = 256;          // rows per workgroup
constexpr int GPTQ_MAXC = 128;        // columns per launch
constexpr int GPTQ_BARS = 6;          // counter barriers per column (1 + 4 digit passes + tie pass)

struct GptqArgs {
    const float* Wt;      // [cols][O]  current values of the block's columns (column-major scratch)
    const float* U;       // upper Cholesky factor of H^-1, pointing at U[c0][c0], row stride ldu
    float* Qt;            // [cols][O]  quantised + pruned columns
    float* Et;            // [cols][O]  error columns (w - q) / d
    double* loss;         // += sum (w - q)^2 / d^2 / 2
    unsigned long long* pruned;   // += entries zeroed by the pruning step
    int* status;
    unsigned long long* colstat;  // [cols] packed: outliers (bits 0-23) | zero entries (24-47) | "zero importance, nonzero q" flag (bit 48+)
    unsigned int* bars;           // [cols][GPTQ_BARS] arrival counters (zeroed by the caller)
    unsigned int* hist;           // [cols][4][256] digit histograms of the radix select (zeroed by the caller)
    unsigned int* ties;           // [cols][nwg] entries equal to the threshold per workgroup
    int64_t O;
    int cols, ldu, nwg;
    OutlierArgs qa;
};

MSQ_D unsigned long long aload64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
MSQ_D unsigned int aload32(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// All workgroups of the launch must be co-resident: msq_gptq_block checks the grid against the device's CU count and the
// occupancy of this kernel at its LDS size before launching (and refuses otherwise), but it cannot see other streams or
// processes that hold CUs, so the spin is BOUNDED: after ~2^23 polls (seconds) the waiter raises MSQ_STATUS_TIMEOUT and goes on
// -- the launch then ends with a wrong result and a status bit instead of hanging the device.  Arrive after this workgroup's
// published atomics have RETURNED (their values are consumed), then poll with relaxed agent-scope loads.
MSQ_D void grid_barrier(unsigned int* ctr, int nwg, int tid, int* status) {
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_s_waitcnt(0x0070);                                  // vmcnt(0) lgkmcnt(0): every earlier atomic has completed
        const unsigned int old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)old + 1 < nwg) {
            unsigned int spins = 0;
            while ((int)aload32(ctr) < nwg) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 23)) { if (status) atomicOr(status, MSQ_STATUS_TIMEOUT); break; }
            }
        }
    }
    __syncthreads();
}

template <int BS>
__global__ void __launch_bounds__(GPTQ_WG)
k_gptq_block(GptqArgs A) {
    extern __shared__ float err_lds[];                    // [cols][GPTQ_WG]
    __shared__ unsigned int s_cnt[4];                     // outliers, zero entries, flag, scratch
    __shared__ unsigned int s_hist[256];
    __shared__ unsigned int s_wave[GPTQ_WG / 64];
    __shared__ unsigned int s_sel[2];
    const int tid = threadIdx.x, lane = tid & 63, li = lane % BS;
    const int64_t r = (int64_t)blockIdx.x * GPTQ_WG + tid;
    const bool live = r < A.O;
    const int64_t nb = r / BS;                             // quantisation block index along the column
    double loss = 0.0;
    unsigned long long n_pruned = 0;
    int status = 0;
    for (int j = 0; j < A.cols; ++j) {
        // ---- the column as the reference's running updates leave it
        float w = live ? A.Wt[(int64_t)j * A.O + r] : 0.f;
        for (int k = 0; k < j; ++k) {
            const float t = err_lds[k * GPTQ_WG + tid] * A.U[(int64_t)k * A.ldu + j];      // llm/gptq.py:156 (outer product entry)
            w = w - t;
        }
        const float d = A.U[(int64_t)j * A.ldu + j];
        // ---- quantise the BS rows of this lane's block (zero padded past O, utils/quant.py:563-583)
        float a[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) a[i] = __shfl(w, (lane / BS) * BS + i, 64);
        uint32_t mkw[(BS + 31) / 32];
        float se_in, se_out;
        status |= outlier_block_fast<BS, -1>(a, mkw, se_in, se_out, A.qa, 1, nullptr, nullptr, 1);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < BS; ++i) q = (i == li) ? a[i] : q;
        const float d2 = d * d;
        const float imp = (q * q) / d2;                                                    // :144
        // ---- column totals: outliers of every BS-th block (utils/quant.py:66), zero entries, awkward-zero flag
        if (tid < 4) s_cnt[tid] = 0;
        __syncthreads();
        {
            unsigned int c = 0;
            if (li == 0 && live && (nb % BS) == 0) {
#pragma unroll
                for (int wd = 0; wd < (BS + 31) / 32; ++wd) c += __builtin_popcount(mkw[wd]);
                c &= 0xFF;                                                                 // int8 per block (:66), <= BS anyway
            }
            const unsigned int z = (live && imp == 0.f && q == 0.f) ? 1u : 0u;
            const unsigned int f = (live && imp == 0.f && q != 0.f) ? 1u : 0u;
            if (c) atomicAdd(&s_cnt[0], c);
            const unsigned long long zb = __ballot(z != 0), fb = __ballot(f != 0);
            if (lane == 0) { if (zb) atomicAdd(&s_cnt[1], (unsigned int)__builtin_popcountll(zb)); if (fb) atomicAdd(&s_cnt[2], 1u); }
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned long long pk = (unsigned long long)s_cnt[0] | ((unsigned long long)s_cnt[1] << 24) | ((unsigned long long)(s_cnt[2] ? 1 : 0) << 48);
            const unsigned long long old = __hip_atomic_fetch_add(&A.colstat[j], pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(old));
        }
        grid_barrier(&A.bars[j * GPTQ_BARS + 0], A.nwg, tid, A.status);
        if (tid == 0) {
            const unsigned long long tot = aload64(&A.colstat[j]);
            s_sel[0] = (unsigned int)(tot & 0xFFFFFFu);
            s_sel[1] = (unsigned int)((tot >> 24) & 0xFFFFFFu) | (((tot >> 48) != 0) ? 0x80000000u : 0u);
        }
        __syncthreads();
        int n = (int)(int16_t)(uint16_t)s_sel[0];                                          // .sum().to(torch.int16), :145
        const unsigned int zeros = s_sel[1] & 0x7FFFFFFFu;
        const bool awkward = (s_sel[1] >> 31) != 0;
        if (n > (int)A.O) n = (int)A.O;
        bool drop = false;
        if (n > 0 && (awkward || (unsigned int)n > zeros)) {
            // ---- exact selection of the n smallest importances, ties by lowest row index: radix select on the bits
            const uint32_t key = (imp != imp) ? 0xFFFFFFFFu : f2u(imp);                    // NaN counts as the largest
            uint32_t prefix = 0, pmask = 0;
            unsigned int remaining = (unsigned int)n;
            for (int p = 3; p >= 0; --p) {
                s_hist[tid] = 0;
                __syncthreads();
                const bool cand = live && ((key & pmask) == prefix);
                if (cand) atomicAdd(&s_hist[(key >> (8 * p)) & 0xFF], 1u);
                __syncthreads();
                unsigned int* gh = A.hist + ((int64_t)j * 4 + p) * 256;
                if (s_hist[tid]) { const unsigned int o = __hip_atomic_fetch_add(&gh[tid], s_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("" ::"v"(o)); }
                grid_barrier(&A.bars[j * GPTQ_BARS + 1 + (3 - p)], A.nwg, tid, A.status);
                s_hist[tid] = aload32(&gh[tid]);
                __syncthreads();
                if (tid == 0) {
                    unsigned int cum = 0, dsel = 255;
                    for (int b = 0; b < 256; ++b) { if (cum + s_hist[b] >= remaining) { dsel = b; break; } cum += s_hist[b]; }
                    s_sel[0] = dsel; s_sel[1] = cum;
                }
                __syncthreads();
                prefix |= s_sel[0] << (8 * p); pmask |= 0xFFu << (8 * p);
                remaining -= s_sel[1];
                __syncthreads();
            }
            // entries below the threshold go; `remaining` entries equal to it go in row order
            const bool below = live && key < prefix;
            const bool tie = live && key == prefix;
            const unsigned long long tb = __ballot(tie);
            if (lane == 0) s_wave[tid >> 6] = (unsigned int)__builtin_popcountll(tb);
            __syncthreads();
            unsigned int wg_ties = 0, before = 0;
            for (int wv = 0; wv < GPTQ_WG / 64; ++wv) { if (wv < (tid >> 6)) before += s_wave[wv]; wg_ties += s_wave[wv]; }
            if (tid == 0) __hip_atomic_store(&A.ties[(int64_t)j * A.nwg + blockIdx.x], wg_ties, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            grid_barrier(&A.bars[j * GPTQ_BARS + 5], A.nwg, tid, A.status);
            if (tid == 0) {
                unsigned int base = 0;
                for (int g = 0; g < (int)blockIdx.x; ++g) base += aload32(&A.ties[(int64_t)j * A.nwg + g]);
                s_sel[0] = base;
            }
            __syncthreads();
            const unsigned int rank = s_sel[0] + before + (unsigned int)__builtin_popcountll(tb & ((1ull << lane) - 1ull));
            drop = below || (tie && rank < remaining);
            __syncthreads();
        }
        if (drop) { if (q != 0.f) ++n_pruned; q = 0.f; }                                   // :150
        // ---- error feedback (:152-157)
        const float diff = w - q;
        const float e = diff / d;
        err_lds[j * GPTQ_WG + tid] = e;
        if (live) {
            A.Qt[(int64_t)j * A.O + r] = q;
            A.Et[(int64_t)j * A.O + r] = e;
            loss += (double)((diff * diff) / d2) * 0.5;                                    // Losses1 / 2
        }
        __syncthreads();
    }
    // ---- per-launch totals
    for (int o = 32; o > 0; o >>= 1) { loss += __shfl_xor(loss, o, 64); n_pruned += __shfl_xor(n_pruned, o, 64); }
    if (lane == 0) {
        if (loss != 0.0) atomicAdd(A.loss, loss);
        if (n_pruned) atomicAdd(A.pruned, n_pruned);
    }
    if (status && A.status) atomicOr(A.status, status);
}

}  // namespace

extern "C" void msq_set_error_(const char* msg);
static int gfail(int code, const char* msg) { msq_set_error_(msg); return code; }

extern "C" {

int64_t msq_gptq_block_workspace_bytes(int64_t O, int cols) {
    if (O <= 0 || cols <= 0) return 0;
    const int64_t nwg = (O + GPTQ_WG - 1) / GPTQ_WG;
    return (int64_t)cols * 8 + (int64_t)cols * GPTQ_BARS * 4 + (int64_t)cols * 4 * 256 * 4 + (int64_t)cols * nwg * 4;
}

int msq_gptq_block(const float* Wt, const float* U, int ldu, float* Qt, float* Et, double* loss, unsigned long long* pruned,
                   int* status_flag, void* workspace, int64_t workspace_bytes, int64_t O, int cols, int block,
                   int inlier_fmt, int outlier_fmt, int inlier_scale_bits, int outlier_scale_bits, float std_dev, int rmode,
                   int flush_fp32_subnorms, void* stream) {
    if (O <= 0 || cols <= 0) return (O == 0 || cols == 0) ? MSQ_OK : gfail(MSQ_ERR_BAD_ARG, "msq_gptq_block: negative size");
    if (!Wt || !U || !Qt || !Et || !loss || !pruned || !workspace) return gfail(MSQ_ERR_BAD_ARG, "msq_gptq_block: null buffer");
    if (cols > GPTQ_MAXC) return gfail(MSQ_ERR_UNSUPPORTED, "msq_gptq_block: at most 128 columns per launch");
    if (workspace_bytes < msq_gptq_block_workspace_bytes(O, cols)) return gfail(MSQ_ERR_BAD_ARG, "msq_gptq_block: workspace too small (msq_gptq_block_workspace_bytes)");
    if (inlier_scale_bits <= 0 || outlier_scale_bits <= 0 || inlier_scale_bits > 8 || outlier_scale_bits > 8)
        return gfail(MSQ_ERR_BAD_ARG, "msq_gptq_block: scale bits must be in [1,8]");
    if (rmode < 0 || rmode > 2) return gfail(MSQ_ERR_BAD_ARG, "msq_gptq_block: bad rounding mode");
    msq_host::FmtInfo fi, fo;
    if (!msq_host::format_info(inlier_fmt, &fi) || !msq_host::format_info(outlier_fmt, &fo))
        return gfail(MSQ_ERR_BAD_ARG, "msq_gptq_block: unknown element format");
    if (fi.kind != 0) return gfail(MSQ_ERR_UNSUPPORTED, "msq_gptq_block: posit inliers take the per-column path");
    const int64_t nwg = (O + GPTQ_WG - 1) / GPTQ_WG;
    if (nwg > 200) return gfail(MSQ_ERR_UNSUPPORTED, "msq_gptq_block: more than 51200 output rows");   // all workgroups must be co-resident
    const size_t lds = (size_t)cols * GPTQ_WG * 4;
    // ... on THIS device: the hand-rolled grid barrier needs every workgroup resident at once.  Ask the runtime how many blocks
    // of this kernel fit a CU at this LDS size and how many CUs the device (or partition: CPX exposes 32) has; the occupancy API
    // can answer one block per CU high (MI355X_MICROARCH.md, residency), so one is taken off whenever it says more than one.
    // A grid that does not fit is refused -- harness/gptq.py then takes the per-column path.
    {
        int dev = 0, cus = 0, per_cu = 0;
        const void* kfn = nullptr;
        switch (block) {
            case 8: kfn = (const void*)k_gptq_block<8>; break;
            case 16: kfn = (const void*)k_gptq_block<16>; break;
            case 32: kfn = (const void*)k_gptq_block<32>; break;
            case 64: kfn = (const void*)k_gptq_block<64>; break;
            default: return gfail(MSQ_ERR_UNSUPPORTED, "msq_gptq_block: quantiser block must be 8, 16, 32 or 64");
        }
        if (lds > 65536 && hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return gfail(MSQ_ERR_UNSUPPORTED, "msq_gptq_block: the device refuses this much dynamic LDS per workgroup");
        }
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, GPTQ_WG, lds) != hipSuccess) {
            (void)hipGetLastError();
            return gfail(MSQ_ERR_LAUNCH, "msq_gptq_block: occupancy query failed");
        }
        if (per_cu > 1) --per_cu;
        if (per_cu < 1 || nwg > (int64_t)cus * per_cu) {
            char b[200];
            snprintf(b, sizeof(b), "msq_gptq_block: %lld workgroups cannot all be resident on this device (%d CUs x %d)", (long long)nwg, cus, per_cu);
            return gfail(MSQ_ERR_UNSUPPORTED, b);
        }
    }
    GptqArgs A;
    A.Wt = Wt; A.U = U; A.Qt = Qt; A.Et = Et; A.loss = loss; A.pruned = pruned; A.status = status_flag;
    char* ws = (char*)workspace;
    A.colstat = (unsigned long long*)ws; ws += (int64_t)cols * 8;
    A.bars = (unsigned int*)ws; ws += (int64_t)cols * GPTQ_BARS * 4;
    A.hist = (unsigned int*)ws; ws += (int64_t)cols * 4 * 256 * 4;
    A.ties = (unsigned int*)ws;
    A.O = O; A.cols = cols; A.ldu = ldu; A.nwg = (int)nwg;
    OutlierArgs& q = A.qa;
    q.fi = Fmt{fi.kind, fi.ebits, fi.mbits, fi.emax, fi.max_norm};
    q.fo = Fmt{fo.kind, fo.ebits, fo.mbits, fo.emax, fo.max_norm};
    q.in_sb = inlier_scale_bits; q.out_sb = outlier_scale_bits; q.k = std_dev; q.rmode = rmode; q.flush = flush_fp32_subnorms;
    q.variant = 0; q.pre = 1; q.axis_len = O; q.post = 1; q.nblk = (O + block - 1) / block;
    q.mask = nullptr; q.e_in = nullptr; q.e_out = nullptr; q.n_out = nullptr; q.status = nullptr; q.vmean = nullptr; q.vstd = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)msq_gptq_block_workspace_bytes(O, cols), st) != hipSuccess)
        return gfail(MSQ_ERR_LAUNCH, "msq_gptq_block: clearing the workspace failed");
#define MSQ_GPTQ(BSV)                                                                                                   \
    case BSV: {                                                                                                        \
        hipLaunchKernelGGL(k_gptq_block<BSV>, dim3((unsigned)nwg), dim3(GPTQ_WG), lds, st, A);                         \
        break; }
    switch (block) { MSQ_GPTQ(8) MSQ_GPTQ(16) MSQ_GPTQ(32) MSQ_GPTQ(64)
        default: return gfail(MSQ_ERR_UNSUPPORTED, "msq_gptq_block: quantiser block must be 8, 16, 32 or 64"); }
#undef MSQ_GPTQ
    return hipGetLastError() == hipSuccess ? MSQ_OK : gfail(MSQ_ERR_LAUNCH, "msq_gptq_block: launch failed");
}

}  // extern "C"
