// msq_kv.hip -- KV-cache group quantisation at the GEAR hook (BASELINE config 4).
//
// The reference's kv_quant/ tree has integer min / max group fake-quant only
// (kv_quant/GEARLM/Simulated/compress_function.py:8-38 fake_groupwise_token_asymmetric_quantization: groups along
// head.dim of one token; :41-70 fake_groupwise_channel_asymmetric_quantization_new: groups along the tokens of one
// channel), applied to a [batch, heads, seq, head_dim] cache tensor through a permute / view / float round trip:
//     scale = (max - min) / (2^bits - 1);  q = round_half_even(relu((x - min) / scale));  y = q * scale + min
// Both run here as ONE pass over the tensor in its own [B, H, S, D] layout (no permute copies): every group's
// min / max is reduced across lanes (token groups) or kept per lane (channel groups), the elements are re-read from
// cache, quantised with the reference's fp32 op sequence (no FMA contraction, RNE rounding) and written back in the
// tensor dtype.  Bit-exact against the reference, NaNs of constant groups (0 / 0) included.
// HBM-bound: bytes = 2 x numel x sizeof(dtype).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msq.h"
#include "msq_device.h"

using namespace msq;

namespace {

template <int DT> struct KvIO;
template <> struct KvIO<0> {
    typedef float T;
    static MSQ_D float ld(const T* p, int64_t i) { return p[i]; }
    static MSQ_D void st(T* p, int64_t i, float v) { p[i] = v; }
};
template <> struct KvIO<1> {
    typedef uint16_t T;
    static MSQ_D float ld(const T* p, int64_t i) { return (float)__builtin_bit_cast(_Float16, p[i]); }
    static MSQ_D void st(T* p, int64_t i, float v) { p[i] = __builtin_bit_cast(uint16_t, (_Float16)v); }
};
template <> struct KvIO<2> {
    typedef uint16_t T;
    static MSQ_D float ld(const T* p, int64_t i) { return u2f((uint32_t)p[i] << 16); }
    static MSQ_D void st(T* p, int64_t i, float v) { p[i] = __builtin_bit_cast(uint16_t, (__bf16)v); }
};

// torch.max / torch.min propagate NaN
MSQ_D float nmax(float a, float b) { return (a != a || b != b) ? u2f(0x7FC00000u) : (a > b ? a : b); }
MSQ_D float nmin(float a, float b) { return (a != a || b != b) ? u2f(0x7FC00000u) : (a < b ? a : b); }

MSQ_D float kv_codec(float x, float mn, float scale) {
    float v = (x - mn) / scale;                          // compress_function.py:27 / :58
    v = (v < 0.f) ? 0.f : v;                             // F.relu (keeps NaN)
    v = __builtin_rintf(v);                              // Tensor.round_(): half to even
    return v * scale + mn;                               // :30 / :61 (two roundings: compiled with -ffp-contract=off)
}

// ---- groups along head.dim of one token.  LPG lanes per group (power of two <= 64); element i of group g of token
// (b, s) is hd = g * gs + i -> address ((b * H + hd / D) * S + s) * D + hd % D.
template <int DT>
__global__ void __launch_bounds__(256)
k_kv_token(const typename KvIO<DT>::T* __restrict__ in, typename KvIO<DT>::T* __restrict__ out, int64_t B, int64_t H, int64_t S,
           int64_t D, int64_t gs, int lpg, float levels) {
    const int64_t HD = H * D, ngrp = HD / gs;
    const int64_t total = B * S * ngrp;
    const int gpw = 64 / lpg;                                       // groups per wave
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t grp = wave * gpw + lane / lpg;
    const int li = lane % lpg;
    const bool live = grp < total;
    const int64_t g = live ? grp % ngrp : 0, bs = live ? grp / ngrp : 0;
    const int64_t s = bs % S, b = bs / S;
    float mx = -__builtin_inff(), mn = __builtin_inff();
    if (live)
        for (int64_t i = li; i < gs; i += lpg) {
            const int64_t hd = g * gs + i;
            const float x = KvIO<DT>::ld(in, ((b * H + hd / D) * S + s) * D + hd % D);
            mx = nmax(mx, x); mn = nmin(mn, x);
        }
    for (int o = 1; o < lpg; o <<= 1) {                            // butterfly inside the lpg-lane group
        mx = nmax(mx, __shfl_xor(mx, o, 64));
        mn = nmin(mn, __shfl_xor(mn, o, 64));
    }
    if (!live) return;
    const float scale = (mx - mn) / levels;                        // :26
    for (int64_t i = li; i < gs; i += lpg) {
        const int64_t hd = g * gs + i;
        const int64_t a = ((b * H + hd / D) * S + s) * D + hd % D;
        KvIO<DT>::st(out, a, kv_codec(KvIO<DT>::ld(in, a), mn, scale));
    }
}

// ---- groups along the tokens of one channel: one lane per (b, token group, h, d), d fastest (coalesced rows)
template <int DT>
__global__ void __launch_bounds__(256)
k_kv_channel(const typename KvIO<DT>::T* __restrict__ in, typename KvIO<DT>::T* __restrict__ out, int64_t B, int64_t H, int64_t S,
             int64_t D, int64_t gs, float levels) {
    const int64_t ngrp = S / gs;
    const int64_t total = B * H * ngrp * D;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t d = t % D, g = (t / D) % ngrp, bh = t / (D * ngrp);
    const int64_t base = (bh * S + g * gs) * D + d;
    float mx = -__builtin_inff(), mn = __builtin_inff();
    for (int64_t i = 0; i < gs; ++i) {
        const float x = KvIO<DT>::ld(in, base + i * D);
        mx = nmax(mx, x); mn = nmin(mn, x);
    }
    const float scale = (mx - mn) / levels;                        // :57
    for (int64_t i = 0; i < gs; ++i)
        KvIO<DT>::st(out, base + i * D, kv_codec(KvIO<DT>::ld(in, base + i * D), mn, scale));
}

}  // namespace

extern "C" void msq_set_error_(const char* msg);      // msq_quant.hip: the message msq_last_error() returns
static int kv_fail(int code, const char* msg) { msq_set_error_(msg); return code; }

extern "C" {

int msq_kv_group_quant(const void* in, void* out, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int quantize_bit,
                       int64_t group_size, int along_tokens, void* stream) {
    if (B < 0 || H < 0 || S < 0 || D < 0) return kv_fail(MSQ_ERR_BAD_ARG, "msq_kv_group_quant: negative size");
    if (B * H * S * D == 0) return MSQ_OK;
    if (!in || !out) return kv_fail(MSQ_ERR_BAD_ARG, "msq_kv_group_quant: null buffer");
    if (quantize_bit < 1 || quantize_bit > 16) return kv_fail(MSQ_ERR_BAD_ARG, "msq_kv_group_quant: quantize_bit must be in [1, 16]");
    if (dtype < 0 || dtype > 2) return kv_fail(MSQ_ERR_UNSUPPORTED, "msq_kv_group_quant: dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    if (group_size <= 0) return kv_fail(MSQ_ERR_BAD_ARG, "msq_kv_group_quant: group_size must be positive");
    const float levels = (float)((1 << quantize_bit) - 1);
    hipStream_t st = (hipStream_t)stream;
    if (along_tokens) {
        // compress_function.py:50-52: group_num = seq // group_size, then .view(batch, group_num, group_size, H * D)
        if (S % group_size) return kv_fail(MSQ_ERR_BAD_ARG, "msq_kv_group_quant: group_size must divide the sequence length (the reference's view() raises)");
        const int64_t total = B * H * (S / group_size) * D;
        const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
        if (dtype == 0) hipLaunchKernelGGL(k_kv_channel<0>, grid, blk, 0, st, (const float*)in, (float*)out, B, H, S, D, group_size, levels);
        else if (dtype == 1) hipLaunchKernelGGL(k_kv_channel<1>, grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, B, H, S, D, group_size, levels);
        else hipLaunchKernelGGL(k_kv_channel<2>, grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, B, H, S, D, group_size, levels);
    } else {
        // :15-17: "group_size should be a factor of the last dimension size"
        if ((H * D) % group_size) return kv_fail(MSQ_ERR_BAD_ARG, "group_size should be a factor of the last dimension size");
        int lpg = 64;
        while (lpg > 1 && lpg > group_size) lpg >>= 1;                // largest power of two <= min(64, group_size)
        const int64_t groups = B * S * ((H * D) / group_size);
        const int64_t waves = (groups + (64 / lpg) - 1) / (64 / lpg);
        const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
        if (dtype == 0) hipLaunchKernelGGL(k_kv_token<0>, grid, blk, 0, st, (const float*)in, (float*)out, B, H, S, D, group_size, lpg, levels);
        else if (dtype == 1) hipLaunchKernelGGL(k_kv_token<1>, grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, B, H, S, D, group_size, lpg, levels);
        else hipLaunchKernelGGL(k_kv_token<2>, grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, B, H, S, D, group_size, lpg, levels);
    }
    return hipGetLastError() == hipSuccess ? MSQ_OK : kv_fail(MSQ_ERR_LAUNCH, "msq_kv_group_quant: launch failed");
}

}  // extern "C"
