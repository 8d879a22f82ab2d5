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

// The same value without the IEEE division in the common case.  t = (x - mn) * rcp(scale) is within levels * 2^-22 of the
// quotient the reference rounds (v_rcp_f32: one ulp, the product: half an ulp, the IEEE quotient itself: half an ulp; x - mn <=
// levels * scale), so round(t) is the reference's integer unless t lies within thr = levels * 2^-21 of k + 1/2: those elements,
// and every group whose scale is zero, subnormal, huge or NaN (constant groups: 0 / 0; NaN / Inf members), go through kv_codec.
// x - mn >= +0 in the fast case (mn is the group's minimum and no member is NaN): the relu is the identity there.
struct KvG { float mn, scale, r, thr; bool fast; };
MSQ_D KvG kv_group(float mx, float mn, float levels) {
    KvG g;
    g.mn = mn; g.scale = (mx - mn) / levels;                       // :26 / :57
    g.fast = g.scale > 1e-30f && g.scale < 1e30f;
    g.r = __builtin_amdgcn_rcpf(g.scale);
    g.thr = levels * 4.76837158203125e-07f;
    return g;
}
template <int N>
MSQ_D void kv_codec_n(const float* x, float* y, const KvG& g) {
    float v[N];
    bool redo = !g.fast;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float t = (x[k] - g.mn) * g.r;
        v[k] = __builtin_rintf(t);
        redo |= __builtin_fabsf(__builtin_fabsf(t - v[k]) - 0.5f) < g.thr;
    }
    if (redo) {
#pragma unroll
        for (int k = 0; k < N; ++k) y[k] = kv_codec(x[k], g.mn, g.scale);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) y[k] = v[k] * g.scale + g.mn;
    }
}
// one group per element (channel groups: a lane's N channels)
template <int N>
MSQ_D void kv_codec_each(const float* x, float* y, const KvG* g) {
    float v[N];
    bool redo = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float t = (x[k] - g[k].mn) * g[k].r;
        v[k] = __builtin_rintf(t);
        redo |= !g[k].fast | (__builtin_fabsf(__builtin_fabsf(t - v[k]) - 0.5f) < g[k].thr);
    }
    if (redo) {
#pragma unroll
        for (int k = 0; k < N; ++k) y[k] = kv_codec(x[k], g[k].mn, g[k].scale);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) y[k] = v[k] * g[k].scale + g[k].mn;
    }
}
// min / max of a lane's own elements: plain v_max / v_min and a NaN flag (torch.max / torch.min propagate NaN: mm_done)
struct MinMax { float mx, mn; bool nan; };
MSQ_D MinMax mm_init() { return MinMax{-__builtin_inff(), __builtin_inff(), false}; }
MSQ_D void mm_acc(MinMax& m, float x) { m.mx = x > m.mx ? x : m.mx; m.mn = x < m.mn ? x : m.mn; m.nan |= (x != x); }
MSQ_D void mm_done(MinMax& m) { if (m.nan) { m.mx = u2f(0x7FC00000u); m.mn = u2f(0x7FC00000u); } }

// the same on raw half-precision pairs: packed min / max (two entries per instruction, no conversion) and the largest magnitude
// pattern, which exceeds Inf's exactly when a NaN was seen
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef uint16_t us2_t __attribute__((ext_vector_type(2)));
struct MinMaxH { uint32_t mx, mn, mag; };
MSQ_D MinMaxH mmh_init() { return MinMaxH{0xFC00FC00u, 0x7C007C00u, 0u}; }
MSQ_D void mmh_acc(MinMaxH& m, uint32_t w) {
    uint32_t a, b;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(a) : "v"(m.mx), "v"(w));
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(b) : "v"(m.mn), "v"(w));
    m.mx = a; m.mn = b;
    m.mag = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2_t, m.mag), __builtin_bit_cast(us2_t, w & 0x7FFF7FFFu)));
}
MSQ_D MinMax mmh_done(const MinMaxH& m) {
    const float a = (float)__builtin_bit_cast(_Float16, (uint16_t)(m.mx & 0xFFFFu)), b = (float)__builtin_bit_cast(_Float16, (uint16_t)(m.mx >> 16));
    const float c = (float)__builtin_bit_cast(_Float16, (uint16_t)(m.mn & 0xFFFFu)), d = (float)__builtin_bit_cast(_Float16, (uint16_t)(m.mn >> 16));
    MinMax r;
    r.mx = a > b ? a : b; r.mn = c < d ? c : d;
    r.nan = ((m.mag & 0xFFFFu) > 0x7C00u) | ((m.mag >> 16) > 0x7C00u);
    mm_done(r);
    return r;
}
template <int DT> MSQ_D MinMax minmax_chunk(const uint4& raw) {       // one 16-byte chunk
    if constexpr (DT == 1) {
        MinMaxH h = mmh_init();
        mmh_acc(h, raw.x); mmh_acc(h, raw.y); mmh_acc(h, raw.z); mmh_acc(h, raw.w);
        return mmh_done(h);
    } else {
        constexpr int N = DT == 0 ? 4 : 8;
        float x[N];
        MinMax m = mm_init();
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
        if (DT == 0) { x[0] = u2f(w[0]); x[1] = u2f(w[1]); x[2] = u2f(w[2]); x[3] = u2f(w[3]); }
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { x[2 * k] = u2f(w[k] << 16); x[2 * k + 1] = u2f(w[k] & 0xFFFF0000u); }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) mm_acc(m, x[k]);
        mm_done(m);
        return m;
    }
}

// ---- groups along head.dim of one token.  A lane owns V = 8 consecutive head.dim entries (16 bytes of a half-precision
// cache: one dwordx4 access; two for f32), LPG lanes share a group (LPG = gs / 8 rounded up to a power of two, at most 64;
// longer groups loop), groups packed 64 / LPG per wave.  hd = g * gs + i -> address ((b * H + hd / D) * S + s) * D + hd % D;
// D % 8 == 0 and gs % 8 == 0 keep a lane's 8 entries inside one head (the launcher falls back to V = 1 otherwise).
template <int DT, int V>
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
    constexpr int MAXIT = 16;                                        // chunks of V per lane kept in registers (gs <= 64 * V * MAXIT = 8192)
    float x[MAXIT][V];
    float mx = -__builtin_inff(), mn = __builtin_inff();
    const int64_t nchunk = gs / V;                                   // V-element chunks of the group
#pragma unroll
    for (int nit = 0; nit < MAXIT; ++nit) {
        const int64_t c = li + (int64_t)nit * lpg;
        if (live && c < nchunk) {
            const int64_t hd = g * gs + c * V;
            const int64_t a = ((b * H + hd / D) * S + s) * D + hd % D;
            if constexpr (V == 8 && DT != 0) {
                const uint4 raw = *reinterpret_cast<const uint4*>(in + a);
                const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (DT == 1) { x[nit][2 * k] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[k] & 0xFFFFu)); x[nit][2 * k + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[k] >> 16)); }
                    else { x[nit][2 * k] = u2f(w[k] << 16); x[nit][2 * k + 1] = u2f(w[k] & 0xFFFF0000u); }
                }
            } else if constexpr (V == 8) {
                const float4 r0 = *reinterpret_cast<const float4*>(in + a), r1 = *reinterpret_cast<const float4*>(in + a + 4);
                x[nit][0] = r0.x; x[nit][1] = r0.y; x[nit][2] = r0.z; x[nit][3] = r0.w; x[nit][4] = r1.x; x[nit][5] = r1.y; x[nit][6] = r1.z; x[nit][7] = r1.w;
            } else {
                x[nit][0] = KvIO<DT>::ld(in, a);
            }
#pragma unroll
            for (int k = 0; k < V; ++k) { mx = nmax(mx, x[nit][k]); mn = nmin(mn, x[nit][k]); }
        }
    }
    for (int o = 1; o < lpg; o <<= 1) {                            // butterfly inside the lpg-lane group
        mx = nmax(mx, __shfl_xor(mx, o, 64));
        mn = nmin(mn, __shfl_xor(mn, o, 64));
    }
    if (!live) return;
    const float scale = (mx - mn) / levels;                        // :26
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int64_t c = li + (int64_t)it * lpg;
        if (c >= nchunk) break;
        const int64_t hd = g * gs + c * V;
        const int64_t a = ((b * H + hd / D) * S + s) * D + hd % D;
        float y[V];
#pragma unroll
        for (int k = 0; k < V; ++k) y[k] = kv_codec(x[it][k], mn, scale);
        if constexpr (V == 8 && DT != 0) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (DT == 1) w[k] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)y[2 * k]) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)y[2 * k + 1]) << 16);
                else w[k] = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)y[2 * k]) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)y[2 * k + 1]) << 16);
            }
            *reinterpret_cast<uint4*>(out + a) = make_uint4(w[0], w[1], w[2], w[3]);
        } else if constexpr (V == 8) {
            *reinterpret_cast<float4*>(out + a) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(out + a + 4) = make_float4(y[4], y[5], y[6], y[7]);
        } else {
            KvIO<DT>::st(out, a, y[0]);
        }
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


// ---------------------------------------------------------------------------------------------------------------------
// Fast forms for tensors of fewer than 2^31 16-byte chunks: 32-bit index arithmetic with host-computed multiply-shift
// division (the 64-bit / and % of the general kernels cost more than the quantiser itself).
// ---------------------------------------------------------------------------------------------------------------------
struct FastDiv { uint32_t d, mul, shr; };                            // n / d for n < 2^31: d == 1 ? n : umulhi(n, mul) >> shr
MSQ_D uint32_t fdiv(uint32_t n, const FastDiv& f) { return f.d == 1u ? n : (__umulhi(n, f.mul) >> f.shr); }

template <int DT> struct Chunk;                                      // one 16-byte access: 8 half / bf16 values or 4 floats
template <> struct Chunk<0> { static constexpr int N = 4; };
template <> struct Chunk<1> { static constexpr int N = 8; };
template <> struct Chunk<2> { static constexpr int N = 8; };
template <int DT> MSQ_D void unpack16(const uint4& raw, float* x) {
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    if (DT == 0) { x[0] = u2f(w[0]); x[1] = u2f(w[1]); x[2] = u2f(w[2]); x[3] = u2f(w[3]); return; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (DT == 1) { x[2 * k] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[k] & 0xFFFFu)); x[2 * k + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[k] >> 16)); }
        else { x[2 * k] = u2f(w[k] << 16); x[2 * k + 1] = u2f(w[k] & 0xFFFF0000u); }
    }
}
template <int DT> MSQ_D uint4 pack16(const float* y) {
    if (DT == 0) return make_uint4(f2u(y[0]), f2u(y[1]), f2u(y[2]), f2u(y[3]));
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (DT == 1) w[k] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)y[2 * k]) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)y[2 * k + 1]) << 16);
        else w[k] = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)y[2 * k]) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)y[2 * k + 1]) << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// groups along head.dim of one token, LPG = group_size / Chunk::N lanes per group (power of two <= 64): every lane owns
// one 16-byte chunk; chunk q -> row = q / CPR (b, s), cr = q % CPR -> head cr / DC, offset cr % DC (CPR = H D / N, DC = D / N)
template <int DT, int LPG>
__global__ void __launch_bounds__(256)
k_kv_token_fast(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t nq, uint32_t H, uint32_t S, FastDiv cpr, FastDiv dc,
                FastDiv sdiv, float levels) {
    constexpr int N = Chunk<DT>::N;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t q0 = blockIdx.x * 256u; q0 < nq; q0 += stride) {
        const uint32_t q = q0 + threadIdx.x;
        const bool live = q < nq;
        const uint32_t qq = live ? q : 0u;
        const uint32_t row = fdiv(qq, cpr), cr = qq - row * cpr.d;
        const uint32_t h = fdiv(cr, dc), dv = cr - h * dc.d;
        const uint32_t b = fdiv(row, sdiv), s = row - b * sdiv.d;
        const uint32_t a = ((b * H + h) * S + s) * dc.d + dv;          // in 16-byte chunks
        float x[N];
        MinMax m = mm_init();
        if (live) {
            const uint4 raw = in[a];
            unpack16<DT>(raw, x);
            m = minmax_chunk<DT>(raw);
        }
        float mx = m.mx, mn = m.mn;
#pragma unroll
        for (int o = 1; o < LPG; o <<= 1) {
            mx = nmax(mx, __shfl_xor(mx, o, 64));
            mn = nmin(mn, __shfl_xor(mn, o, 64));
        }
        if (live) {
            const KvG g = kv_group(mx, mn, levels);
            float y[N];
            kv_codec_n<N>(x, y, g);
            out[a] = pack16<DT>(y);
        }
    }
}

// groups of gs <= 32 tokens of one channel: a lane owns one 16-byte chunk of channels over the gs tokens of its group, the
// raw chunks stay in registers between the min / max scan and the quantisation.  t -> dv = t % DC, g = (t / DC) % ngrp, bh
template <int DT>
__global__ void __launch_bounds__(256)
k_kv_channel_fast(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t total, uint32_t S, uint32_t gs, FastDiv dc,
                  FastDiv ng, float levels) {
    constexpr int N = Chunk<DT>::N;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    const uint32_t u = fdiv(t, dc), dv = t - u * dc.d;
    const uint32_t bh = fdiv(u, ng), g = u - bh * ng.d;
    const uint32_t base = (bh * S + g * gs) * dc.d + dv;
    uint4 raw[32];
    MinMax m[N];
#pragma unroll
    for (int k = 0; k < N; ++k) m[k] = mm_init();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        raw[i] = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)i < gs) raw[i] = in[base + (uint32_t)i * dc.d];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if ((uint32_t)i < gs) {
            float x[N];
            unpack16<DT>(raw[i], x);
#pragma unroll
            for (int k = 0; k < N; ++k) mm_acc(m[k], x[k]);
        }
    }
    KvG kg[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { mm_done(m[k]); kg[k] = kv_group(m[k].mx, m[k].mn, levels); }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if ((uint32_t)i < gs) {
            float x[N], y[N];
            unpack16<DT>(raw[i], x);
            kv_codec_each<N>(x, y, kg);
            out[base + (uint32_t)i * dc.d] = pack16<DT>(y);
        }
    }
}

// the same for gs = 8 LG tokens (LG a power of two <= 64): LG neighbouring lanes share a group, 8 tokens each -- four
// times the waves of the one-lane form at gs = 32 (the [1, 32, 4096, 128] cache has only 1024 waves there) and groups of
// up to 512 tokens.  t -> j = t % LG (token eighth), dv = (t / LG) % DC, g, bh
template <int DT, int LG>
__global__ void __launch_bounds__(256)
k_kv_channel_fast8(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t total, uint32_t S, FastDiv dc, FastDiv ng, float levels) {
    constexpr int N = Chunk<DT>::N;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool live = t < total;                                      // total is a multiple of LG: groups are never partial
    const uint32_t tt = live ? t : 0u;
    const uint32_t j = tt % LG, r = tt / LG;
    const uint32_t u = fdiv(r, dc), dv = r - u * dc.d;
    const uint32_t bh = fdiv(u, ng), g = u - bh * ng.d;
    const uint32_t base = (bh * S + g * (8u * LG) + j * 8u) * dc.d + dv;
    uint4 raw[8];
    float mx[N], mn[N];
    {
        MinMax m[N];
#pragma unroll
        for (int k = 0; k < N; ++k) m[k] = mm_init();
        if (live) {
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = in[base + (uint32_t)i * dc.d];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x[N];
                unpack16<DT>(raw[i], x);
#pragma unroll
                for (int k = 0; k < N; ++k) mm_acc(m[k], x[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) { mm_done(m[k]); mx[k] = m[k].mx; mn[k] = m[k].mn; }
    }
#pragma unroll
    for (int o = 1; o < LG; o <<= 1) {
#pragma unroll
        for (int k = 0; k < N; ++k) { mx[k] = nmax(mx[k], __shfl_xor(mx[k], o, 64)); mn[k] = nmin(mn[k], __shfl_xor(mn[k], o, 64)); }
    }
    if (!live) return;
    KvG kg[N];
#pragma unroll
    for (int k = 0; k < N; ++k) kg[k] = kv_group(mx[k], mn[k], levels);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float x[N], y[N];
        unpack16<DT>(raw[i], x);
        kv_codec_each<N>(x, y, kg);
        out[base + (uint32_t)i * dc.d] = pack16<DT>(y);
    }
}

// groups of exactly 64 * N * NIT entries along head.dim of one token (the whole token of a Llama-2-7B cache: 4096 = 64 * 8 * 8
// half values): one wave per group, NIT 16-byte chunks per lane kept raw in registers between the min / max scan and the
// quantisation, 32-bit index arithmetic.  chunk cr of token (b, s) -> head cr / DC, offset cr % DC.
template <int DT, int NIT>
__global__ void __launch_bounds__(256)
k_kv_token_wide(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t ngroups, uint32_t H, uint32_t S, FastDiv gpt, FastDiv dc,
                FastDiv sdiv, float levels) {
    constexpr int N = Chunk<DT>::N;
    const uint32_t grp = (blockIdx.x * 256u + threadIdx.x) >> 6;
    if (grp >= ngroups) return;                                       // whole waves
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row = fdiv(grp, gpt), gi = grp - row * gpt.d;
    const uint32_t b = fdiv(row, sdiv), s = row - b * sdiv.d;
    uint4 raw[NIT];
    uint32_t a[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const uint32_t cr = gi * (64u * NIT) + (uint32_t)i * 64u + lane;
        const uint32_t h = fdiv(cr, dc), dv = cr - h * dc.d;
        a[i] = ((b * H + h) * S + s) * dc.d + dv;
        raw[i] = in[a[i]];
    }
    MinMax m;
    if constexpr (DT == 1) {
        MinMaxH h = mmh_init();
#pragma unroll
        for (int i = 0; i < NIT; ++i) { mmh_acc(h, raw[i].x); mmh_acc(h, raw[i].y); mmh_acc(h, raw[i].z); mmh_acc(h, raw[i].w); }
        m = mmh_done(h);
    } else {
        m = mm_init();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            float x[N];
            unpack16<DT>(raw[i], x);
#pragma unroll
            for (int k = 0; k < N; ++k) mm_acc(m, x[k]);
        }
        mm_done(m);
    }
    float mx = m.mx, mn = m.mn;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        mx = nmax(mx, __shfl_xor(mx, o, 64));
        mn = nmin(mn, __shfl_xor(mn, o, 64));
    }
    const KvG g = kv_group(mx, mn, levels);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        float x[N], y[N];
        unpack16<DT>(raw[i], x);
        kv_codec_n<N>(x, y, g);
        out[a[i]] = pack16<DT>(y);
    }
}

}  // namespace

extern "C" void msq_set_error_(const char* msg);      // msq_quant.hip: the message msq_last_error() returns
static int kv_fail(int code, const char* msg) { msq_set_error_(msg); return code; }
static FastDiv make_fastdiv(uint32_t d) {             // exact for n < 2^31 (Granlund-Montgomery round-up multiplier)
    FastDiv f{d, 0u, 0u};
    if (d <= 1u) return f;
    uint32_t l = 0; while ((1ull << l) < d) ++l;      // ceil(log2 d)
    const uint32_t p = 31u + l;
    f.mul = (uint32_t)(((1ull << p) + d - 1ull) / d);
    f.shr = p - 32u;
    return f;
}

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
        const int cn = dtype == 0 ? 4 : 8;
        const bool al = ((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
        const int64_t lg = group_size / 8;
        if (al && D % cn == 0 && group_size % 8 == 0 && lg <= 64 && (lg & (lg - 1)) == 0 && B * H * S * D / cn < (int64_t)0x7FFFFFFF) {
            const uint32_t DC = (uint32_t)(D / cn), ngrp = (uint32_t)(S / group_size);
            const uint32_t tot = (uint32_t)(B * H) * ngrp * DC * (uint32_t)lg;
            const dim3 grid((tot + 255u) / 256u), blk(256);
            const FastDiv fdc = make_fastdiv(DC), fng = make_fastdiv(ngrp);
#define MSQ_KC(DTV, L) hipLaunchKernelGGL((k_kv_channel_fast8<DTV, L>), grid, blk, 0, st, (const uint4*)in, (uint4*)out, tot, (uint32_t)S, fdc, fng, levels)
#define MSQ_KCL(DTV) do { switch ((int)lg) { case 1: MSQ_KC(DTV, 1); break; case 2: MSQ_KC(DTV, 2); break; case 4: MSQ_KC(DTV, 4); break; \
                       case 8: MSQ_KC(DTV, 8); break; case 16: MSQ_KC(DTV, 16); break; case 32: MSQ_KC(DTV, 32); break; default: MSQ_KC(DTV, 64); break; } } while (0)
            if (dtype == 0) MSQ_KCL(0); else if (dtype == 1) MSQ_KCL(1); else MSQ_KCL(2);
#undef MSQ_KCL
#undef MSQ_KC
            return hipGetLastError() == hipSuccess ? MSQ_OK : kv_fail(MSQ_ERR_LAUNCH, "msq_kv_group_quant: launch failed");
        }
        if (al && D % cn == 0 && group_size <= 32 && B * H * S * D / cn < (int64_t)0x7FFFFFFF) {
            const uint32_t DC = (uint32_t)(D / cn), ngrp = (uint32_t)(S / group_size);
            const uint32_t tot = (uint32_t)(B * H) * ngrp * DC;
            const dim3 grid((tot + 255u) / 256u), blk(256);
            const FastDiv fdc = make_fastdiv(DC), fng = make_fastdiv(ngrp);
            if (dtype == 0) hipLaunchKernelGGL(k_kv_channel_fast<0>, grid, blk, 0, st, (const uint4*)in, (uint4*)out, tot, (uint32_t)S, (uint32_t)group_size, fdc, fng, levels);
            else if (dtype == 1) hipLaunchKernelGGL(k_kv_channel_fast<1>, grid, blk, 0, st, (const uint4*)in, (uint4*)out, tot, (uint32_t)S, (uint32_t)group_size, fdc, fng, levels);
            else hipLaunchKernelGGL(k_kv_channel_fast<2>, grid, blk, 0, st, (const uint4*)in, (uint4*)out, tot, (uint32_t)S, (uint32_t)group_size, fdc, fng, levels);
            return hipGetLastError() == hipSuccess ? MSQ_OK : kv_fail(MSQ_ERR_LAUNCH, "msq_kv_group_quant: launch failed");
        }
        const int64_t total = B * H * (S / group_size) * D;
        const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
        if (dtype == 0) hipLaunchKernelGGL(k_kv_channel<0>, grid, blk, 0, st, (const float*)in, (float*)out, B, H, S, D, group_size, levels);
        else if (dtype == 1) hipLaunchKernelGGL(k_kv_channel<1>, grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, B, H, S, D, group_size, levels);
        else hipLaunchKernelGGL(k_kv_channel<2>, grid, blk, 0, st, (const uint16_t*)in, (uint16_t*)out, B, H, S, D, group_size, levels);
    } else {
        // :15-17: "group_size should be a factor of the last dimension size"
        if ((H * D) % group_size) return kv_fail(MSQ_ERR_BAD_ARG, "group_size should be a factor of the last dimension size");
        // V = 8 entries per lane (16-byte accesses) when the geometry and the alignment allow it, else one entry per lane;
        // a lane keeps at most 16 chunks in registers: groups of more than 64 * V * 16 entries are not built
        {
            const int cn = dtype == 0 ? 4 : 8;
            const bool al = ((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
            const int64_t lp = group_size / cn;
            if (al && D % cn == 0 && group_size % cn == 0 && lp >= 1 && lp <= 64 && (lp & (lp - 1)) == 0 &&
                B * H * S * D / cn < (int64_t)0x7FFFFFFF) {
                const uint32_t nq = (uint32_t)(B * H * S * D / cn);
                const FastDiv fcpr = make_fastdiv((uint32_t)(H * D / cn)), fdc = make_fastdiv((uint32_t)(D / cn)), fs = make_fastdiv((uint32_t)S);
                uint32_t gblocks = (nq + 255u) / 256u; if (gblocks > 256u * 16u) gblocks = 256u * 16u;
                const dim3 grid(gblocks), blk(256);
#define MSQ_KVF(DTV, L) hipLaunchKernelGGL((k_kv_token_fast<DTV, L>), grid, blk, 0, st, (const uint4*)in, (uint4*)out, nq, (uint32_t)H, (uint32_t)S, fcpr, fdc, fs, levels)
#define MSQ_KVFL(DTV) do { switch ((int)lp) { case 1: MSQ_KVF(DTV, 1); break; case 2: MSQ_KVF(DTV, 2); break; case 4: MSQ_KVF(DTV, 4); break; \
                        case 8: MSQ_KVF(DTV, 8); break; case 16: MSQ_KVF(DTV, 16); break; case 32: MSQ_KVF(DTV, 32); break; default: MSQ_KVF(DTV, 64); break; } } while (0)
                if (dtype == 0) MSQ_KVFL(0); else if (dtype == 1) MSQ_KVFL(1); else MSQ_KVFL(2);
#undef MSQ_KVFL
#undef MSQ_KVF
                return hipGetLastError() == hipSuccess ? MSQ_OK : kv_fail(MSQ_ERR_LAUNCH, "msq_kv_group_quant: launch failed");
            }
        }
        {
            const int cn = dtype == 0 ? 4 : 8;
            const bool al = ((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
            const int64_t nit = group_size / (64 * cn);
            if (al && D % cn == 0 && group_size == nit * 64 * cn && (nit == 2 || nit == 4 || nit == 8 || nit == 16) &&
                B * H * S * D / cn < (int64_t)0x7FFFFFFF) {
                const uint32_t ngroups = (uint32_t)(B * S * ((H * D) / group_size));
                const FastDiv fgpt = make_fastdiv((uint32_t)((H * D) / group_size)), fdc = make_fastdiv((uint32_t)(D / cn)), fs = make_fastdiv((uint32_t)S);
                const dim3 grid((ngroups + 3u) / 4u), blk(256);
#define MSQ_KVW(DTV, NI) hipLaunchKernelGGL((k_kv_token_wide<DTV, NI>), grid, blk, 0, st, (const uint4*)in, (uint4*)out, ngroups, (uint32_t)H, (uint32_t)S, fgpt, fdc, fs, levels)
#define MSQ_KVWL(DTV) do { switch ((int)nit) { case 2: MSQ_KVW(DTV, 2); break; case 4: MSQ_KVW(DTV, 4); break; case 8: MSQ_KVW(DTV, 8); break; default: MSQ_KVW(DTV, 16); break; } } while (0)
                if (dtype == 0) MSQ_KVWL(0); else if (dtype == 1) MSQ_KVWL(1); else MSQ_KVWL(2);
#undef MSQ_KVWL
#undef MSQ_KVW
                return hipGetLastError() == hipSuccess ? MSQ_OK : kv_fail(MSQ_ERR_LAUNCH, "msq_kv_group_quant: launch failed");
            }
        }
        const bool v8 = (D % 8 == 0) && (group_size % 8 == 0) && ((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
        const int V = v8 ? 8 : 1;
        const int64_t chunks = group_size / V;
        if (chunks > 64 * 16) return kv_fail(MSQ_ERR_UNSUPPORTED, "msq_kv_group_quant: group too long for the per-token kernel (max 8192 entries, 1024 unaligned)");
        int lpg = 1;
        while (lpg < 64 && lpg < chunks) lpg <<= 1;                   // lanes per group: power of two covering the chunks (<= 64)
        const int64_t groups = B * S * ((H * D) / group_size);
        const int64_t waves = (groups + (64 / lpg) - 1) / (64 / lpg);
        const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
#define MSQ_KVT(DTV, TY)                                                                                               \
        do { if (v8) hipLaunchKernelGGL((k_kv_token<DTV, 8>), grid, blk, 0, st, (const TY*)in, (TY*)out, B, H, S, D, group_size, lpg, levels); \
             else hipLaunchKernelGGL((k_kv_token<DTV, 1>), grid, blk, 0, st, (const TY*)in, (TY*)out, B, H, S, D, group_size, lpg, levels); } while (0)
        if (dtype == 0) MSQ_KVT(0, float); else if (dtype == 1) MSQ_KVT(1, uint16_t); else MSQ_KVT(2, uint16_t);
#undef MSQ_KVT
    }
    return hipGetLastError() == hipSuccess ? MSQ_OK : kv_fail(MSQ_ERR_LAUNCH, "msq_kv_group_quant: launch failed");
}

}  // extern "C"
