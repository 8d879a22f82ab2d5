// msq_outlier_kernels.h -- the two fake-quant kernels (block along a strided axis / along the contiguous
// axis), shared by the translation units that instantiate them (msq_quant.hip: arithmetic codec variants,
// msq_quant_hw.hip: hardware-convert variants).  FAST: 0 generic maths (posit inliers), 1 nearest rounding,
// 2 any rounding mode, 3 nearest + hardware converts for inliers and outliers, 4 nearest + hardware converts
// for the inliers and posit outliers.
#pragma once
#include "msq_outlier_core.h"

template <typename T> struct IO;
template <> struct IO<float> {
    static MSQ_D float ld(const float* p, int64_t i) { return p[i]; }
    static MSQ_D void st(float* p, int64_t i, float v) { p[i] = v; }
};

// bfloat16 tensors: read as is (every bf16 is an fp32 value), written with round-to-nearest-even like torch's .to(bfloat16)
// (exact whenever the formats have <= 8 significant bits)
struct bf16io_t { uint16_t v; };
template <> struct IO<bf16io_t> {
    static MSQ_D float ld(const bf16io_t* p, int64_t i) { return u2f((uint32_t)p[i].v << 16); }
    static MSQ_D void st(bf16io_t* p, int64_t i, float v) { p[i].v = __builtin_bit_cast(uint16_t, (__bf16)v); }
};

// float16 tensors: likewise (every fp16 is an fp32 value); written with v_cvt_f16_f32 = torch's .to(float16): nearest even, subnormals kept,
// overflow to Inf.  Same result as upcast -> compute -> downcast, without the two cast passes (the MicroScopiQ KV cache runs this: kvcache.py)
struct f16io_t { uint16_t v; };
template <> struct IO<f16io_t> {
    static MSQ_D float ld(const f16io_t* p, int64_t i) { return (float)__builtin_bit_cast(_Float16, p[i].v); }
    static MSQ_D void st(f16io_t* p, int64_t i, float v) { p[i].v = __builtin_bit_cast(uint16_t, (_Float16)v); }
};
// two values of a dword of a 16-bit tensor (low half first), and back
template <typename T> MSQ_D void unpack2(uint32_t w, float& lo, float& hi) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16io_t)) {
        lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
    } else { lo = u2f(w << 16); hi = u2f(w & 0xFFFF0000u); }
}
template <typename T> MSQ_D uint32_t pack2(float lo, float hi) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16io_t))
        return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)lo) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)hi) << 16);
    else return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)lo) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)hi) << 16);
}

template <int BS>
MSQ_D void outlier_side_outputs(const OutlierArgs& A, const uint32_t (&mkw)[(BS + 31) / 32], float se_in,
                                float se_out, int status, int64_t p, int64_t nb, int64_t q) {
    if (A.e_in) A.e_in[(p * A.nblk + nb) * A.post + q] = se_in;
    if (A.e_out) A.e_out[(p * A.nblk + nb) * A.post + q] = se_out;
    if (A.n_out && A.pre == 1 && (nb % BS) == 0) {           // utils/quant.py:66
        int c = 0;
#pragma unroll
        for (int w = 0; w < (BS + 31) / 32; ++w) c += __builtin_popcount(mkw[w]);
        A.n_out[(nb / BS) * A.post + q] = (int8_t)c;
    }
    if (status && A.status) atomicOr(A.status, status);
}

// --- layout A: post > 1.  lane <-> (p, nb, q), q fastest: each of the BS row reads
// of a wave is one contiguous 256-byte segment.
template <int BS, typename T, int FAST>
__global__ void __launch_bounds__(256)
k_outlier_strided(const T* __restrict__ in, T* __restrict__ out, OutlierArgs A) {
    const int64_t total = A.pre * A.nblk * A.post;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t q = t % A.post;
    const int64_t nb = (t / A.post) % A.nblk;
    const int64_t p = t / (A.post * A.nblk);
    const int64_t a0 = nb * BS;
    const int64_t base = (p * A.axis_len + a0) * A.post + q;
    float a[BS];
#pragma unroll
    for (int b = 0; b < BS; ++b)
        a[b] = (a0 + b < A.axis_len) ? IO<T>::ld(in, base + (int64_t)b * A.post) : 0.f;   // zero padding, :563-583
    int order;
    {   // torch's summation order for this column (oracle/msq_oracle.c sum_order_for)
        const int64_t lim = (A.post >= 8) ? (A.post / 32) * 32 : (A.post / 4) * 4;
        order = (q < lim) ? 0 : 2;
    }
    uint32_t mkw[(BS + 31) / 32];
    float se_in, se_out;
    const float* vm = A.vmean ? A.vmean + (p * BS) * A.post + q : nullptr;
    const float* vs = A.vstd ? A.vstd + (p * BS) * A.post + q : nullptr;
    int status;
    if (FAST == 1) status = outlier_block_fast<BS, 0>(a, mkw, se_in, se_out, A, order, vm, vs, A.post);
    else if (FAST == 2) status = outlier_block_fast<BS, -1>(a, mkw, se_in, se_out, A, order, vm, vs, A.post);
    else if (FAST == 3) status = outlier_block_fast<BS, 0, false, 1>(a, mkw, se_in, se_out, A, order, vm, vs, A.post);
    else if (FAST == 4) status = outlier_block_fast<BS, 0, false, 2>(a, mkw, se_in, se_out, A, order, vm, vs, A.post);
    else status = outlier_block<BS>(a, mkw, se_in, se_out, A, order, vm, vs, A.post);
#pragma unroll
    for (int b = 0; b < BS; ++b) {
        if (a0 + b < A.axis_len) {
            IO<T>::st(out, base + (int64_t)b * A.post, a[b]);
            if (A.mask) A.mask[base + (int64_t)b * A.post] = (uint8_t)((mkw[b >> 5] >> (b & 31)) & 1u);
        }
    }
    outlier_side_outputs<BS>(A, mkw, se_in, se_out, status, p, nb, q);
}

// --- layout A for 16-bit tensors with an even contiguous extent (the K cache: blocks of BS tokens of one channel, post = head_dim): one lane
// per PAIR of neighbouring columns, 4-byte accesses -- a wave moves 256 contiguous bytes per row instead of 128, half as many memory
// instructions per byte.  The two columns are two blocks, quantised one after the other; the results wait in the dwords they came from.
// Worth 2-4 % only (fp16 keys [1, 32, 4096, 128], fp4 + fp8 outliers: 48.7 -> 46.6 us; bf16 W[16384,4096] blocks of 32 along out_features
// 161.9 -> 158.4 us): these launches are one round of waves whose length is a lane's serial walk through its blocks, not memory instructions.
template <int BS, typename T, int FAST>
__global__ void __launch_bounds__(256)
k_outlier_strided_pair(const T* __restrict__ in, T* __restrict__ out, OutlierArgs A) {
    static_assert(sizeof(T) == 2, "16-bit tensors");
    const int64_t hp = A.post / 2;
    const int64_t total = A.pre * A.nblk * hp;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t q = (t % hp) * 2;
    const int64_t nb = (t / hp) % A.nblk;
    const int64_t p = t / (hp * A.nblk);
    const int64_t a0 = nb * BS;
    const int64_t base = (p * A.axis_len + a0) * A.post + q;
    const uint16_t* in16 = reinterpret_cast<const uint16_t*>(in);
    uint16_t* out16 = reinterpret_cast<uint16_t*>(out);
    uint32_t raw[BS];
#pragma unroll
    for (int b = 0; b < BS; ++b)
        raw[b] = (a0 + b < A.axis_len) ? *reinterpret_cast<const uint32_t*>(in16 + base + (int64_t)b * A.post) : 0u;   // zero padding, :563-583
    const int64_t lim = (A.post >= 8) ? (A.post / 32) * 32 : (A.post / 4) * 4;     // (even: both columns on the same side)
    const int order = (q < lim) ? 0 : 2;
#pragma unroll
    for (int col = 0; col < 2; ++col) {
        float a[BS];
#pragma unroll
        for (int b = 0; b < BS; ++b) { float lo, hi; unpack2<T>(raw[b], lo, hi); a[b] = col ? hi : lo; }
        uint32_t mkw[(BS + 31) / 32];
        float se_in, se_out;
        const float* vm = A.vmean ? A.vmean + (p * BS) * A.post + q + col : nullptr;
        const float* vs = A.vstd ? A.vstd + (p * BS) * A.post + q + col : nullptr;
        int status;
        if (FAST == 3) status = outlier_block_fast<BS, 0, false, 1>(a, mkw, se_in, se_out, A, order, vm, vs, A.post);
        else status = outlier_block_fast<BS, 0, false, 2>(a, mkw, se_in, se_out, A, order, vm, vs, A.post);
#pragma unroll
        for (int b = 0; b < BS; ++b) {
            const uint32_t h = pack2<T>(a[b], 0.f) & 0xFFFFu;
            raw[b] = col ? ((raw[b] & 0x0000FFFFu) | (h << 16)) : ((raw[b] & 0xFFFF0000u) | h);
            if (A.mask && a0 + b < A.axis_len) A.mask[base + (int64_t)b * A.post + col] = (uint8_t)((mkw[b >> 5] >> (b & 31)) & 1u);
        }
        outlier_side_outputs<BS>(A, mkw, se_in, se_out, status, p, nb, q + col);
    }
#pragma unroll
    for (int b = 0; b < BS; ++b)
        if (a0 + b < A.axis_len) *reinterpret_cast<uint32_t*>(out16 + base + (int64_t)b * A.post) = raw[b];
}

// --- layout B: post == 1 (block contiguous).  A wave owns 64 consecutive blocks.
// When axis_len % BS == 0 they are one contiguous run of 64*BS floats: the wave
// streams it with 16-byte coalesced accesses and transposes through LDS (row stride
// BS+4 floats: conflict-free ds_read_b128 for 16-lane groups) so that each lane ends
// up with its own block in registers; results go back the same way.
template <int BS, typename T, int FAST>
__global__ void __launch_bounds__(256)
k_outlier_contig(const T* __restrict__ in, T* __restrict__ out, OutlierArgs A) {
    constexpr int LDS_STRIDE = BS + 4;
    __shared__ __attribute__((aligned(16))) float tile[4][64 * LDS_STRIDE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t nblocks = A.pre * A.nblk;
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + wv) * 64;      // first block of this wave
    if (g0 >= nblocks) return;
    const bool fast = (A.axis_len % BS == 0) && (g0 + 64 <= nblocks);
    float a[BS];
    const int64_t g = g0 + lane;
    const int64_t p = g / A.nblk, nb = g % A.nblk;
    const int64_t a0 = nb * BS;
    const int64_t base = p * A.axis_len + a0;
    float* tl = tile[wv];
    if (fast) {
        if constexpr (sizeof(T) == 2) {                           // 8 bf16 / fp16 per 16-byte load
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(in) + g0 * BS);
#pragma unroll
            for (int t = 0; t < BS / 8; ++t) {
                const int f = lane + 64 * t;
                const int row = f / (BS / 8), c8 = f % (BS / 8);
                const uint4 v = src[f];
                float4 x, y;
                unpack2<T>(v.x, x.x, x.y); unpack2<T>(v.y, x.z, x.w); unpack2<T>(v.z, y.x, y.y); unpack2<T>(v.w, y.z, y.w);
                *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c8 * 8) = x;
                *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c8 * 8 + 4) = y;
            }
        } else {
        const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(in) + g0 * BS);
#pragma unroll
        for (int t = 0; t < BS / 4; ++t) {
            const int f = lane + 64 * t;                         // float4 index inside the 64xBS tile
            const int row = f / (BS / 4), c4 = f % (BS / 4);
            *reinterpret_cast<float4*>(tl + row * LDS_STRIDE + c4 * 4) = src[f];
        }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
#pragma unroll
        for (int c = 0; c < BS / 4; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(tl + lane * LDS_STRIDE + c * 4);
            a[c * 4 + 0] = v.x; a[c * 4 + 1] = v.y; a[c * 4 + 2] = v.z; a[c * 4 + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int b = 0; b < BS; ++b)
            a[b] = (g < nblocks && a0 + b < A.axis_len) ? IO<T>::ld(in, base + b) : 0.f;
    }
    uint32_t mkw[(BS + 31) / 32];
    float se_in, se_out;
    int status = 0;
    if (g < nblocks) {
        const float* vm = A.vmean ? A.vmean + p * BS : nullptr;
        const float* vs = A.vstd ? A.vstd + p * BS : nullptr;
        if (FAST == 1) status = outlier_block_fast<BS, 0>(a, mkw, se_in, se_out, A, 1, vm, vs, 1);
        else if (FAST == 2) status = outlier_block_fast<BS, -1>(a, mkw, se_in, se_out, A, 1, vm, vs, 1);
        else if (FAST == 3) status = outlier_block_fast<BS, 0, false, 1>(a, mkw, se_in, se_out, A, 1, vm, vs, 1);
        else if (FAST == 4) status = outlier_block_fast<BS, 0, false, 2>(a, mkw, se_in, se_out, A, 1, vm, vs, 1, nullptr, 0, 0,
                                                                         fast ? tl + lane * LDS_STRIDE : nullptr);
        else status = outlier_block<BS>(a, mkw, se_in, se_out, A, 1, vm, vs, 1);
    }
    if (fast) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < BS / 4; ++c)
            *reinterpret_cast<float4*>(tl + lane * LDS_STRIDE + c * 4) =
                make_float4(a[c * 4 + 0], a[c * 4 + 1], a[c * 4 + 2], a[c * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if constexpr (sizeof(T) == 2) {
            uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out) + g0 * BS);
#pragma unroll
            for (int t = 0; t < BS / 8; ++t) {
                const int f = lane + 64 * t;
                const int row = f / (BS / 8), c8 = f % (BS / 8);
                const float4 x = *reinterpret_cast<const float4*>(tl + row * LDS_STRIDE + c8 * 8);
                const float4 y = *reinterpret_cast<const float4*>(tl + row * LDS_STRIDE + c8 * 8 + 4);
                dst[f] = make_uint4(pack2<T>(x.x, x.y), pack2<T>(x.z, x.w), pack2<T>(y.x, y.y), pack2<T>(y.z, y.w));
            }
        } else {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + g0 * BS);
#pragma unroll
        for (int t = 0; t < BS / 4; ++t) {
            const int f = lane + 64 * t;
            const int row = f / (BS / 4), c4 = f % (BS / 4);
            dst[f] = *reinterpret_cast<const float4*>(tl + row * LDS_STRIDE + c4 * 4);
        }
        }
    } else if (g < nblocks) {
#pragma unroll
        for (int b = 0; b < BS; ++b)
            if (a0 + b < A.axis_len) IO<T>::st(out, base + b, a[b]);
    }
    if (g < nblocks) {
        if (A.mask) {
#pragma unroll
            for (int b = 0; b < BS; ++b)
                if (a0 + b < A.axis_len) A.mask[base + b] = (uint8_t)((mkw[b >> 5] >> (b & 31)) & 1u);
        }
        outlier_side_outputs<BS>(A, mkw, se_in, se_out, status, p, nb, 0);
    }
}

// launch one of the two kernels for the compile-time FAST variant; returns false for an unsupported block size
template <int FAST, typename T = float>
static inline bool launch_outlier_variant(const void* in, void* out, const OutlierArgs& A, int block, hipStream_t st) {
    const int64_t nthreads = A.pre * A.nblk * A.post;
    int64_t g = (nthreads + 255) / 256; if (g < 1) g = 1;
    const dim3 grid((unsigned)g), blk(256);
    if constexpr (sizeof(T) == 2 && (FAST == 3 || FAST == 4)) {
        if (A.post >= 2 && (A.post % 2) == 0 && block <= 32 && (((uintptr_t)in | (uintptr_t)out) & 3) == 0) {
            const dim3 gp((unsigned)((nthreads / 2 + 255) / 256));
            switch (block) {
                case 8: hipLaunchKernelGGL((k_outlier_strided_pair<8, T, FAST>), gp, blk, 0, st, (const T*)in, (T*)out, A); return true;
                case 16: hipLaunchKernelGGL((k_outlier_strided_pair<16, T, FAST>), gp, blk, 0, st, (const T*)in, (T*)out, A); return true;
                case 32: hipLaunchKernelGGL((k_outlier_strided_pair<32, T, FAST>), gp, blk, 0, st, (const T*)in, (T*)out, A); return true;
                default: break;
            }
        }
    }
#define MSQ_OL(BS)                                                                                          \
    case BS:                                                                                                \
        if (A.post == 1) hipLaunchKernelGGL((k_outlier_contig<BS, T, FAST>), grid, blk, 0, st, (const T*)in, (T*)out, A);  \
        else hipLaunchKernelGGL((k_outlier_strided<BS, T, FAST>), grid, blk, 0, st, (const T*)in, (T*)out, A);             \
        return true;
    switch (block) { MSQ_OL(8) MSQ_OL(16) MSQ_OL(32) MSQ_OL(64) MSQ_OL(128) default: return false; }
#undef MSQ_OL
}
