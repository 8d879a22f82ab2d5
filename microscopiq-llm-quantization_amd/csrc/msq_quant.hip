// msq_quant.hip -- gfx950 kernels for the quant/dequant half of the hot path and
// their C-ABI entry points (include/msq.h).  HBM-bound byte/bit work: one read and
// one write per element, wave64-coalesced, block statistics kept in registers.
//
// Compiled with -ffp-contract=off: the fp32 results must match the reference's
// CPU arithmetic bit for bit (see msq_device.h).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
#include <string.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"

using namespace msq;

// ===========================================================================
// dtype helpers
// ===========================================================================
#include "msq_outlier_kernels.h"
template <> struct IO<__half> {
    static MSQ_D float ld(const __half* p, int64_t i) { return __half2float(p[i]); }
    static MSQ_D void st(__half* p, int64_t i, float v) { p[i] = __float2half(v); }
};
template <> struct IO<__hip_bfloat16> {
    static MSQ_D float ld(const __hip_bfloat16* p, int64_t i) { return __bfloat162float(p[i]); }
    static MSQ_D void st(__hip_bfloat16* p, int64_t i, float v) { p[i] = __float2bfloat16(v); }
};

// ===========================================================================
// elementwise quantise (replaces cpp/elemwise.cuh:17-38).  Grid-stride, 4 elements
// per lane per step for f32 (16-byte accesses), HBM-bound.
// ===========================================================================
__global__ void __launch_bounds__(256)
k_elemwise_f32(const float* __restrict__ in, float* __restrict__ out, int64_t n, int bits, int ebits,
               float max_norm, int rmode, int saturate, int allow_denorm) {
    const int64_t nvec = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        float4 v = reinterpret_cast<const float4*>(in)[i];
        v.x = quant_bits(v.x, bits, ebits, max_norm, rmode, saturate, allow_denorm);
        v.y = quant_bits(v.y, bits, ebits, max_norm, rmode, saturate, allow_denorm);
        v.z = quant_bits(v.z, bits, ebits, max_norm, rmode, saturate, allow_denorm);
        v.w = quant_bits(v.w, bits, ebits, max_norm, rmode, saturate, allow_denorm);
        reinterpret_cast<float4*>(out)[i] = v;
    }
    const int64_t tail = nvec << 2;
    const int64_t t = tail + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = quant_bits(in[t], bits, ebits, max_norm, rmode, saturate, allow_denorm);
}

// 8 half / bfloat16 elements per lane per access (16 bytes), tail by the scalar kernel below
template <typename T>
__global__ void __launch_bounds__(256)
k_elemwise_16x8(const T* __restrict__ in, T* __restrict__ out, int64_t nvec, int bits, int ebits,
                float max_norm, int rmode, int saturate, int allow_denorm) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        union { uint4 u; T e[8]; } v;
        v.u = reinterpret_cast<const uint4*>(in)[i];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            IO<T>::st(v.e, k, quant_bits(IO<T>::ld(v.e, k), bits, ebits, max_norm, rmode, saturate, allow_denorm));
        reinterpret_cast<uint4*>(out)[i] = v.u;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_elemwise_16(const T* __restrict__ in, T* __restrict__ out, int64_t n, int bits, int ebits,
              float max_norm, int rmode, int saturate, int allow_denorm) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        IO<T>::st(out, i, quant_bits(IO<T>::ld(in, i), bits, ebits, max_norm, rmode, saturate, allow_denorm));
}

// elementwise quantise to a NAMED format (posit<n,es> included) with the saturating,
// denorm-keeping codec the MicroScopiQ quantiser uses (utils/quant.py:218-221)
__global__ void __launch_bounds__(256)
k_elem_format(const float* __restrict__ in, float* __restrict__ out, int64_t n, Fmt f, int rmode) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = quant_elem(in[i], f, rmode);
}

// ===========================================================================
// MX shared scale with the native semantics (cpp/shared_exp.cuh:14-53): NaN scale on
// overflow, 2^-emax floor, subnormal/NaN scale mantissa bit.
// ===========================================================================
MSQ_D float mx_shared_scale(int shared_exp, int scale_bits, float elem_max_norm) {
    const int elem_emax = (int)((f2u(elem_max_norm) >> 23) & 0xFF) - 127;
    if (shared_exp != 255) shared_exp -= elem_emax;
    const int emax = scale_bits != 0 ? (1 << (scale_bits - 1)) - 1 : 255;
    const int ub = shared_exp - 127;
    if (ub > emax) shared_exp = 255;
    if (ub < -emax) shared_exp = 127 - emax;
    const uint32_t mant = (shared_exp == 0 || shared_exp == 255) ? (1u << 22) : 0u;
    return u2f(((uint32_t)shared_exp << 23) | mant);
}

// exponent of one element for the block maximum: the exponent field (the reference's native kernels, cpp/shared_exp.cuh) or, with pyexp,
// what its Python path gets from floor(torch.log2(.)) (msq_device.h biased_exp_py); and the element codec: the native bit codec, or --
// Python path under truncation only, where the private exponent of elemwise_ops.py:139-144 decides the grid -- the arithmetic one
// PY is a template parameter like EPS: as a run-time argument (round 5) it put both codecs into every kernel and the
// register tile of k_mx_tile_cols4 into scratch (272 / 528 bytes per lane, 2.5-2.9x slower); `make check-resources` gates it
template <bool PY> MSQ_D int mx_exp_of(uint32_t bits) { return PY ? biased_exp_py(bits) : (int)((bits >> 23) & 0xFF); }
template <bool PY> MSQ_D float mx_elem(float si, int mbits, int ebits, float max_norm, int rmode) {
    if (PY && rmode == 1) return quant_core_sat(si, mbits, ebits, max_norm, rmode);
    const float q = quant_bits(si, mbits, ebits, max_norm, rmode, true, true);
    if (PY) {                                                    // the Python path's floor(|x| + 0.5) in float32: msq_device.h half_away_quirk_bits
        const int t = (ebits ? 2 - (1 << (ebits - 1)) : 0) - mbits + 1;
        if ((f2u(si) & 0x7FFFFFFFu) == half_away_quirk_bits(t)) return __builtin_copysignf(pow2i(t + 1), si);
    }
    return q;
}

// quantize_mx with precomputed max values (replaces cpp/mx.cuh:15-53)
__global__ void __launch_bounds__(256)
k_mx_maxvals(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ maxv,
             int64_t total, int64_t axis_len, int64_t post, int scale_bits, int ebits, int mbits,
             float max_norm, int flush, int rmode) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t q = i % post;
        const int64_t p = i / (post * axis_len);
        const int se = (int)((f2u(maxv[p * post + q]) >> 23) & 0xFF);
        const bool fl = (se == 0) && flush;
        const float scale = mx_shared_scale(se, scale_bits, max_norm);
        const float si = fl ? 0.f : in[i] / scale;
        out[i] = quant_bits(si, mbits, ebits, max_norm, rmode, true, true) * scale;
    }
}

// quantize_mx_by_tile (replaces cpp/mx.cuh:63-170).  One kernel for both layouts:
//  post > 1 : one lane per (tile, q) column, lanes run along q -> every row access of
//             the wave is one coalesced segment; the tile lives in registers (<= 128)
//             or is re-read (L2-resident) for larger tiles.
//  post == 1: TS lanes of the wave share a tile (tile innermost, power of two <= 64):
//             max biased exponent by a DPP/shuffle butterfly inside the wave.
// EPS: the divisor of the reference's PYTHON path, `2**shared_exp + 1e-6` in fp32 (mx_ops.py:444; custom_cuda = False) --
// the native path the entry replaces divides by the scale itself (cpp/mx.cuh:132).
template <int TS, bool EPS = false, bool PY = false>
__global__ void __launch_bounds__(256)
k_mx_tile_inner(const float* __restrict__ in, float* __restrict__ out, int64_t total, int scale_bits,
                int ebits, int mbits, float max_norm, int flush, int rmode) {
    // axis_len % TS == 0, TS power of two <= 64: element i belongs to tile i / TS
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float v = (i < total) ? in[i] : 0.f;
    uint32_t um = f2u(v) & 0x7FFFFFFFu;
#pragma unroll
    for (int m = TS / 2; m > 0; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)um, m, 64);
        um = o > um ? o : um;
    }
    const int se = mx_exp_of<PY>(um);
    const bool fl = (se == 0) && flush;
    const float scale = mx_shared_scale(se, scale_bits, max_norm);
    const float si = fl ? 0.f : v / (EPS ? scale + 1e-6f : scale);
    if (i < total) out[i] = mx_elem<PY>(si, mbits, ebits, max_norm, rmode) * scale;
}

template <bool EPS = false, bool PY = false>
__global__ void __launch_bounds__(256)
k_mx_tile_generic(const float* __restrict__ in, float* __restrict__ out, int64_t pre, int64_t axis_len,
                  int64_t post, int tile, int64_t ntiles, int scale_bits, int ebits, int mbits,
                  float max_norm, int flush, int rmode) {
    const int64_t total = pre * ntiles * post;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t q = t % post;
        const int64_t ti = (t / post) % ntiles;
        const int64_t p = t / (post * ntiles);
        const int64_t a0 = ti * tile;
        int64_t a1 = a0 + tile; a1 = a1 > axis_len ? axis_len : a1;
        const int64_t base = (p * axis_len) * post + q;
        uint32_t um = 0;
        for (int64_t a = a0; a < a1; ++a) {
            const uint32_t u = f2u(in[base + a * post]) & 0x7FFFFFFFu;
            um = u > um ? u : um;
        }
        const int se = mx_exp_of<PY>(um);
        const bool fl = (se == 0) && flush;
        const float scale = mx_shared_scale(se, scale_bits, max_norm);
        for (int64_t a = a0; a < a1; ++a) {
            const float si = fl ? 0.f : in[base + a * post] / (EPS ? scale + 1e-6f : scale);
            out[base + a * post] = mx_elem<PY>(si, mbits, ebits, max_norm, rmode) * scale;
        }
    }
}

// 16-byte versions (HBM-bound: 4 elements per lane per access, like k_elemwise_f32).  The division by the power-of-two
// scale is a multiplication by its exact reciprocal: both are the correctly rounded value of the same real number
// (the reciprocal 2^-k, k in [-127, 127], is always a float, subnormal at k = 127), NaN scales stay NaN.
MSQ_D float mx_scale_recip(float scale) {
    const uint32_t u = f2u(scale);
    const int e = (int)((u >> 23) & 0xFF);
    if (e == 255) return scale;                                  // NaN scale (overflow of the scale format)
    if (e == 0) return u2f(254u << 23);                          // 2^-127 -> 2^127
    return e == 254 ? u2f(1u << 22) : u2f((uint32_t)(254 - e) << 23);
}

template <bool EPS, bool PY>
MSQ_D float mx_apply(float v, float scale, float rs, bool fl, int mbits, int ebits, float max_norm, int rmode) {
    const float si = fl ? 0.f : (EPS ? v / (scale + 1e-6f) : v * rs);
    return mx_elem<PY>(si, mbits, ebits, max_norm, rmode) * scale;
}

// tile innermost (post == 1), TS in {4, 8, 16, 32, 64}: TS / 4 neighbouring lanes share a tile
template <int TS, bool EPS, bool PY>
__global__ void __launch_bounds__(256)
k_mx_tile_inner4(const float* __restrict__ in, float* __restrict__ out, int64_t nvec, int scale_bits,
                 int ebits, int mbits, float max_norm, int flush, int rmode) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x; b < nvec; b += stride) {
        const int64_t i = b + threadIdx.x;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nvec) v = reinterpret_cast<const float4*>(in)[i];
        const uint32_t m01 = max(f2u(v.x) & 0x7FFFFFFFu, f2u(v.y) & 0x7FFFFFFFu);
        const uint32_t m23 = max(f2u(v.z) & 0x7FFFFFFFu, f2u(v.w) & 0x7FFFFFFFu);
        uint32_t um = max(m01, m23);
#pragma unroll
        for (int m = TS / 8; m > 0; m >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)um, m, 64);
            um = o > um ? o : um;
        }
        const int se = mx_exp_of<PY>(um);                         // monotone in |bits|: the rule once, on the maximum
        const bool fl = (se == 0) && flush;
        const float scale = mx_shared_scale(se, scale_bits, max_norm);
        const float rs = mx_scale_recip(scale);
        v.x = mx_apply<EPS, PY>(v.x, scale, rs, fl, mbits, ebits, max_norm, rmode);
        v.y = mx_apply<EPS, PY>(v.y, scale, rs, fl, mbits, ebits, max_norm, rmode);
        v.z = mx_apply<EPS, PY>(v.z, scale, rs, fl, mbits, ebits, max_norm, rmode);
        v.w = mx_apply<EPS, PY>(v.w, scale, rs, fl, mbits, ebits, max_norm, rmode);
        if (i < nvec) reinterpret_cast<float4*>(out)[i] = v;
    }
}

// tile along an outer axis (post % 4 == 0, tile <= TILE): one lane per (tile, 4 neighbouring columns); the TILE x 4
// values stay in registers between the exponent scan and the quantisation, every access is a coalesced 16-byte one
template <int TILE, bool EPS, bool PY>
__global__ void __launch_bounds__(256)
k_mx_tile_cols4(const float* __restrict__ in, float* __restrict__ out, int64_t pre, int64_t axis_len, int64_t post4,
                int tile, int64_t ntiles, int scale_bits, int ebits, int mbits, float max_norm, int flush, int rmode) {
    const int64_t total = pre * ntiles * post4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t q = t % post4;
    const int64_t ti = (t / post4) % ntiles;
    const int64_t p = t / (post4 * ntiles);
    const int64_t a0 = ti * tile;
    int rows = (int)(axis_len - a0); rows = rows > tile ? tile : rows;
    const float4* src = reinterpret_cast<const float4*>(in) + (p * axis_len + a0) * post4 + q;
    float4* dst = reinterpret_cast<float4*>(out) + (p * axis_len + a0) * post4 + q;
    float4 r[TILE];
    // the exponent rule (field, or the Python path's floor(log2)) is monotone in |bits|: one unsigned maximum per element,
    // the rule once per column (inside the unrolled loop its rare log2f branch x 4 x TILE kept hipcc from unrolling at all)
    uint32_t ux = 0, uy = 0, uz = 0, uw = 0;
#pragma unroll
    for (int j = 0; j < TILE; ++j) {
        r[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < rows) r[j] = src[(int64_t)j * post4];
        ux = max(ux, f2u(r[j].x) & 0x7FFFFFFFu); uy = max(uy, f2u(r[j].y) & 0x7FFFFFFFu);
        uz = max(uz, f2u(r[j].z) & 0x7FFFFFFFu); uw = max(uw, f2u(r[j].w) & 0x7FFFFFFFu);
    }
    const int ex = mx_exp_of<PY>(ux), ey = mx_exp_of<PY>(uy), ez = mx_exp_of<PY>(uz), ew = mx_exp_of<PY>(uw);
    const float sx = mx_shared_scale(ex, scale_bits, max_norm), sy = mx_shared_scale(ey, scale_bits, max_norm);
    const float sz = mx_shared_scale(ez, scale_bits, max_norm), sw = mx_shared_scale(ew, scale_bits, max_norm);
    const float rx = mx_scale_recip(sx), ry = mx_scale_recip(sy), rz = mx_scale_recip(sz), rw = mx_scale_recip(sw);
    const bool fx = (ex == 0) && flush, fy = (ey == 0) && flush, fz = (ez == 0) && flush, fw = (ew == 0) && flush;
    if (PY && rmode == 1) {
        // Python path under truncation: the arithmetic codec (log2f per element).  Kept OUT of the unrolled register-tile
        // loop -- with both codecs in its body hipcc gave up unrolling and put r[] into scratch -- and fed from the
        // (L2-resident) source again: the rare configuration pays a second read, the common one keeps its registers.
        for (int j = 0; j < rows; ++j) {
            const float4 v = src[(int64_t)j * post4];
            float4 o;
            o.x = quant_core_sat(fx ? 0.f : (EPS ? v.x / (sx + 1e-6f) : v.x * rx), mbits, ebits, max_norm, 1) * sx;
            o.y = quant_core_sat(fy ? 0.f : (EPS ? v.y / (sy + 1e-6f) : v.y * ry), mbits, ebits, max_norm, 1) * sy;
            o.z = quant_core_sat(fz ? 0.f : (EPS ? v.z / (sz + 1e-6f) : v.z * rz), mbits, ebits, max_norm, 1) * sz;
            o.w = quant_core_sat(fw ? 0.f : (EPS ? v.w / (sw + 1e-6f) : v.w * rw), mbits, ebits, max_norm, 1) * sw;
            dst[(int64_t)j * post4] = o;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TILE; ++j) {
        float4 o;
        o.x = mx_apply<EPS, PY>(r[j].x, sx, rx, fx, mbits, ebits, max_norm, rmode);
        o.y = mx_apply<EPS, PY>(r[j].y, sy, ry, fy, mbits, ebits, max_norm, rmode);
        o.z = mx_apply<EPS, PY>(r[j].z, sz, rz, fz, mbits, ebits, max_norm, rmode);
        o.w = mx_apply<EPS, PY>(r[j].w, sw, rw, fw, mbits, ebits, max_norm, rmode);
        if (j < rows) dst[(int64_t)j * post4] = o;
    }
}

// quantize_mx with precomputed max values, max over whole rows (post == 1, axis_len % 4 == 0): 16-byte accesses,
// one row-index division per 4 elements
__global__ void __launch_bounds__(256)
k_mx_maxvals_rows4(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ maxv,
                   int64_t nvec, int64_t row_vecs, int scale_bits, int ebits, int mbits, float max_norm, int flush, int rmode) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const int64_t p = (nvec >> 32) ? i / row_vecs : (int64_t)((uint32_t)i / (uint32_t)row_vecs);
        const int se = (int)((f2u(maxv[p]) >> 23) & 0xFF);
        const bool fl = (se == 0) && flush;
        const float scale = mx_shared_scale(se, scale_bits, max_norm);
        const float rs = mx_scale_recip(scale);
        float4 v = reinterpret_cast<const float4*>(in)[i];
        v.x = mx_apply<false, false>(v.x, scale, rs, fl, mbits, ebits, max_norm, rmode);
        v.y = mx_apply<false, false>(v.y, scale, rs, fl, mbits, ebits, max_norm, rmode);
        v.z = mx_apply<false, false>(v.z, scale, rs, fl, mbits, ebits, max_norm, rmode);
        v.w = mx_apply<false, false>(v.w, scale, rs, fl, mbits, ebits, max_norm, rmode);
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// ===========================================================================
// inner-dim reductions (replace cpp/reduce.cuh:154-210): one wave per row,
// 16-byte loads, wave64 shuffle tree; rows of any length >= 1.
// ===========================================================================
template <bool IS_MAX>
__global__ void __launch_bounds__(256)
k_reduce_inner(const float* __restrict__ in, float* __restrict__ out, int64_t outer, int64_t inner) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= outer) return;
    const float* r = in + row * inner;
    float acc = IS_MAX ? -__builtin_inff() : 0.f;
    const bool al = ((reinterpret_cast<uintptr_t>(r) & 15) == 0);
    int64_t j = 0;
    if (al) {
        const int64_t nv = inner >> 2;
        for (int64_t v = lane; v < nv; v += 64) {
            const float4 x = reinterpret_cast<const float4*>(r)[v];
            if (IS_MAX) { acc = fmaxf(acc, fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w))); }
            else { acc += (x.x + x.y) + (x.z + x.w); }
        }
        j = nv << 2;
    }
    for (int64_t k = j + lane; k < inner; k += 64) {
        if (IS_MAX) acc = fmaxf(acc, r[k]); else acc += r[k];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const float o = __shfl_xor(acc, m, 64);
        acc = IS_MAX ? fmaxf(acc, o) : acc + o;
    }
    if (lane == 0) out[row] = acc;
}

// short rows (inner = 4 G, G a power of two <= 16, 16-byte aligned): G neighbouring lanes share a row, so a wave
// covers 64 / G rows per access instead of one; the additions happen in the same order as in k_reduce_inner.
template <bool IS_MAX, int G>
__global__ void __launch_bounds__(256)
k_reduce_inner_short(const float* __restrict__ in, float* __restrict__ out, int64_t nvec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x; b < nvec; b += stride) {
        const int64_t i = b + threadIdx.x;
        float acc = IS_MAX ? -__builtin_inff() : 0.f;
        if (i < nvec) {
            const float4 x = reinterpret_cast<const float4*>(in)[i];
            acc = IS_MAX ? fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)) : (x.x + x.y) + (x.z + x.w);
        }
#pragma unroll
        for (int m = G / 2; m > 0; m >>= 1) {
            const float o = __shfl_xor(acc, m, 64);
            acc = IS_MAX ? fmaxf(acc, o) : acc + o;
        }
        if (i < nvec && (i & (G - 1)) == 0) out[i / G] = acc;
    }
}

// variant 1 statistics (mx_ops.py:62-66,248): mean / unbiased std of the SIGNED values
// over the block-count axis, one lane per (p, b, q).
__global__ void __launch_bounds__(256)
k_mxops_stats(const float* __restrict__ in, float* __restrict__ vmean, float* __restrict__ vstd,
              int64_t pre, int64_t axis_len, int64_t post, int bs, int64_t nblk, int* status) {
    const int64_t total = pre * bs * post;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t q = t % post;
    const int b = (int)((t / post) % bs);
    const int64_t p = t / (post * bs);
    const int64_t cols = (int64_t)bs * post, col = (int64_t)b * post + q;
    const int64_t lim = (cols >= 8) ? (cols / 32) * 32 : (cols / 4) * 4;
    const int order = (cols == 1) ? 1 : ((col < lim) ? 0 : 2);
    auto ld = [&](int64_t nb) -> float {
        const int64_t ai = nb * bs + b;
        return ai < axis_len ? in[(p * axis_len + ai) * post + q] : 0.f;
    };
    float s;
    if (order == 0) {                       // cascade, step 16 (nblk < 65536)
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        int64_t i = 0;
        while (i + 16 <= nblk) {
            for (int j = 0; j < 16; ++j, ++i) acc0 += ld(i);
            acc1 += acc0; acc0 = 0.f;
            if ((i & (15 << 4)) == 0) { acc2 += acc1; acc1 = 0.f;
                if ((i & (15 << 8)) == 0) { acc3 += acc2; acc2 = 0.f; } }
        }
        for (; i < nblk; ++i) acc0 += ld(i);
        acc0 += acc1; acc0 += acc2; acc0 += acc3;
        s = acc0;
    } else if (order == 2) {                // ilp4
        const int64_t S = nblk / 4;
        float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
        int64_t i = 0;
        while (i + 16 <= S) {
            for (int j = 0; j < 16; ++j, ++i) for (int k = 0; k < 4; ++k) a0[k] += ld(i * 4 + k);
            for (int k = 0; k < 4; ++k) { a1[k] += a0[k]; a0[k] = 0.f; }
            if ((i & (15 << 4)) == 0) for (int k = 0; k < 4; ++k) { a2[k] += a1[k]; a1[k] = 0.f; }
        }
        for (; i < S; ++i) for (int k = 0; k < 4; ++k) a0[k] += ld(i * 4 + k);
        for (int k = 0; k < 4; ++k) { a0[k] += a1[k]; a0[k] += a2[k]; }
        for (int64_t r = S * 4; r < nblk; ++r) a0[0] += ld(r);
        a0[0] += a0[1]; a0[0] += a0[2]; a0[0] += a0[3];
        s = a0[0];
    } else {                                // inner8 can only happen for bs*post == 1
        s = 0.f;
        for (int64_t i = 0; i < nblk; ++i) s += ld(i);
    }
    double mean = 0.0, m2 = 0.0;
    for (int64_t i = 0; i < nblk; ++i) {
        const double d = (double)ld(i);
        const double delta = d - mean;
        mean = mean + delta / (double)(i + 1);
        m2 = m2 + delta * (d - mean);
    }
    double den = (double)nblk - 1.0; den = den < 0 ? 0 : den;
    const float sd = (float)__builtin_sqrt(m2 / den);
    vmean[t] = s / (float)nblk;
    vstd[t] = sd;
    if (sd != sd && status) atomicOr(status, MSQ_STATUS_NAN);   // mx_ops.py:66 assert
}

// Row-parallel form of the same statistics for contiguous blocks (post == 1, BS a multiple of 32, K % BS == 0):
// one workgroup per row; thread (r, b) sums the 16-block runs r, r+R, ... of column b exactly as torch's
// cascade does (each run starts from 0), thread (0, b) then folds the run sums in cascade order.  The std is a
// two-pass variance in double (column mean first, then squared deviations, every thread over its own runs); it is
// rounded to float only when it is provably on the same side of the rounding boundary as torch's sequential
// Welford result (both are within a few hundred double ulps of the exact value), else the column is redone
// sequentially (about one column in 500 000).  Reads each row twice (second time from L2), coalesced.
// XT = float, or uint16_t for bfloat16 rows (every bf16 is an fp32 value: the same statistics as after a cast pass)
template <int BS, typename XT = float>
__global__ void __launch_bounds__(256)
k_mxops_stats_rows(const XT* __restrict__ in, float* __restrict__ vmean, float* __restrict__ vstd,
                   int64_t axis_len, int64_t nblk, int* status) {
    constexpr int R = 256 / BS;                 // threads per column
    constexpr int MAXRUNS = 64;                 // nblk <= 1024
    __shared__ float run_sum[MAXRUNS + 1][BS];
    __shared__ double part_s[R][BS], part_q[R][BS], part_d[R][BS];
    const int b = threadIdx.x % BS, r = threadIdx.x / BS;
    const int64_t p = blockIdx.x;
    const XT* rowp = in + p * axis_len;
    auto row = [&](int64_t i) -> float { if constexpr (sizeof(XT) == 4) return (float)rowp[i]; else return u2f((uint32_t)rowp[i] << 16); };
    const int nruns = (int)(nblk / 16), tail = (int)(nblk % 16);
    const bool has_tail = (r == (nruns % R));
    double ds = 0.0;
    for (int j = r; j < nruns; j += R) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = row((int64_t)(j * 16 + i) * BS + b);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc += v[i]; ds += (double)v[i]; }
        run_sum[j][b] = acc;
    }
    if (has_tail) {                              // the tail run (may be empty)
        float acc = 0.f;
        for (int i = 0; i < tail; ++i) { const float v = row((int64_t)(nruns * 16 + i) * BS + b); acc += v; ds += (double)v; }
        run_sum[MAXRUNS][b] = acc;
    }
    part_s[r][b] = ds;
    __syncthreads();
    double S = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) S += part_s[k][b];
    const double n = (double)nblk;
    const double dmean = S / n;
    double dq = 0.0, dd = 0.0;                   // sum of squared / plain deviations from dmean
    for (int j = r; j < nruns; j += R) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = row((int64_t)(j * 16 + i) * BS + b);
#pragma unroll
        for (int i = 0; i < 16; ++i) { const double d = (double)v[i] - dmean; dq = __builtin_fma(d, d, dq); dd += d; }
    }
    if (has_tail)
        for (int i = 0; i < tail; ++i) { const double d = (double)row((int64_t)(nruns * 16 + i) * BS + b) - dmean; dq = __builtin_fma(d, d, dq); dd += d; }
    part_q[r][b] = dq; part_d[r][b] = dd;
    __syncthreads();
    if (r != 0) return;
    // cascade fold (ATen: level-1 accumulator folded every 256 elements, level-2 every 4096)
    float acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int j = 0; j < nruns; ++j) {
        acc1 += run_sum[j][b];
        const int i = (j + 1) * 16;
        if ((i & (15 << 4)) == 0) { acc2 += acc1; acc1 = 0.f;
            if ((i & (15 << 8)) == 0) { acc3 += acc2; acc2 = 0.f; } }
    }
    float acc0 = run_sum[MAXRUNS][b];
    acc0 += acc1; acc0 += acc2; acc0 += acc3;
    double Q = 0.0, D = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) { Q += part_q[k][b]; D += part_d[k][b]; }
    double den = n - 1.0; den = den < 0 ? 0 : den;
    double m2 = Q - D * D / n; m2 = m2 < 0 ? 0 : m2;      // corrected two-pass: no cancellation
    double sdd = __builtin_sqrt(m2 / den);
    const uint64_t bits = __builtin_bit_cast(uint64_t, sdd);
    const uint32_t dropped = (uint32_t)(bits & 0x1FFFFFFFull);
    const uint32_t dist = dropped > 0x10000000u ? dropped - 0x10000000u : 0x10000000u - dropped;
    // two-pass error <= ~(n / 2 + 3) ulps, sequential Welford <= ~2 n ulps: a band of 4 n + 64 ulps is safe
    const bool safe = (dist > 4u * (uint32_t)nblk + 64u) && (sdd == sdd) && (sdd > 1e-150) && (sdd < 1e150);
    if (!safe) {
        double mean = 0.0; m2 = 0.0;
        int64_t i = 0;
        for (; i + 16 <= nblk; i += 16) {                    // loads in batches, the recurrence stays sequential
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = row((i + u) * BS + b);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const double d = (double)v[u];
                const double delta = d - mean;
                mean = mean + delta / (double)(i + u + 1);
                m2 = m2 + delta * (d - mean);
            }
        }
        for (; i < nblk; ++i) {
            const double d = (double)row(i * BS + b);
            const double delta = d - mean;
            mean = mean + delta / (double)(i + 1);
            m2 = m2 + delta * (d - mean);
        }
        sdd = __builtin_sqrt(m2 / den);
    }
    const float sd = (float)sdd;
    vmean[p * BS + b] = acc0 / (float)nblk;
    vstd[p * BS + b] = sd;
    if (sd != sd && status) atomicOr(status, MSQ_STATUS_NAN);
}

// ===========================================================================
// C ABI
// ===========================================================================
static thread_local char g_err[256] = "";
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
extern "C" void msq_set_error_(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return MSQ_ERR_LAUNCH;
    }
    return MSQ_OK;
}
static inline int grid_for(int64_t n, int block, int64_t cap = 1 << 30) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// implemented in msq_quant_hw.hip (hardware-convert variants, own translation unit)
extern "C" int msq_launch_outlier_lowp_(const void* in, void* out, const void* args, int block, int dt, void* ws, int64_t ws_bytes, void* stream);   // msq_quant_lowp.hip
extern "C" int64_t msq_outlier_lowp_ws_bytes_(int64_t pre, int64_t axis_len, int64_t post, int block);
extern "C" int msq_launch_outlier_f32sem_(const void* in, void* out, const void* args, int block, int dt, void* ws, int64_t ws_bytes, void* stream);
extern "C" int msq_launch_outlier_hw_(const void* in, void* out, const OutlierArgs* A, int block, int mode, int dtype, void* stream);

// dtype 1 / 2 (fp16 / bf16 tensors, computed in fp32) is built for round-to-nearest with float / int inliers (the hardware-convert variants and
// the nearest-specialised arithmetic one); everything else is f32 only (the host shim upcasts)
static int launch_outlier(const void* in, void* out, OutlierArgs& A, int block, hipStream_t st, int dtype = 0, void* ws = nullptr, int64_t ws_bytes = 0) {
    bool ok;
    // fp16 / bf16 tensors computed in float32: the packed kernels of msq_quant_lowp.hip where they apply (same bits, 2-3 x faster)
    if ((dtype == 1 || dtype == 2) && msq_launch_outlier_f32sem_(in, out, &A, block, dtype, ws, ws_bytes, (void*)st)) return MSQ_OK;
    if (dtype != 0 && !((dtype == 1 || dtype == 2) && A.fi.kind == 0 && A.rmode == 0))
        return fail(MSQ_ERR_UNSUPPORTED, "msq_outlier_fakequant: this dtype / format / rounding combination is f32 only; the host shim upcasts");
    if (A.fi.kind == 0) {                                      // float/int inliers; outliers float/int or posit
        if (A.rmode == 0) {
            const int ih = hw_codec_kind(A.fi), oh = hw_codec_kind(A.fo);
            if (ih && oh) return msq_launch_outlier_hw_(in, out, &A, block, 1, dtype, (void*)st);            // both through the converts
            if (ih && A.fo.kind == 1) return msq_launch_outlier_hw_(in, out, &A, block, 2, dtype, (void*)st); // inliers only, posit outliers
            if (dtype == 2) ok = launch_outlier_variant<1, bf16io_t>(in, out, A, block, st);
            else if (dtype == 1) ok = launch_outlier_variant<1, f16io_t>(in, out, A, block, st);
            else ok = launch_outlier_variant<1>(in, out, A, block, st);
        } else ok = launch_outlier_variant<2>(in, out, A, block, st);
    } else ok = launch_outlier_variant<0>(in, out, A, block, st);   // posit inliers: generic maths
    if (!ok) return fail(MSQ_ERR_UNSUPPORTED, "msq_outlier_fakequant: block size must be 8, 16, 32, 64 or 128");
    return MSQ_OK;
}

template <bool IS_MAX>
static bool launch_reduce_short(const float* in, float* out, int64_t outer, int64_t inner, hipStream_t st) {
    if (inner < 4 || inner > 64 || (inner & (inner - 1)) || (reinterpret_cast<uintptr_t>(in) & 15)) return false;
    const int64_t nvec = outer * (inner / 4);
    const dim3 g(grid_for(nvec, 256, 2048 * 4)), b(256);
    switch (inner / 4) {
        case 1: hipLaunchKernelGGL((k_reduce_inner_short<IS_MAX, 1>), g, b, 0, st, in, out, nvec); break;
        case 2: hipLaunchKernelGGL((k_reduce_inner_short<IS_MAX, 2>), g, b, 0, st, in, out, nvec); break;
        case 4: hipLaunchKernelGGL((k_reduce_inner_short<IS_MAX, 4>), g, b, 0, st, in, out, nvec); break;
        case 8: hipLaunchKernelGGL((k_reduce_inner_short<IS_MAX, 8>), g, b, 0, st, in, out, nvec); break;
        default: hipLaunchKernelGGL((k_reduce_inner_short<IS_MAX, 16>), g, b, 0, st, in, out, nvec); break;
    }
    return true;
}

template <typename T>
static void launch_elemwise_16(const void* in, void* out, int64_t n, int bits, int ebits, float max_norm, int rmode,
                               int saturate, int allow_denorm, hipStream_t st) {
    int64_t done = 0;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && n >= 8) {
        const int64_t nvec = n / 8;
        hipLaunchKernelGGL(k_elemwise_16x8<T>, dim3(grid_for(nvec, 256, 2048 * 4)), dim3(256), 0, st, (const T*)in, (T*)out,
                           nvec, bits, ebits, max_norm, rmode, saturate, allow_denorm);
        done = nvec * 8;
    }
    if (done < n)
        hipLaunchKernelGGL(k_elemwise_16<T>, dim3(grid_for(n - done, 256, 8192)), dim3(256), 0, st, (const T*)in + done,
                           (T*)out + done, n - done, bits, ebits, max_norm, rmode, saturate, allow_denorm);
}

template <bool EPS, bool PY>
static int launch_mx_by_tile(const float* in, float* out, int64_t pre, int64_t axis_len, int64_t post,
                             int tile_size, int scale_bits, int elem_ebits, int elem_mbits,
                             float elem_max_norm, int flush_fp32_subnorms, int rmode, void* stream) {
    if (pre < 0 || axis_len < 0 || post < 0) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_mx_by_tile: negative size");
    const int64_t total = pre * axis_len * post;
    if (total == 0) return MSQ_OK;
    if (!in || !out) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_mx_by_tile: null buffer");
    if (rmode < 0 || rmode > 2) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_mx_by_tile: bad rounding mode");
    if (tile_size <= 0) tile_size = (int)axis_len;
    hipStream_t st = (hipStream_t)stream;
    const bool pow2 = (tile_size & (tile_size - 1)) == 0;
    const bool al16 = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (post == 1 && pow2 && tile_size >= 4 && tile_size <= 64 && axis_len % tile_size == 0 && al16) {
        const int64_t nvec = total / 4;
        const int g = grid_for(nvec, 256, 2048 * 4);
#define MSQ_TI(TS) case TS: hipLaunchKernelGGL((k_mx_tile_inner4<TS, EPS, PY>), dim3(g), dim3(256), 0, st, in, out, nvec, \
                        scale_bits, elem_ebits, elem_mbits, elem_max_norm, flush_fp32_subnorms, rmode); break;
        switch (tile_size) { MSQ_TI(4) MSQ_TI(8) MSQ_TI(16) MSQ_TI(32) MSQ_TI(64) }
#undef MSQ_TI
    } else if (post > 1 && post % 4 == 0 && tile_size <= 32 && al16 && pre * ((axis_len + tile_size - 1) / tile_size) * (post / 4) < (int64_t)0x7FFFFFFF * 256) {
        const int64_t ntiles = (axis_len + tile_size - 1) / tile_size;
        const int64_t nthreads = pre * ntiles * (post / 4);
        const unsigned g = (unsigned)((nthreads + 255) / 256);
        if (tile_size <= 16)
            hipLaunchKernelGGL((k_mx_tile_cols4<16, EPS, PY>), dim3(g), dim3(256), 0, st, in, out, pre, axis_len, post / 4, tile_size,
                               ntiles, scale_bits, elem_ebits, elem_mbits, elem_max_norm, flush_fp32_subnorms, rmode);
        else
            hipLaunchKernelGGL((k_mx_tile_cols4<32, EPS, PY>), dim3(g), dim3(256), 0, st, in, out, pre, axis_len, post / 4, tile_size,
                               ntiles, scale_bits, elem_ebits, elem_mbits, elem_max_norm, flush_fp32_subnorms, rmode);
    } else if (post == 1 && pow2 && tile_size <= 64 && axis_len % tile_size == 0) {
        const int g = grid_for(total, 256);
#define MSQ_TI(TS) case TS: hipLaunchKernelGGL((k_mx_tile_inner<TS, EPS, PY>), dim3(g), dim3(256), 0, st, in, out, total, \
                        scale_bits, elem_ebits, elem_mbits, elem_max_norm, flush_fp32_subnorms, rmode); break;
        switch (tile_size) { MSQ_TI(1) MSQ_TI(2) MSQ_TI(4) MSQ_TI(8) MSQ_TI(16) MSQ_TI(32) MSQ_TI(64) }
#undef MSQ_TI
    } else {
        const int64_t ntiles = (axis_len + tile_size - 1) / tile_size;
        hipLaunchKernelGGL((k_mx_tile_generic<EPS, PY>), dim3(grid_for(pre * ntiles * post, 256, 16384)), dim3(256), 0, st,
                           in, out, pre, axis_len, post, tile_size, ntiles, scale_bits, elem_ebits, elem_mbits,
                           elem_max_norm, flush_fp32_subnorms, rmode);
    }
    return check_launch("msq_quantize_mx_by_tile");
}

extern "C" {

int msq_version(void) { return 200; }
const char* msq_last_error(void) { return g_err; }

int msq_format_id(const char* name) { return msq_host::format_id(name); }

int msq_format_params(int fmt, int* ebits, int* mbits, int* emax, float* max_norm, float* min_norm, int* kind) {
    msq_host::FmtInfo f;
    if (!msq_host::format_info(fmt, &f)) return fail(MSQ_ERR_BAD_ARG, "msq_format_params: unknown format id");
    if (ebits) *ebits = f.ebits;
    if (mbits) *mbits = f.mbits;
    if (emax) *emax = f.emax;
    if (max_norm) *max_norm = f.max_norm;
    if (min_norm) *min_norm = f.min_norm;
    if (kind) *kind = f.kind;
    return MSQ_OK;
}

int msq_quantize_elemwise(const void* in, void* out, int64_t n, int dtype, int bits, int exp_bits,
                          float max_norm, int rmode, int saturate_normals, int allow_denorm, void* stream) {
    if (n < 0 || (n > 0 && (!in || !out))) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_elemwise: null buffer");
    if (bits > 24 || bits < 2) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_elemwise: bits must be in [2,24]");  // funcs.cpp:193
    if (rmode < 0 || rmode > 2) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_elemwise: bad rounding mode");
    if (n == 0) return MSQ_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15)
            return fail(MSQ_ERR_BAD_ARG, "msq_quantize_elemwise: f32 buffers must be 16-byte aligned");
        const int g = grid_for((n + 3) / 4, 256, 2048 * 4);
        hipLaunchKernelGGL(k_elemwise_f32, dim3(g), dim3(256), 0, st, (const float*)in, (float*)out, n, bits,
                           exp_bits, max_norm, rmode, saturate_normals, allow_denorm);
    } else if (dtype == 1) {
        launch_elemwise_16<__half>(in, out, n, bits, exp_bits, max_norm, rmode, saturate_normals, allow_denorm, st);
    } else if (dtype == 2) {
        launch_elemwise_16<__hip_bfloat16>(in, out, n, bits, exp_bits, max_norm, rmode, saturate_normals, allow_denorm, st);
    } else return fail(MSQ_ERR_UNSUPPORTED, "msq_quantize_elemwise: dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    return check_launch("msq_quantize_elemwise");
}

int msq_quantize_format(const float* in, float* out, int64_t n, int fmt, int rmode, void* stream) {
    if (n < 0 || (n > 0 && (!in || !out))) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_format: null buffer");
    if (rmode < 0 || rmode > 2) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_format: bad rounding mode");
    msq_host::FmtInfo fi;
    if (!msq_host::format_info(fmt, &fi)) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_format: unknown element format");
    if (n == 0) return MSQ_OK;
    hipLaunchKernelGGL(k_elem_format, dim3(grid_for(n, 256, 16384)), dim3(256), 0, (hipStream_t)stream, in, out, n,
                       Fmt{fi.kind, fi.ebits, fi.mbits, fi.emax, fi.max_norm}, rmode);
    return check_launch("msq_quantize_format");
}

int msq_quantize_mx(const float* in, float* out, const float* max_values, int64_t pre, int64_t axis_len,
                    int64_t post, int scale_bits, int elem_ebits, int elem_mbits, float elem_max_norm,
                    int flush_fp32_subnorms, int rmode, void* stream) {
    if (pre < 0 || axis_len < 0 || post < 0) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_mx: negative size");
    const int64_t total = pre * axis_len * post;
    if (total == 0) return MSQ_OK;
    if (!in || !out || !max_values) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_mx: null buffer");
    if (rmode < 0 || rmode > 2) return fail(MSQ_ERR_BAD_ARG, "msq_quantize_mx: bad rounding mode");
    if (post == 1 && axis_len % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(k_mx_maxvals_rows4, dim3(grid_for(total / 4, 256, 2048 * 4)), dim3(256), 0, (hipStream_t)stream, in,
                           out, max_values, total / 4, axis_len / 4, scale_bits, elem_ebits, elem_mbits,
                           elem_max_norm, flush_fp32_subnorms, rmode);
    else
        hipLaunchKernelGGL(k_mx_maxvals, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, in,
                           out, max_values, total, axis_len, post, scale_bits, elem_ebits, elem_mbits,
                           elem_max_norm, flush_fp32_subnorms, rmode);
    return check_launch("msq_quantize_mx");
}

int msq_quantize_mx_by_tile(const float* in, float* out, int64_t pre, int64_t axis_len, int64_t post,
                            int tile_size, int scale_bits, int elem_ebits, int elem_mbits,
                            float elem_max_norm, int flush_fp32_subnorms, int rmode, void* stream) {
    return launch_mx_by_tile<false, false>(in, out, pre, axis_len, post, tile_size, scale_bits, elem_ebits, elem_mbits,
                                           elem_max_norm, flush_fp32_subnorms, rmode, stream);
}

int msq_quantize_mx_by_tile_py(const float* in, float* out, int64_t pre, int64_t axis_len, int64_t post,
                               int tile_size, int scale_bits, int elem_ebits, int elem_mbits,
                               float elem_max_norm, int flush_fp32_subnorms, int rmode, void* stream) {
    return launch_mx_by_tile<true, true>(in, out, pre, axis_len, post, tile_size, scale_bits, elem_ebits, elem_mbits,
                                         elem_max_norm, flush_fp32_subnorms, rmode, stream);
}

int msq_quantize_mx_by_tile_ex(const float* in, float* out, int64_t pre, int64_t axis_len, int64_t post,
                               int tile_size, int scale_bits, int elem_ebits, int elem_mbits,
                               float elem_max_norm, int flush_fp32_subnorms, int rmode, int py_divisor, int py_exponent, void* stream) {
#define MSQ_MXT(E, P) launch_mx_by_tile<E, P>(in, out, pre, axis_len, post, tile_size, scale_bits, elem_ebits, elem_mbits, \
                                             elem_max_norm, flush_fp32_subnorms, rmode, stream)
    if (py_divisor) return py_exponent ? MSQ_MXT(true, true) : MSQ_MXT(true, false);
    return py_exponent ? MSQ_MXT(false, true) : MSQ_MXT(false, false);
#undef MSQ_MXT
}

int msq_reduce_sum_inner(const float* in, float* out, int64_t outer, int64_t inner, void* stream) {
    if (outer < 0 || inner < 0) return fail(MSQ_ERR_BAD_ARG, "msq_reduce_sum_inner: negative size");
    if (outer == 0) return MSQ_OK;
    if (!in || !out) return fail(MSQ_ERR_BAD_ARG, "msq_reduce_sum_inner: null buffer");
    if (!launch_reduce_short<false>(in, out, outer, inner, (hipStream_t)stream))
        hipLaunchKernelGGL(k_reduce_inner<false>, dim3(grid_for(outer, 4)), dim3(256), 0, (hipStream_t)stream, in,
                           out, outer, inner);
    return check_launch("msq_reduce_sum_inner");
}

int msq_reduce_max_inner(const float* in, float* out, int64_t outer, int64_t inner, void* stream) {
    if (outer < 0 || inner < 0) return fail(MSQ_ERR_BAD_ARG, "msq_reduce_max_inner: negative size");
    if (outer == 0) return MSQ_OK;
    if (!in || !out) return fail(MSQ_ERR_BAD_ARG, "msq_reduce_max_inner: null buffer");
    if (!launch_reduce_short<true>(in, out, outer, inner, (hipStream_t)stream))
        hipLaunchKernelGGL(k_reduce_inner<true>, dim3(grid_for(outer, 4)), dim3(256), 0, (hipStream_t)stream, in,
                           out, outer, inner);
    return check_launch("msq_reduce_max_inner");
}

// internal (msq_outlier_pack / msq_act_quant_bf16 with variant 1)
int msq_mxops_stats_x_(const void* in, int x_bf16, float* vmean, float* vstd, int64_t pre, int64_t axis_len, int64_t post, int block,
                       int* status, void* stream);
int msq_mxops_stats_(const float* in, float* vmean, float* vstd, int64_t pre, int64_t axis_len, int64_t post, int block,
                     int* status, void* stream) {
    return msq_mxops_stats_x_(in, 0, vmean, vstd, pre, axis_len, post, block, status, stream);
}
// x_bf16: the rows hold bfloat16 (contiguous blocks only: post == 1, block 32 / 64 / 128 dividing the axis)
int msq_mxops_stats_x_(const void* inv, int x_bf16, float* vmean, float* vstd, int64_t pre, int64_t axis_len, int64_t post, int block,
                       int* status, void* stream) {
    const float* in = (const float*)inv;
    const int64_t nblk = (axis_len + block - 1) / block;
    hipStream_t st = (hipStream_t)stream;
    if (post == 1 && axis_len % block == 0 && nblk <= 1024 && nblk >= 2 && pre < (1ll << 31) &&
        (block == 32 || block == 64 || block == 128)) {
        const dim3 grid((unsigned)pre), blk(256);
        if (x_bf16) {
            const uint16_t* ih = (const uint16_t*)inv;
            if (block == 32) hipLaunchKernelGGL((k_mxops_stats_rows<32, uint16_t>), grid, blk, 0, st, ih, vmean, vstd, axis_len, nblk, status);
            else if (block == 64) hipLaunchKernelGGL((k_mxops_stats_rows<64, uint16_t>), grid, blk, 0, st, ih, vmean, vstd, axis_len, nblk, status);
            else hipLaunchKernelGGL((k_mxops_stats_rows<128, uint16_t>), grid, blk, 0, st, ih, vmean, vstd, axis_len, nblk, status);
            return check_launch("mx_ops statistics");
        }
        if (block == 32) hipLaunchKernelGGL(k_mxops_stats_rows<32>, grid, blk, 0, st, in, vmean, vstd, axis_len, nblk, status);
        else if (block == 64) hipLaunchKernelGGL(k_mxops_stats_rows<64>, grid, blk, 0, st, in, vmean, vstd, axis_len, nblk, status);
        else hipLaunchKernelGGL(k_mxops_stats_rows<128>, grid, blk, 0, st, in, vmean, vstd, axis_len, nblk, status);
    } else {
        if (x_bf16) return fail(MSQ_ERR_UNSUPPORTED, "mx_ops statistics: bfloat16 rows need contiguous blocks of 32 / 64 / 128 and 2 ... 1024 blocks per row");
        hipLaunchKernelGGL(k_mxops_stats, dim3(grid_for(pre * block * post, 256)), dim3(256), 0, st,
                           in, vmean, vstd, pre, axis_len, post, block, nblk, status);
    }
    return check_launch("mx_ops statistics");
}

int64_t msq_outlier_workspace_bytes(int64_t pre, int64_t axis_len, int64_t post, int block, int variant) {
    if (block <= 0) block = (int)axis_len;
    // utils/quant.py variant: the list of waves the packed half-precision kernels hand back to the op-by-op kernel (fp16 / bf16 tensors
    // computed in their dtype; without it that call runs the op-by-op kernel only -- same results, a third of the speed)
    if (variant != MSQ_VARIANT_MXOPS) return msq_outlier_lowp_ws_bytes_(pre, axis_len, post, block);
    return 2 * (int64_t)sizeof(float) * pre * block * post;
}

int msq_outlier_fakequant(const void* in, void* out, uint8_t* mask, float* e_in, float* e_out,
                          int8_t* num_outliers, int* status_flag, void* workspace, int64_t workspace_bytes,
                          int dtype, int64_t pre, int64_t axis_len, int64_t post, int block, int inlier_fmt,
                          int outlier_fmt, int inlier_scale_bits, int outlier_scale_bits, float std_dev,
                          int rmode, int flush_fp32_subnorms, int variant, void* stream) {
    if (pre < 0 || axis_len < 0 || post < 0) return fail(MSQ_ERR_BAD_ARG, "msq_outlier_fakequant: negative size");
    if (pre * axis_len * post == 0) return MSQ_OK;
    if (!in || !out) return fail(MSQ_ERR_BAD_ARG, "msq_outlier_fakequant: null buffer");
    if (inlier_scale_bits <= 0 || outlier_scale_bits <= 0 || inlier_scale_bits > 8 || outlier_scale_bits > 8)
        return fail(MSQ_ERR_BAD_ARG, "msq_outlier_fakequant: scale bits must be in [1,8]");   // utils/quant.py:168
    if (rmode < 0 || rmode > 2) return fail(MSQ_ERR_BAD_ARG, "msq_outlier_fakequant: bad rounding mode");
    if (variant != MSQ_VARIANT_QUANT && variant != MSQ_VARIANT_MXOPS)
        return fail(MSQ_ERR_BAD_ARG, "msq_outlier_fakequant: bad variant");
    msq_host::FmtInfo fi, fo;
    if (!msq_host::format_info(inlier_fmt, &fi) || !msq_host::format_info(outlier_fmt, &fo))
        return fail(MSQ_ERR_BAD_ARG, "msq_outlier_fakequant: unknown element format");
    if (block <= 0) block = (int)axis_len;
    OutlierArgs A;
    A.fi = Fmt{fi.kind, fi.ebits, fi.mbits, fi.emax, fi.max_norm};
    A.fo = Fmt{fo.kind, fo.ebits, fo.mbits, fo.emax, fo.max_norm};
    A.in_sb = inlier_scale_bits; A.out_sb = outlier_scale_bits;
    A.k = std_dev; A.rmode = rmode; A.flush = flush_fp32_subnorms; A.variant = variant;
    A.pre = pre; A.axis_len = axis_len; A.post = post; A.nblk = (axis_len + block - 1) / block;
    A.mask = mask; A.e_in = e_in; A.e_out = e_out; A.n_out = num_outliers; A.status = status_flag;
    A.vmean = nullptr; A.vstd = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (variant == MSQ_VARIANT_MXOPS) {
        if (dtype != 0) return fail(MSQ_ERR_UNSUPPORTED, "msq_outlier_fakequant: variant mx_ops is f32 only");
        const int64_t need = msq_outlier_workspace_bytes(pre, axis_len, post, block, variant);
        if (!workspace || workspace_bytes < need)
            return fail(MSQ_ERR_BAD_ARG, "msq_outlier_fakequant: workspace too small (msq_outlier_workspace_bytes)");
        float* vmean = (float*)workspace;
        float* vstd = vmean + pre * block * post;
        int rc = msq_mxops_stats_((const float*)in, vmean, vstd, pre, axis_len, post, block, status_flag, stream);
        if (rc) return rc;
        A.vmean = vmean; A.vstd = vstd;
    }
    int rc;
    if (dtype == MSQ_DTYPE_F16_NATIVE || dtype == MSQ_DTYPE_BF16_NATIVE) {
        // compute in the tensor dtype, every op rounded as ATen's CPU half kernels do (llm/llama.py:238)
        if (variant != MSQ_VARIANT_QUANT || fi.kind != 0 || fo.kind != 0 || num_outliers)
            return fail(MSQ_ERR_UNSUPPORTED, "msq_outlier_fakequant: native half-precision compute covers utils/quant.py:147-266 with float / int element formats");
        if (!msq_launch_outlier_lowp_(in, out, &A, block, dtype & 3, workspace, workspace_bytes, (void*)st))
            return fail(MSQ_ERR_UNSUPPORTED, "msq_outlier_fakequant: block size must be 8, 16, 32, 64 or 128");
        return check_launch("msq_outlier_fakequant(native half)");
    }
    if (dtype == 0 || dtype == 1 || dtype == 2) rc = launch_outlier(in, out, A, block, st, dtype, workspace, workspace_bytes);
    else return fail(MSQ_ERR_BAD_ARG, "msq_outlier_fakequant: dtype must be 0 (f32), 1 (f16), 2 (bf16) or MSQ_DTYPE_F16_NATIVE / _BF16_NATIVE");
    if (rc) return rc;
    return check_launch("msq_outlier_fakequant");
}

}  // extern "C"
