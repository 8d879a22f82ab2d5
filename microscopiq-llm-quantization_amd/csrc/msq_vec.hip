// msq_vec.hip -- the bfloat-rounded vector ops that surround the MX Linear in the reference's emulation library
// (number_system/mx/vector_ops.py: every op = the torch op followed by quantize_elemwise_op):
//   LayerNorm  layernorm.py:18-42 -> norm_utils.py:27-113 _norm_forward over the last axis (14 rounded ops per element)
//   gelu       activations.py:460-512 (sigmoid form with bf16 coefficients, 10 rounded ops; first-order variant)
//   simd_add   simd_ops.py:85-106
// The reference issues one eager torch kernel per op (and one quantise pass after each): here each function is ONE
// launch, read once / write once, with the rounding Q() applied in registers after every arithmetic step exactly
// where the reference applies it, so the results are those of the op-by-op emulation.
// Q() = the native element codec (quant_bits: cpp/quantize.cuh:88-149 semantics, saturate_normals = false) for
// (bits, exp_bits, max_norm): bfloat16 = (9, 8, bf16 max), bfloat12 = (5, 8, ...), fp16-style = (bits, 5, ...).
// LayerNorm: one wavefront per row; the row sums follow ATen's order for a contiguous inner dimension (32 interleaved
// partial sums = 4 ILP x 8 vector lanes with its cascade levels, combined in ATen's order), so that the mean and
// variance carry the same fp32 rounding as torch.sum on the CPU.  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_mx_pack_core.h"
#include "msq_host.h"

using namespace msq;

namespace {

struct VQ { int bits, ebits, rmode, dn; float max_norm; };
// bfloatX (8 exponent bits, subnormals kept): the target shares fp32's exponent range, so rounding to `bits - 2` explicit
// mantissa bits is integer arithmetic on the fp32 pattern, subnormals and the carry into the exponent (up to Inf: the
// reference does not saturate here) included: nearest = half away from zero = add half a quantum to the magnitude and
// truncate; even = add (half - 1 + kept lsb); floor = truncate.  Bit-identical to quant_bits for every finite input
// (tests/test_gpu_f4_vector_ops_golden.py::test_vector_rounding_fast_path_equals_codec); Inf / NaN pass through.
MSQ_D float Qbf(float a, int drop, int rmode) {
    const uint32_t u = f2u(a);
    if ((u & 0x7F800000u) == 0x7F800000u) return a;
    if (rmode == 1) {
        // Truncation is the one mode in which the PYTHON path's private exponent shows (elemwise_ops.py:139-140: floor(torch.log2(|a|)) in
        // float32): for the K largest floats below a power of two 2^u it is u, one binade high (msq_device.h ilog2f_torch), the value is
        // scaled by 2^-u instead of 2^-(u-1) and loses one more mantissa bit -- (2^(m-1) - 1) / 2^(m-1) 2^u instead of the float below 2^u
        // on the grid.  Nearest / even round such a value to 2^u either way.  The reference's CPU outputs hold this
        // (tests/golden/vec_rmsnorm_modes.npz: `b - tiny` under floor); the native codec (quant_bits, msq_vec_round's force_codec) does not.
        const uint32_t E = (u >> 23) & 0xFFu;
        if (E != 0u && E < 254u && ilog2f_torch(u2f(u & 0x7FFFFFFFu)) + 127 > (int)E) drop += 1;
    }
    const uint32_t mag = u & 0x7FFFFFFFu, half = 1u << (drop - 1);
    const uint32_t add = (rmode == 0) ? half : ((rmode == 2) ? (half - 1u + ((mag >> drop) & 1u)) : 0u);
    const uint32_t r = (mag + add) & ~((1u << drop) - 1u);
    return r ? u2f((u & 0x80000000u) | r) : 0.0f;                  // zero is +0, as the codec returns it
}
MSQ_D float Q(float a, const VQ& q) {
    if (q.bits <= 0) return a;
    if (q.ebits == 8 && q.dn && q.bits < 24) return Qbf(a, 25 - q.bits, q.rmode);
    return quant_bits(a, q.bits, q.ebits, q.max_norm, q.rmode, false, q.dn != 0);
}

// FAST = 1: bfloat16, round to nearest (the run_mx_fp6.sh spec) as two integer instructions on the fp32 pattern: half a quantum
// added to the whole word (the carry cannot reach the sign below the NaN range; up to Inf as in Qbf), low half cleared.  That is
// Qbf(a, 16, 0) except for (i) NaNs whose payload sits in the low half / overflows, and (ii) negative values that round to zero
// (-0 here, +0 there).  (i) is closed by Qin on every value that enters an op from memory (a NaN becomes the canonical quiet
// NaN, which arithmetic propagates and the two instructions keep); (ii) by Qout on every value that leaves: in between a zero
// of either sign gives the same non-zero results (no op of the three functions divides by, or takes the root of, such a value:
// the variance is a sum of squares, phi >= 1), so the outputs are those of Qbf at every step.
// tests/test_gpu_f4_vector_ops_golden.py::test_vector_ops_fast_rounding_equals_generic runs both variants on every bfloat16 pattern.
MSQ_D float Qmid16(float a) { return u2f((f2u(a) + 0x8000u) & 0xFFFF0000u); }
MSQ_D float Qin16(float a) {
    const uint32_t u = f2u(a);
    uint32_t r = (u + 0x8000u) & 0xFFFF0000u;
    r = ((u << 1) > 0xFF000000u) ? 0x7FC00000u : r;
    return u2f(r == 0x80000000u ? 0u : r);
}
MSQ_D float Qout16(float a) {
    const uint32_t r = (f2u(a) + 0x8000u) & 0xFFFF0000u;
    return u2f(r == 0x80000000u ? 0u : r);
}
template <int FAST> MSQ_D float QT(float a, const VQ& q) { return FAST ? Qmid16(a) : Q(a, q); }
template <int FAST> MSQ_D float QI(float a, const VQ& q) { return FAST ? Qin16(a) : Q(a, q); }
template <int FAST> MSQ_D float QO(float a, const VQ& q) { return FAST ? Qout16(a) : Q(a, q); }
// a value that enters from a BFLOAT16 tensor under the bfloat16 / nearest rounding is already on the grid: Qin16 would only make a NaN
// canonical (its payload sits in the upper half: the two-instruction rounding keeps it a NaN) and a -0 a +0 (which no op below tells
// apart: products and sums with a zero of either sign end in Qout16's +0).  XS: 0 float32, 1 bfloat16, 2 float16 source.
template <int FAST, int XS> MSQ_D float QIx(float a, const VQ& q) { return (FAST && XS == 1) ? a : QI<FAST>(a, q); }

MSQ_D int ceil_log2_i64(int64_t x) { int l = 0; while (((int64_t)1 << l) < x) ++l; return l; }

// ATen's inner-dimension sum of row[0..n) (oracle sum_inner_v8): lane t < 32 owns the partial sum of elements
// i * 32 + t (t = 8 k + l), accumulated with multi_row_sum's cascade; the caller's functor gives element values.
template <typename F>
MSQ_D float row_sum_inner8(F elem, int64_t n, int lane) {
    const int64_t vec_size = n / 8, size_ilp = vec_size / 4;
    float part = 0.f;
    if (lane < 32) {
        int lp = ceil_log2_i64(size_ilp > 0 ? size_ilp : 1) / 4; if (lp < 4) lp = 4;
        const int64_t step = (int64_t)1 << lp, lmask = step - 1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int64_t i = 0;
        while (i + step <= size_ilp) {
            for (int64_t j = 0; j < step; j += 8, i += 8) {          // step is a multiple of 16: fetch 8, then add in order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = elem((i + u) * 32 + lane);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[0] += v[u];
            }
            for (int j = 1; j < 4; ++j) {
                acc[j] += acc[j - 1]; acc[j - 1] = 0.f;
                if ((i & (lmask << (j * lp))) != 0) break;
            }
        }
        for (; i < size_ilp; ++i) acc[0] += elem(i * 32 + lane);
        for (int j = 1; j < 4; ++j) acc[0] += acc[j];
        part = acc[0];
        // vectors beyond the last full group of four go to ILP slot 0 (lanes 0..7)
        if (lane < 8)
            for (int64_t v = size_ilp * 4; v < vec_size; ++v) part += elem(v * 8 + lane);
    }
    // part[0][l] += part[k][l], k = 1..3, in that order
    float p0 = part;
    for (int k = 1; k < 4; ++k) { const float o = __shfl(part, (lane & 7) + 8 * k, 64); p0 += o; }
    // fin = scalar tail, then + part[0][l] for l = 0..7
    float fin = 0.f;
    for (int64_t k = vec_size * 8; k < n; ++k) fin += elem(k);
    for (int l = 0; l < 8; ++l) fin += __shfl(p0, l, 64);
    return fin;                                            // identical in every lane
}

__global__ void __launch_bounds__(64)
k_vec_layernorm(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
                int64_t rows, int64_t H, float eps, VQ q) {
    extern __shared__ float xs[];
    const int64_t r = blockIdx.x;
    const int lane = threadIdx.x;
    if (r >= rows) return;
    const float* xr = x + r * H;
    for (int64_t i = lane; i < H; i += 64) xs[i] = Q(xr[i], q);                       // layernorm.py:24
    __syncthreads();
    float mean = Q(row_sum_inner8([&](int64_t i) { return xs[i]; }, H, lane), q);     // vec_reduce_sum
    mean = Q(mean / (float)H, q);                                                     // vec_div(s, denom)
    __syncthreads();
    for (int64_t i = lane; i < H; i += 64) xs[i] = Q(xs[i] - mean, q);                // x_shift
    __syncthreads();
    float var = Q(row_sum_inner8([&](int64_t i) { return Q(xs[i] * xs[i], q); }, H, lane), q);
    var = Q(var / (float)H, q);
    const float vare = Q(var + eps, q);                                               // norm_utils.py:92
    const float sd = Q(__builtin_sqrtf(vare), q);
    const float inv = Q(1.0f / sd, q);
    float* orow = out + r * H;
    for (int64_t i = lane; i < H; i += 64) {
        const float xn = Q(xs[i] * inv, q);
        const float sc = Q(Q(w[i], q) * xn, q);
        orow[i] = Q(sc + Q(b[i], q), q);
    }
}

// Rows with H = 512 G, G <= 16, held in registers in the layout of the sum itself: thread (k = tid / 32, t = tid % 32) owns the
// sixteen elements (g * 16 + j) * 32 + t of groups g = k (and k + 8), i.e. exactly the addends of its level-0 partial sum, so the
// row never goes through the LDS (only the G * 32 partial sums do).  Global accesses are 4 B per lane, two full 128 B lines per
// wave instruction.  The rounded weight and bias of the thread's columns are kept for all rows of the block (FAST: as one
// register per column, bfloat16 pair); the next row is fetched while the current one is reduced.
template <int FAST, int GP>
__global__ void __launch_bounds__(256)
k_vec_layernorm_reg(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
                    int64_t rows, int H, float eps, VQ q) {
    __shared__ float part[512];
    __shared__ float bc[1];
    const int tid = threadIdx.x, k = tid >> 5, t = tid & 31, G = H / 512;
    float wq[GP][16], bq[GP][16], cur[GP][16], nxt[GP][16];
    uint32_t wb[GP][16];
#pragma unroll
    for (int gi = 0; gi < GP; ++gi)
        if (k + 8 * gi < G) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = ((k + 8 * gi) * 16 + j) * 32 + t;
                const float a = QI<FAST>(w[c], q), d = QI<FAST>(b[c], q);
                if (FAST) wb[gi][j] = f2u(a) | (f2u(d) >> 16);
                else { wq[gi][j] = a; bq[gi][j] = d; }
            }
        }
    auto fetch = [&](int64_t r) {
        const float* xr = x + r * H;
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
            if (k + 8 * gi < G) {
#pragma unroll
                for (int j = 0; j < 16; ++j) nxt[gi][j] = xr[((k + 8 * gi) * 16 + j) * 32 + t];
            }
    };
    // level 1 and the 32 interleave slots, combined as in row_sum_inner8 (ATen's order)
    auto combine = [&]() -> float {
        __syncthreads();
        if (tid < 64) {
            float p = 0.f;
            if (tid < 32) { for (int g = 0; g < G; ++g) p += part[g * 32 + tid]; p = 0.f + p; }
            float p0 = p;
            for (int kk = 1; kk < 4; ++kk) { const float o = __shfl(p, (tid & 7) + 8 * kk, 64); p0 += o; }
            float fin = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) fin += u2f(__builtin_amdgcn_readlane(f2u(p0), l));
            if (tid == 0) bc[0] = fin;
        }
        __syncthreads();
        return bc[0];
    };
    if ((int64_t)blockIdx.x < rows) fetch(blockIdx.x);
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
#pragma unroll
            for (int j = 0; j < 16; ++j) cur[gi][j] = QI<FAST>(nxt[gi][j], q);
        if (r + gridDim.x < rows) fetch(r + gridDim.x);
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
            if (k + 8 * gi < G) {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) sm += cur[gi][j];
                part[(k + 8 * gi) * 32 + t] = sm;
            }
        float mean = QT<FAST>(combine(), q);
        mean = QT<FAST>(mean / (float)H, q);
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
            if (k + 8 * gi < G) {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    cur[gi][j] = QT<FAST>(cur[gi][j] - mean, q);
                    sm += QT<FAST>(cur[gi][j] * cur[gi][j], q);
                }
                part[(k + 8 * gi) * 32 + t] = sm;
            }
        float var = QT<FAST>(combine(), q);
        var = QT<FAST>(var / (float)H, q);
        const float vare = QT<FAST>(var + eps, q);
        const float sd = QT<FAST>(__builtin_sqrtf(vare), q);
        const float inv = QT<FAST>(1.0f / sd, q);
        float* orow = out + r * H;
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
            if (k + 8 * gi < G) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float ww = FAST ? u2f(wb[gi][j] & 0xFFFF0000u) : wq[gi][j];
                    const float bb = FAST ? u2f(wb[gi][j] << 16) : bq[gi][j];
                    orow[((k + 8 * gi) * 16 + j) * 32 + t] = QO<FAST>(QT<FAST>(ww * QT<FAST>(cur[gi][j] * inv, q), q) + bb, q);
                }
            }
    }
}

template <int FAST>
MSQ_D float gelu_one(float x, int first_order, const VQ& q) {
    const float qi = QI<FAST>(x, q);
    float s;
    if (first_order) s = QT<FAST>(1.703125f * qi, q);
    else {
        s = QT<FAST>(qi * qi, q); s = QT<FAST>(s * qi, q); s = QT<FAST>(0.044677734f * s, q);
        s = QT<FAST>(s + qi, q); s = QT<FAST>(1.59375f * s, q);
    }
    float phi = QT<FAST>(expf(-s), q);                             // torch.exp (vec_use_exp2 off)
    phi = QT<FAST>(phi + 1.0f, q);
    // 1 / phi: phi >= 1 carries 8 significant bits, so the exact quotient is at least 2^-17 (relative) away from every rounding
    // boundary of the 8-bit result (1 / m = y needs m y = 2^e with both odd): v_rcp_f32's one ulp cannot move the rounded value.
    // Quotients in fp32's subnormal range (phi > 2^126) and non-finite phi take the IEEE division.
    if (FAST) phi = QT<FAST>((phi < 8.0e37f) ? __builtin_amdgcn_rcpf(phi) : 1.0f / phi, q);
    else phi = QT<FAST>(1.0f / phi, q);
    return QO<FAST>(qi * phi, q);
}

// 16-byte accesses (n4 float4 vectors; the tail goes through k_vec_gelu)
template <int FAST>
__global__ void __launch_bounds__(256)
k_vec_gelu4(const float* __restrict__ x, float* __restrict__ out, int64_t n4, int first_order, VQ q) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = gelu_one<FAST>(v.x, first_order, q); v.y = gelu_one<FAST>(v.y, first_order, q);
        v.z = gelu_one<FAST>(v.z, first_order, q); v.w = gelu_one<FAST>(v.w, first_order, q);
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

template <int FAST>
__global__ void __launch_bounds__(256)
k_vec_add4(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n4, int b_is_scalar, float bs, VQ q) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 va = reinterpret_cast<const float4*>(a)[i];
        float4 vb = make_float4(bs, bs, bs, bs);
        if (!b_is_scalar) { vb = reinterpret_cast<const float4*>(b)[i]; vb.x = QI<FAST>(vb.x, q); vb.y = QI<FAST>(vb.y, q); vb.z = QI<FAST>(vb.z, q); vb.w = QI<FAST>(vb.w, q); }
        float4 o;
        o.x = QO<FAST>(QI<FAST>(va.x, q) + vb.x, q); o.y = QO<FAST>(QI<FAST>(va.y, q) + vb.y, q);
        o.z = QO<FAST>(QI<FAST>(va.z, q) + vb.z, q); o.w = QO<FAST>(QI<FAST>(va.w, q) + vb.w, q);
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

__global__ void __launch_bounds__(256)
k_vec_gelu(const float* __restrict__ x, float* __restrict__ out, int64_t n, int first_order, VQ q) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float qi = Q(x[i], q);
        float s;
        if (first_order) s = Q(1.703125f * qi, q);
        else {
            s = Q(qi * qi, q); s = Q(s * qi, q); s = Q(0.044677734f * s, q);
            s = Q(s + qi, q); s = Q(1.59375f * s, q);
        }
        float phi = Q(expf(-s), q);                                // torch.exp (vec_use_exp2 off)
        phi = Q(phi + 1.0f, q);
        phi = Q(1.0f / phi, q);
        out[i] = Q(qi * phi, q);
    }
}

__global__ void __launch_bounds__(256)
k_vec_add(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n, int b_is_scalar, float bs, VQ q) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = Q(Q(a[i], q) + (b_is_scalar ? bs : Q(b[i], q)), q);                  // simd_ops.py:95-106
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the activation PRODUCERS in front of the MX Linear, each able to hand its result on as the MX-FP8 operand of the scaled-MFMA
// GEMM (e4m3 codes row-major + one scale byte per 32, the layout of msq_mx_pack_a8) instead of -- or besides -- the float32 tensor:
//   RMSNorm      layernorm.py:177 -> RMSNormFunction.forward :98-128 (9 rounded ops per element, one row sum in ATen's order)
//   silu x up    activations.py:76 -> :420-434 (5 rounded ops), simd_mul simd_ops.py:445 -> :154-187 (Q(Q(a) Q(b)))
// The reference runs ~12 / ~9 eager torch kernels for these and then re-reads the result in the Linear to quantise it (quantize_mx_op, mx_ops.py:460-490);
// unfused here it is two launches (producer: 4 B in + 4 B out per element; packer: 4 B in + 1 B out), fused one (4 B in, 1 B out).
// The pack is msq_mx_pack_core.h's: the bytes are those of msq_mx_pack_a8 on the producer's output.
// ---------------------------------------------------------------------------------------------------------------------------
struct PackOut { uint8_t* codes; uint8_t* scales; int* status; int flush; };
// element i of a float32 (XS 0), bfloat16 (1) or float16 (2) tensor as float (every 16-bit value is a float32 value: the results of casting first)
template <int XS> MSQ_D float ldx(const void* p, int64_t i) {
    if (XS == 1) return u2f((uint32_t)reinterpret_cast<const uint16_t*>(p)[i] << 16);
    if (XS == 2) return (float)reinterpret_cast<const _Float16*>(p)[i];
    return reinterpret_cast<const float*>(p)[i];
}
// eight consecutive elements starting at element index i (16-byte aligned for 16-bit sources, 32-byte pieces for float32)
template <int XS> MSQ_D void ldx8(const void* p, int64_t i, float (&a)[8]) {
    if (XS == 0) {
        const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
        const float4 v0 = q[0], v1 = q[1];
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    } else if (XS == 1) {
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p) + i);
        a[0] = u2f(v.x << 16); a[1] = u2f(v.x & 0xFFFF0000u); a[2] = u2f(v.y << 16); a[3] = u2f(v.y & 0xFFFF0000u);
        a[4] = u2f(v.z << 16); a[5] = u2f(v.z & 0xFFFF0000u); a[6] = u2f(v.w << 16); a[7] = u2f(v.w & 0xFFFF0000u);
    } else {
        union { uint4 u; _Float16 h[8]; } r;
        r.u = *reinterpret_cast<const uint4*>(reinterpret_cast<const _Float16*>(p) + i);
#pragma unroll
        for (int b = 0; b < 8; ++b) a[b] = (float)r.h[b];
    }
}

// one chunk of eight consecutive values of row-major [rows, H] (chunk index i over the whole tensor); the quad's lanes call together
MSQ_D void pack_chunk(const float (&a)[8], int64_t i, const PackOut& P, int& status) {
    uint32_t cw[2];
    int sb;
    mx_pack8_e4m3_quad(a, cw, sb, P.flush, status);
    reinterpret_cast<uint2*>(P.codes)[i] = make_uint2(cw[0], cw[1]);
    if ((i & 3) == 0) P.scales[i >> 2] = (uint8_t)sb;
}

// any H: one wavefront per row, the row in LDS (as k_vec_layernorm)
template <int XS>
__global__ void __launch_bounds__(64)
k_vec_rmsnorm(const void* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
              int64_t rows, int64_t H, float eps, VQ q, PackOut P) {
    extern __shared__ float xs[];
    const int64_t r = blockIdx.x;
    const int lane = threadIdx.x;
    if (r >= rows) return;
    for (int64_t i = lane; i < H; i += 64) xs[i] = Q(ldx<XS>(x, r * H + i), q);       // layernorm.py:104
    __syncthreads();
    float ms = Q(row_sum_inner8([&](int64_t i) { return Q(xs[i] * xs[i], q); }, H, lane), q);     // :107, vec_reduce_sum
    ms = Q(ms / (float)H, q);                                                         // vec_div(s, denom)
    const float mse = Q(ms + eps, q);                                                 // :114
    const float rms = Q(__builtin_sqrtf(mse), q);                                     // :116
    const float inv = Q(1.0f / rms, q);                                               // :119
    __syncthreads();
    float* orow = out ? out + r * H : nullptr;
    for (int64_t i = lane; i < H; i += 64) {
        const float xn = Q(xs[i] * inv, q);                                           // :120
        const float sc = Q(Q(w[i], q) * xn, q);                                       // :122, :126
        const float y = Q(sc + (b ? Q(b[i], q) : 0.f), q);                            // :124, :128
        if (orow) orow[i] = y;
        xs[i] = y;
    }
    if (P.codes) {                                                                    // H % 32 == 0 (checked on the host)
        __syncthreads();
        int status = 0;
        for (int64_t c = lane; c < H / 8; c += 64) {
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = xs[c * 8 + u];
            pack_chunk(a, r * (H / 8) + c, P, status);
        }
        if (status && P.status) atomicOr(P.status, status);
    }
}

// H = 512 G, G <= 16: the register layout of k_vec_layernorm_reg (thread (k, t) owns the addends of its level-0 partial sums).  PACK: the
// finished row goes through an LDS row buffer into the packer's layout (eight consecutive values per lane, a quad per block).
template <int FAST, int GP, int PACK, int XS>
__global__ void __launch_bounds__(256)
k_vec_rmsnorm_reg(const void* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
                  int64_t rows, int H, float eps, VQ q, PackOut P) {
    __shared__ float part[512];
    __shared__ float bc[1];
    __shared__ __attribute__((aligned(16))) float ys[PACK ? GP * 4096 : 4];
    const int tid = threadIdx.x, k = tid >> 5, t = tid & 31, G = H / 512;
    float wq[GP][16], bq[GP][16], cur[GP][16], nxt[GP][16];
    uint32_t wb[GP][16];
#pragma unroll
    for (int gi = 0; gi < GP; ++gi)
        if (k + 8 * gi < G) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = ((k + 8 * gi) * 16 + j) * 32 + t;
                const float a = QI<FAST>(w[c], q), d = b ? QI<FAST>(b[c], q) : 0.f;
                if (FAST) wb[gi][j] = f2u(a) | (f2u(d) >> 16);
                else { wq[gi][j] = a; bq[gi][j] = d; }
            }
        }
    auto fetch = [&](int64_t r) {
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
            if (k + 8 * gi < G) {
#pragma unroll
                for (int j = 0; j < 16; ++j) nxt[gi][j] = ldx<XS>(x, r * H + ((k + 8 * gi) * 16 + j) * 32 + t);
            }
    };
    auto combine = [&]() -> float {                                // level 1 and the 32 interleave slots, ATen's order (row_sum_inner8)
        __syncthreads();
        if (tid < 64) {
            float p = 0.f;
            if (tid < 32) { for (int g = 0; g < G; ++g) p += part[g * 32 + tid]; p = 0.f + p; }
            float p0 = p;
            for (int kk = 1; kk < 4; ++kk) { const float o = __shfl(p, (tid & 7) + 8 * kk, 64); p0 += o; }
            float fin = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) fin += u2f(__builtin_amdgcn_readlane(f2u(p0), l));
            if (tid == 0) bc[0] = fin;
        }
        __syncthreads();
        return bc[0];
    };
    int status = 0;
    if ((int64_t)blockIdx.x < rows) fetch(blockIdx.x);
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
#pragma unroll
            for (int j = 0; j < 16; ++j) cur[gi][j] = QIx<FAST, XS>(nxt[gi][j], q);
        if (r + gridDim.x < rows) fetch(r + gridDim.x);
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
            if (k + 8 * gi < G) {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) sm += QT<FAST>(cur[gi][j] * cur[gi][j], q);
                part[(k + 8 * gi) * 32 + t] = sm;
            }
        float ms = QT<FAST>(combine(), q);
        ms = QT<FAST>(ms / (float)H, q);
        const float mse = QT<FAST>(ms + eps, q);
        const float rms = QT<FAST>(__builtin_sqrtf(mse), q);
        const float inv = QT<FAST>(1.0f / rms, q);
        float* orow = out ? out + r * H : nullptr;
#pragma unroll
        for (int gi = 0; gi < GP; ++gi)
            if (k + 8 * gi < G) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float ww = FAST ? u2f(wb[gi][j] & 0xFFFF0000u) : wq[gi][j];
                    const float bb = FAST ? u2f(wb[gi][j] << 16) : bq[gi][j];
                    const float y = QO<FAST>(QT<FAST>(ww * QT<FAST>(cur[gi][j] * inv, q), q) + bb, q);
                    const int c = ((k + 8 * gi) * 16 + j) * 32 + t;
                    if (orow) orow[c] = y;
                    if (PACK) ys[c] = y;
                }
            }
        if (PACK) {
            __syncthreads();
            for (int c = tid; c < H / 8; c += 256) {
                const float4 v0 = *reinterpret_cast<const float4*>(ys + c * 8), v1 = *reinterpret_cast<const float4*>(ys + c * 8 + 4);
                const float a[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                pack_chunk(a, r * (H / 8) + c, P, status);
            }
            // (the next row's writes to ys come after its two combine() barriers)
        }
    }
    if (PACK && status && P.status) atomicOr(P.status, status);
}

// silu in two stages, so that the choice between v_rcp_f32 and the IEEE division is ONE wave-uniform branch per eight values instead of a
// divergent one per value (the division's expansion + exec juggling was a quarter of the kernel's instructions):
//   silu_phi: qi = Q(x), phi = Q(Q(exp(-qi)) + 1)            (activations.py:427-429; vec_exp with vec_use_exp2 off)
//   silu_fin: Q(qi Q(1 / phi))                               (:430-432)
template <int FAST, int XS>
MSQ_D float silu_phi(float x, const VQ& q, float& qi) {
    qi = QIx<FAST, XS>(x, q);
    const float phi = QT<FAST>(expf(-qi), q);
    return QT<FAST>(phi + 1.0f, q);
}
// RCP: phi >= 1 carries 8 significant bits, so v_rcp_f32's one ulp cannot move the rounded value (as gelu_one); only for phi < 8e37
// (beyond it the quotient is a float32 subnormal: the IEEE division)
template <int FAST, bool RCP>
MSQ_D float silu_fin(float qi, float phi, const VQ& q) {
    const float s = QT<FAST>(RCP ? __builtin_amdgcn_rcpf(phi) : 1.0f / phi, q);
    return QO<FAST>(qi * s, q);
}

// out = silu(g) (u == nullptr), Q(Q(g) Q(u)) (MODE 1: simd_mul) or simd_mul(silu(g), u) (MODE 2); eight consecutive values per lane, rows of
// I values at row strides ldg / ldu (gate and up may be the two halves of one [M, 2 I] tensor); out (row stride I) and / or the MX pack.
template <int FAST, int MODE, int XS>
__global__ void __launch_bounds__(256)
k_vec_act8(const void* __restrict__ g, const void* __restrict__ u, int64_t ldg, int64_t ldu, float* __restrict__ out, int64_t n8, int I8,
           VQ q, PackOut P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;                                          // n8 is a multiple of 4 whenever a pack is asked for
    const int64_t row = i / I8, c8 = i % I8;
    float a[8];
    ldx8<XS>(g, row * ldg + c8 * 8, a);
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE != 0) ldx8<XS>(u, row * ldu + c8 * 8, b8);
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = QO<FAST>(QIx<FAST, XS>(a[e], q) * QIx<FAST, XS>(b8[e], q), q);
    } else {
        float qi[8], phi[8];
        bool small = true;
#pragma unroll
        for (int e = 0; e < 8; ++e) { phi[e] = silu_phi<FAST, XS>(a[e], q, qi[e]); small = small && (phi[e] < 8.0e37f); }
        if (FAST && __builtin_amdgcn_ballot_w64(!small) == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = silu_fin<FAST, true>(qi[e], phi[e], q);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = silu_fin<FAST, false>(qi[e], phi[e], q);
        }
        if (MODE == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = QO<FAST>(a[e] * QIx<FAST, XS>(b8[e], q), q);      // Q(silu) = silu: vec_quantize of a rounded value
        }
    }
    if (out) {
        float4* op = reinterpret_cast<float4*>(out + i * 8);
        op[0] = make_float4(a[0], a[1], a[2], a[3]); op[1] = make_float4(a[4], a[5], a[6], a[7]);
    }
    if (P.codes) {
        int status = 0;
        pack_chunk(a, i, P, status);
        if (status && P.status) atomicOr(P.status, status);
    }
}

// scalar tail / unaligned form of the three element-wise ops (no pack)
template <int MODE>
__global__ void __launch_bounds__(256)
k_vec_act1(const float* __restrict__ g, const float* __restrict__ u, float* __restrict__ out, int64_t n, VQ q) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE == 1) { out[i] = Q(Q(g[i], q) * Q(u[i], q), q); continue; }
        const float qi = Q(g[i], q);
        float phi = Q(expf(-qi), q);
        phi = Q(phi + 1.0f, q);
        phi = Q(1.0f / phi, q);
        const float s = Q(qi * phi, q);
        out[i] = (MODE == 0) ? s : Q(s * Q(u[i], q), q);
    }
}

}  // namespace

extern "C" void msq_set_error_(const char* msg);
static int vfail(int code, const char* msg) { msq_set_error_(msg); return code; }
static int vq_check(int bits, int exp_bits, int rmode) {
    if (bits != 0 && (bits < 2 || bits > 24)) return vfail(MSQ_ERR_BAD_ARG, "vector op: bits must be 0 (no rounding) or in [2, 24]");
    if (exp_bits < 0 || exp_bits > 8) return vfail(MSQ_ERR_BAD_ARG, "vector op: exp_bits must be in [0, 8]");
    if (rmode < 0 || rmode > 2) return vfail(MSQ_ERR_BAD_ARG, "vector op: bad rounding mode");
    return MSQ_OK;
}
// MSQ_VEC_GENERIC=1 in the environment sends bfloat16 / nearest through the run-time-parameter kernels too (the parity tests compare the two)
static msq_host::TuneKey g_vec_generic("MSQ_VEC_GENERIC");
static msq_host::TuneKey g_rms_rpb("MSQ_RMS_RPB");          // rows per block of the register RMSNorm kernel (tuning)
extern "C" int msq_set_tuning_vec_(const char* key, int value) { return g_vec_generic.set_if(key, value) || g_rms_rpb.set_if(key, value); }
static bool vq_is_fast(int bits, int exp_bits, int rmode, int allow_denorm) {
    if (g_vec_generic.value(0) == 1) return false;
    return bits == 9 && exp_bits == 8 && allow_denorm && rmode == 0;
}
static int grid1(int64_t n) { int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 * 4 ? 65535 * 4 : g)); }

namespace {
__global__ void __launch_bounds__(256) k_vec_round(const float* __restrict__ x, float* __restrict__ out, int64_t n, VQ q, int force_codec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = force_codec ? quant_bits(x[i], q.bits, q.ebits, q.max_norm, q.rmode, false, q.dn != 0) : Q(x[i], q);
}
}  // namespace

extern "C" {

/* the rounding Q() of the vector ops on its own (force_codec = 1: always through the generic element codec) */
int msq_vec_round(const float* x, float* out, int64_t n, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm,
                  int force_codec, void* stream) {
    if (n <= 0) return MSQ_OK;
    if (!x || !out) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_round: null buffer");
    if (int rc = vq_check(bits, exp_bits, rmode)) return rc;
    hipLaunchKernelGGL(k_vec_round, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, VQ{bits, exp_bits, rmode, allow_denorm, max_norm}, force_codec);
    return hipGetLastError() == hipSuccess ? MSQ_OK : vfail(MSQ_ERR_LAUNCH, "msq_vec_round: launch failed");
}

int msq_vec_layernorm(const float* x, const float* weight, const float* bias, float* out, int64_t rows, int64_t H, float eps,
                      int bits, int exp_bits, float max_norm, int rmode, int allow_denorm, void* stream) {
    if (rows < 0 || H < 0) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_layernorm: negative size");
    if (rows * H == 0) return MSQ_OK;
    if (!x || !weight || !bias || !out) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_layernorm: null buffer");
    if (int rc = vq_check(bits, exp_bits, rmode)) return rc;
    if (H * 4 > 160 * 1024 - 1024) return vfail(MSQ_ERR_UNSUPPORTED, "msq_vec_layernorm: a row must fit the CU's LDS (H <= 40704)");
    const VQ vq{bits, exp_bits, rmode, allow_denorm, max_norm};
    if (H % 512 == 0 && H <= 8192) {
        const int rpb = rows >= 2048 ? 2 : 1;                                 // rows per block (see the kernel; measured 1 / 2 / 3 / 4 / 8: 16.0 / 15.5 / 16.3 / 16.4 / 21.6 us)
        const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
        const bool fast = vq_is_fast(bits, exp_bits, rmode, allow_denorm);
        const hipStream_t st = (hipStream_t)stream;
        if (H <= 4096) {
            if (fast) hipLaunchKernelGGL((k_vec_layernorm_reg<1, 1>), dim3(grid), dim3(256), 0, st, x, weight, bias, out, rows, (int)H, eps, vq);
            else hipLaunchKernelGGL((k_vec_layernorm_reg<0, 1>), dim3(grid), dim3(256), 0, st, x, weight, bias, out, rows, (int)H, eps, vq);
        } else {
            if (fast) hipLaunchKernelGGL((k_vec_layernorm_reg<1, 2>), dim3(grid), dim3(256), 0, st, x, weight, bias, out, rows, (int)H, eps, vq);
            else hipLaunchKernelGGL((k_vec_layernorm_reg<0, 2>), dim3(grid), dim3(256), 0, st, x, weight, bias, out, rows, (int)H, eps, vq);
        }
        return hipGetLastError() == hipSuccess ? MSQ_OK : vfail(MSQ_ERR_LAUNCH, "msq_vec_layernorm: launch failed");
    }
    const size_t lds = (size_t)H * 4;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_vec_layernorm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_vec_layernorm, dim3((unsigned)rows), dim3(64), lds, (hipStream_t)stream, x, weight, bias, out, rows, H, eps,
                       VQ{bits, exp_bits, rmode, allow_denorm, max_norm});
    return hipGetLastError() == hipSuccess ? MSQ_OK : vfail(MSQ_ERR_LAUNCH, "msq_vec_layernorm: launch failed");
}

int msq_vec_gelu(const float* x, float* out, int64_t n, int first_order, int bits, int exp_bits, float max_norm, int rmode,
                 int allow_denorm, void* stream) {
    if (n < 0) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_gelu: negative size");
    if (n == 0) return MSQ_OK;
    if (!x || !out) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_gelu: null buffer");
    if (int rc = vq_check(bits, exp_bits, rmode)) return rc;
    const VQ vq{bits, exp_bits, rmode, allow_denorm, max_norm};
    int64_t done = 0;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && n >= 4) {
        const int64_t n4 = n / 4;
        if (vq_is_fast(bits, exp_bits, rmode, allow_denorm))
            hipLaunchKernelGGL(k_vec_gelu4<1>, dim3(grid1(n4)), dim3(256), 0, (hipStream_t)stream, x, out, n4, first_order, vq);
        else
            hipLaunchKernelGGL(k_vec_gelu4<0>, dim3(grid1(n4)), dim3(256), 0, (hipStream_t)stream, x, out, n4, first_order, vq);
        done = n4 * 4;
    }
    if (done < n)
        hipLaunchKernelGGL(k_vec_gelu, dim3(grid1(n - done)), dim3(256), 0, (hipStream_t)stream, x + done, out + done, n - done, first_order, vq);
    return hipGetLastError() == hipSuccess ? MSQ_OK : vfail(MSQ_ERR_LAUNCH, "msq_vec_gelu: launch failed");
}

int msq_vec_add(const float* a, const float* b, float b_scalar, float* out, int64_t n, int bits, int exp_bits, float max_norm,
                int rmode, int allow_denorm, void* stream) {
    if (n < 0) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_add: negative size");
    if (n == 0) return MSQ_OK;
    if (!a || !out) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_add: null buffer");
    if (int rc = vq_check(bits, exp_bits, rmode)) return rc;
    const VQ vq{bits, exp_bits, rmode, allow_denorm, max_norm};
    int64_t done = 0;
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && n >= 4) {
        const int64_t n4 = n / 4;
        if (vq_is_fast(bits, exp_bits, rmode, allow_denorm))
            hipLaunchKernelGGL(k_vec_add4<1>, dim3(grid1(n4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n4, b ? 0 : 1, b_scalar, vq);
        else
            hipLaunchKernelGGL(k_vec_add4<0>, dim3(grid1(n4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n4, b ? 0 : 1, b_scalar, vq);
        done = n4 * 4;
    }
    if (done < n)
        hipLaunchKernelGGL(k_vec_add, dim3(grid1(n - done)), dim3(256), 0, (hipStream_t)stream, a + done, b ? b + done : nullptr, out + done, n - done, b ? 0 : 1, b_scalar, vq);
    return hipGetLastError() == hipSuccess ? MSQ_OK : vfail(MSQ_ERR_LAUNCH, "msq_vec_add: launch failed");
}


/* RMSNorm (layernorm.py:98-128) and, with codes / scales given, the MX-FP8 activation pack of its output in the same launch.
 * x_dtype 0 float32, 1 float16, 2 bfloat16 (the library's dtype codes): 16-bit activations are read as they are. */
static int rmsnorm_impl(const void* x, int x_dtype, const float* weight, const float* bias, float* out, void* codes, void* scales,
                        int* status_flag, int64_t rows, int64_t H, float eps, int bits, int exp_bits, float max_norm, int rmode,
                        int allow_denorm, int flush, void* stream) {
    if (rows < 0 || H < 0) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_rmsnorm: negative size");
    if (rows * H == 0) return MSQ_OK;
    if (x_dtype < 0 || x_dtype > 2) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_rmsnorm: x_dtype must be 0 (float32), 1 (float16) or 2 (bfloat16)");
    if (!x || !weight || (!out && !codes)) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_rmsnorm: null buffer (x, weight and one of out / codes are required)");
    if ((codes != nullptr) != (scales != nullptr)) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_rmsnorm_mx_pack_a8: codes and scales go together");
    if (codes && (H % 128)) return vfail(MSQ_ERR_UNSUPPORTED, "msq_vec_rmsnorm_mx_pack_a8: H must be a multiple of 128 (the GEMM's K)");
    if (codes && ((uintptr_t)codes & 7)) return vfail(MSQ_ERR_UNSUPPORTED, "msq_vec_rmsnorm_mx_pack_a8: codes must be 8-byte aligned");
    if (int rc = vq_check(bits, exp_bits, rmode)) return rc;
    if (H * 4 > 160 * 1024 - 1024) return vfail(MSQ_ERR_UNSUPPORTED, "msq_vec_rmsnorm: a row must fit the CU's LDS (H <= 40704)");
    const VQ vq{bits, exp_bits, rmode, allow_denorm, max_norm};
    const PackOut P{(uint8_t*)codes, (uint8_t*)scales, status_flag, flush};
    const hipStream_t st = (hipStream_t)stream;
    const int xs = x_dtype == 2 ? 1 : (x_dtype == 1 ? 2 : 0);               // kernel source code: 0 f32, 1 bf16, 2 f16
    if (H % 512 == 0 && H <= 8192) {
        int rpb = rows >= 8192 ? 4 : (rows >= 2048 ? 2 : 1);      // rows per block, measured (scripts/experiments/rms_rpb.py): M 2048: 13.5 / 13.3 / 14.5 / 15.6 us for 1 / 2 / 3 / 4; M 8192: 42.2 / 36.9 / 36.7 / 35.7
        { const int t = g_rms_rpb.value(0); if (t > 0) rpb = t; }
        const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
        const bool fast = vq_is_fast(bits, exp_bits, rmode, allow_denorm);
#define MSQ_RMS3(FAST, GP, PK, XS) hipLaunchKernelGGL((k_vec_rmsnorm_reg<FAST, GP, PK, XS>), dim3(grid), dim3(256), 0, st, x, weight, bias, out, rows, (int)H, eps, vq, P)
#define MSQ_RMS2(FAST, GP, PK) do { if (xs == 0) MSQ_RMS3(FAST, GP, PK, 0); else if (xs == 1) MSQ_RMS3(FAST, GP, PK, 1); else MSQ_RMS3(FAST, GP, PK, 2); } while (0)
#define MSQ_RMS(FAST, GP) do { if (codes) MSQ_RMS2(FAST, GP, 1); else MSQ_RMS2(FAST, GP, 0); } while (0)
        if (H <= 4096) { if (fast) MSQ_RMS(1, 1); else MSQ_RMS(0, 1); }
        else { if (fast) MSQ_RMS(1, 2); else MSQ_RMS(0, 2); }
#undef MSQ_RMS
#undef MSQ_RMS2
#undef MSQ_RMS3
        return hipGetLastError() == hipSuccess ? MSQ_OK : vfail(MSQ_ERR_LAUNCH, "msq_vec_rmsnorm: launch failed");
    }
    const size_t lds = (size_t)H * 4;
#define MSQ_RMSG(XS) do { if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_vec_rmsnorm<XS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                          hipLaunchKernelGGL(k_vec_rmsnorm<XS>, dim3((unsigned)rows), dim3(64), lds, st, x, weight, bias, out, rows, H, eps, vq, P); } while (0)
    if (xs == 0) MSQ_RMSG(0); else if (xs == 1) MSQ_RMSG(1); else MSQ_RMSG(2);
#undef MSQ_RMSG
    return hipGetLastError() == hipSuccess ? MSQ_OK : vfail(MSQ_ERR_LAUNCH, "msq_vec_rmsnorm: launch failed");
}
int msq_vec_rmsnorm(const float* x, const float* weight, const float* bias, float* out, int64_t rows, int64_t H, float eps,
                    int bits, int exp_bits, float max_norm, int rmode, int allow_denorm, void* stream) {
    if (rows * H != 0 && !out) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_rmsnorm: null buffer");
    return rmsnorm_impl(x, 0, weight, bias, out, nullptr, nullptr, nullptr, rows, H, eps, bits, exp_bits, max_norm, rmode, allow_denorm, 0, stream);
}
int msq_vec_rmsnorm_mx_pack_a8(const float* x, const float* weight, const float* bias, float* out, void* codes, void* scales, int* status_flag,
                               int64_t rows, int64_t H, float eps, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm,
                               int flush_fp32_subnorms, void* stream) {
    if (rows * H != 0 && (!codes || !scales)) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_rmsnorm_mx_pack_a8: null buffer");
    return rmsnorm_impl(x, 0, weight, bias, out, codes, scales, status_flag, rows, H, eps, bits, exp_bits, max_norm, rmode,
                        allow_denorm, flush_fp32_subnorms, stream);
}
int msq_vec_rmsnorm_mx_pack_a8_x16(const void* x, int x_dtype, const float* weight, const float* bias, float* out, void* codes, void* scales,
                                   int* status_flag, int64_t rows, int64_t H, float eps, int bits, int exp_bits, float max_norm, int rmode,
                                   int allow_denorm, int flush_fp32_subnorms, void* stream) {
    if (x_dtype != 1 && x_dtype != 2) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_rmsnorm_mx_pack_a8_x16: x_dtype must be 1 (float16) or 2 (bfloat16)");
    return rmsnorm_impl(x, x_dtype, weight, bias, out, codes, scales, status_flag, rows, H, eps, bits, exp_bits, max_norm, rmode,
                        allow_denorm, flush_fp32_subnorms, stream);
}

/* mode 0: silu(gate) (activations.py:420-434); 1: simd_mul(gate, up) (simd_ops.py:154-187); 2: simd_mul(silu(gate), up).  Rows of I values at
 * row strides ld_gate / ld_up (elements); out [M, I] and / or the MX-FP8 pack of the result.  x_dtype as above (gate and up alike). */
static int act_impl(int mode, const void* gate, const void* up, int x_dtype, int64_t ld_gate, int64_t ld_up, float* out, void* codes, void* scales,
                    int* status_flag, int64_t M, int64_t I, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm, int flush,
                    void* stream) {
    if (M < 0 || I < 0) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_silu_mul: negative size");
    if (M * I == 0) return MSQ_OK;
    if (x_dtype < 0 || x_dtype > 2) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_silu_mul: x_dtype must be 0 (float32), 1 (float16) or 2 (bfloat16)");
    if (!gate || (mode != 0 && !up) || (!out && !codes)) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_silu_mul: null buffer");
    if ((codes != nullptr) != (scales != nullptr)) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_silu_mul_mx_pack_a8: codes and scales go together");
    if (ld_gate < I || (mode != 0 && ld_up < I)) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_silu_mul: a row stride is shorter than the row");
    if (int rc = vq_check(bits, exp_bits, rmode)) return rc;
    const VQ vq{bits, exp_bits, rmode, allow_denorm, max_norm};
    const hipStream_t st = (hipStream_t)stream;
    const int ldm = x_dtype ? 8 : 4;                                        // row strides that keep 16-byte pieces aligned
    const bool al = (((uintptr_t)gate | (uintptr_t)up | (uintptr_t)out) & 15) == 0 && (ld_gate % ldm) == 0 && (ld_up % ldm) == 0 && (I % 8) == 0;
    if (codes) {
        if (I % 128) return vfail(MSQ_ERR_UNSUPPORTED, "msq_vec_silu_mul_mx_pack_a8: I must be a multiple of 128 (the GEMM's K)");
        if (!al || ((uintptr_t)codes & 7)) return vfail(MSQ_ERR_UNSUPPORTED, "msq_vec_silu_mul_mx_pack_a8: buffers must be 16-byte aligned, row strides multiples of 16 bytes");
    }
    const PackOut P{(uint8_t*)codes, (uint8_t*)scales, status_flag, flush};
    const int xs = x_dtype == 2 ? 1 : (x_dtype == 1 ? 2 : 0);
    if (al) {
        const int64_t n8 = M * I / 8;
        const dim3 grid((unsigned)((n8 + 255) / 256)), blk(256);
        const bool fast = vq_is_fast(bits, exp_bits, rmode, allow_denorm);
#define MSQ_ACT3(FAST, MODE, XS) hipLaunchKernelGGL((k_vec_act8<FAST, MODE, XS>), grid, blk, 0, st, gate, up, ld_gate, ld_up, out, n8, (int)(I / 8), vq, P)
#define MSQ_ACT2(FAST, MODE) do { if (xs == 0) MSQ_ACT3(FAST, MODE, 0); else if (xs == 1) MSQ_ACT3(FAST, MODE, 1); else MSQ_ACT3(FAST, MODE, 2); } while (0)
#define MSQ_ACT(MODE) do { if (fast) MSQ_ACT2(1, MODE); else MSQ_ACT2(0, MODE); } while (0)
        if (mode == 0) MSQ_ACT(0); else if (mode == 1) MSQ_ACT(1); else MSQ_ACT(2);
#undef MSQ_ACT
#undef MSQ_ACT2
#undef MSQ_ACT3
    } else {
        if (x_dtype != 0) return vfail(MSQ_ERR_UNSUPPORTED, "msq_vec_silu_mul: 16-bit inputs need 16-byte aligned buffers, I % 8 == 0 and row strides % 8 == 0");
        if (ld_gate != I || (mode != 0 && ld_up != I)) return vfail(MSQ_ERR_UNSUPPORTED, "msq_vec_silu_mul: strided rows need 16-byte aligned buffers and I % 8 == 0");
        const int64_t n = M * I;
        const float* g32 = (const float*)gate; const float* u32 = (const float*)up;
        if (mode == 0) hipLaunchKernelGGL(k_vec_act1<0>, dim3(grid1(n)), dim3(256), 0, st, g32, u32, out, n, vq);
        else if (mode == 1) hipLaunchKernelGGL(k_vec_act1<1>, dim3(grid1(n)), dim3(256), 0, st, g32, u32, out, n, vq);
        else hipLaunchKernelGGL(k_vec_act1<2>, dim3(grid1(n)), dim3(256), 0, st, g32, u32, out, n, vq);
    }
    return hipGetLastError() == hipSuccess ? MSQ_OK : vfail(MSQ_ERR_LAUNCH, "msq_vec_silu_mul: launch failed");
}
int msq_vec_silu(const float* x, float* out, int64_t n, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm, void* stream) {
    if (n < 0) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_silu: negative size");
    if (n > 0 && !out) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_silu: null buffer");
    return act_impl(0, x, nullptr, 0, n, n, out, nullptr, nullptr, nullptr, n ? 1 : 0, n, bits, exp_bits, max_norm, rmode, allow_denorm, 0, stream);
}
int msq_vec_mul(const float* a, const float* b, float* out, int64_t n, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm, void* stream) {
    if (n < 0) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_mul: negative size");
    if (n > 0 && !out) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_mul: null buffer");
    return act_impl(1, a, b, 0, n, n, out, nullptr, nullptr, nullptr, n ? 1 : 0, n, bits, exp_bits, max_norm, rmode, allow_denorm, 0, stream);
}
int msq_vec_silu_mul_mx_pack_a8(const float* gate, const float* up, int64_t ld_gate, int64_t ld_up, float* out, void* codes, void* scales,
                                int* status_flag, int64_t M, int64_t I, int bits, int exp_bits, float max_norm, int rmode, int allow_denorm,
                                int flush_fp32_subnorms, void* stream) {
    return act_impl(2, gate, up, 0, ld_gate, ld_up, out, codes, scales, status_flag, M, I, bits, exp_bits, max_norm, rmode, allow_denorm,
                    flush_fp32_subnorms, stream);
}
int msq_vec_silu_mul_mx_pack_a8_x16(const void* gate, const void* up, int x_dtype, int64_t ld_gate, int64_t ld_up, float* out, void* codes,
                                    void* scales, int* status_flag, int64_t M, int64_t I, int bits, int exp_bits, float max_norm, int rmode,
                                    int allow_denorm, int flush_fp32_subnorms, void* stream) {
    if (x_dtype != 1 && x_dtype != 2) return vfail(MSQ_ERR_BAD_ARG, "msq_vec_silu_mul_mx_pack_a8_x16: x_dtype must be 1 (float16) or 2 (bfloat16)");
    return act_impl(2, gate, up, x_dtype, ld_gate, ld_up, out, codes, scales, status_flag, M, I, bits, exp_bits, max_norm, rmode, allow_denorm,
                    flush_fp32_subnorms, stream);
}

}  // extern "C"
