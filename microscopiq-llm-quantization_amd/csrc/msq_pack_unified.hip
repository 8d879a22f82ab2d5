// msq_pack_unified.hip -- pack kernels of the unified layout MSQ-U1 (one e4m3 code per weight, one scale per
// 32 k, optional extension bit): fused quantise + pack (msq_outlier_pack, layout unified) and packing of given
// values (msq_pack_values).  Own translation unit (heavy templates).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msq.h"
#include "msq_device.h"
#include "msq_host.h"
#include "msq_outlier_core.h"

extern "C" void msq_set_error_(const char* msg);

// ===========================================================================
// Fused pack, unified layout (MSQ-U1): as k_pack_tile, but every 32-k half row gets ONE scale and every
// weight one e4m3 code (+ one extension bit when EXT: the 4th fraction bit of posit<8,1> outliers).
//   scale s = floor(log2(max|v|)) - 8 (+1 when max|v| 2^-s > 448): when an e4m3 outlier holds the maximum
//   this is the outlier's own scale, so its codes are unchanged; inlier e2m1 values widen exactly.
// Every code is decoded back with the GEMM's own instruction and compared with the fake-quant value.
// ===========================================================================
// BS == 0: the input already holds fake-quant VALUES (any block direction, GPTQ output, ...): no quantiser runs,
// the values are only encoded and checked (msq_pack_values).
// A wave owns 64 n x KW k of a packed tile: KW = 32 (one 32-k half: its own code slots, scale dword and extension word, so
// the two halves of a tile are independent work items) whenever the quantiser block fits (BS <= 32 or values only), else
// the whole 64 x 64 tile.  The half tile needs 9.2 KB of LDS per wave instead of 17.4: four workgroups per CU instead of
// two -- the kernel is latency-bound (one row per lane, LDS round trips between its phases), occupancy is what it lacks.
template <int BS, bool EXT, int HW>
#ifndef MSQ_PACKU_MINB
#define MSQ_PACKU_MINB 4
#endif
__global__ void __launch_bounds__(256, (BS <= 32 ? MSQ_PACKU_MINB : 2))
k_pack_tile_u(const float* __restrict__ W, uint8_t* __restrict__ ext_plane, uint8_t* __restrict__ code_plane,
              uint8_t* __restrict__ scl_plane, OutlierArgs A, int64_t N, int64_t K) {
    // Lane r owns row r of the tile in LDS (KW + 4 words per row) and works on it IN PLACE, one block / one 32-k half at
    // a time, so that no 64-element register array is needed: the fake-quant values overwrite the inputs; the codes of
    // half h then go to bytes 32 h .. 32 h + 31 of the row (already consumed), the scale bytes to word KW and the
    // extension bytes to words KW + 1 .. once the halves are encoded.
    constexpr int KW = (BS <= 32) ? 32 : 64;                     // k columns per wave
    constexpr int NH = KW / 32;                                  // 32-k halves per wave
    constexpr int ROW_W = KW + 4;                                // words per row
    constexpr int WAVE_LDS = 64 * ROW_W * 4;
    __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t KT = K / 64, NT = N / 64;
    const int64_t unit = (int64_t)blockIdx.x * 4 + wv;
    const int64_t tile = unit / (64 / KW);
    const int kf0 = (int)(unit % (64 / KW)) * NH;                // first half of the tile this wave owns
    if (tile >= KT * NT) return;
    const int64_t nt = tile / KT, kt = tile % KT;
    float* ft = reinterpret_cast<float*>(lds + wv * WAVE_LDS);
    {
        const float* src = W + (nt * 64) * K + kt * 64 + kf0 * 32;
        constexpr int C4 = KW / 4, RPI = 64 / C4;                // float4 per row, rows per iteration
#pragma unroll
        for (int i = 0; i < 64 / RPI; ++i) {
            const int row = i * RPI + lane / C4, c4 = lane % C4;
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
        for (int j = 0; j < KW / B; ++j) {
            float a[B];
#pragma unroll
            for (int c = 0; c < B / 4; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(myrow + j * B + c * 4);
                a[c * 4 + 0] = v.x; a[c * 4 + 1] = v.y; a[c * 4 + 2] = v.z; a[c * 4 + 3] = v.w;
            }
            uint32_t mkw[(B + 31) / 32];
            float se_in, se_out;
            status |= outlier_block_fast<B, 0, false, HW>(a, mkw, se_in, se_out, A, /*inner order*/ 1, nullptr, nullptr, 1, nullptr, 0, 0,
                                                          (HW == 2) ? myrow + j * B : nullptr);
#pragma unroll
            for (int c = 0; c < B / 4; ++c)
                *reinterpret_cast<float4*>(myrow + j * B + c * 4) = make_float4(a[c * 4 + 0], a[c * 4 + 1], a[c * 4 + 2], a[c * 4 + 3]);
        }
    }
    // 2. one scale + 32 codes (+ 32 extension bits) per half
    uint32_t sbytes = 0, eb_w[NH];
#pragma nounroll
    for (int h = 0; h < NH; ++h) {
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
        eb_w[h] = eb;
    }
    reinterpret_cast<uint32_t*>(myrow)[KW] = sbytes;             // the scale byte(s)
    if (EXT) {
#pragma unroll
        for (int h = 0; h < NH; ++h) reinterpret_cast<uint32_t*>(myrow)[KW + 1 + h] = eb_w[h];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // 3. fragment gather + slot stores
    const int c = lane & 15, g = lane >> 4;
    const uint32_t* rows = reinterpret_cast<const uint32_t*>(ft);
    uint32_t sc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int kf = kf0 + h;
        uint32_t ew = 0;
        sc[h] = 0;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int n = nf * 16 + c, k8 = h * 4 + g;
            const uint32_t* r = rows + n * ROW_W;
            sc[h] |= ((r[KW] >> (8 * h)) & 0xFFu) << (8 * nf);
            const uint2 o = *reinterpret_cast<const uint2*>(r + k8 * 2);
            *reinterpret_cast<uint2*>(code_plane + ((tile * 4 + kf * 2 + (nf >> 1)) * 64 + lane) * 16 + (nf & 1) * 8) = o;
            if (EXT) {
                const uint32_t eb = (r[KW + 1 + h] >> (8 * g)) & 0xFFu;    // elements 8 g .. 8 g + 7 of half kf
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    ew |= ((eb >> j) & 1u) << ((3 + 16 * (j & 1) + 4 * nf + (j >> 1)) & 31);
            }
        }
        if (EXT) *reinterpret_cast<uint32_t*>(ext_plane + ((tile * 2 + kf) * 64 + lane) * 4) = ew;
        if (g == 0) *reinterpret_cast<uint32_t*>(scl_plane + (tile * 16 + c) * 8 + kf * 4) = sc[h];
    }
    if (status && A.status) atomicOr(A.status, status);
}

// quantise + pack; hw = quantiser codec mode (see msq_pack_fused_)
extern "C" int msq_pack_unified_(const float* W, void* ext_plane, void* code_plane, void* scale_plane, const OutlierArgs* Ap,
                                 int64_t N, int64_t K, int block, int out_kind, int hw, void* stream) {
    const OutlierArgs A = *Ap;
    const int64_t tiles = (N / 64) * (K / 64) * (block <= 32 ? 2 : 1);     // work items: half tiles for blocks <= 32
    const dim3 grid((unsigned)((tiles + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define MSQ_PU(BS, EXTV, HWV) hipLaunchKernelGGL((k_pack_tile_u<BS, EXTV, HWV>), grid, blk, 0, st, W, (uint8_t*)ext_plane, \
                                                 (uint8_t*)code_plane, (uint8_t*)scale_plane, A, N, K)
#define MSQ_PUB(EXTV, HWV) do { switch (block) { case 8: MSQ_PU(8, EXTV, HWV); break; case 16: MSQ_PU(16, EXTV, HWV); break; \
                                                 case 32: MSQ_PU(32, EXTV, HWV); break; default: MSQ_PU(64, EXTV, HWV); break; } } while (0)
    if (out_kind == MSQ_PLANE_U8) { if (hw == 1) MSQ_PUB(false, 1); else MSQ_PUB(false, 0); }
    else { if (hw == 2) MSQ_PUB(true, 2); else MSQ_PUB(true, 0); }
#undef MSQ_PUB
#undef MSQ_PU
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

// values -> unified planes (no quantiser): out_kind MSQ_PLANE_U8 or MSQ_PLANE_U8X
extern "C" int msq_pack_values_u_(const float* W, void* ext_plane, void* code_plane, void* scale_plane, int* status,
                                  int64_t N, int64_t K, int out_kind, void* stream) {
    OutlierArgs A = {};
    A.status = status;
    const int64_t tiles = (N / 64) * (K / 64) * 2;                        // half tiles
    const dim3 grid((unsigned)((tiles + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (out_kind == MSQ_PLANE_U8) hipLaunchKernelGGL((k_pack_tile_u<0, false, 0>), grid, blk, 0, st, W, (uint8_t*)ext_plane, (uint8_t*)code_plane, (uint8_t*)scale_plane, A, N, K);
    else hipLaunchKernelGGL((k_pack_tile_u<0, true, 0>), grid, blk, 0, st, W, (uint8_t*)ext_plane, (uint8_t*)code_plane, (uint8_t*)scale_plane, A, N, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { msq_set_error_(hipGetErrorString(e)); return MSQ_ERR_LAUNCH; }
    return MSQ_OK;
}

