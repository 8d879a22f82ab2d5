"""QuantLinear -- the packed MicroScopiQ Linear the reference only sketches
(`make_quant3` / `Quant3Linear` are called at llm/opt.py:255-294 but defined nowhere).

Contract: for x in bf16, ``QuantLinear.forward(x)`` equals ``F.linear(x, Wq)`` where
``Wq = quantize_mx_outlier_v1(W, ...)`` is the reference fake-quant (utils/quant.py:147-266)
-- the dequantised values are exact in bf16, only the fp32 summation order differs.
The weight lives in HBM as GEMM-ready tile-major planes (include/msq.h "MSQ-T1") and is
dequantised inside the MFMA kernel (csrc/msq_gemm.hip), never materialised.

Row-parallel (K-split) sharding for the 70B configuration: RowParallelQuantLinear keeps
W[:, rank*K/G:(rank+1)*K/G] packed on every rank and all-reduces the partial outputs over
RCCL.  Blocks run along K, so a K split on a multiple of the block size
never cuts a block and every shard's masks / scales equal the unsharded ones.
"""
import ctypes as C

import torch
import torch.nn as nn

from ._lib import MsqError, check, current_stream, lib, ptr
from .formats import RoundingMode, format_id

_PLANE_NONE = 0


N_MULT, K_MULT, K_MULT_MX = 256, 64, 128       # tile grid of the GEMM kernels (csrc/msq_gemm.hip: BN, BK; MX K-step)


def _ceil_to(v, m):
    return (int(v) + m - 1) // m * m


def _pad2d(W, n_mult, k_mult):
    """Zero-pad a [N, K] weight to the kernels' tile grid.  Blocks run along K and every block is quantised on its own
    (utils/quant.py:477-492 takes mean / std per block), so zero rows and zero k-columns change no other value: a ragged
    last block sees the same zeros the reference's _reshape_to_blocks pads it with, and the extra all-zero blocks
    quantise to zero."""
    N, K = W.shape
    Np, Kp = _ceil_to(N, n_mult), _ceil_to(K, k_mult)
    if (Np, Kp) == (N, K):
        return W
    Wp = W.new_zeros(Np, Kp)
    Wp[:N, :K] = W
    return Wp


class PackedWeight:
    """Device buffers + metadata of one packed weight.  ``N`` / ``K`` are the PADDED dimensions the planes were built
    for (multiples of 256 / 64: what the kernels see), ``n`` / ``k`` the logical [out_features, in_features] of the
    layer: qlinear zero-pads the activations to K and returns the first n columns."""

    def __init__(self, inl, out, scl, N, K, block, in_kind, out_kind, n=None, k=None):
        self.inl, self.out, self.scl = inl, out, scl
        self.N, self.K, self.block, self.in_kind, self.out_kind = N, K, block, in_kind, out_kind
        self.n, self.k = (N if n is None else int(n)), (K if k is None else int(k))

    @property
    def nbytes(self):
        return sum(int(t.numel()) for t in (self.inl, self.out, self.scl) if t is not None)

    @property
    def bits_per_element(self):
        return 8.0 * self.nbytes / (self.n * self.k)


LAYOUT_PLANES, LAYOUT_UNIFIED = 0, 1
_LAYOUTS = {"planes": LAYOUT_PLANES, "unified": LAYOUT_UNIFIED, LAYOUT_PLANES: LAYOUT_PLANES,
            LAYOUT_UNIFIED: LAYOUT_UNIFIED}


def packed_kinds(inlier_elem_format, outlier_elem_format, layout="planes"):
    """Plane kinds of a packed layout: "planes" = MSQ-T1 (inlier plane + outlier plane, two scales per
    block), "unified" = MSQ-U1 (one e4m3 code per weight [+ extension bit], one scale per 32 k)."""
    ik, ok = C.c_int(), C.c_int()
    check(lib().msq_packed_kinds_layout(format_id(inlier_elem_format), format_id(outlier_elem_format),
                                        _LAYOUTS[layout], C.byref(ik), C.byref(ok)), "msq_packed_kinds_layout")
    return ik.value, ok.value


def packed_sizes(N, K, block, in_kind, out_kind):
    ib, ob, sb, wb = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    check(lib().msq_packed_sizes(N, K, block, in_kind, out_kind, C.byref(ib), C.byref(ob), C.byref(sb), C.byref(wb)),
          "msq_packed_sizes")
    return ib.value, ob.value, sb.value, wb.value


def pack_weight(W, inlier_scale_bits=8, outlier_scale_bits=8, inlier_elem_format="fp4_e2m1",
                outlier_elem_format="fp8_e4m3", std_dev=2, block_size=32, round="nearest",
                flush_fp32_subnorms=False, variant=0, layout="planes", compute_dtype="input"):
    """Quantise W [N, K] (blocks along K = the reference's axes=[-1]) and pack it.  Any N, K: the weight is zero-padded
    to the kernels' tile grid (``PackedWeight.N / K``), the logical shape is kept in ``PackedWeight.n / k``.

    layout: "planes" (MSQ-T1), "unified" (MSQ-U1, smaller and faster; raises MsqError if a 32-k group
    cannot be represented exactly) or "auto" (unified when the formats allow it and every group is
    exact, otherwise planes).

    ``compute_dtype`` as quant.outlier_fakequant: an fp16 / bf16 weight is by default quantised IN its dtype (what the
    reference's RTN harness does with a half checkpoint, llm/llama.py:238), so that the packed layer holds exactly the
    values ``MXQuantizer.quantize`` / ``quantize_mx_outlier_v1`` give for the same tensor; those values are then packed
    as they are (pack_values; layout "planes": MSQ-T1 planes of those values when they reproduce them exactly, else the unified
    planes of those values, MsqError if neither fits -- never a 16-bit plane).  "float32" upcasts first (one fused quantise +
    pack launch)."""
    if not W.is_cuda:
        raise MsqError("pack_weight needs a CUDA/HIP tensor (no CPU fallback)")
    if W.ndim != 2:
        raise MsqError("pack_weight expects a 2-D [out_features, in_features] weight")
    if compute_dtype not in ("input", "float32"):
        raise MsqError("compute_dtype must be 'input' or 'float32'")
    n, k = W.shape
    if (compute_dtype == "input" and W.dtype in (torch.float16, torch.bfloat16) and variant == 0
            and not str(inlier_elem_format).startswith("posit") and not str(outlier_elem_format).startswith("posit")):
        from .quant import outlier_fakequant
        Wq = outlier_fakequant(W.detach(), inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                               std_dev, -1, block_size, round, flush_fp32_subnorms, compute_dtype="input")["out"]
        if layout == "planes":
            # MSQ-T1 (fp4 plane + outlier plane + two scales per block) is what "planes" names -- and never a silent 16-bit plane.
            # The T1 packer quantises once more in float32; its result is kept only if it reproduces the in-dtype values exactly.
            # On a real matrix that often fails (an element changes sides of the outlier bounds once its block is quantised: among
            # the 524288 blocks of a 4096 x 4096 weight practically always), so the values are then packed AS THEY ARE into the
            # unified planes (8.25 / 9.25 bits per weight: smaller than T1's 12.5, the same GEMM kernels) -- the default call on a
            # half weight succeeds and decodes to exactly what `MXQuantizer.quantize` gives for it (advisor, round 4).  Only a tensor
            # whose groups fit neither raises.
            try:
                P = _pack(_pad2d(Wq.detach().contiguous().float(), N_MULT, K_MULT), inlier_scale_bits, outlier_scale_bits,
                          inlier_elem_format, outlier_elem_format, std_dev, block_size, round, flush_fp32_subnorms, variant, "planes", n, k)
                if bool((unpack_weight(P) == Wq.float()).all()):
                    return P
            except MsqError as e:
                # only "the planes do not reproduce these values" (the pack kernel's own exactness proof, MSQ_STATUS_INEXACT) falls through to the
                # unified planes; a configuration the packer does not support is the caller's error and is reported as such (advisor, round 5)
                if "INEXACT" not in str(e).upper() and "exact" not in str(e).lower():
                    raise
            try:
                return pack_values(Wq, (PLANE_U8, PLANE_U8X))
            except MsqError:
                raise MsqError("pack_weight(layout='planes'): the %s weight's in-dtype fake-quant values fit neither the MSQ-T1 planes nor "
                               "the unified planes exactly; use layout='auto' (falls back to a 16-bit plane of those values) or "
                               "compute_dtype='float32'" % str(W.dtype).replace("torch.", ""))
        kinds = {"unified": (PLANE_U8, PLANE_U8X), "auto": (PLANE_U8, PLANE_U8X, PLANE_BF16)}[layout]
        return pack_values(Wq, kinds)
    Wf = _pad2d(W.detach().contiguous().float(), N_MULT, K_MULT)
    if layout == "auto":
        if variant == 0 and round == "nearest" and block_size <= 64:
            try:
                return _pack(Wf, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format, std_dev,
                             block_size, round, flush_fp32_subnorms, variant, "unified", n, k)
            except MsqError:
                pass
        layout = "planes"
    return _pack(Wf, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format, std_dev,
                 block_size, round, flush_fp32_subnorms, variant, layout, n, k)


def _pack(Wf, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format, std_dev, block_size,
          round, flush_fp32_subnorms, variant, layout, n=None, k=None):
    N, K = Wf.shape
    ik, ok = packed_kinds(inlier_elem_format, outlier_elem_format, layout)
    ib, ob, sb, wb = packed_sizes(N, K, block_size, ik, ok)
    dev = Wf.device
    inl = torch.empty(ib, dtype=torch.uint8, device=dev) if ib else None
    out = torch.empty(ob, dtype=torch.uint8, device=dev)
    scl = torch.empty(sb, dtype=torch.uint8, device=dev) if sb else None
    ws = torch.empty(wb, dtype=torch.uint8, device=dev) if wb else None
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().msq_outlier_pack(ptr(Wf), ptr(inl), ptr(out), ptr(scl), ptr(status), ptr(ws), wb, N, K, block_size,
                                 format_id(inlier_elem_format), format_id(outlier_elem_format),
                                 int(inlier_scale_bits), int(outlier_scale_bits), float(std_dev),
                                 int(RoundingMode[round]), int(bool(flush_fp32_subnorms)), int(variant),
                                 _LAYOUTS[layout], current_stream(dev)),
          "msq_outlier_pack")
    st = int(status.item())
    if st & 1:
        raise AssertionError("shared_exp contains NaN values (scale overflow) while packing")
    if st & 2:
        raise MsqError("pack_weight: a value is not exactly code * 2^scale in the %s layout "
                       "(degenerate block scale / too wide a range inside one group)" % layout)
    return PackedWeight(inl, out, scl, N, K, block_size, ik, ok, n, k)


PLANE_NONE, PLANE_BF16, PLANE_U8, PLANE_U8X = 0, 4, 5, 6


def pack_values(Wq, kinds=(PLANE_U8, PLANE_U8X, PLANE_BF16)):
    """Pack a dense [N, K] tensor that already holds fake-quant values -- blocks along out_features
    (the reference harness default, llm/llama.py:229-237), the GPTQ solver's output (llm/gptq.py:166),
    anything -- into the first single-plane kind of `kinds` that represents every value exactly:
    unified e4m3 (8.25 bits/weight), unified + extension bit (9.25) or plain bf16 (16).  Nothing is rounded."""
    if not Wq.is_cuda:
        raise MsqError("pack_values needs a CUDA/HIP tensor (no CPU fallback)")
    if Wq.ndim != 2:
        raise MsqError("pack_values expects a 2-D [out_features, in_features] tensor")
    n, k = Wq.shape
    Wf = _pad2d(Wq.detach().contiguous().float(), N_MULT, K_MULT)
    N, K = Wf.shape
    dev = Wf.device
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    for ok in kinds:
        ib, ob, sb, _ = packed_sizes(N, K, 32, PLANE_NONE, ok)
        inl = torch.empty(ib, dtype=torch.uint8, device=dev) if ib else None
        out = torch.empty(ob, dtype=torch.uint8, device=dev)
        scl = torch.empty(sb, dtype=torch.uint8, device=dev) if sb else None
        status.zero_()
        check(lib().msq_pack_values(ptr(Wf), ptr(inl), ptr(out), ptr(scl), ptr(status), N, K, PLANE_NONE, ok,
                                    current_stream(dev)), "msq_pack_values")
        st = int(status.item())
        if st & 1:
            raise AssertionError("pack_values: the tensor contains NaN / Inf")
        if st == 0:
            return PackedWeight(inl, out, scl, N, K, 32, PLANE_NONE, ok, n, k)
    raise MsqError("pack_values: the values fit none of the requested plane kinds %r exactly" % (tuple(kinds),))


def unpack_weight(P, dtype=torch.float32):
    """Dense dequantised weight [n, k] (the logical shape; exact in f32 and bf16)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise MsqError("unpack_weight: dtype must be float32 or bfloat16")
    W = torch.empty(P.N, P.K, dtype=dtype, device=P.out.device)
    check(lib().msq_outlier_unpack(ptr(P.inl), ptr(P.out), ptr(P.scl), ptr(W), 0 if dtype == torch.float32 else 2,
                                   P.N, P.K, P.block, P.in_kind, P.out_kind, current_stream(W.device)),
          "msq_outlier_unpack")
    return W if (P.n, P.k) == (P.N, P.K) else W[:P.n, :P.k].contiguous()


_YD = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}     # y_dtype codes of include/msq.h


def _pad_k(xb, K):
    """[M, k] -> [M, K] with zero columns (off-grid in_features)"""
    return xb if xb.shape[-1] == K else torch.nn.functional.pad(xb, (0, K - xb.shape[-1]))


def _out_buffer(out, M, N, out_dtype, dev, who):
    """Caller-provided result buffer (a contiguous [M, N] view, e.g. a row slice of a larger tensor) or a fresh one."""
    if out is None:
        return torch.empty(M, N, dtype=out_dtype, device=dev)
    if tuple(out.shape) != (M, N) or out.dtype != out_dtype or not out.is_contiguous() or out.device != dev:
        raise MsqError("%s: out must be a contiguous [%d, %d] %s tensor on %s" % (who, M, N, out_dtype, dev))
    return out


def _cast_f16_bf16(x):
    """x.to(torch.bfloat16) for a float16 tensor as one bandwidth-bound launch (msq_cast_f16_bf16: same values)"""
    xc = x.contiguous()
    y = torch.empty(xc.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().msq_cast_f16_bf16(ptr(xc), ptr(y), xc.numel(), current_stream(x.device)), "msq_cast_f16_bf16")
    return y


def qlinear(x, P, bias=None, out_dtype=torch.bfloat16, out=None):
    """y = x . Wq^T (+ bias): the fused unpack-dequant-GEMM.  x: [..., K] (cast to bf16)."""
    if not x.is_cuda:
        raise MsqError("qlinear needs CUDA/HIP tensors (no CPU fallback)")
    k = x.shape[-1]
    if k != P.k:
        raise MsqError("qlinear: in_features mismatch (%d vs %d)" % (k, P.k))
    K = P.K
    xb = x.reshape(-1, k)
    # fp16 activations at decode sizes go to the kernels as they are (msq_qlinear_f16x converts while loading); everything else is
    # cast to bf16 first
    f16x = xb.dtype == torch.float16 and xb.shape[0] <= 64
    if xb.dtype == torch.float16 and not f16x:
        xb = _cast_f16_bf16(xb)
    elif xb.dtype != torch.bfloat16 and not f16x:
        xb = xb.to(torch.bfloat16)
    xb = _pad_k(xb, K).contiguous()
    M = xb.shape[0]
    if out_dtype not in _YD:
        raise MsqError("qlinear: out_dtype must be float32, float16 or bfloat16")
    padded_n = P.n != P.N
    if out is not None and padded_n:
        _out_buffer(out, M, P.n, out_dtype, x.device, "qlinear")          # validates the caller's buffer
    y = _out_buffer(None if padded_n else out, M, P.N, out_dtype, x.device, "qlinear")
    b = None
    if bias is not None:
        b = bias.detach().float().contiguous()
        if padded_n:
            b = torch.nn.functional.pad(b, (0, P.N - P.n))
    wsb = lib().msq_qlinear_workspace_bytes(M, P.N, K)        # > 0 only for small M (split-K partial tiles)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb > 0 else None
    rc = -2
    if f16x:
        rc = lib().msq_qlinear_f16x(ptr(xb), ptr(P.inl), ptr(P.out), ptr(P.scl), ptr(b), ptr(y), _YD[out_dtype], M, P.N, K, P.block,
                                    P.in_kind, P.out_kind, ptr(ws), wsb, current_stream(x.device))
        if rc not in (0, -2):
            check(rc, "msq_qlinear_f16x")
        if rc == -2:                                                      # not a decode shape after all: cast
            xb = _cast_f16_bf16(xb)
    if rc != 0:
        check(lib().msq_qlinear_bf16(ptr(xb), ptr(P.inl), ptr(P.out), ptr(P.scl), ptr(b), ptr(y),
                                     _YD[out_dtype], M, P.N, K, P.block, P.in_kind, P.out_kind,
                                     ptr(ws), wsb, current_stream(x.device)), "msq_qlinear_bf16")
    if padded_n:
        if out is not None:
            out.copy_(y[:, :P.n]); y = out
        else:
            y = y[:, :P.n].contiguous()
    return y.reshape(*x.shape[:-1], P.n)


def act_quant(x, inlier_scale_bits=8, outlier_scale_bits=8, inlier_elem_format="fp8_e4m3",
              outlier_elem_format="fp8_e4m3", std_dev=2, block_size=32, round="nearest",
              flush_fp32_subnorms=False, variant=0):
    """MicroScopiQ outlier-aware MX fake-quant of activations along the last axis, returned as bf16
    (exact for element formats of at most 8 bits).  variant 0 = utils/quant.py:147-266,
    1 = number_system/mx/mx_ops.py:210-330 (what MXLinear applies, linear.py:66-73)."""
    if not x.is_cuda:
        raise MsqError("act_quant needs a CUDA/HIP tensor (no CPU fallback)")
    K = x.shape[-1]
    x16 = x.dtype == torch.bfloat16 and (int(variant) == 0 or block_size in (32, 64))   # read as is: no cast pass
    xf = x.reshape(-1, K).contiguous() if x16 else x.reshape(-1, K).float().contiguous()
    if xf.data_ptr() % 16:                                # a contiguous view at an odd storage offset: the kernels read 16-byte pieces
        xf = xf.clone()
    M = xf.shape[0]
    xq = torch.empty(M, K, dtype=torch.bfloat16, device=x.device)
    wsb = lib().msq_act_quant_workspace_bytes(M, K, block_size, int(variant))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb > 0 else None
    status = torch.zeros(1, dtype=torch.int32, device=x.device)
    fn = lib().msq_act_quant_bf16_x16 if x16 else lib().msq_act_quant_bf16
    check(fn(ptr(xf), ptr(xq), ptr(status), ptr(ws), wsb, M, K, block_size,
             format_id(inlier_elem_format), format_id(outlier_elem_format),
             int(inlier_scale_bits), int(outlier_scale_bits), float(std_dev),
             int(RoundingMode[round]), int(bool(flush_fp32_subnorms)), int(variant),
             current_stream(x.device)), "msq_act_quant_bf16")
    return xq.reshape(x.shape), status


def qlinear_w4a8(x, P, bias=None, out_dtype=torch.float32, a_elem_format="fp8_e4m3", a_scale_bits=8, a_std_dev=2,
                 a_block_size=32, a_round="nearest", a_flush_fp32_subnorms=False, a_variant=0,
                 a_outlier_elem_format=None, check_status=False):
    """y = Q_a(x) . Wq^T (+ bias): activation quantisation (one pass, bf16, exact) + fused dequant-GEMM
    (BASELINE config 3, W4A8).  x: [..., K] float."""
    if not x.is_cuda:
        raise MsqError("qlinear_w4a8 needs CUDA/HIP tensors (no CPU fallback)")
    k = x.shape[-1]
    if k != P.k:
        raise MsqError("qlinear_w4a8: in_features mismatch (%d vs %d)" % (k, P.k))
    K = P.K
    x16 = x.dtype == torch.bfloat16 and (int(a_variant) == 0 or a_block_size in (32, 64))   # read as is: no cast pass
    xf = _pad_k(x.reshape(-1, k), K).contiguous() if x16 else _pad_k(x.reshape(-1, k).float(), K).contiguous()
    M = xf.shape[0]
    if out_dtype not in _YD:
        raise MsqError("qlinear_w4a8: out_dtype must be float32, float16 or bfloat16")
    y = torch.empty(M, P.N, dtype=out_dtype, device=x.device)
    b = bias.detach().float().contiguous() if bias is not None else None
    if b is not None and P.n != P.N:
        b = torch.nn.functional.pad(b, (0, P.N - P.n))
    wsb = lib().msq_qlinear_w4a8_workspace_bytes(M, P.N, K, a_block_size, int(a_variant))
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x.device)
    status = torch.zeros(1, dtype=torch.int32, device=x.device)
    fo = a_outlier_elem_format or a_elem_format
    fn = lib().msq_qlinear_w4a8_x16 if x16 else lib().msq_qlinear_w4a8
    check(fn(ptr(xf), ptr(P.inl), ptr(P.out), ptr(P.scl), ptr(b), ptr(y),
             _YD[out_dtype], M, P.N, K, P.block, P.in_kind, P.out_kind,
             a_block_size, format_id(a_elem_format), format_id(fo), int(a_scale_bits),
             int(a_scale_bits), float(a_std_dev), int(RoundingMode[a_round]),
             int(bool(a_flush_fp32_subnorms)), int(a_variant), ptr(status), ptr(ws), wsb,
             current_stream(x.device)), "msq_qlinear_w4a8")
    if check_status:
        st = int(status.item())
        if st & 1:
            raise AssertionError("shared_exp contains NaN values (activation scale overflow)")
        if st & 2:
            raise MsqError("qlinear_w4a8: a quantised activation is not exact in bf16")
    if P.n != P.N:
        y = y[:, :P.n].contiguous()
    return y.reshape(*x.shape[:-1], P.n)


# ---------------------------------------------------------------------------------------------------------
# MX-native W4A8 (BASELINE config 3): plain OCP-MX operands, no outlier split, multiplied by the scaled fp8 / fp4 MFMA
# ---------------------------------------------------------------------------------------------------------
_FP6_IDS = {"e3m2": 6, "e2m3": 7}                        # MSQ_FMT_FP6_E3M2 / MSQ_FMT_FP6_E2M3 (include/msq.h)
_MX_FMT_ALIASES = {"fp4": "e2m1", "fp4_e2m1": "e2m1", "fp6_e3m2": "e3m2", "fp6_e2m3": "e2m3", "fp8_e4m3": "e4m3"}


def _mx_fmt(name):
    name = str(name).lower()
    name = _MX_FMT_ALIASES.get(name, name)
    if name not in ("e2m1", "e3m2", "e2m3", "e4m3"):
        raise MsqError("MX operand format must be e2m1 / e3m2 / e2m3 / e4m3 (got %r)" % (name,))
    return name


class MXPackedWeight:
    """Codes in the operand order of the scaled MFMA + E8M0 block scales of one [N, K] weight: ``w_fmt`` "e2m1"
    (plain MX-FP4, 4.25 bits per weight), "e3m2" / "e2m3" (plain MX-FP6, a true 6-bit plane: 6.25 bits per weight)
    or "e4m3" (exactly packed fake-quant values, 8.25 bits per weight)."""

    def __init__(self, codes, scales, N, K, w_fmt="e2m1", n=None, k=None):
        self.codes, self.scales, self.N, self.K, self.w_fmt = codes, scales, N, K, w_fmt      # N, K: padded (256 / 128 grid)
        self.n, self.k = (N if n is None else int(n)), (K if k is None else int(k))              # logical shape

    @property
    def nbytes(self):
        return int(self.codes.numel() + self.scales.numel())

    @property
    def bits_per_element(self):
        return 8.0 * self.nbytes / (self.n * self.k)


def _mx_status(status, what):
    if int(status.item()) & 1:
        raise AssertionError("%s: a block holds Inf / NaN or its shared exponent overflows" % what)


def mx_pack_weight(W, flush_fp32_subnorms=False, w_fmt="e2m1"):
    """W [N, K] -> plain OCP-MX codes (block 32 along K, scale_bits 8, round nearest: mx_ops.py:332-457): MX-FP4
    (``w_fmt`` "e2m1", the default) or MX-FP6 ("e3m2" / "e2m3": six bits per code in HBM)."""
    if not W.is_cuda:
        raise MsqError("mx_pack_weight needs a CUDA/HIP tensor (no CPU fallback)")
    w_fmt = _mx_fmt(w_fmt)
    if w_fmt == "e4m3":
        raise MsqError("mx_pack_weight: e4m3 operands hold exact VALUES, use mx_pack_values")
    n, k = W.shape
    Wf = _pad2d(W.detach().contiguous().float(), N_MULT, K_MULT_MX)
    N, K = Wf.shape
    scales = torch.empty(N * K // 32, dtype=torch.uint8, device=Wf.device)
    status = torch.zeros(1, dtype=torch.int32, device=Wf.device)
    if w_fmt == "e2m1":
        codes = torch.empty(N * K // 2, dtype=torch.uint8, device=Wf.device)
        check(lib().msq_mx_pack_w4(ptr(Wf), ptr(codes), ptr(scales), ptr(status), N, K, int(bool(flush_fp32_subnorms)),
                                   current_stream(Wf.device)), "msq_mx_pack_w4")
    else:
        codes = torch.empty(N * K * 3 // 4, dtype=torch.uint8, device=Wf.device)
        check(lib().msq_mx_pack_w6(ptr(Wf), ptr(codes), ptr(scales), ptr(status), N, K, _FP6_IDS[w_fmt],
                                   int(bool(flush_fp32_subnorms)), current_stream(Wf.device)), "msq_mx_pack_w6")
    _mx_status(status, "mx_pack_weight")
    return MXPackedWeight(codes, scales, N, K, w_fmt, n, k)


def mx_pack_values(Wq, allow_inexact=False):
    """Fake-quant VALUES [N, K] (MicroScopiQ inliers + outliers, the mx_ops variant, GPTQ output ...) -> one e4m3
    code per weight + one E8M0 scale per 32 k in the fp8 operand order of the scaled MFMA (msq_mx_pack_w8).  Every
    code is decoded back and compared on the GPU: raises MsqError when a value is not representable (e.g. posit8
    outliers with 4 fraction bits, blocks spanning more than e4m3's range) unless ``allow_inexact``."""
    if not Wq.is_cuda:
        raise MsqError("mx_pack_values needs a CUDA/HIP tensor (no CPU fallback)")
    n, k = Wq.shape
    Wf = _pad2d(Wq.detach().contiguous().float(), N_MULT, K_MULT_MX)
    N, K = Wf.shape
    codes = torch.empty(N * K, dtype=torch.uint8, device=Wf.device)
    scales = torch.empty(N * K // 32, dtype=torch.uint8, device=Wf.device)
    status = torch.zeros(1, dtype=torch.int32, device=Wf.device)
    check(lib().msq_mx_pack_w8(ptr(Wf), ptr(codes), ptr(scales), ptr(status), N, K, current_stream(Wf.device)),
          "msq_mx_pack_w8")
    st = int(status.item())
    if st & 1:
        raise AssertionError("mx_pack_values: the values hold Inf / NaN")
    if (st & 2) and not allow_inexact:              # MSQ_STATUS_INEXACT
        raise MsqError("mx_pack_values: values not representable as e4m3 x 2^s per 32-block (MSQ_STATUS_INEXACT)")
    return MXPackedWeight(codes, scales, N, K, "e4m3", n, k)


def mx_pack_act(x, flush_fp32_subnorms=False, check_status=False, a_fmt="e4m3"):
    """x [..., K] -> MX-FP8 (e4m3) codes [M, K] + scales [M, K / 32].  ``a_fmt`` "e3m2" / "e2m3": the activations are
    quantised to MX-FP6 and the fp6 values travel as e4m3 codes with the fp6 block scale (exact)."""
    if not x.is_cuda:
        raise MsqError("mx_pack_act needs a CUDA/HIP tensor (no CPU fallback)")
    a_fmt = _mx_fmt(a_fmt)
    if a_fmt == "e2m1":
        raise MsqError("mx_pack_act: the activation operand is e4m3 (or fp6 values as e4m3)")
    K = x.shape[-1]
    bf = x.dtype in (torch.bfloat16, torch.float16) and a_fmt == "e4m3"   # read as is (every half value is an fp32 value: same codes)
    xf = x.reshape(-1, K).contiguous() if bf else x.reshape(-1, K).float().contiguous()
    if xf.data_ptr() % 16:                                # a contiguous view at an odd storage offset: the vector packers read 16-byte pieces
        xf = xf.clone()
    M = xf.shape[0]
    codes = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    scales = torch.empty(M, K // 32, dtype=torch.uint8, device=x.device)
    status = torch.zeros(1, dtype=torch.int32, device=x.device) if check_status else None
    if a_fmt != "e4m3":
        check(lib().msq_mx_pack_a6(ptr(xf), ptr(codes), ptr(scales), ptr(status), M, K, _FP6_IDS[a_fmt],
                                   int(bool(flush_fp32_subnorms)), current_stream(x.device)), "msq_mx_pack_a6")
        if check_status:
            _mx_status(status, "mx_pack_act")
        return codes, scales
    fn = (lib().msq_mx_pack_a8_f16 if x.dtype == torch.float16 else lib().msq_mx_pack_a8_bf16) if bf else lib().msq_mx_pack_a8
    check(fn(ptr(xf), ptr(codes), ptr(scales), ptr(status), M, K, int(bool(flush_fp32_subnorms)),
             current_stream(x.device)), "msq_mx_pack_a8")
    if check_status:
        _mx_status(status, "mx_pack_act")
    return codes, scales


def qlinear_mx_w4a8(x, P, bias=None, out_dtype=torch.bfloat16, check_status=False, out=None, a_fmt="e4m3"):
    """y = MXFP8(x) . MX(W)^T (+ bias) on v_mfma_scale_f32_16x16x128_f8f6f4: one pass packs the activations, the GEMM
    consumes codes and scale bytes directly.  The weight operand is whatever ``P`` holds (MX-FP4, MX-FP6 or exact e4m3
    values); ``a_fmt`` "e3m2" / "e2m3" quantises the activations to MX-FP6 instead of MX-FP8 (W6A6 of the fp6 spec)."""
    if isinstance(x, (tuple, list)):                     # activations already packed by mx_pack_act: q / k / v or
        xc, xs = x                                       # gate / up projections of one input share the pack
        K, lead, xdev = xc.shape[-1], tuple(xc.shape[:-1]), xc.device
        xc, xs = xc.reshape(-1, K), xs.reshape(-1, K // 32)
        if K != P.K:
            raise MsqError("qlinear_mx_w4a8: packed activations must cover the padded in_features (%d vs %d)" % (K, P.K))
    else:
        k, lead, xdev = x.shape[-1], tuple(x.shape[:-1]), x.device
        if k != P.k:
            raise MsqError("qlinear_mx_w4a8: in_features mismatch (%d vs %d)" % (k, P.k))
        K = P.K
        # zero columns up to the padded K: whole zero blocks, and a ragged last block padded as mx_ops.py:332-457 pads it
        xc, xs = mx_pack_act(_pad_k(x.reshape(-1, k), K), check_status=check_status, a_fmt=a_fmt)
    M = xc.shape[0]
    if out_dtype not in _YD:
        raise MsqError("qlinear_mx_w4a8: out_dtype must be float32, float16 or bfloat16")
    padded_n = P.n != P.N
    if out is not None and padded_n:
        _out_buffer(out, M, P.n, out_dtype, xdev, "qlinear_mx_w4a8")
    y = _out_buffer(None if padded_n else out, M, P.N, out_dtype, xdev, "qlinear_mx_w4a8")
    b = bias.detach().float().contiguous() if bias is not None else None
    if b is not None and padded_n:
        b = torch.nn.functional.pad(b, (0, P.N - P.n))

    def _ret(y):
        if padded_n:
            if out is not None:
                out.copy_(y[:, :P.n]); y = out
            else:
                y = y[:, :P.n].contiguous()
        return y.reshape(*lead, P.n)
    wsb = lib().msq_qlinear_mx_w4a8_workspace_bytes(M, P.N, K)     # > 0 only for small M (split-K partial tiles)
    ws = torch.empty(wsb, dtype=torch.uint8, device=xdev) if wsb > 0 else None
    yd = _YD[out_dtype]
    if P.w_fmt in _FP6_IDS:
        check(lib().msq_qlinear_mx_w6a8(ptr(xc), ptr(xs), ptr(P.codes), ptr(P.scales), ptr(b), ptr(y), yd, M, P.N, K,
                                        _FP6_IDS[P.w_fmt], ptr(ws), wsb, current_stream(xdev)), "msq_qlinear_mx_w6a8")
        return _ret(y)
    fn = lib().msq_qlinear_mx_w8a8 if P.w_fmt == "e4m3" else lib().msq_qlinear_mx_w4a8
    check(fn(ptr(xc), ptr(xs), ptr(P.codes), ptr(P.scales), ptr(b), ptr(y), yd,
             M, P.N, K, ptr(ws), wsb, current_stream(xdev)), "msq_qlinear_mx_w%sa8" % ("8" if P.w_fmt == "e4m3" else "4"))
    return _ret(y)


class MXLinearW4A8(nn.Module):
    """Linear whose weight lives as MX-FP4 codes (4.25 bits/weight) and whose forward quantises the activations
    to MX-FP8 and multiplies on the scaled MFMA (plain OCP-MX semantics: quantize_mx_op on both operands,
    number_system/mx/mx_ops.py:460-490, block 32 along in_features)."""

    def __init__(self, in_features, out_features, bias=True, out_dtype=torch.bfloat16, device=None, w_fmt="e2m1", a_fmt="e4m3"):
        super().__init__()
        w_fmt, a_fmt = _mx_fmt(w_fmt), _mx_fmt(a_fmt)
        self.in_features, self.out_features, self.out_dtype, self.w_fmt, self.a_fmt = in_features, out_features, out_dtype, w_fmt, a_fmt
        Np, Kp = _ceil_to(out_features, N_MULT), _ceil_to(in_features, K_MULT_MX)           # the planes cover the padded grid
        nb = Np * Kp * {"e2m1": 4, "e3m2": 6, "e2m3": 6, "e4m3": 8}[w_fmt] // 8
        self.register_buffer("w_codes", torch.zeros(nb, dtype=torch.uint8, device=device))
        self.register_buffer("w_scales", torch.zeros(Np * Kp // 32, dtype=torch.uint8, device=device))
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float32, device=device))
        else:
            self.bias = None

    @classmethod
    def from_linear(cls, linear, out_dtype=torch.bfloat16, w_fmt="e2m1", a_fmt="e4m3"):
        """``w_fmt`` "e3m2" / "e2m3": MX-FP6 weights as a 6-bit plane; ``a_fmt`` likewise quantises the activations to
        MX-FP6 (run_mx_fp6.sh's formats with plain OCP-MX quantisers)."""
        m = cls(linear.in_features, linear.out_features, linear.bias is not None, out_dtype, linear.weight.device, w_fmt, a_fmt)
        P = mx_pack_weight(linear.weight.data, w_fmt=m.w_fmt)
        m.w_codes.copy_(P.codes); m.w_scales.copy_(P.scales)
        if m.bias is not None:
            m.bias.copy_(linear.bias.data.float())
        return m

    @classmethod
    def from_values(cls, Wq, bias=None, out_dtype=torch.bfloat16):
        """Fake-quant values of any quantiser (e.g. quant.outlier_fakequant / GPTQ output): MicroScopiQ weights with
        their outliers, exact, on the scaled-MFMA path with MX-FP8 activations."""
        N, K = Wq.shape
        m = cls(K, N, bias is not None, out_dtype, Wq.device, w_fmt="e4m3")
        P = mx_pack_values(Wq)
        m.w_codes.copy_(P.codes); m.w_scales.copy_(P.scales)
        if bias is not None:
            m.bias.copy_(bias.detach().float())
        return m

    def _packed(self):
        return MXPackedWeight(self.w_codes, self.w_scales, _ceil_to(self.out_features, N_MULT), _ceil_to(self.in_features, K_MULT_MX),
                              self.w_fmt, self.out_features, self.in_features)

    def forward(self, x, out=None):
        # an fp16 / bf16 model gets its own dtype back, written by the kernel's epilogue (no cast pass); pre-packed activations
        # (a tuple of codes and scales) carry no dtype: those callers take out_dtype
        od = self.out_dtype
        if out is None and torch.is_tensor(x) and x.dtype in (torch.float16, torch.bfloat16):
            od = x.dtype
        return qlinear_mx_w4a8(x, self._packed(), self.bias, od, out=out, a_fmt=getattr(self, "a_fmt", "e4m3"))


class QuantLinear(nn.Module):
    """Packed Linear.  ``pack(linear, quantizer)`` consumes an nn.Linear and an MXQuantizer
    (the GPTQ-style contract implied by llm/opt.py:255-264); state_dict round-trips the
    packed planes (llm/opt.py:510-512, :290)."""

    def __init__(self, in_features, out_features, bias=True, block_size=32, inlier_elem_format="fp4_e2m1",
                 outlier_elem_format="fp8_e4m3", out_dtype=torch.bfloat16, device=None, layout="planes"):
        super().__init__()
        self.layout = layout
        self.in_features, self.out_features = in_features, out_features
        self.block_size = block_size
        self.inlier_elem_format, self.outlier_elem_format = inlier_elem_format, outlier_elem_format
        self.out_dtype = out_dtype
        ik, ok = packed_kinds(inlier_elem_format, outlier_elem_format, layout)
        ib, ob, sb, _ = packed_sizes(_ceil_to(out_features, N_MULT), _ceil_to(in_features, K_MULT), block_size, ik, ok)
        self.in_kind, self.out_kind = ik, ok
        self.register_buffer("inl_plane", torch.zeros(ib, dtype=torch.uint8, device=device))
        self.register_buffer("out_plane", torch.zeros(ob, dtype=torch.uint8, device=device))
        self.register_buffer("scale_plane", torch.zeros(sb, dtype=torch.uint8, device=device))
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float32, device=device))
        else:
            self.bias = None

    def _packed(self):
        return PackedWeight(self.inl_plane if self.inl_plane.numel() else None, self.out_plane,
                            self.scale_plane if self.scale_plane.numel() else None, _ceil_to(self.out_features, N_MULT),
                            _ceil_to(self.in_features, K_MULT), self.block_size, self.in_kind, self.out_kind,
                            self.out_features, self.in_features)

    def pack(self, linear, quantizer=None):
        q = quantizer
        kw = dict(inlier_elem_format=self.inlier_elem_format, outlier_elem_format=self.outlier_elem_format,
                  block_size=self.block_size)
        if q is not None:
            axes = q.axes if isinstance(q.axes, (list, tuple)) else [q.axes]
            if [a % 2 for a in axes] != [1]:
                raise MsqError("QuantLinear packs blocks along in_features (axes=[-1]); "
                               "use quantize_mx_outlier_v1 + a bf16 plane for axes=[0]")
            kw.update(inlier_scale_bits=q.inlier_scale_bits, outlier_scale_bits=q.outlier_scale_bits,
                      inlier_elem_format=q.inlier_elem_format, outlier_elem_format=q.outlier_elem_format,
                      std_dev=q.std_dev, block_size=q.block_size, round=q.round,
                      flush_fp32_subnorms=q.flush_fp32_subnorms)
        # float32 compute: the planes this module was sized for come from the fused quantise + pack launch.  (from_linear
        # routes half-precision weights through quantise-in-dtype + pack_values instead: QuantLinear.from_dense.)
        P = pack_weight(linear.weight.data, layout=self.layout, compute_dtype="float32", **kw)
        if (P.in_kind, P.out_kind, P.block) != (self.in_kind, self.out_kind, self.block_size):
            raise MsqError("quantizer formats do not match the formats this QuantLinear was built for")
        if P.inl is not None:
            self.inl_plane.copy_(P.inl)
        if P.scl is not None:
            self.scale_plane.copy_(P.scl)
        self.out_plane.copy_(P.out)
        if self.bias is not None and linear.bias is not None:
            self.bias.copy_(linear.bias.data.float())
        return self

    @classmethod
    def from_linear(cls, linear, quantizer=None, compute_dtype="input", **kw):
        """``compute_dtype="input"`` (default): an fp16 / bf16 Linear is quantised in its own dtype, op by op, exactly as
        ``quantizer.quantize(linear.weight)`` -- the reference's RTN harness on a half checkpoint (llm/llama.py:238) -- would,
        and those values are packed as they are; "float32" upcasts first (the fused quantise + pack launch)."""
        W = linear.weight.data
        posit = quantizer is not None and (str(quantizer.inlier_elem_format).startswith("posit") or str(quantizer.outlier_elem_format).startswith("posit"))
        if compute_dtype == "input" and quantizer is not None and W.dtype in (torch.float16, torch.bfloat16) and not posit:
            axes = quantizer.axes if isinstance(quantizer.axes, (list, tuple)) else [quantizer.axes]
            if [a % 2 for a in axes] == [1]:
                P = pack_weight(W, quantizer.inlier_scale_bits, quantizer.outlier_scale_bits, quantizer.inlier_elem_format,
                                quantizer.outlier_elem_format, quantizer.std_dev, quantizer.block_size, quantizer.round,
                                quantizer.flush_fp32_subnorms, layout=kw.get("layout", "auto"), compute_dtype="input")
                return cls.from_packed(P, linear.bias.data if linear.bias is not None else None, kw.get("out_dtype", torch.bfloat16))
        if quantizer is not None:
            kw.setdefault("block_size", quantizer.block_size)
            kw.setdefault("inlier_elem_format", quantizer.inlier_elem_format)
            kw.setdefault("outlier_elem_format", quantizer.outlier_elem_format)
        kw.setdefault("layout", "auto")
        if kw["layout"] == "auto":              # unified when the formats allow it and every group is exact
            q = quantizer
            if q is None or (q.round == "nearest" and q.block_size <= 64):
                try:
                    return cls.from_linear(linear, quantizer, **dict(kw, layout="unified"))
                except MsqError:
                    pass
            kw = dict(kw, layout="planes")
        m = cls(linear.in_features, linear.out_features, linear.bias is not None, device=linear.weight.device, **kw)
        return m.pack(linear, quantizer)

    @classmethod
    def empty_single_plane(cls, in_features, out_features, bias, out_kind, out_dtype=torch.bfloat16, device=None):
        """Unfilled module for a layer packed from dense values (checkpoint loading)."""
        m = cls.__new__(cls)
        nn.Module.__init__(m)
        m.in_features, m.out_features, m.block_size = in_features, out_features, 32
        m.inlier_elem_format = m.outlier_elem_format = "values"
        m.out_dtype = out_dtype
        m.layout = {PLANE_U8: "unified", PLANE_U8X: "unified", PLANE_BF16: "planes"}[out_kind]
        m.in_kind, m.out_kind = PLANE_NONE, out_kind
        ib, ob, sb, _ = packed_sizes(_ceil_to(out_features, N_MULT), _ceil_to(in_features, K_MULT), 32, PLANE_NONE, out_kind)
        m.register_buffer("inl_plane", torch.zeros(ib, dtype=torch.uint8, device=device))
        m.register_buffer("out_plane", torch.zeros(ob, dtype=torch.uint8, device=device))
        m.register_buffer("scale_plane", torch.zeros(sb, dtype=torch.uint8, device=device))
        if bias:
            m.register_buffer("bias", torch.zeros(out_features, dtype=torch.float32, device=device))
        else:
            m.bias = None
        return m

    @classmethod
    def from_dense(cls, linear, out_dtype=torch.bfloat16):
        """Pack a Linear whose weight ALREADY holds fake-quant values (RTN with any `axes`, or the GPTQ
        solver's output): the layer is stored in the smallest single-plane kind that is exact."""
        P = pack_values(linear.weight.data)
        m = cls.__new__(cls)
        nn.Module.__init__(m)
        m.in_features, m.out_features, m.block_size = linear.in_features, linear.out_features, 32
        m.inlier_elem_format = m.outlier_elem_format = "values"
        m.out_dtype = out_dtype
        m.layout = {PLANE_U8: "unified", PLANE_U8X: "unified", PLANE_BF16: "planes"}[P.out_kind]
        m.in_kind, m.out_kind = P.in_kind, P.out_kind
        dev = linear.weight.device
        m.register_buffer("inl_plane", P.inl if P.inl is not None else torch.zeros(0, dtype=torch.uint8, device=dev))
        m.register_buffer("out_plane", P.out)
        m.register_buffer("scale_plane", P.scl if P.scl is not None else torch.zeros(0, dtype=torch.uint8, device=dev))
        if linear.bias is not None:
            m.register_buffer("bias", linear.bias.data.float().clone())
        else:
            m.bias = None
        return m

    @classmethod
    def from_packed(cls, P, bias=None, out_dtype=torch.bfloat16):
        """Module around an existing PackedWeight (the planes are adopted, not copied)."""
        m = cls.__new__(cls)
        nn.Module.__init__(m)
        m.in_features, m.out_features, m.block_size = P.k, P.n, P.block
        # single-plane kinds are what pack_values / empty_single_plane produce: a checkpoint reloads them as such
        m.inlier_elem_format = m.outlier_elem_format = "values" if (P.in_kind == PLANE_NONE and P.block == 32) else "packed"
        m.out_dtype = out_dtype
        m.layout = "unified" if P.out_kind in (PLANE_U8, PLANE_U8X) else "planes"
        m.in_kind, m.out_kind = P.in_kind, P.out_kind
        dev = P.out.device
        m.register_buffer("inl_plane", P.inl if P.inl is not None else torch.zeros(0, dtype=torch.uint8, device=dev))
        m.register_buffer("out_plane", P.out)
        m.register_buffer("scale_plane", P.scl if P.scl is not None else torch.zeros(0, dtype=torch.uint8, device=dev))
        if bias is not None:
            m.register_buffer("bias", bias.detach().float().clone())
        else:
            m.bias = None
        return m

    def dequantize(self, dtype=torch.float32):
        return unpack_weight(self._packed(), dtype)

    def forward(self, x, out=None):
        # half-precision models get their own dtype from the kernel's epilogue (fp16 included: no cast pass over the output)
        od = self.out_dtype if (out is not None or x.dtype == torch.float32) else x.dtype
        return qlinear(x, self._packed(), self.bias, od, out=out)


class FusedProjections(nn.Module):
    """Several Linears of ONE input (q / k / v, gate / up) as a single packed GEMM: their weights are concatenated along
    out_features before packing.  Blocks run along in_features, so every row is quantised exactly as it would be alone: the
    values do not change, only the launch count (and at decode sizes the weight stream is read by one kernel instead of
    three).  The children of the model stay callable one by one (`ProjectionSlice`): the first call with a given input
    tensor runs the fused GEMM, the others return their column slice of the same result."""

    def __init__(self, qlinear_module, splits):
        super().__init__()
        self.proj = qlinear_module                        # QuantLinear or MXLinearW4A8 with out_features = sum(splits)
        self.splits = [int(v) for v in splits]
        self.offsets = [sum(self.splits[:i]) for i in range(len(self.splits) + 1)]
        self._x = self._y = None
        self._ver = -1
        self._left = 0
        self._last = -1

    def slice(self, i, x):
        # the cached result is valid for this very tensor object only (holding the reference keeps its storage from being
        # reused by another tensor) at the same version counter, and for one read per sibling
        # (inference tensors carry no version counter -- `x._version` raises under torch.inference_mode() -- and cannot be
        # written in place outside inference mode: identity alone is a sufficient key for them)
        #
        # RESTRICTION (advisor, round 4): an inference tensor CAN be written in place inside torch.inference_mode(), and nothing
        # on the host sees it -- `q_proj(x); x.mul_(s); k_proj(x)` under inference mode returns k from the old x.  Ordinary
        # tensors are covered by the version counter.  What is checked for both: every sibling is served at most once per
        # fused result and in ascending order (q, k, v / gate, up: the order every HF block calls them in); a repeated or
        # out-of-order call recomputes.  Do not modify the shared input in place between sibling calls under inference mode.
        ver = None if x.is_inference() else x._version
        if self._x is not x or self._ver != ver or self._left <= 0 or i <= self._last:
            self._y = self.proj(x)
            self._x, self._ver, self._left, self._last = x, ver, len(self.splits), -1
        y = self._y[..., self.offsets[i]:self.offsets[i + 1]]
        self._left -= 1
        self._last = i
        if self._left == 0:
            self._x = self._y = None
            self._last = -1
        return y


class ProjectionSlice(nn.Module):
    """Stand-in for one of the fused Linears (same in_features / out_features): forward(x) = columns of the fused result.
    The first slice owns the shared module as a submodule (its planes appear once in the state_dict, under
    ``<first child>.fused.proj.*``); the others reference it."""

    def __init__(self, fused, index, owner):
        super().__init__()
        if owner:
            self.fused = fused
        else:
            object.__setattr__(self, "_fused_ref", fused)
        self.index = index
        self.in_features = fused.proj.in_features
        self.out_features = fused.splits[index]

    def shared(self):
        return self.fused if "fused" in self._modules else self._fused_ref

    def forward(self, x):
        return self.shared().slice(self.index, x)


def fuse_projections(parent, names, quantizer=None, layout="auto", path="bf16"):
    """Replace the nn.Linear children ``names`` of ``parent`` (same in_features, same bias-ness) by ProjectionSlices of one
    packed module.  ``quantizer`` None = the weights already hold fake-quant values (pack_values / from_values)."""
    lins = [getattr(parent, n) for n in names]
    if not all(isinstance(l, nn.Linear) for l in lins):
        raise MsqError("fuse_projections: %r are not all nn.Linear" % (names,))
    if len({l.in_features for l in lins}) != 1 or len({l.bias is None for l in lins}) != 1 or len({l.weight.dtype for l in lins}) != 1:
        raise MsqError("fuse_projections: %r differ in in_features / bias / dtype" % (names,))
    W = torch.cat([l.weight.data for l in lins], 0)
    cat = nn.Linear(lins[0].in_features, W.shape[0], bias=lins[0].bias is not None, device="meta")
    cat.weight = nn.Parameter(W, requires_grad=False)
    if lins[0].bias is not None:
        cat.bias = nn.Parameter(torch.cat([l.bias.data for l in lins], 0), requires_grad=False)
    if path == "mx":
        od = W.dtype if W.dtype == torch.bfloat16 else torch.float32
        proj = MXLinearW4A8.from_values(W, cat.bias.data if cat.bias is not None else None, out_dtype=od) if quantizer is None \
            else MXLinearW4A8.from_linear(cat, out_dtype=od)
    else:
        proj = QuantLinear.from_dense(cat) if quantizer is None else QuantLinear.from_linear(cat, quantizer, layout=layout)
    fused = FusedProjections(proj, [l.out_features for l in lins])
    for i, n in enumerate(names):
        setattr(parent, n, ProjectionSlice(fused, i, owner=(i == 0)))
    return fused


def make_quant(module, quantizers, name='', layout="auto", fuse=None):
    """Swap every nn.Linear whose qualified name is in `quantizers` (name -> MXQuantizer) for a
    packed QuantLinear (the make_quant3 contract of llm/opt.py:258-264).  A value of None means "the weight
    already holds fake-quant values" (RTN along any axis / GPTQ): it is packed as it is (from_dense).

    ``fuse``: groups of sibling names, e.g. ``[("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj")]``: wherever a
    module has all the Linears of a group as children (all listed in `quantizers`, with equal quantiser settings and blocks
    along in_features, or all None) they are packed as ONE weight (fuse_projections): one GEMM launch instead of two or
    three, the same values."""
    children = dict(module.named_children())
    for group in (fuse or ()):
        if not all(g in children and isinstance(children[g], nn.Linear) for g in group):
            continue
        fulls = [(name + '.' + g if name != '' else g) for g in group]
        if not all(f in quantizers for f in fulls):
            continue
        qs = [quantizers[f] for f in fulls]
        if any(q is None for q in qs):
            if not all(q is None for q in qs):
                continue
            q0 = None
        else:
            q0 = qs[0]
            keys = ("inlier_scale_bits", "outlier_scale_bits", "inlier_elem_format", "outlier_elem_format", "std_dev",
                    "block_size", "round", "flush_fp32_subnorms")
            axes = q0.axes if isinstance(q0.axes, (list, tuple)) else [q0.axes]
            if [a % 2 for a in axes] != [1] or any(getattr(q, k_) != getattr(q0, k_) for q in qs for k_ in keys):
                continue
        try:
            fuse_projections(module, list(group), q0, layout)
        except MsqError:
            continue
        children = dict(module.named_children())
    for attr in list(children.keys()):
        child = getattr(module, attr)
        full = name + '.' + attr if name != '' else attr
        if isinstance(child, nn.Linear) and full in quantizers:
            if quantizers[full] is None:
                setattr(module, attr, QuantLinear.from_dense(child))
            else:
                setattr(module, attr, QuantLinear.from_linear(child, quantizers[full], layout=layout))
        elif not isinstance(child, (ProjectionSlice, QuantLinear, MXLinearW4A8)):
            make_quant(child, quantizers, full, layout, fuse)
    return module


class RowParallelQuantLinear(nn.Module):
    """K-split (row-parallel) packed Linear for the 70B configuration: rank r owns in_features
    [r K/G, (r+1) K/G) as a packed shard (``QuantLinear``, or ``MXLinearW4A8`` on the MX matrix path) and the ranks'
    partial products are summed over ``torch.distributed`` (backend "nccl" == RCCL over xGMI on MI355X).

    Communication.  xGMI is point-to-point, so a ring all-reduce is bound by one link (2 (G-1)/G S / 153 GB/s = 383 us
    for the 33.5 MB bf16 output of down_proj at M = 2048) while reduce-scatter + all-gather over all seven links move
    2 S / (G 153 GB/s) = 55 us -- still the same order as the shard's GEMM (~96 us).  The forward therefore
      * sums with ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` (``comm="rs_ag"``; ``"all_reduce"`` keeps the
        single collective, and is what non-NCCL backends such as gloo get), and
      * cuts the M rows into chunks: the GEMM of chunk i + 1 runs on the compute stream while RCCL's stream moves
        chunk i (async collectives; every chunk is a contiguous row slice of the one output tensor, so nothing is
        copied and the GEMM writes straight into the buffer the collective reduces in place).
    ``reduce_dtype`` is the wire / accumulation dtype of the sum (float32: exact fp32 sum in rank order up to RCCL's
    reduction tree; bfloat16 halves the bytes)."""

    def __init__(self, shard, world_size, rank, process_group=None, comm="rs_ag", chunks=0, reduce_dtype=torch.float32,
                 single_rank_collectives=False):
        super().__init__()
        if comm not in ("rs_ag", "all_reduce"):
            raise MsqError("RowParallelQuantLinear: comm must be 'rs_ag' or 'all_reduce'")
        self.shard = shard                    # packed Linear over the local K slice (bias only on rank 0)
        self.world_size, self.rank, self.process_group = world_size, rank, process_group
        self.comm, self.chunks, self.reduce_dtype = comm, int(chunks), reduce_dtype
        self.single_rank_collectives = bool(single_rank_collectives)   # run the collectives with one rank too (profiling on one GPU)
        self._parts = {}                      # reduce-scatter pieces, kept across calls (see forward)

    def _part(self, chunk_index, rows, N, dtype, device):
        """The reduce-scatter piece of chunk `chunk_index`: ONE buffer per chunk position (the chunks of a call are in flight together, so they
        cannot share one), grown to the largest piece seen and sliced -- a serving loop with varying M keeps as many buffers as it has chunks,
        not one per (r0, r1) it ever saw (advisor, round 5).  Assumes ONE communication stream per module instance: the buffer is written and read
        by the collectives of that stream only, in order."""
        key = (chunk_index, N, dtype, device)
        buf = self._parts.get(key)
        if buf is None or buf.shape[0] < rows:
            buf = self._parts[key] = torch.empty(rows, N, dtype=dtype, device=device)
        return buf[:rows]

    @staticmethod
    def shard_bounds(in_features, world_size, rank, block_size, multiple=64):
        """[k0, k1) of a rank; the split must fall on a tile multiple (64; 128 on the MX path) and a block multiple."""
        if in_features % world_size:
            raise MsqError("in_features must divide evenly over the ranks")
        per = in_features // world_size
        if per % multiple or per % block_size:
            raise MsqError("per-rank in_features (%d) must be a multiple of %d and of the block size" % (per, multiple))
        return rank * per, (rank + 1) * per

    @classmethod
    def from_linear(cls, linear, quantizer, world_size, rank, process_group=None, path="bf16", **kw):
        """Quantise and pack this rank's K slice of a dense Linear.  ``path="bf16"``: fused dequant-GEMM shard
        (QuantLinear); ``"mx"``: the fake-quant values as an exact e4m3 operand with MX-FP8 activations
        (MXLinearW4A8.from_values).  Blocks run along K, so every shard's masks and scales equal the unsharded ones."""
        k0, k1 = cls.shard_bounds(linear.in_features, world_size, rank, quantizer.block_size, 128 if path == "mx" else 64)
        has_bias = linear.bias is not None and rank == 0
        Wl = linear.weight.data[:, k0:k1].contiguous()
        kw.setdefault("reduce_dtype", torch.float32)
        if path == "mx":
            from .quant import outlier_fakequant
            axes = quantizer.axes if isinstance(quantizer.axes, (list, tuple)) else [quantizer.axes]
            if [a % 2 for a in axes] != [1]:
                raise MsqError("RowParallelQuantLinear: blocks must run along in_features (axes=[-1])")
            Wq = outlier_fakequant(Wl.float(), quantizer.inlier_scale_bits, quantizer.outlier_scale_bits,
                                   quantizer.inlier_elem_format, quantizer.outlier_elem_format, quantizer.std_dev, -1,
                                   quantizer.block_size, quantizer.round, quantizer.flush_fp32_subnorms)["out"]
            shard = MXLinearW4A8.from_values(Wq, linear.bias.data if has_bias else None, out_dtype=kw["reduce_dtype"])
        elif path == "bf16":
            local = nn.Linear(k1 - k0, linear.out_features, bias=has_bias, device=linear.weight.device, dtype=linear.weight.dtype)
            with torch.no_grad():
                local.weight.copy_(Wl)
                if has_bias:
                    local.bias.copy_(linear.bias)
            shard = QuantLinear.from_linear(local, quantizer, out_dtype=kw["reduce_dtype"])
        else:
            raise MsqError("RowParallelQuantLinear.from_linear: path must be 'bf16' or 'mx'")
        return cls(shard, world_size, rank, process_group, **kw)

    def chunks_for(self, M):
        """Row chunks the forward overlaps with the collective: the configured count, or one per 1024 rows up to four
        (70B down_proj shard [8192 x 3584] on one MI355X: 2048 rows 91 us, 2 x 1024 rows 103 us, 4 x 512 rows 164 us -- a
        512-row chunk is a split-K launch; with ~28 us of reduce-scatter + all-gather per 1024-row chunk two chunks win)."""
        if self.world_size == 1 and not self.single_rank_collectives:
            return 1
        n = self.chunks if self.chunks > 0 else max(1, min(4, M // 1024))
        return max(1, min(n, (M + 127) // 128))

    def chunk_bounds(self, M):
        """[(r0, r1)] row chunks of the forward (multiples of the GEMM's 128-row block tile)."""
        nch = self.chunks_for(M)
        rows = (((M + nch - 1) // nch) + 127) // 128 * 128
        return [(r0, min(M, r0 + rows)) for r0 in range(0, M, rows)]

    def comm_only(self, M, dtype=None, device=None, buf=None):
        """The collectives of one forward on an [M, N] buffer, without the GEMMs (bench.py: what the wire alone costs).  ``buf``: a caller-owned
        [M, N] buffer to run them on (zero-filled here), see forward's ``out``."""
        import torch.distributed as dist
        N, G, pg = self.shard.out_features, self.world_size, self.process_group
        dev = device if device is not None else next(self.shard.buffers()).device
        y = buf.zero_() if buf is not None else torch.zeros(M, N, dtype=dtype or self.reduce_dtype, device=dev)
        use_rs = self.comm == "rs_ag" and dist.get_backend(pg) == "nccl"
        pending = []
        for ci, (r0, r1) in enumerate(self.chunk_bounds(M)):
            yc = y[r0:r1]
            if use_rs and (r1 - r0) % G == 0:
                part = self._part(ci, (r1 - r0) // G, N, y.dtype, dev)
                pending.append(dist.reduce_scatter_tensor(part, yc, op=dist.ReduceOp.SUM, group=pg, async_op=True))
                pending.append(dist.all_gather_into_tensor(yc, part, group=pg, async_op=True))
            else:
                pending.append(dist.all_reduce(yc, op=dist.ReduceOp.SUM, group=pg, async_op=True))
        for h in pending:
            h.wait()
        return y

    def forward(self, x_local, gemm_events=None, out=None):
        """x_local: [..., K/G] (this rank's slice of the activations) -> the full sum [..., N] in ``reduce_dtype``.
        ``gemm_events``: a list that receives one (start, end) event pair per GEMM chunk, recorded on the compute stream
        (bench.py: GEMM time against the step time = the communication that is NOT hidden).
        ``out``: a caller-owned [M, N] buffer of ``reduce_dtype`` for the result.  A serving loop should pass one: every tensor an async
        collective touches is held back by the caching allocator until the communication stream's event has been SEEN complete, so a host
        that runs ahead of the GPU gets a fresh hipMalloc (an implicit device synchronisation, tens of ms) every few calls instead of the
        block it just freed.  The reduce-scatter pieces are kept by the module for the same reason."""
        import torch.distributed as dist
        lead = tuple(x_local.shape[:-1])
        x2 = x_local.reshape(-1, x_local.shape[-1])
        M, N, G = x2.shape[0], self.shard.out_features, self.world_size
        if out is not None:
            if tuple(out.shape) != (M, N) or out.dtype != self.reduce_dtype or out.device != x2.device or not out.is_contiguous():
                raise MsqError("RowParallelQuantLinear: out must be a contiguous [%d, %d] %s tensor on %s" % (M, N, self.reduce_dtype, x2.device))
            y = out
        else:
            y = torch.empty(M, N, dtype=self.reduce_dtype, device=x2.device)
        direct = isinstance(self.shard, (QuantLinear, MXLinearW4A8)) and getattr(self.shard, "out_dtype", None) == self.reduce_dtype

        def partial(r0, r1):                     # the shard's GEMM on rows [r0, r1), written into y[r0:r1]
            if gemm_events is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if direct:
                self.shard(x2[r0:r1], out=y[r0:r1])
            else:
                y[r0:r1].copy_(self.shard(x2[r0:r1]))
            if gemm_events is not None:
                e1.record()
                gemm_events.append((e0, e1))

        if G == 1 and not self.single_rank_collectives:
            partial(0, M)
            return y.reshape(*lead, N)       # (y is `out` when one was given)
        pg = self.process_group
        use_rs = self.comm == "rs_ag" and dist.get_backend(pg) == "nccl"
        pending = []
        for ci, (r0, r1) in enumerate(self.chunk_bounds(M)):
            partial(r0, r1)
            yc = y[r0:r1]
            if use_rs and (r1 - r0) % G == 0:
                # (kept across calls: written and read on the communication stream only, in order; the caller's stream waits below)
                part = self._part(ci, (r1 - r0) // G, N, self.reduce_dtype, y.device)
                pending.append(dist.reduce_scatter_tensor(part, yc, op=dist.ReduceOp.SUM, group=pg, async_op=True))
                pending.append(dist.all_gather_into_tensor(yc, part, group=pg, async_op=True))
            else:
                pending.append(dist.all_reduce(yc, op=dist.ReduceOp.SUM, group=pg, async_op=True))
        for h in pending:
            h.wait()
        return y.reshape(*lead, N)
