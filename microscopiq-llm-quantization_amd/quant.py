"""MicroScopiQ quantisers -- the surface of the reference's utils/quant.py
(quantize_mx_outlier_v1 :147, quantize_mx_outlier_hessian :23, quantize :268,
Quantizer :274, MXQuantizer :393 and the helpers :460-616), executed on the MI355X by
libmsq_hip.so.  Same names, positional order, attribute names and error behaviour,
so the reference harness (llm/llama.py:226-253, llm/gptq.py:130-143) runs unchanged.

All block statistics, masks, shared exponents and element codecs run inside ONE fused
HIP kernel (csrc/msq_quant.hip k_outlier_*); there is no CPU path.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import MsqError, check, current_stream, lib, ptr
from .formats import (ElemFormat, FP32_EXPONENT_BIAS, FP32_MIN_NORMAL, RoundingMode,  # noqa: F401
                      _get_format_params, format_id)
from .elemwise_ops import _quantize_elemwise_core  # noqa: F401  (re-exported like utils/quant.py:15-19)
from .specs import finalize_mx_specs

# The reference asserts on NaNs after every stage (utils/quant.py:225-250).  The kernels
# OR a flag instead; reading it back costs one stream sync per call.
CHECK_NAN = os.environ.get("MSQ_CHECK_NAN", "1") != "0"

VARIANT_QUANT = 0
VARIANT_MXOPS = 1
_SUPPORTED_BLOCKS = (8, 16, 32, 64, 128)


def _norm_axes(axes, ndim):
    axes = [axes] if type(axes) == int else axes
    if axes is None:
        raise Exception("axes required in order to determine which dimension toapply block size to")
    axes = [x + ndim if x < 0 else x for x in axes]
    if len(axes) != 1:
        raise MsqError("the MI355X kernels quantise along exactly one axis (got %r)" % (axes,))
    return axes


def _pap(shape, axis):
    pre = 1
    for s in shape[:axis]:
        pre *= int(s)
    post = 1
    for s in shape[axis + 1:]:
        post *= int(s)
    return pre, int(shape[axis]), post


def outlier_fakequant(A, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                      std_dev=2, axis=0, block_size=0, round="nearest", flush_fp32_subnorms=False,
                      variant=VARIANT_QUANT, want_mask=False, want_exps=False, want_num_outliers=False,
                      compute_dtype="input"):
    """One fused launch of the MicroScopiQ fake-quant.  Returns a dict with 'out' and the
    requested side outputs.

    fp16 / bf16 tensors: ``compute_dtype="input"`` (default) computes IN the tensor dtype, every op rounded back as
    ATen's CPU half kernels do -- what the reference does when the RTN harness quantises an fp16 checkpoint
    (llm/llama.py:238) -- and is bit-exact against the reference on half tensors (utils/quant.py variant, float /
    int element formats).  ``compute_dtype="float32"`` upcasts, computes in fp32 and rounds once at the end (more
    accurate, not what the reference does); posit formats, the mx_ops variant and num_outliers always take it."""
    if not isinstance(A, torch.Tensor) or not A.is_cuda:
        raise MsqError("input must be a CUDA/HIP tensor (the MI355X library has no CPU fallback)")
    if round not in RoundingMode.string_enums():
        raise Exception("Unrecognized round method %s" % (round))
    assert (inlier_scale_bits > 0 and outlier_scale_bits > 0)            # utils/quant.py:168
    orig_dtype = A.dtype
    x = A.contiguous()
    if compute_dtype not in ("input", "float32"):
        raise MsqError("compute_dtype must be 'input' or 'float32'")
    native_half = (compute_dtype == "input" and x.dtype in (torch.float16, torch.bfloat16) and variant == VARIANT_QUANT
                   and not want_num_outliers and not str(inlier_elem_format).startswith("posit")
                   and not str(outlier_elem_format).startswith("posit"))
    # bf16 / fp16 tensors computed in fp32 go through natively (read as fp32 values, one RNE rounding on the way out: the same
    # result as the upcast / downcast shim) where the kernels are built for it: round-to-nearest with float / int inliers,
    # quant.py variant; everything else (other rounding modes, posit inliers) is upcast
    native_bf16 = (x.dtype in (torch.bfloat16, torch.float16) and not native_half and round == "nearest" and variant == VARIANT_QUANT
                   and not str(inlier_elem_format).startswith("posit"))
    if x.dtype != torch.float32 and not native_bf16 and not native_half:
        x = x.float()
    axis = axis % x.ndim
    pre, axis_len, post = _pap(x.shape, axis)
    blk = int(block_size) if block_size and block_size > 0 else axis_len
    if blk not in _SUPPORTED_BLOCKS:
        raise MsqError("block size %d not supported by the HIP kernels (supported: %s)" % (blk, _SUPPORTED_BLOCKS))
    nblk = (axis_len + blk - 1) // blk
    dev = x.device
    out = torch.empty_like(x)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=dev) if want_mask else None
    e_in = torch.empty((pre, nblk, post), dtype=torch.float32, device=dev) if want_exps else None
    e_out = torch.empty((pre, nblk, post), dtype=torch.float32, device=dev) if want_exps else None
    n_out = None
    if want_num_outliers:
        if pre != 1:
            raise MsqError("num_outliers needs the blocked axis to be the first one")
        n_out = torch.zeros((((nblk + blk - 1) // blk) * post,), dtype=torch.int8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev) if CHECK_NAN else None
    L = lib()
    if native_half:
        dtype_code = 0x11 if x.dtype == torch.float16 else 0x12      # MSQ_DTYPE_F16_NATIVE / MSQ_DTYPE_BF16_NATIVE
    else:
        dtype_code = (1 if x.dtype == torch.float16 else 2) if native_bf16 else 0
    wsb = L.msq_outlier_workspace_bytes(pre, axis_len, post, blk, variant)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb > 0 else None
    check(L.msq_outlier_fakequant(ptr(x), ptr(out), ptr(mask), ptr(e_in), ptr(e_out), ptr(n_out), ptr(status),
                                  ptr(ws), wsb, dtype_code, pre, axis_len, post, blk, format_id(inlier_elem_format),
                                  format_id(outlier_elem_format), int(inlier_scale_bits), int(outlier_scale_bits),
                                  float(std_dev), int(RoundingMode[round]), int(bool(flush_fp32_subnorms)),
                                  int(variant), current_stream(dev)), "msq_outlier_fakequant")
    if CHECK_NAN and int(status.item()) & 1:
        # utils/quant.py:225-250 / mx_ops.py:66: a shared scale overflowed to NaN
        raise AssertionError("outlier_val / inlier_val / shared_exp contains NaN values")
    r = {"out": out if out.dtype == orig_dtype else out.to(orig_dtype)}
    if want_mask:
        r["mask"] = mask
    if want_exps:
        r["e_in"], r["e_out"] = e_in, e_out
    if want_num_outliers:
        r["num_outliers"] = n_out
    return r


def quantize_mx_outlier_v1(A, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                           shared_exp_method="max", std_dev=2, axes=None, block_size=0, round="nearest",
                           flush_fp32_subnorms=False, custom_cuda=False):
    """utils/quant.py:147-266 -- MX* outlier quantisation (fake-quant), same positional order."""
    if inlier_elem_format == None:
        return A
    if shared_exp_method != "max":
        raise Exception("Unrecognized shared exponent selection method %s" % (shared_exp_method))
    axes = _norm_axes(axes, A.ndim)
    return outlier_fakequant(A, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                             std_dev, axes[0], block_size, round, flush_fp32_subnorms)["out"]


def quantize_mx_outlier_hessian(A, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                                shared_exp_method="max", std_dev=2, axes=None, block_size=0, round="nearest",
                                flush_fp32_subnorms=False, prune_inliers=False, custom_cuda=False):
    """utils/quant.py:23-146 -- same maths, additionally returns num_outliers (:66, :146)."""
    if inlier_elem_format == None:
        return A
    if shared_exp_method != "max":
        raise Exception("Unrecognized shared exponent selection method %s" % (shared_exp_method))
    axes = _norm_axes(axes, A.ndim)
    r = outlier_fakequant(A, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                          std_dev, axes[0], block_size, round, flush_fp32_subnorms, want_num_outliers=True)
    return r["out"], r["num_outliers"]


# -------------------------------------------------------------------------
# legacy GPTQ uniform quantiser (utils/quant.py:268-390) -- elementwise torch ops on
# whatever device the tensors live on; not part of the MicroScopiQ hot path.
# -------------------------------------------------------------------------
def quantize(x, scale, zero, maxq):
    if maxq < 0:
        return (x > scale / 2).float() * scale + (x < zero / 2).float() * zero
    q = torch.clamp(torch.round(x / scale) + zero, 0, maxq)
    return scale * (q - zero)


class Quantizer(nn.Module):
    """utils/quant.py:274-390: asymmetric / symmetric min-max uniform quantiser."""

    def __init__(self, shape=1):
        super().__init__()
        self.register_buffer('maxq', torch.tensor(0))
        self.register_buffer('scale', torch.zeros(shape))
        self.register_buffer('zero', torch.zeros(shape))

    def configure(self, bits, perchannel=False, sym=True, mse=False, norm=2.4, grid=100, maxshrink=.8, trits=False):
        self.maxq = torch.tensor(2 ** bits - 1)
        self.perchannel, self.sym, self.mse = perchannel, sym, mse
        self.norm, self.grid, self.maxshrink = norm, grid, maxshrink
        if trits:
            self.maxq = torch.tensor(-1)

    def _rows(self, x, weight):
        shape = x.shape
        if not self.perchannel:
            return x.flatten().unsqueeze(0)
        if weight:
            return x.flatten(1)
        if len(shape) == 4:
            return x.permute([1, 0, 2, 3]).flatten(1)
        if len(shape) == 3:
            return x.reshape((-1, shape[-1])).t()
        return x.t()

    def find_params(self, x, weight=False):
        dev = x.device
        self.maxq = self.maxq.to(dev)
        shape = x.shape
        x = self._rows(x, weight)
        zeros = torch.zeros(x.shape[0], device=dev)
        xmin = torch.minimum(x.min(1)[0], zeros)
        xmax = torch.maximum(x.max(1)[0], zeros)
        if self.sym:
            xmax = torch.maximum(torch.abs(xmin), xmax)
            neg = xmin < 0
            if torch.any(neg):
                xmin[neg] = -xmax[neg]
        dead = (xmin == 0) & (xmax == 0)
        xmin[dead] = -1
        xmax[dead] = +1
        if self.maxq < 0:
            self.scale, self.zero = xmax, xmin
        else:
            self.scale = (xmax - xmin) / self.maxq
            self.zero = (torch.full_like(self.scale, (self.maxq + 1) / 2) if self.sym
                         else torch.round(-xmin / self.scale))
        if self.mse:
            best = torch.full([x.shape[0]], float('inf'), device=dev)
            for i in range(int(self.maxshrink * self.grid)):
                p = 1 - i / self.grid
                xmin1, xmax1 = p * xmin, p * xmax
                scale1 = (xmax1 - xmin1) / self.maxq
                zero1 = torch.round(-xmin1 / scale1) if not self.sym else self.zero
                q = quantize(x, scale1.unsqueeze(1), zero1.unsqueeze(1), self.maxq)
                err = torch.sum((q - x).abs_().pow_(self.norm), 1)
                better = err < best
                if torch.any(better):
                    best[better] = err[better]
                    self.scale[better] = scale1[better]
                    self.zero[better] = zero1[better]
        if not self.perchannel:
            reps = shape[0] if weight else (shape[1] if len(shape) != 3 else shape[2])
            self.scale = self.scale.repeat(reps)
            self.zero = self.zero.repeat(reps)
        if weight:
            view = [-1] + [1] * (len(shape) - 1)
        elif len(shape) == 4:
            view = (1, -1, 1, 1)
        elif len(shape) == 3:
            view = (1, 1, -1)
        else:
            view = (1, -1)
        self.scale = self.scale.reshape(view)
        self.zero = self.zero.reshape(view)

    def quantize(self, x):
        if self.ready():
            return quantize(x, self.scale, self.zero, self.maxq)
        return x

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return torch.all(self.scale != 0)


class MXQuantizer(nn.Module):
    """utils/quant.py:393-454: configuration holder read attribute-by-attribute by the
    harness (llm/llama.py:242-252) and the GPTQ solver (llm/gptq.py:132-142)."""

    def __init__(self, shape=1):
        super().__init__()
        self.mx_specs = finalize_mx_specs({'w_elem_format': 'int2', 'a_elem_format': 'fp16', 'block_size': 128,
                                           'custom_cuda': False, 'quantize_backprop': False})

    def configure(self, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                  shared_exp_method="max", std_dev=2, axes=None, block_size=0, round="nearest",
                  flush_fp32_subnorms=False, custom_cuda=False):
        self.inlier_scale_bits = inlier_scale_bits
        self.inlier_elem_format = inlier_elem_format
        self.outlier_scale_bits = outlier_scale_bits
        self.outlier_elem_format = outlier_elem_format
        self.shared_exp_method = shared_exp_method
        self.std_dev = std_dev
        self.axes = axes
        self.block_size = block_size
        self.round = round
        self.flush_fp32_subnorms = flush_fp32_subnorms
        self.custom_cuda = custom_cuda

    def find_params(self, x, weight=False):
        pass

    def quantize(self, x):
        if self.ready():
            return quantize_mx_outlier_v1(x, self.inlier_scale_bits, self.outlier_scale_bits,
                                          self.inlier_elem_format, self.outlier_elem_format, self.shared_exp_method,
                                          self.std_dev, self.axes, self.block_size, self.round,
                                          self.flush_fp32_subnorms, self.custom_cuda)
        return x

    def enabled(self):
        pass

    def ready(self):
        return True


# -------------------------------------------------------------------------
# helpers with the reference's names (utils/quant.py:460-616)
# -------------------------------------------------------------------------
def _extract_outlier_indices(A, std_dev=1, axes=None):
    """utils/quant.py:460-495 on an already blocked tensor: 0/1 mask in A.dtype."""
    if axes is None:
        raise MsqError("whole-tensor statistics (axes=None) are not on the MicroScopiQ path")
    axes = [a % A.ndim for a in ([axes] if type(axes) == int else axes)]
    if len(axes) != 1:
        raise MsqError("exactly one statistics axis is supported")
    # the blocked axis is one whole block: any element format works for the mask
    r = outlier_fakequant(A, 8, 8, "fp8_e4m3", "fp8_e4m3", std_dev, axes[0], int(A.shape[axes[0]]),
                          want_mask=True)
    return r["mask"].to(A.dtype)


def _shared_exponents(A, method="max", axes=None, ebits=0):
    """utils/quant.py:498-541: floor(log2(max|A|)) (exact exponent, as cpp/mx.cuh:81-85)."""
    if method == "max":
        if axes is None:
            shared = torch.max(torch.abs(A))
        else:
            shared = A
            for axis in axes:
                shared, _ = torch.max(torch.abs(shared), dim=axis, keepdim=True)
    elif method == "none":
        shared = torch.abs(A)
    else:
        raise Exception("Unrecognized shared exponent selection method %s" % (method))
    shared = shared.float()
    shared = shared + FP32_MIN_NORMAL * (shared == 0).float()
    exp = (torch.frexp(shared)[1] - 1).to(A.dtype)
    if ebits > 0:
        emax = 2 ** (ebits - 1) - 1
        exp[exp > emax] = float("NaN")
        exp[exp < -emax] = -emax
    return exp


def _reshape_to_blocks(A, axes, block_size):
    """utils/quant.py:544-603: insert a tile dimension behind every blocked axis, zero-pad
    the axis to a multiple of block_size.  Returns (A, axes, orig_shape, padded_shape)."""
    if axes is None:
        raise Exception("axes required in order to determine which dimension toapply block size to")
    if block_size == 0:
        raise Exception("block_size == 0 in _reshape_to_blocks")
    axes = sorted((x + A.ndim if x < 0 else x) for x in axes)
    for i in range(len(axes)):
        axes[i] += i
        A = A.unsqueeze(axes[i] + 1)
    orig_shape = A.size()
    pad_right = {}
    for ax in axes:
        rem = orig_shape[ax] % block_size
        if rem:
            pad_right[ax] = block_size - rem
    if pad_right:
        spec = []
        for d in reversed(range(A.ndim)):
            spec += [0, pad_right.get(d, 0)]
        A = torch.nn.functional.pad(A, spec, mode="constant")
    padded_shape = A.size()
    view = list(padded_shape)
    for ax in axes:
        if view[ax] >= block_size:
            assert view[ax] % block_size == 0
            view[ax + 1] = block_size
            view[ax] = view[ax] // block_size
        else:
            view[ax + 1] = view[ax]
            view[ax] = 1
    return A.view(view), axes, orig_shape, padded_shape


def _undo_reshape_to_blocks(A, padded_shape, orig_shape, axes):
    """utils/quant.py:606-616"""
    A = A.view(padded_shape)
    if list(padded_shape) != list(orig_shape):
        A = A[tuple(slice(0, x) for x in orig_shape)]
    for ax in reversed(axes):
        A = torch.squeeze(A, dim=ax + 1)
    return A
