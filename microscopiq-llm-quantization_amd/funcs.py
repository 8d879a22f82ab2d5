"""Drop-in for the reference's pybind module ``custom_extensions.funcs``
(number_system/mx/cpp/funcs.cpp:218-226): the same seven function names and
argument order, implemented by libmsq_hip.so through the C ABI (include/msq.h).

Tensors must live on the GPU and be contiguous (cpp/funcs.h:11-13); outputs are
allocated here with ``torch.empty_like`` exactly as the reference wrapper does
(cpp/mx.cu:26) -- the native side never allocates.  There is NO CPU path: the
``*_cpp`` names exist for API compatibility and raise for CPU tensors.
"""
import torch

from . import _lib
from ._lib import MsqError, check, current_stream, lib, ptr


def _check_input(t, name):
    if not isinstance(t, torch.Tensor):
        raise MsqError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise MsqError(f"{name} must be a CUDA/HIP tensor (the MI355X library has no CPU fallback)")
    if not t.is_contiguous():
        raise MsqError(f"{name} must be contiguous")


def _pap(shape, axis):
    nd = len(shape)
    if not (0 <= axis < nd):
        raise MsqError("shared axis < 0 or shared axis >= ndim")       # funcs.cpp:38-39
    pre = 1
    for s in shape[:axis]:
        pre *= int(s)
    post = 1
    for s in shape[axis + 1:]:
        post *= int(s)
    return pre, int(shape[axis]), post


def quantize_elemwise_func_cuda(A, bits, exp_bits, max_norm, rmode=0, saturate_normals=False, allow_denorm=True):
    """cpp/funcs.cpp:183-200"""
    _check_input(A, "A")
    did = _lib.DTYPE_ID.get(str(A.dtype))
    if did is None:
        raise MsqError(f"unsupported dtype {A.dtype}")
    out = torch.empty_like(A)
    check(lib().msq_quantize_elemwise(ptr(A), ptr(out), A.numel(), did, int(bits), int(exp_bits), float(max_norm),
                                      int(rmode), int(bool(saturate_normals)), int(bool(allow_denorm)),
                                      current_stream(A.device)), "quantize_elemwise_func_cuda")
    return out


def quantize_mx_func_cuda(A, scale_bits, ebits, mbits, max_norm, max_values, axis, flush_fp32_subnorms=False, rmode=0):
    """cpp/funcs.cpp:138-159"""
    _check_input(A, "A")
    _check_input(max_values, "max_values")
    if A.dtype != torch.float32:
        raise MsqError("quantize_mx_func_cuda: only float32 is supported (cpp/mx.cu:51-52)")
    pre, axis_len, post = _pap(A.shape, int(axis))
    if max_values.numel() != pre * post:
        raise MsqError("max_values must hold one value per (pre, post) position")
    out = torch.empty_like(A)
    check(lib().msq_quantize_mx(ptr(A), ptr(out), ptr(max_values.float()), pre, axis_len, post, int(scale_bits),
                                int(ebits), int(mbits), float(max_norm), int(bool(flush_fp32_subnorms)), int(rmode),
                                current_stream(A.device)), "quantize_mx_func_cuda")
    return out


def quantize_mx_by_tile_func_cuda(A, scale_bits, ebits, mbits, max_norm, tile_size, axis, flush_fp32_subnorms=False,
                                  rmode=0, python_divisor=False, python_exponent=False):
    """cpp/funcs.cpp:161-181.  python_divisor: divide by `2**e + 1e-6` like the reference's PYTHON path (mx_ops.py:444)
    instead of by the scale like its native kernel (cpp/mx.cuh:132).  python_exponent: the shared exponent is
    floor(torch.log2(max)) in fp32 like the Python path (mx_ops.py:66-77) instead of the exponent field of the maximum like the
    native kernel (cpp/shared_exp.cuh:14-53) -- they differ for the 88 largest floats under a power of two."""
    _check_input(A, "A")
    if A.dtype != torch.float32:
        raise MsqError("quantize_mx_by_tile_func_cuda: only float32 is supported (cpp/mx.cu:124-125)")
    pre, axis_len, post = _pap(A.shape, int(axis))
    out = torch.empty_like(A)
    check(lib().msq_quantize_mx_by_tile_ex(ptr(A), ptr(out), pre, axis_len, post, int(tile_size), int(scale_bits), int(ebits), int(mbits),
                                           float(max_norm), int(bool(flush_fp32_subnorms)), int(rmode), int(bool(python_divisor)),
                                           int(bool(python_exponent)), current_stream(A.device)), "quantize_mx_by_tile_func_cuda")
    return out


def _reduce(A, fn, name):
    _check_input(A, "A")
    if A.dtype != torch.float32:
        raise MsqError(f"{name}: only float32 is supported")
    inner = int(A.shape[-1])
    outer = A.numel() // max(inner, 1)
    out = torch.empty(A.shape[:-1], dtype=A.dtype, device=A.device)
    check(fn(ptr(A), ptr(out), outer, inner, current_stream(A.device)), name)
    return out


def reduce_sum_inner_dim(A):
    """cpp/funcs.cpp:203-208"""
    return _reduce(A, lib().msq_reduce_sum_inner, "reduce_sum_inner_dim")


def reduce_max_inner_dim(A):
    """cpp/funcs.cpp:210-215"""
    return _reduce(A, lib().msq_reduce_max_inner, "reduce_max_inner_dim")


def quantize_elemwise_func_cpp(A, bits, exp_bits, max_norm, rmode=0, saturate_normals=False, allow_denorm=True):
    """cpp/funcs.cpp:98-133 ran on the host; this build has no CPU path."""
    return quantize_elemwise_func_cuda(A, bits, exp_bits, max_norm, rmode, saturate_normals, allow_denorm)


def quantize_mx_func_cpp(A, scale_bits, ebits, mbits, max_norm, max_values, axis, flush_fp32_subnorms=False, rmode=0):
    """cpp/funcs.cpp:26-96 ran on the host; this build has no CPU path."""
    return quantize_mx_func_cuda(A, scale_bits, ebits, mbits, max_norm, max_values, axis, flush_fp32_subnorms, rmode)


def quantize_format(A, fmt_id, rmode=0):
    """NEW: round to a named format id (posit included), saturating, denorms kept."""
    _check_input(A, "A")
    x = A.float()
    out = torch.empty_like(x)
    check(lib().msq_quantize_format(ptr(x), ptr(out), x.numel(), int(fmt_id), int(rmode), current_stream(A.device)),
          "quantize_format")
    return out.to(A.dtype)
