"""MX matmul / bmm -- the surface of number_system/mx/matmul.py:194 `matmul` (MatMulFunction.forward :30-94) and
number_system/mx/bmm.py:127 `bmm` (BMMFunction.forward :19-63): both operands are rounded to the vector format,
MX-quantised along the DOT-PRODUCT axis (last axis of in1, second-to-last of in2 -- the strided `k_mx_tile_*`
kernel, no transpose copy), multiplied in fp32, and the product (and the bias sum) rounded again.  The reference
reaches these from attention score / context products; the Linear path (linear.py) does not call them.  Forward
only, like the rest of the hot path; the quantisers are the HIP kernels (no CPU fallback)."""
import torch

from ._lib import MsqError
from .elemwise_ops import quantize_elemwise_op
from .mx_ops import quantize_mx_op
from .specs import apply_mx_specs, mx_assert_test

torch_matmul = torch.matmul
torch_addmm = torch.addmm
torch_bmm = torch.bmm


def _need_gpu(who, *ts):
    for t in ts:
        if t is not None and not (torch.is_tensor(t) and t.is_cuda):
            raise MsqError("%s needs CUDA/HIP tensors (no CPU fallback)" % who)


def _mx_product(in1, in2, fmt1, fmt2, mx_specs, mm):
    bf_in1 = quantize_elemwise_op(in1, mx_specs=mx_specs, round=mx_specs["round_output"])
    bf_in2 = quantize_elemwise_op(in2, mx_specs=mx_specs, round=mx_specs["round_output"])
    qin1 = quantize_mx_op(bf_in1, mx_specs, elem_format=fmt1, axes=[-1], round=mx_specs["round_mx_output"])
    qin2 = quantize_mx_op(bf_in2, mx_specs, elem_format=fmt2, axes=[-2], round=mx_specs["round_mx_output"])
    return quantize_elemwise_op(mm(qin1, qin2), mx_specs=mx_specs, round=mx_specs["round_output"])


def matmul(in1, in2, bias=None, mx_specs=None, name=None, mode_config='aa'):
    """matmul.py:194-205.  in1 (..., rows, features) x in2 (..., features, cols) or (features, cols);
    mode_config picks the element format of each operand: 'a' -> a_elem_format, 'w' -> w_elem_format (:31-42)."""
    mx_assert_test(mx_specs)
    if mx_specs is None:
        return torch_matmul(in1, in2) if bias is None else torch_addmm(bias, in1, in2)
    mx_specs = apply_mx_specs(mx_specs)
    assert mode_config in ["aa", "aw", "wa"]
    _need_gpu("matmul", in1, in2, bias)
    fmt = {"a": mx_specs["a_elem_format"], "w": mx_specs["w_elem_format"]}
    with torch.no_grad():
        out = _mx_product(in1, in2, fmt[mode_config[0]], fmt[mode_config[1]], mx_specs, torch_matmul)
        if bias is not None:
            bf_bias = quantize_elemwise_op(bias, mx_specs=mx_specs, round=mx_specs["round_weight"])
            out = quantize_elemwise_op(out + bf_bias, mx_specs=mx_specs, round=mx_specs["round_output"])
    return out


def bmm(in1, in2, mx_specs=None, name=None):
    """bmm.py:127-134.  Any number of outer dims (bmm.py:21-26 says so; torch.bmm itself takes exactly one, and the
    reference calls torch.bmm, so more than 3 dims raise there as here); both operands use a_elem_format (:38-51)."""
    mx_assert_test(mx_specs)
    if mx_specs is None:
        return torch_bmm(in1, in2)
    mx_specs = apply_mx_specs(mx_specs)
    _need_gpu("bmm", in1, in2)
    with torch.no_grad():
        return _mx_product(in1, in2, mx_specs["a_elem_format"], mx_specs["a_elem_format"], mx_specs, torch_bmm)
