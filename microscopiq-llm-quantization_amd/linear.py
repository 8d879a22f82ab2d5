"""Mixed-precision Linear -- the surface of number_system/mx/linear.py (linear :204, MXLinear
:220, LinearFunction.forward :20-108): bfloat/fp elementwise cast of input, weight and bias,
MicroScopiQ outlier quantisation of activations and weights (the mx_ops variant, axes=[1],
inlier format == outlier format), dense GEMM, cast, + bias, cast.

Every quantisation step is one HIP launch; the GEMM runs in fp32 on the GPU exactly like the
reference's F.linear (linear.py:91).  Forward only: the reference's backward is broken as
shipped (linear.py:129-136 passes unknown kwargs).

``MXLinear.pack()`` (new) freezes the weight into the packed tile-major planes once and turns the
forward into the W4A8 hot path: activation quantisation (one pass, bf16 out) + fused
dequant-GEMM (msq_qlinear_w4a8) instead of re-quantising the weight on every call."""
import torch
import torch.nn.functional as F

from .elemwise_ops import quantize_elemwise_op
from .mx_ops import quantize_mx_outlier_op
from .specs import apply_mx_specs, mx_assert_test


def _forward(input, weight, bias, mx_specs):
    bf_in = quantize_elemwise_op(input, mx_specs=mx_specs, round=mx_specs["round_output"])        # linear.py:29
    bf_weight = quantize_elemwise_op(weight, mx_specs=mx_specs, round=mx_specs["round_weight"])  # :39
    bf_bias = None
    if bias is not None:
        bf_bias = quantize_elemwise_op(bias, mx_specs=mx_specs, round=mx_specs["round_weight"])  # :50
    qis_input = quantize_mx_outlier_op(bf_in, mx_specs, inlier_elem_format=mx_specs['a_elem_format'],
                                       outlier_elem_format=mx_specs['a_elem_format'], axes=[1],
                                       round=mx_specs["round_mx_output"])                         # :66-73
    qis_weight = quantize_mx_outlier_op(bf_weight, mx_specs, inlier_elem_format=mx_specs['w_elem_format'],
                                        outlier_elem_format=mx_specs['w_elem_format'], axes=[1],
                                        round=mx_specs["round_mx_output"])                        # :78-85
    output = F.linear(qis_input, qis_weight)                                                      # :91
    output = quantize_elemwise_op(output, mx_specs=mx_specs, round=mx_specs["round_output"])      # :92
    if bias is not None:
        output = quantize_elemwise_op(output + bf_bias, mx_specs=mx_specs, round=mx_specs["round_output"])  # :99-102
    return output


def _mx_scale_bits(mx_specs):
    return 4 if mx_specs["scale_bits"] == 0 else mx_specs["scale_bits"]          # mx_ops.py:519-524


def pack_mx_weight(weight, mx_specs):
    """weight -> (bf16 elementwise cast, linear.py:39) -> mx_ops outlier quantiser along in_features
    (linear.py:78-85) -> packed planes."""
    from .qlinear import pack_values
    mx_specs = apply_mx_specs(mx_specs)
    bf_weight = quantize_elemwise_op(weight.detach(), mx_specs=mx_specs, round=mx_specs["round_weight"])
    qis_weight = quantize_mx_outlier_op(bf_weight.float(), mx_specs, inlier_elem_format=mx_specs['w_elem_format'],
                                        outlier_elem_format=mx_specs['w_elem_format'], axes=[1],
                                        round=mx_specs["round_mx_output"])               # linear.py:78-85
    # the fake-quant values go into the smallest exact single-plane kind (8.25 bits/weight for fp4 weights)
    return pack_values(qis_weight)


def _forward_packed(input, P, bias, mx_specs):
    """Same dataflow as _forward with the weight side precomputed.  The reference quantises the activations along
    ``axes=[1]`` (linear.py:66-73).  For a 2-D input [M, K] that is the feature axis and the two middle steps run
    fused (activation quantiser + dequant-GEMM, msq_qlinear_w4a8).  For [B, S, K] and beyond axis 1 is NOT the last
    axis (it is the sequence axis): the activations then go through the same quantiser call as ``_forward``
    (quantize_mx_outlier_op, axes=[1]) and only the GEMM uses the packed weight, so pack() never changes which
    elements share a block."""
    from .formats import _get_format_params
    from .qlinear import qlinear, qlinear_w4a8
    from ._lib import MsqError
    bf_in = quantize_elemwise_op(input, mx_specs=mx_specs, round=mx_specs["round_output"])
    sb = _mx_scale_bits(mx_specs)
    if input.ndim == 2:
        output = qlinear_w4a8(bf_in, P, None, torch.float32, a_elem_format=mx_specs["a_elem_format"], a_scale_bits=sb,
                              a_std_dev=5, a_block_size=mx_specs["block_size"], a_round=mx_specs["round_mx_output"],
                              a_flush_fp32_subnorms=mx_specs["mx_flush_fp32_subnorms"], a_variant=1)
    else:
        if _get_format_params(mx_specs["a_elem_format"])[1] > 9:     # mbits counts sign + implicit bit: bf16 holds 9
            raise MsqError("MXLinear.pack(): a_elem_format %s is not exact in bf16 (fused GEMM operand)" % mx_specs["a_elem_format"])
        qis_input = quantize_mx_outlier_op(bf_in, mx_specs, inlier_elem_format=mx_specs['a_elem_format'],
                                           outlier_elem_format=mx_specs['a_elem_format'], axes=[1],
                                           round=mx_specs["round_mx_output"])
        output = qlinear(qis_input, P, None, torch.float32)
    output = quantize_elemwise_op(output, mx_specs=mx_specs, round=mx_specs["round_output"])
    if bias is not None:
        bf_bias = quantize_elemwise_op(bias, mx_specs=mx_specs, round=mx_specs["round_weight"])
        output = quantize_elemwise_op(output + bf_bias, mx_specs=mx_specs, round=mx_specs["round_output"])
    return output.to(input.dtype) if output.dtype != input.dtype else output


def linear(input, weight, bias=None, mx_specs=None, name=None):
    """linear.py:204-217"""
    mx_assert_test(mx_specs)
    if mx_specs is None:
        return F.linear(input, weight, bias=bias)
    mx_specs = apply_mx_specs(mx_specs)
    with torch.no_grad():
        return _forward(input, weight, bias, mx_specs)


class MXLinear(torch.nn.Linear):
    """linear.py:220-254"""

    def __init__(self, in_features, out_features, bias=True, mx_specs=None, name=None):
        mx_assert_test(mx_specs)
        self.mx_none = mx_specs is None
        self.name = name
        self.mx_specs = apply_mx_specs(mx_specs)
        super().__init__(in_features, out_features, bias)

    def apply_mx_specs(self, mx_specs):
        mx_assert_test(mx_specs)
        self.mx_none = mx_specs is None
        self.mx_specs = apply_mx_specs(mx_specs)

    def append_name(self, postfix):
        self.name += postfix

    def pack(self):
        """Freeze the current weight into packed planes; forward then runs the fused W4A8 kernels."""
        if self.mx_none:
            raise ValueError("MXLinear.pack needs mx_specs")
        self._packed = pack_mx_weight(self.weight.data, self.mx_specs)
        return self

    def forward(self, inputs):
        if self.mx_none:
            return super().forward(inputs)
        if getattr(self, "_packed", None) is not None:
            with torch.no_grad():
                return _forward_packed(inputs, self._packed, self.bias, self.mx_specs)
        return linear(input=inputs, weight=self.weight, bias=self.bias, mx_specs=self.mx_specs, name=self.name)
