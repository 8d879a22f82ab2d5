"""Mixed-precision Linear -- the surface of number_system/mx/linear.py (linear :204, MXLinear
:220, LinearFunction.forward :20-108): bfloat/fp elementwise cast of input, weight and bias,
MicroScopiQ outlier quantisation of activations and weights (the mx_ops variant, axes=[1],
inlier format == outlier format), dense GEMM, cast, + bias, cast.

Every quantisation step is one HIP launch; the GEMM runs in fp32 on the GPU exactly like the
reference's F.linear (linear.py:91).  Forward only: the reference's backward is broken as
shipped (linear.py:129-136 passes unknown kwargs)."""
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

    def forward(self, inputs):
        if self.mx_none:
            return super().forward(inputs)
        return linear(input=inputs, weight=self.weight, bias=self.bias, mx_specs=self.mx_specs, name=self.name)
