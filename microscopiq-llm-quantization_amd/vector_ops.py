"""Bfloat-rounded vector ops -- the surface of number_system/mx/{layernorm.py:68 LayerNorm, activations.py:85 gelu,
simd_ops.py:427 simd_add / :463 simd_split}: the reference's chains of "torch op + quantize_elemwise_op"
(vector_ops.py) run here as ONE HIP launch per function (csrc/msq_vec.hip) with the rounding applied after every
step.  Forward only, like the rest of the hot path."""
import torch

from ._lib import MsqError, check, current_stream, lib, ptr
from .formats import RoundingMode, _get_max_norm
from .specs import apply_mx_specs, mx_assert_test


def _rounding(mx_specs, round=None):
    """(bits, exp_bits, max_norm, rmode, allow_denorm) of quantize_elemwise_op (elemwise_ops.py:237-266) for these specs"""
    if round is None:
        round = mx_specs['round']
    rm = int(RoundingMode[round])
    if mx_specs['bfloat'] > 0 and mx_specs['fp'] > 0:
        raise ValueError("Cannot set both [bfloat] and [fp] in mx_specs.")
    if mx_specs['bfloat'] > 9:
        b = mx_specs['bfloat']
        if b == 32:
            return 0, 8, 0.0, rm, 1
        return b - 7, 8, float(_get_max_norm(8, b - 7)), rm, int(bool(mx_specs['bfloat_subnorms']))
    if 0 < mx_specs['bfloat'] <= 9:
        raise ValueError("Cannot set [bfloat] <= 9 in mx_specs.")
    if mx_specs['fp'] > 6:
        m = mx_specs['fp'] - 6
        return m + 2, 5, float(_get_max_norm(5, m + 2)), rm, int(bool(mx_specs['bfloat_subnorms']))
    if 0 < mx_specs['fp'] <= 6:
        raise ValueError("Cannot set [fp] <= 6 in mx_specs.")
    return 0, 8, 0.0, rm, 1                       # no vector rounding configured


def _f32c(t, who):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise MsqError("%s needs CUDA/HIP tensors (no CPU fallback)" % who)
    return t.detach().float().contiguous()


def layer_norm(x, weight, bias, eps, mx_specs):
    xs = _f32c(x, "LayerNorm")
    H = xs.shape[-1]
    out = torch.empty_like(xs)
    bits, eb, mn, rm, dn = _rounding(mx_specs)
    check(lib().msq_vec_layernorm(ptr(xs), ptr(_f32c(weight, "LayerNorm")), ptr(_f32c(bias, "LayerNorm")), ptr(out),
                                  xs.numel() // H, H, float(eps), bits, eb, mn, rm, dn, current_stream(xs.device)),
          "msq_vec_layernorm")
    return out if x.dtype == torch.float32 else out.to(x.dtype)


class LayerNorm(torch.nn.LayerNorm):
    """layernorm.py:68-99 (TF-style epsilon inside the square root, default 1e-12)."""

    def __init__(self, hidden_size, eps=1e-12, mx_specs=None, name=None):
        mx_assert_test(mx_specs)
        self.mx_none = (mx_specs is None)
        self.name = name
        self.mx_specs = apply_mx_specs(mx_specs)
        super().__init__(normalized_shape=hidden_size, eps=eps)

    def apply_mx_specs(self, mx_specs):
        self.mx_none = (mx_specs is None)
        self.mx_specs = apply_mx_specs(mx_specs)

    def append_name(self, postfix):
        self.name += postfix

    def forward(self, x):
        if self.mx_none:
            return super().forward(x)
        with torch.no_grad():
            return layer_norm(x, self.weight, self.bias, self.eps, self.mx_specs)


def gelu(input, mx_specs=None, first_order_gelu=False, approximate=None, name=None):
    """activations.py:85-104"""
    mx_assert_test(mx_specs)
    if mx_specs is None and first_order_gelu == False:                 # noqa: E712
        return torch.nn.functional.gelu(input, approximate='tanh')
    mx_specs = apply_mx_specs(mx_specs)
    xs = _f32c(input, "gelu")
    out = torch.empty_like(xs)
    bits, eb, mn, rm, dn = _rounding(mx_specs)
    check(lib().msq_vec_gelu(ptr(xs), ptr(out), xs.numel(), int(bool(first_order_gelu)), bits, eb, mn, rm, dn,
                             current_stream(xs.device)), "msq_vec_gelu")
    return out if input.dtype == torch.float32 else out.to(input.dtype)


def simd_add(in1, in2, mx_specs=None):
    """simd_ops.py:427-433, :80-106 (tensor + tensor with full two-way broadcasting -- "Shape broadcasting is fully supported" --, or
    tensor + python scalar); the result has the broadcast shape and torch's promoted dtype, as `in1 + in2` has in the reference."""
    mx_assert_test(mx_specs)
    if mx_specs is None:
        return in1 + in2
    mx_specs = apply_mx_specs(mx_specs)
    assert isinstance(in1, torch.Tensor)
    bits, eb, mn, rm, dn = _rounding(mx_specs)
    if isinstance(in2, torch.Tensor):
        out_dtype = torch.result_type(in1, in2)
        if in2.shape != in1.shape:
            in1, in2 = torch.broadcast_tensors(in1, in2)
        a = _f32c(in1, "simd_add")
        b, bs = _f32c(in2, "simd_add"), 0.0
    else:
        out_dtype = torch.result_type(in1, in2)
        a = _f32c(in1, "simd_add")
        b, bs = None, float(in2)
    out = torch.empty_like(a)
    check(lib().msq_vec_add(ptr(a), ptr(b), bs, ptr(out), a.numel(), bits, eb, mn, rm, dn, current_stream(a.device)),
          "msq_vec_add")
    return out if out_dtype == torch.float32 else out.to(out_dtype)


def simd_split(in1, mx_specs=None):
    """simd_ops.py:463-469: two copies (the backward adds the two gradients; forward only here)"""
    mx_assert_test(mx_specs)
    if mx_specs is None:
        return in1, in1
    return in1.clone(), in1.clone()
