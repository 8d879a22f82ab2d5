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
    # vector_ops.py:61-81 of the reference switch `exp` to vec_exp2(Q(LOG2_E_BF16 x)) under 'vec_use_exp2' and `a / b` to Q(a Q(1 / b)) under
    # 'vec_use_recip': other arithmetic than the kernels implement -- refuse instead of silently returning different values (advisor, round 5)
    for key in ('vec_use_exp2', 'vec_use_recip'):
        try:
            on = bool(mx_specs[key])
        except (KeyError, TypeError):
            on = False
        if on:
            raise MsqError("mx_specs['%s'] is not implemented by the HIP vector ops (they compute exp(x) and a / b directly, the reference's default)" % key)
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


# ---------------------------------------------------------------------------------------------------------------------------
# Round 5: the activation producers in front of the MX Linear -- RMSNorm (layernorm.py:177), silu (activations.py:76), simd_mul
# (simd_ops.py:445) -- and their fused forms that hand the result on as the packed MX-FP8 operand of qlinear.qlinear_mx_w4a8
# (`(codes, scales)`, what qlinear.mx_pack_act returns) without the float32 tensor in between.
# ---------------------------------------------------------------------------------------------------------------------------
def _rows2d(t, who):
    """(tensor kept alive, data pointer, row stride in elements, M, I) of a float32 / float16 / bfloat16 CUDA tensor seen as rows of its last dimension;
    the rows may sit at a stride (a slice of a wider tensor along the last dimension), the values of a row are contiguous."""
    if not torch.is_tensor(t) or not t.is_cuda:
        raise MsqError("%s needs CUDA/HIP tensors (no CPU fallback)" % who)
    t = t.detach()
    if t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        t = t.float()
    I = t.shape[-1]
    try:
        v = t.view(-1, I)                     # a last-dimension slice of a contiguous tensor keeps one row stride
    except RuntimeError:
        v = t.reshape(-1, I)
    if v.stride(-1) != 1 or (v.shape[0] > 1 and v.stride(0) < I):
        v = v.contiguous()
    return v, v.data_ptr(), (v.stride(0) if v.shape[0] > 1 else I), v.shape[0], I


def rms_norm(x, weight, bias, eps, mx_specs):
    """RMSNormFunction.forward (layernorm.py:98-128) as one launch; bias None = zeros."""
    xs = _f32c(x, "RMSNorm")
    H = xs.shape[-1]
    out = torch.empty_like(xs)
    bits, eb, mn, rm, dn = _rounding(mx_specs)
    b = _f32c(bias, "RMSNorm") if bias is not None else None
    check(lib().msq_vec_rmsnorm(ptr(xs), ptr(_f32c(weight, "RMSNorm")), ptr(b), ptr(out), xs.numel() // H, H, float(eps),
                                bits, eb, mn, rm, dn, current_stream(xs.device)), "msq_vec_rmsnorm")
    return out if x.dtype == torch.float32 else out.to(x.dtype)


_X16 = {torch.float16: 1, torch.bfloat16: 2}


def rms_norm_mx_pack(x, weight, bias, eps, mx_specs, return_out=False, flush_fp32_subnorms=False, check_status=False):
    """RMSNorm and the MX-FP8 (e4m3, block 32, 8-bit scale) pack of its output in one launch: `(codes [M, H] uint8, scales [M, H / 32]
    uint8)` -- the activation operand qlinear.qlinear_mx_w4a8 accepts in place of x (one pack shared by q / k / v or gate / up) --
    and, with return_out, the float32 output too.  The bytes are those of qlinear.mx_pack_act(rms_norm(x, ...)).  A float16 / bfloat16 x is
    read as it is (no cast pass; the values of casting first)."""
    if not torch.is_tensor(x) or not x.is_cuda:
        raise MsqError("RMSNorm needs CUDA/HIP tensors (no CPU fallback)")
    xs = x.detach().contiguous() if x.dtype in _X16 else _f32c(x, "RMSNorm")
    if xs.data_ptr() % 16:
        xs = xs.clone()
    H = xs.shape[-1]
    M = xs.numel() // H
    out = torch.empty(xs.shape, dtype=torch.float32, device=xs.device) if return_out else None
    codes = torch.empty(M, H, dtype=torch.uint8, device=xs.device)
    scales = torch.empty(M, H // 32, dtype=torch.uint8, device=xs.device)
    status = torch.zeros(1, dtype=torch.int32, device=xs.device) if check_status else None
    bits, eb, mn, rm, dn = _rounding(mx_specs)
    b = _f32c(bias, "RMSNorm") if bias is not None else None
    w = _f32c(weight, "RMSNorm")
    if xs.dtype in _X16:
        check(lib().msq_vec_rmsnorm_mx_pack_a8_x16(ptr(xs), _X16[xs.dtype], ptr(w), ptr(b), ptr(out), ptr(codes), ptr(scales), ptr(status),
                                                   M, H, float(eps), bits, eb, mn, rm, dn, int(bool(flush_fp32_subnorms)),
                                                   current_stream(xs.device)), "msq_vec_rmsnorm_mx_pack_a8_x16")
    else:
        check(lib().msq_vec_rmsnorm_mx_pack_a8(ptr(xs), ptr(w), ptr(b), ptr(out), ptr(codes), ptr(scales), ptr(status),
                                               M, H, float(eps), bits, eb, mn, rm, dn, int(bool(flush_fp32_subnorms)),
                                               current_stream(xs.device)), "msq_vec_rmsnorm_mx_pack_a8")
    if check_status:
        from .qlinear import _mx_status
        _mx_status(status, "rms_norm_mx_pack")
    return ((codes, scales), out) if return_out else (codes, scales)


class RMSNorm(torch.nn.LayerNorm):
    """layernorm.py:177-202 ("There's no torch equivalent for RMSNorm"): a LayerNorm's parameters (weight, bias), RMSNormFunction's forward.
    `forward_packed(x)` returns the packed MX-FP8 operand of the output instead (rms_norm_mx_pack)."""

    def __init__(self, hidden_size, eps=1e-12, mx_specs=None, name=None):
        mx_assert_test(mx_specs)
        self.name = name
        self.mx_none = (mx_specs is None)
        self.mx_specs = apply_mx_specs(mx_specs)
        super().__init__(normalized_shape=hidden_size, eps=eps)

    def apply_mx_specs(self, mx_specs):
        self.mx_none = (mx_specs is None)
        self.mx_specs = apply_mx_specs(mx_specs)

    def append_name(self, postfix):
        self.name += postfix

    def forward(self, x):
        with torch.no_grad():
            return rms_norm(x, self.weight, self.bias, self.eps, self.mx_specs)

    def forward_packed(self, x, return_out=False):
        with torch.no_grad():
            return rms_norm_mx_pack(x, self.weight, self.bias, self.eps, self.mx_specs, return_out=return_out)


def silu(input, inplace=False, mx_specs=None, name=None):
    """activations.py:76-82"""
    mx_assert_test(mx_specs)
    if mx_specs is None:
        return torch.nn.functional.silu(input, inplace=inplace)
    mx_specs = apply_mx_specs(mx_specs)
    xs = _f32c(input, "silu")
    out = torch.empty_like(xs)
    bits, eb, mn, rm, dn = _rounding(mx_specs)
    check(lib().msq_vec_silu(ptr(xs), ptr(out), xs.numel(), bits, eb, mn, rm, dn, current_stream(xs.device)), "msq_vec_silu")
    out = out if input.dtype == torch.float32 else out.to(input.dtype)
    if inplace:
        input.copy_(out)
        return input
    return out


class SiLU(torch.nn.SiLU):
    """activations.py:195-209"""

    def __init__(self, inplace=False, mx_specs=None, name=None):
        super().__init__(inplace=inplace)
        mx_assert_test(mx_specs)
        self.mx_none = (mx_specs is None)
        self.name = name
        self.mx_specs = apply_mx_specs(mx_specs)

    def forward(self, input):
        if self.mx_none:
            return super().forward(input)
        return silu(input, inplace=self.inplace, mx_specs=self.mx_specs, name=self.name)


def simd_mul(in1, in2, mx_specs=None):
    """simd_ops.py:445-451, :154-187: tensor x tensor (broadcast both ways) or tensor x python scalar (the scalar is not rounded)."""
    mx_assert_test(mx_specs)
    if mx_specs is None:
        return in1 * in2
    mx_specs = apply_mx_specs(mx_specs)
    assert isinstance(in1, torch.Tensor)
    bits, eb, mn, rm, dn = _rounding(mx_specs)
    out_dtype = torch.result_type(in1, in2)
    if not isinstance(in2, torch.Tensor):
        # Q(Q(a) c): the scalar path of the reference; through the rounding entry (two launches: a scalar multiplier is rare on this path)
        a = _f32c(in1, "simd_mul")
        q = torch.empty_like(a)
        check(lib().msq_vec_round(ptr(a), ptr(q), a.numel(), bits, eb, mn, rm, dn, 0, current_stream(a.device)), "msq_vec_round")
        q.mul_(float(in2))
        out = torch.empty_like(q)
        check(lib().msq_vec_round(ptr(q), ptr(out), q.numel(), bits, eb, mn, rm, dn, 0, current_stream(a.device)), "msq_vec_round")
        return out if out_dtype == torch.float32 else out.to(out_dtype)
    if in2.shape != in1.shape:
        in1, in2 = torch.broadcast_tensors(in1, in2)
    a, b = _f32c(in1, "simd_mul"), _f32c(in2, "simd_mul")
    out = torch.empty_like(a)
    check(lib().msq_vec_mul(ptr(a), ptr(b), ptr(out), a.numel(), bits, eb, mn, rm, dn, current_stream(a.device)), "msq_vec_mul")
    return out if out_dtype == torch.float32 else out.to(out_dtype)


def silu_mul(gate, up, mx_specs, pack=False, return_out=None, flush_fp32_subnorms=False, check_status=False):
    """simd_mul(silu(gate), up) -- the gated-MLP activation -- as one launch.  gate / up: float32 (or float16 / bfloat16: read as they are) CUDA tensors of one shape, possibly the two
    halves of one projection output (`gu[..., :I]`, `gu[..., I:]`: read in place at their row stride).  pack=True returns the packed MX-FP8
    operand `(codes, scales)` of the result (the bytes of qlinear.mx_pack_act on it), with return_out=True `((codes, scales), out)`."""
    mx_specs = apply_mx_specs(mx_specs)
    if return_out is None:
        return_out = not pack
    if gate.shape != up.shape:
        raise MsqError("silu_mul: gate and up must have one shape")
    if gate.dtype != up.dtype:
        gate, up = gate.float(), up.float()
    gv, gp, ldg, M, I = _rows2d(gate, "silu_mul")
    uv, up_, ldu, _, _ = _rows2d(up, "silu_mul")
    dev = gv.device
    out = torch.empty(*gate.shape, dtype=torch.float32, device=dev) if return_out else None
    codes = torch.empty(M, I, dtype=torch.uint8, device=dev) if pack else None
    scales = torch.empty(M, I // 32, dtype=torch.uint8, device=dev) if pack else None
    status = torch.zeros(1, dtype=torch.int32, device=dev) if (pack and check_status) else None
    bits, eb, mn, rm, dn = _rounding(mx_specs)
    if gv.dtype in _X16:
        if (gp | up_) % 16 or ldg % 8 or ldu % 8 or I % 8:      # the 16-bit kernel reads 16-byte pieces: fall back to the float32 entry
            return silu_mul(gate.float(), up.float(), mx_specs, pack=pack, return_out=return_out, flush_fp32_subnorms=flush_fp32_subnorms,
                            check_status=check_status)
        check(lib().msq_vec_silu_mul_mx_pack_a8_x16(gp, up_, _X16[gv.dtype], ldg, ldu, ptr(out), ptr(codes), ptr(scales), ptr(status), M, I,
                                                    bits, eb, mn, rm, dn, int(bool(flush_fp32_subnorms)), current_stream(dev)),
              "msq_vec_silu_mul_mx_pack_a8_x16")
    else:
        check(lib().msq_vec_silu_mul_mx_pack_a8(gp, up_, ldg, ldu, ptr(out), ptr(codes), ptr(scales), ptr(status), M, I, bits, eb, mn, rm, dn,
                                                int(bool(flush_fp32_subnorms)), current_stream(dev)), "msq_vec_silu_mul_mx_pack_a8")
    if status is not None:
        from .qlinear import _mx_status
        _mx_status(status, "silu_mul")
    if pack:
        return ((codes, scales), out) if return_out else (codes, scales)
    return out
