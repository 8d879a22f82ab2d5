"""MX block quantisation -- the surface of number_system/mx/mx_ops.py (_quantize_mx :332,
quantize_mx_op :460, _quantize_mx_outlier_v1 :210, quantize_mx_outlier_op :492), on the GPU.

``_quantize_mx`` selects its arithmetic from the ``custom_cuda`` flag, as the reference does:

* ``custom_cuda=True``  -> the reference's NATIVE kernel (cpp/mx.cuh:107-170, cpp/shared_exp.cuh:14-53): shared exponent = the
  exponent FIELD of the block maximum, divisor = the scale.  This is what the upstream KATs pin.
* ``custom_cuda=False`` -> the reference's PYTHON path (mx_ops.py:332-457): shared exponent = ``floor(torch.log2(max))`` (one
  higher for the up to 88 largest floats under a power of two), under ``round="floor"`` the private exponents too.  Its
  divisor ``2**e + 1e-6`` (mx_ops.py:444, a reference defect that breaks three of its own KATs) is NOT
  applied unless ``with reference_python_divisor():`` is active -- inside it the result is the reference's CPU output bit
  for bit; outside it is "the Python path without the defect", which is also what the MX GEMM operand packers encode
  (msq_mx_pack_*: pinned by the GEMM tests against ``oracle.quantize_mx``)."""
import torch

from . import funcs
from ._lib import MsqError
from .formats import ElemFormat, RoundingMode, _get_format_params
from .quant import VARIANT_MXOPS, outlier_fakequant
from .specs import mx_assert_test


# The divisor of the Python path: `scale + 1e-6` (mx_ops.py:444) moves every tie of the scaled element down (3 % of
# bfloat16-rounded activations in fp6).  Off by default (see the module docstring); `with reference_python_divisor():` turns
# it on for custom_cuda=False calls, bit for bit the reference's CPU result.  custom_cuda=True never takes it (cpp/mx.cuh:132).
_PY_DIVISOR = [False]


class reference_python_divisor:
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _PY_DIVISOR[0]
        _PY_DIVISOR[0] = self.on
        return self

    def __exit__(self, *exc):
        _PY_DIVISOR[0] = self.prev
        return False


_LOWP_BLOCKS = (8, 16, 32, 64, 128)


def _quantize_mx(A, scale_bits, elem_format, shared_exp_method="max", axes=None, block_size=0, round="nearest",
                 flush_fp32_subnorms=False, custom_cuda=False, compute_dtype="input"):
    """mx_ops.py:332-457; single-axis, executed by msq_quantize_mx_by_tile_ex.  ``custom_cuda`` picks the reference's native
    kernel arithmetic (True) or its Python path (False) -- module docstring.

    fp16 / bf16 tensors (``compute_dtype="input"``, the default): one launch of msq_quantize_mx_lowp, which computes IN the
    tensor dtype op by op -- what the reference does with a half tensor (its native kernel is float32 only, so the Python path
    runs on the Half tensor, `+ 1e-6` included) -- bit-exact against reference-made fixtures (tests/golden/quantize_mx_lowp.npz).
    ``compute_dtype="float32"`` keeps the upcast -> fp32 kernel -> downcast route (three passes, one rounding)."""
    if elem_format == None:
        return A
    assert (scale_bits > 0)
    if shared_exp_method != "max":
        raise Exception("Unrecognized shared exponent selection method %s" % (shared_exp_method))
    if round not in RoundingMode.string_enums():
        raise Exception("Unrecognized round method %s" % (round))
    axes = [axes] if type(axes) == int else axes
    axes = [x + A.ndim if x < 0 else x for x in axes]
    if len(axes) != 1:
        raise MsqError("the MI355X kernels quantise along exactly one axis (got %r)" % (axes,))
    ebits, mbits, emax, max_norm, _ = _get_format_params(elem_format)
    axis = axes[0]
    tile = block_size if block_size > 0 else A.shape[axis]
    x = A.contiguous()
    if compute_dtype not in ("input", "float32"):
        raise MsqError("compute_dtype must be 'input' or 'float32'")
    if (compute_dtype == "input" and x.dtype in (torch.float16, torch.bfloat16) and x.is_cuda and tile in _LOWP_BLOCKS
            and not str(elem_format).lower().replace("elemformat.", "").startswith("posit")):
        from ._lib import check, current_stream, lib, ptr
        from .formats import format_id
        from . import quant as _quant
        pre = 1
        for s_ in x.shape[:axis]:
            pre *= int(s_)
        post = 1
        for s_ in x.shape[axis + 1:]:
            post *= int(s_)
        out = torch.empty_like(x)
        status = torch.zeros(1, dtype=torch.int32, device=x.device) if _quant.CHECK_NAN else None
        fname = elem_format.name if isinstance(elem_format, ElemFormat) else str(elem_format)
        check(lib().msq_quantize_mx_lowp(ptr(x), ptr(out), 1 if x.dtype == torch.float16 else 2, pre, int(x.shape[axis]), post, int(tile),
                                         int(scale_bits), format_id(fname), int(RoundingMode[round]), int(bool(flush_fp32_subnorms)),
                                         ptr(status), current_stream(x.device)), "msq_quantize_mx_lowp")
        return out
    y = funcs.quantize_mx_by_tile_func_cuda(x.float() if x.dtype != torch.float32 else x, scale_bits, ebits, mbits,
                                            max_norm, tile, axis, flush_fp32_subnorms, int(RoundingMode[round]),
                                            python_divisor=_PY_DIVISOR[0] and not custom_cuda, python_exponent=not custom_cuda)
    return y if A.dtype == torch.float32 else y.to(A.dtype)


def quantize_mx_op(A, mx_specs, elem_format=None, block_size=None, axes=None, round="nearest",
                   expand_and_reshape=False):
    """mx_ops.py:460-490"""
    mx_assert_test(mx_specs)
    if elem_format == None:
        return A
    elif type(elem_format) is str:
        elem_format = ElemFormat.from_str(elem_format)
    if block_size == None:
        block_size = mx_specs["block_size"]
    scale_bits = 8 if mx_specs["scale_bits"] == 0 else mx_specs["scale_bits"]
    return _quantize_mx(A, scale_bits, elem_format, block_size=block_size, axes=axes, round=round,
                        shared_exp_method=mx_specs["shared_exp_method"],
                        flush_fp32_subnorms=mx_specs["mx_flush_fp32_subnorms"], custom_cuda=mx_specs["custom_cuda"])


def _quantize_mx_outlier_v1(A, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                            shared_exp_method="max", std_dev=5, axes=None, block_size=0, round="nearest",
                            flush_fp32_subnorms=False, custom_cuda=False):
    """mx_ops.py:210-330 -- the older outlier quantiser MXLinear uses: statistics of the signed
    values over the block-COUNT axis with unbiased std (:62-66,:248), clamp to -scale_emax (:273)."""
    if inlier_elem_format == None:
        return A
    if shared_exp_method != "max":
        raise Exception("Unrecognized shared exponent selection method %s" % (shared_exp_method))
    axes = [axes] if type(axes) == int else axes
    axes = [x + A.ndim if x < 0 else x for x in axes]
    if len(axes) != 1:
        raise MsqError("the MI355X kernels quantise along exactly one axis (got %r)" % (axes,))
    return outlier_fakequant(A, inlier_scale_bits, outlier_scale_bits, inlier_elem_format, outlier_elem_format,
                             std_dev, axes[0], block_size, round, flush_fp32_subnorms, variant=VARIANT_MXOPS)["out"]


def quantize_mx_outlier_op(A, mx_specs, inlier_elem_format=None, outlier_elem_format=None, block_size=None,
                           axes=None, round="nearest", expand_and_reshape=False):
    """mx_ops.py:492-533"""
    mx_assert_test(mx_specs)
    if inlier_elem_format == None or outlier_elem_format == None:
        return A
    if type(inlier_elem_format) is str:
        inlier_elem_format = ElemFormat.from_str(inlier_elem_format)
    if type(outlier_elem_format) is str:
        outlier_elem_format = ElemFormat.from_str(outlier_elem_format)
    if block_size == None:
        block_size = mx_specs["block_size"]
    inlier_scale_bits = 4 if mx_specs["scale_bits"] == 0 else mx_specs["scale_bits"]     # mx_ops.py:519-524
    return _quantize_mx_outlier_v1(A, inlier_scale_bits, inlier_scale_bits, inlier_elem_format, outlier_elem_format,
                                   block_size=block_size, axes=axes, round=round,
                                   shared_exp_method=mx_specs["shared_exp_method"],
                                   flush_fp32_subnorms=mx_specs["mx_flush_fp32_subnorms"],
                                   custom_cuda=mx_specs["custom_cuda"])
