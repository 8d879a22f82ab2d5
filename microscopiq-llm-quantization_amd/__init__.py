"""MicroScopiQ hot path for AMD MI355X (gfx950): outlier-aware microscaling quant/dequant and
the mixed-precision Linear, behind the reference's own Python surface.

    from msq import quant, mx_ops, linear, qlinear          (see msq.py at the repo root)

The compute lives in libmsq_hip.so (C ABI: include/msq.h).  Nothing here falls back to the
CPU: without the built HIP library every entry point raises."""
from . import _lib  # noqa: F401
from . import formats, funcs, elemwise_ops, specs, quant, mx_ops, linear, qlinear, posit, kvcache, vector_ops, quant_model, matmul as _matmul_mod  # noqa: F401
from .quant import (MXQuantizer, Quantizer, quantize, quantize_mx_outlier_hessian,  # noqa: F401
                    quantize_mx_outlier_v1)
from .qlinear import QuantLinear, RowParallelQuantLinear, make_quant, pack_weight, unpack_weight  # noqa: F401
from .linear import MXLinear  # noqa: F401
from .vector_ops import LayerNorm, RMSNorm, SiLU, gelu, silu, simd_add, simd_mul, simd_split  # noqa: F401
from .quant_model import quantize_model  # noqa: F401
from .matmul import matmul, bmm  # noqa: F401

__version__ = "0.1.0"
