"""SURVEY 8 row a9 (`_quantize_mx` / `quantize_mx_op`, number_system/mx/mx_ops.py:332-490): which arithmetic the `custom_cuda` flag selects
(advisor, round 5) and the strided-axis kernels after the round-6 rework (the exponent rule is applied once per block maximum)."""
import numpy as np
import pytest
import torch

from gpu_common import dev, eq, planted

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt", ["fp8_e4m3", "fp4_e2m1", "fp6_e3m2", "int4"])
@pytest.mark.parametrize("rnd", ["nearest", "floor", "even"])
def test_custom_cuda_flag_selects_native_or_python_arithmetic(msq, O, fmt, rnd):
    """Block maxima 1 .. 90 ulps under a power of two, inner and outer axis:
      custom_cuda=True                                -> the reference's native kernel (exponent field, divisor = scale): oracle.quantize_mx_native
      custom_cuda=False                               -> the Python path without the +1e-6 defect: oracle.quantize_mx
      custom_cuda=False inside reference_python_divisor() -> the reference's CPU result bit for bit: oracle.quantize_mx(plus_eps_defect=True)
    and custom_cuda=True is NOT affected by reference_python_divisor() (cpp/mx.cuh:132 has no epsilon)."""
    e, m, _, mx, _ = msq.formats._get_format_params(fmt)
    rm = {"nearest": 0, "floor": 1, "even": 2}[rnd]
    A = planted(24, 256, 17)
    B = np.ascontiguousarray(A.reshape(24, 8, 32).transpose(0, 2, 1))            # [24, 32, 8]: blocks of 32 along axis 1
    differs = 0
    for X, ax in ((A, 1), (B, 1)):
        Xt = torch.from_numpy(X).to(dev())
        y_nat = msq.mx_ops._quantize_mx(Xt, 8, fmt, axes=[ax], block_size=32, round=rnd, custom_cuda=True).cpu().numpy()
        y_py = msq.mx_ops._quantize_mx(Xt, 8, fmt, axes=[ax], block_size=32, round=rnd, custom_cuda=False).cpu().numpy()
        with msq.mx_ops.reference_python_divisor():
            y_ref = msq.mx_ops._quantize_mx(Xt, 8, fmt, axes=[ax], block_size=32, round=rnd, custom_cuda=False).cpu().numpy()
            y_nat2 = msq.mx_ops._quantize_mx(Xt, 8, fmt, axes=[ax], block_size=32, round=rnd, custom_cuda=True).cpu().numpy()
        assert eq(y_nat, O.quantize_mx_native(X, 8, e, m, mx, 32, ax, False, rm))
        assert eq(y_py, O.quantize_mx(X, 8, fmt, axis=ax, block_size=32, round=rnd))
        assert eq(y_ref, O.quantize_mx(X, 8, fmt, axis=ax, block_size=32, round=rnd, plus_eps_defect=True))
        assert eq(y_nat2, y_nat)
        differs += int((y_nat != y_py).any())
    assert differs == 2                                                           # the planted maxima bite on both kernel families


def test_quantize_mx_op_passes_the_flag_of_the_specs(msq, O):
    A = planted(8, 128, 23)
    At = torch.from_numpy(A).to(dev())
    e, m, _, mx, _ = msq.formats._get_format_params("fp8_e4m3")
    for cc in (False, True):
        specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp8_e4m3", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32,
                                             "custom_cuda": cc})
        y = msq.mx_ops.quantize_mx_op(At, specs, elem_format="fp8_e4m3", axes=[-1]).cpu().numpy()
        yo = O.quantize_mx_native(A, 8, e, m, mx, 32, 1, False, 0) if cc else O.quantize_mx(A, 8, "fp8_e4m3", axis=-1, block_size=32)
        assert eq(y, yo), cc


@pytest.mark.parametrize("tile", [16, 32, 5])
@pytest.mark.parametrize("pyd,pye", [(0, 0), (1, 1), (0, 1), (1, 0)])
@pytest.mark.parametrize("rnd", [0, 1, 2])
def test_strided_axis_tile_kernels_all_four_variants(msq, O, tile, pyd, pye, rnd):
    """k_mx_tile_cols4 / k_mx_tile_generic (quantize_mx_by_tile along an outer axis, cpp/funcs.cpp:161-181) for every (divisor, exponent)
    variant and rounding mode against the oracle, incl. a ragged last tile (axis length 40) and planted maxima: the round-6 kernels take
    the exponent rule once per block maximum (monotone in |bits|) and keep the arithmetic codec of `floor` out of the register-tile loop."""
    A = planted(40, 256, 29)
    A[3, :] = 0.0
    A[7, 5] = np.float32(3e38)
    e, m, _, mx, _ = msq.formats._get_format_params("fp4_e2m1")
    At = torch.from_numpy(A).to(dev())
    y = msq.funcs.quantize_mx_by_tile_func_cuda(At, 8, e, m, mx, tile, 0, False, rnd, python_divisor=bool(pyd), python_exponent=bool(pye)).cpu().numpy()
    if pye:
        yo = O.quantize_mx(A, 8, "fp4_e2m1", axis=0, block_size=tile, round=["nearest", "floor", "even"][rnd], plus_eps_defect=bool(pyd))
        assert eq(y, yo)
    elif not pyd:
        assert eq(y, O.quantize_mx_native(A, 8, e, m, mx, tile, 0, False, rnd))
    elif rnd != 1:
        # (python divisor, native exponent) has no oracle entry of its own: in every block where the two exponent rules agree it equals
        # the full Python path (under `floor` the Python path also changes the element codec, so that mode is left to the other variants)
        names = ["nearest", "floor", "even"]
        yo = O.quantize_mx(A, 8, "fp4_e2m1", axis=0, block_size=tile, round=names[rnd], plus_eps_defect=True)
        yn = O.quantize_mx_native(A, 8, e, m, mx, tile, 0, False, rnd)
        yp = O.quantize_mx(A, 8, "fp4_e2m1", axis=0, block_size=tile, round=names[rnd])
        same = (yn == yp) | (np.isnan(yn) & np.isnan(yp))
        agree = np.ones_like(same)
        for t0 in range(0, 40, tile):
            agree[t0:t0 + tile] = same[t0:t0 + tile].all(axis=0, keepdims=True)
        assert agree.any() and eq(np.where(agree, y, 0), np.where(agree, yo, 0))
