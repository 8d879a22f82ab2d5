"""Elementwise (bfloat / fpX / eXmY) quantisation -- the surface of
number_system/mx/elemwise_ops.py, executed by the HIP library.

``custom_cuda`` is accepted for signature compatibility; every call runs on the
GPU (elemwise_ops.py:117-129 is the only path here)."""
import torch

from . import funcs
from .formats import RoundingMode, _get_format_params, _get_max_norm, _get_min_norm  # noqa: F401


def _quantize_elemwise_core(A, bits, exp_bits, max_norm, round='nearest', saturate_normals=False,
                            allow_denorm=True, custom_cuda=False):
    """elemwise_ops.py:84-174"""
    if round not in RoundingMode.string_enums():
        raise Exception("Unrecognized round method %s" % (round))      # elemwise_ops.py:72 ('dither' has no native path)
    A = A.contiguous()
    return funcs.quantize_elemwise_func_cuda(A, bits, exp_bits, max_norm, int(RoundingMode[round]),
                                             saturate_normals, allow_denorm)


def _quantize_elemwise(A, elem_format, round='nearest', custom_cuda=False, saturate_normals=False,
                       allow_denorm=True):
    """elemwise_ops.py:177-192"""
    if elem_format is None:
        return A
    ebits, mbits, _, max_norm, _ = _get_format_params(elem_format)
    return _quantize_elemwise_core(A, mbits, ebits, max_norm, round=round, allow_denorm=allow_denorm,
                                   saturate_normals=saturate_normals, custom_cuda=custom_cuda)


def _quantize_bfloat(A, bfloat, round='nearest', custom_cuda=False, allow_denorm=True):
    """elemwise_ops.py:195-210"""
    if bfloat == 0 or bfloat == 32:
        return A
    max_norm = _get_max_norm(8, bfloat - 7)
    return _quantize_elemwise_core(A, bits=bfloat - 7, exp_bits=8, max_norm=max_norm, round=round,
                                   allow_denorm=allow_denorm, custom_cuda=custom_cuda)


def _quantize_fp(A, exp_bits=None, mantissa_bits=None, round='nearest', custom_cuda=False, allow_denorm=True):
    """elemwise_ops.py:213-234"""
    if exp_bits is None or mantissa_bits is None:
        return A
    max_norm = _get_max_norm(exp_bits, mantissa_bits + 2)
    return _quantize_elemwise_core(A, bits=mantissa_bits + 2, exp_bits=exp_bits, max_norm=max_norm, round=round,
                                   allow_denorm=allow_denorm, custom_cuda=custom_cuda)


def quantize_elemwise_op(A, mx_specs, round=None):
    """elemwise_ops.py:237-266"""
    if mx_specs is None:
        return A
    elif round is None:
        round = mx_specs['round']
    if mx_specs['bfloat'] > 0 and mx_specs['fp'] > 0:
        raise ValueError("Cannot set both [bfloat] and [fp] in mx_specs.")
    elif mx_specs['bfloat'] > 9:
        A = _quantize_bfloat(A, bfloat=mx_specs['bfloat'], round=round, custom_cuda=mx_specs['custom_cuda'],
                             allow_denorm=mx_specs['bfloat_subnorms'])
    elif mx_specs['bfloat'] > 0 and mx_specs['bfloat'] <= 9:
        raise ValueError("Cannot set [bfloat] <= 9 in mx_specs.")
    elif mx_specs['fp'] > 6:
        A = _quantize_fp(A, exp_bits=5, mantissa_bits=mx_specs['fp'] - 6, round=round,
                         custom_cuda=mx_specs['custom_cuda'], allow_denorm=mx_specs['bfloat_subnorms'])
    elif mx_specs['fp'] > 0 and mx_specs['fp'] <= 6:
        raise ValueError("Cannot set [fp] <= 6 in mx_specs.")
    return A
