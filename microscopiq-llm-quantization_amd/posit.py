"""Posit<n,es> values -- GPU counterpart of number_system/posit/Posit.py used as an outlier
element format (``posit8_es1`` ...).  `posify` mirrors Posit.py:591 (round a tensor to the
nearest posit value, ties to even, never to 0 / NaR)."""
from . import funcs
from .formats import format_id


def posit_round(x, nbits=8, es=1):
    """Round every element of the CUDA/HIP tensor x to the nearest posit<nbits,es> value."""
    return funcs.quantize_format(x.contiguous(), format_id("posit%d_es%d" % (nbits, es)))


def posify(x, nbits=4, es=1):
    return posit_round(x, nbits, es)
