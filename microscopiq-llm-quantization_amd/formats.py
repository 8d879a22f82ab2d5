"""Element formats -- mirror of number_system/mx/formats.py (names, enum values, return
tuples) backed by the native table in libmsq_hip.so (csrc/msq_host.h).

Extension: ``posit<n>_es<k>`` names, e.g. ``posit8_es1``.
"""
from enum import Enum, IntEnum

from . import _lib

FP32_EXPONENT_BIAS = 127                      # formats.py:11
FP32_MIN_NORMAL = 2 ** (-FP32_EXPONENT_BIAS + 1)   # formats.py:12


class RoundingMode(IntEnum):                  # formats.py:15-22 == cpp/common.cuh:130-134
    nearest = 0
    floor = 1
    even = 2

    @staticmethod
    def string_enums():
        return [s.name for s in list(RoundingMode)]


class ElemFormat(Enum):                       # formats.py:25-47
    int8 = 1
    int4 = 2
    int2 = 3
    fp8_e5m2 = 4
    fp8_e4m3 = 5
    fp6_e3m2 = 6
    fp6_e2m3 = 7
    fp4 = 8
    fp4_e2m1 = 8
    float16 = 9
    fp16 = 9
    bfloat16 = 10
    bf16 = 10

    @staticmethod
    def from_str(s):
        assert (s != None), "String elem_format == None"
        s = s.lower()
        if hasattr(ElemFormat, s):
            return getattr(ElemFormat, s)
        raise Exception("Undefined elem format", s)


def format_id(fmt):
    """ElemFormat | str (incl. posit names) -> native format id (include/msq.h MSQ_FMT_*)"""
    if isinstance(fmt, ElemFormat):
        return fmt.value
    if isinstance(fmt, str):
        return _lib.format_id(fmt)
    raise Exception("Unknown element format %s" % fmt)


_FORMAT_CACHE = {}


def _get_format_params(fmt):
    """(ebits, mbits, emax, max_norm, min_norm) -- formats.py:65-129"""
    key = fmt.value if isinstance(fmt, ElemFormat) else str(fmt).lower()
    if key not in _FORMAT_CACHE:
        e, m, ex, mx, mn, _ = _lib.format_params(format_id(fmt))
        _FORMAT_CACHE[key] = (e, m, ex, mx, mn)
    return _FORMAT_CACHE[key]


def _get_min_norm(ebits):                     # formats.py:50-54
    emin = 2 - (2 ** (ebits - 1))
    return 0 if ebits == 0 else 2 ** emin


def _get_max_norm(ebits, mbits):              # formats.py:57-61
    assert (ebits >= 5), "invalid for floats that don't define NaN"
    emax = 0 if ebits == 0 else 2 ** (ebits - 1) - 1
    return 2 ** emax * float(2 ** (mbits - 1) - 1) / 2 ** (mbits - 2)
