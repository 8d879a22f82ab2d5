"""SURVEY 8 row b: the C ABI (include/msq.h): status codes, argument checks, limits

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_b_*.py."""
import pytest

from legacy_gpu_parity import (  # noqa: F401
    test_c_abi_error_codes,
)
from legacy_gpu_round2 import (  # noqa: F401
    test_c_abi_error_codes_round2,
    test_qlinear_rejects_over_4gib_operands,
)

pytestmark = pytest.mark.gpu
