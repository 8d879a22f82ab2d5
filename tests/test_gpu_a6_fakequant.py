"""blocking, `_extract_outlier_indices`, `_shared_exponents`, `quantize_mx_outlier_v1` / `_hessian`, `MXQuantizer` (utils/quant.py:23-616)

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_a6_*.py."""
import pytest

from legacy_gpu_parity import (  # noqa: F401
    test_outlier_fakequant_golden_bit_exact,
    test_reference_nan_assert_is_raised,
    test_hessian_num_outliers_golden,
    test_outlier_fakequant_vs_oracle,
    test_outlier_edge_cases_vs_oracle,
    test_rounding_modes_and_flush_vs_oracle,
    test_low_precision_inputs_bit_exact_and_fp32_mode_delta,
    test_fakequant_bf16_native_equals_upcast,
    test_full_size_properties,
)
from legacy_gpu_round2 import (  # noqa: F401
    test_lowp_floor_log2_exhaustive_gpu,
    test_lowp_outlier_fakequant_golden_gpu,
    test_lowp_llama_sized_weight_vs_oracle,
)
from legacy_gpu_round4 import (  # noqa: F401
    test_shared_exponent_just_below_powers_of_two,
)

pytestmark = pytest.mark.gpu
