"""KV-cache quantisation at the GEAR hook (kv_quant/GEARLM/Simulated/compress_function.py)

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_f3_*.py."""
import pytest

from legacy_gpu_kvcache import (  # noqa: F401
    test_kv_group_quant_golden_bit_exact,
    test_compress_insert_function_golden,
    test_kv_group_quant_llama7b_cache_vs_oracle,
    test_mx_kv_quant_vs_oracle,
    test_streaming_cache_matches_hook_semantics,
)
from legacy_gpu_round3 import (  # noqa: F401
    test_kv_cache_mx_variant_is_one_launch_in_the_cache_dtype,
    test_kv_group_quant_division_free_path_is_exact,
)
from legacy_gpu_round5 import (  # noqa: F401
    test_kv_mx_e4m3_block_setup_every_binade,
    test_kv_mx_fp4_every_tie_and_the_excepted_magnitude,
)

pytestmark = pytest.mark.gpu
