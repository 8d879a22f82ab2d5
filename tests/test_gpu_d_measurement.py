"""SURVEY 8 row d: bench.py's line and its sub-objects

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_d_*.py."""
import pytest

from legacy_gpu_parity import (  # noqa: F401
    test_bench_shape_gemms_against_dense_reference,
)
from legacy_gpu_round3 import (  # noqa: F401
    test_bench_llama7b_e2e_slice,
    test_bench_rowparallel_evidence_on_single_rank_rccl_group,
)
from legacy_gpu_round4 import (  # noqa: F401
    test_bench_other_configs_keys_and_rates,
    test_bench_line_names_its_kernel_sustains_and_carries_the_true_7b_layer,
    test_bench_line_ppl_and_cpu_baseline_parity,
    test_default_bench_line_rowparallel_leg_on_a_single_rank_rccl_group,
)
from legacy_gpu_round4 import bench_line  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
