"""fused unpack-dequant-GEMM on bf16 MFMA (k_qgemm256, k_qgemm3, k_qgemv*, k_qgemm256p; k_qgemm_sk: test_gpu_n2_gemm_midm.py)

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_n2_*.py."""
import pytest

from legacy_gpu_parity import (  # noqa: F401
    test_unified_fused_gemm_vs_oracle_linear,
    test_fused_gemm_vs_oracle_linear,
    test_fused_gemm_llama_shapes_repeatable,
)
from legacy_gpu_round2 import (  # noqa: F401
    test_decode_kernels_on_second_stream_and_repeated,
    test_round2_kernels_run_to_run_identical,
    test_fused_gemm_64_row_tiles_and_k_groups,
)
from legacy_gpu_round3 import (  # noqa: F401
    test_staggered_gemm_short_and_odd_k,
    test_fused_projections_same_values_fewer_launches,
    test_packed_decode_step_is_graph_capturable,
    test_fp16_output_equals_rounded_fp32_output,
    test_decode_kernels_take_fp16_activations,
    test_wide_projection_decode_kernel,
)
from legacy_gpu_round4 import (  # noqa: F401
    test_fused_projections_under_inference_mode,
    test_qgemm256_equals_qgemm3_and_the_dense_product,
)
from legacy_gpu_round5 import (  # noqa: F401
    test_forced_gemm_kernels_against_the_oracle_on_multi_round_grids,
    test_persistent_stream_k_cut_tiles,
    test_persistent_uncut_tiles_equal_qgemm256_bit_for_bit,
    test_persistent_kernel_without_workspace_falls_back,
)

pytestmark = pytest.mark.gpu
