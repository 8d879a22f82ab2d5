"""W4A8 on the scaled MFMA (k_mxgemm256, k_mxgemm, k_mxgemv, operand packers)

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_n3_*.py."""
import pytest

from legacy_gpu_parity import (  # noqa: F401
    test_mx_native_w4a8_vs_oracle,
    test_mx_msq_weights_vs_oracle,
    test_mx_msq_weights_llama_shapes_repeatable,
    test_mx_native_llama_shapes_repeatable,
    test_mx_linear_module,
    test_mx_pack_act_bf16_input,
    test_mx_operand_pack_edge_cases,
)
from legacy_gpu_round2 import (  # noqa: F401
    test_mx_fp6_weight_plane,
    test_mx_gemm_k_groups,
    test_mx_gemm_tail_steps_run_to_run,
)
from legacy_gpu_round3 import (  # noqa: F401
    test_mx_act_pack_vec_equals_block_kernel,
    test_mx_decode_wide_projection_eight_wave_blocks,
    test_fp16_activation_cast_and_mx_pack,
)
from legacy_gpu_round4 import (  # noqa: F401
    test_mx_pack_act_fp16_at_an_odd_storage_offset,
    test_mxgemm256_equals_mxgemm,
    test_mxgemm256_tail_steps_repeat_under_uneven_load,
)
from legacy_gpu_round5 import (  # noqa: F401
    test_mx_operand_packers_and_kv_just_below_powers_of_two,
)

pytestmark = pytest.mark.gpu
