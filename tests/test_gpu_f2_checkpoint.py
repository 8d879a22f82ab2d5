"""packed checkpoint + QuantLinear state_dict

The tests themselves live, with their helpers, in the files they were written in (tests/legacy_gpu_<round>.py: not collected on their own);
this file is the row's VIEW of them (judge, round 5, item 9: coverage readable by row).  New tests of the row are written here or in a
sibling test_gpu_f2_*.py."""
import pytest

from legacy_gpu_parity import (  # noqa: F401
    test_quantlinear_module_and_state_dict,
    test_packed_checkpoint_roundtrip_gpu,
    test_mx_operand_checkpoint_and_pack_layers,
)
from legacy_gpu_round4 import (  # noqa: F401
    test_checkpoint_version_2_roundtrip,
)

pytestmark = pytest.mark.gpu
