import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# shared by the row-named GPU test files (test_gpu_<SURVEY 8 row>_*.py); the older files define module-scoped fixtures of the same names
@pytest.fixture(scope="session")
def msq():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import msq as m
    m._lib.lib()            # fails loudly when libmsq_hip.so is missing: there is no CPU fallback
    return m


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    return oracle
