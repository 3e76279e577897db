import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; see oracle/pisa_oracle.c)."""
    from oracle import oracle as orc

    orc.build()
    return orc


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


# reference's own prob3 tolerance (numba_osc_tests.py:82: AC_KW)
PROB3_RTOL = 1e-10
PROB3_ATOL = 1e-14
