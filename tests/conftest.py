import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure, oracle/): C restatement of the
    reference's advect_scalar2D_cpu, built on demand with gcc."""
    from oracle import oracle as O
    O.build_lib()
    return O


@pytest.fixture(scope="session")
def mpdata():
    import codesign_kernels_amd as M
    return M
