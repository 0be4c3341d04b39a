import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oc():
    """The C oracle (test infrastructure; builds on first use)."""
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c


@pytest.fixture(scope="session")
def gm():
    """The product: HIP library behind the C-ABI.  No fallback: a missing
    or unloadable library is an error, never a skip."""
    import geometric_mapping_amd as g
    g.load_library()
    return g
