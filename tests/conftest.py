import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `pytest -m gpu`")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure, oracle/): built on demand."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def api():
    """The product binding; the HIP library must be built (no CPU fallback)."""
    import __graft_entry__ as g
    from geoflowslam_amd import api as A
    if not os.path.exists(A._LIB_PATH):
        g.build()
    A.lib()
    return A


@pytest.fixture(scope="session")
def gpu_api(api):
    if api.device_count() < 1:
        pytest.fail("no gfx950 device visible: the -m gpu tests must run on an MI355X")
    return api
