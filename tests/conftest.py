import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hostsim():
    from blsutil import HostSim
    return HostSim()


@pytest.fixture(scope="session")
def coracle():
    from blsutil import COracle
    return COracle()


@pytest.fixture(scope="session")
def engine():
    """The product path: libbls381_hip.so through its C ABI.  No fallback -- a missing library or GPU fails."""
    import bls381_amd as B
    eng = B.Engine(0)
    yield eng
    eng.close()
