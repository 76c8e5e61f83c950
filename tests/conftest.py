import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def hip_available():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return False
    n = ctypes.c_int(0)
    return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
