import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """Fails (not skips) when the HIP library or the device is missing: the GPU tests must exercise the native path."""
    import cfd_amd
    n = cfd_amd.lbm.device_count()
    assert n > 0, "no HIP device visible"
    return n
