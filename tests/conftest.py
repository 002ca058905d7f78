import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import rsp_chains_amd as R
        return R.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must not silently pass on a box without a device or without the native library."""
    import rsp_chains_amd as R
    n = R.device_count()
    if n <= 0:
        pytest.fail("test marked gpu but no HIP device is visible / librspchain.so not usable")
    return n
