import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "instancesegmentation-jittor_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def ffi():
    """The product FFI on a GPU box; fails loudly (never falls back) if the HIP library is missing."""
    from isegmi import _ffi
    _ffi.lib()
    if _ffi.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible")
    _ffi.set_device(0)
    return _ffi
