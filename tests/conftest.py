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


def smooth_field(seed, h, w):
    """SURVEY 8d second input suite: per channel a sum of 8 random low-frequency 2-D cosines scaled to [0, 255],
    so that decoded boxes cluster and NMS / fast-NMS see heavy overlap (uniform noise gives scattered boxes)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64) / h, np.arange(w, dtype=np.float64) / w, indexing="ij")
    img = np.zeros((h, w, 3), np.float64)
    for c in range(3):
        for _ in range(8):
            fy, fx = rng.uniform(0.5, 4.0, 2); ph = rng.uniform(0, 2 * np.pi); amp = rng.uniform(0.3, 1.0)
            img[..., c] += amp * np.cos(2 * np.pi * (fy * yy + fx * xx) + ph)
        lo, hi = img[..., c].min(), img[..., c].max()
        img[..., c] = (img[..., c] - lo) / (hi - lo) * 255.0
    return img.astype(np.float32)
