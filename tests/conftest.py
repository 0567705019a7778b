import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bf16_round(x):
    """numpy float32 -> nearest-even bf16, returned as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


TINY = dict(patch=32, width=64, layers=2, heads=2, embed=32, img_res=224, txt_width=64, txt_layers=2, txt_heads=2,
            ctx=77, vocab=512)


@pytest.fixture(scope="session")
def gpu_lib():
    from arp_amd import _ffi
    if _ffi.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (the HIP path has no CPU fallback)")
    return _ffi
