import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "atoma-infer_amd", "bindings"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ah():
    """The ctypes binding of libatoma_hip.so.  GPU tests fail loudly (ImportError) when the
    HIP extension is not built -- there is no CPU fallback to hide behind."""
    import atoma_hip
    return atoma_hip


@pytest.fixture(scope="session")
def gpu(ah):
    if ah.lib.atoma_device_count() < 1:
        pytest.fail("a test marked `gpu` ran without a visible HIP device")
    ah.set_device(0)
    return ah
