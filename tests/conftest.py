import os
import sys

import pytest

# Tests that place up to 8 tensor-parallel ranks on ONE device from one process drive 8 streams whose kernels wait for each other on the
# device (the direct all-reduce): each stream needs a hardware queue of its own.  The HIP runtime multiplexes streams over 4 queues by
# default and a queue starts its packets in order, so rank 0's all-reduce would wait for a rank-4 kernel queued behind it
# (tools/probes/world8_queues_probe.py).  Read once, when the runtime starts -- hence here, before anything loads it.  One rank per
# device (the product's deployment: one host thread or process per GPU) never needs this.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

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


@pytest.fixture(autouse=True)
def _forget_decode_hint(request):
    """atoma_prepare_inputs leaves a per-device hint about its batch's lengths that steers ONE dispatch choice of later decode calls (atoma_hip.h,
    atoma_hint_decode_lengths): tests that assert a kernel's name must not inherit the hint of whichever test packed a batch before them."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import atoma_hip
        atoma_hip.lib.atoma_hint_decode_lengths(0, 0, 0)


@pytest.fixture(scope="session")
def gpu(ah):
    if ah.lib.atoma_device_count() < 1:
        pytest.fail("a test marked `gpu` ran without a visible HIP device")
    ah.set_device(0)
    return ah
