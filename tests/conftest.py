import os
import sys

import pytest

# the HIP runtime's queue-error handler aborts the process from a runtime thread and says why only at log level >= 1 (errors only):
# a GPU-side fault in a test run must not be a silent "Fatal Python error: Aborted" (set before anything initialises the runtime)
os.environ.setdefault("AMD_LOG_LEVEL", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


@pytest.fixture(scope="session", autouse=True)
def _reserve_side_streams():
    """on a GPU box: the side streams of the trainers' two- / three-stream schedules are made before any other test touches the
    GPU (hardware-queue placement follows creation order: nas_3d_unet_amd.train.reserve_side_streams)"""
    try:
        import torch
        if torch.cuda.is_available():
            from nas_3d_unet_amd.train import reserve_side_streams
            reserve_side_streams(torch.device("cuda", torch.cuda.current_device()))
    except Exception:
        pass
    yield
