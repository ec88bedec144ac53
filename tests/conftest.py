import os
import sys

# the loss networks refuse to run without weights (as the reference does); every test loads a seeded synthetic state dict
os.environ.setdefault("E4S_ALLOW_UNINITIALIZED_LOSS_NETS", "1")

import pytest  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    has_gpu = torch.cuda.is_available()
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU visible"))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
    return load


def unz(z):
    """(zlib bytes, shape) of a uint8 fixture (tests/golden/make_golden.py:_z) -> uint8 tensor."""
    import zlib
    import numpy as np
    return torch.from_numpy(np.frombuffer(zlib.decompress(z[0]), dtype=np.uint8).reshape(z[1]).copy())
