import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; when selected without a device, fail loudly instead
    # of skipping (a silent skip would read as "parity green").
    pass


@pytest.fixture(scope="session")
def cuda():
    import torch

    assert torch.cuda.is_available(), "GPU test selected but torch.cuda.is_available() is False"
    from handobjectconsist_amd import _lib

    assert _lib.load().mr_device_ok() == 1, "libmeshraster_hip.so sees no gfx950 device"
    return torch.device("cuda:0")
