import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu():
    """torch + the HIP library on cuda:0; fails (not skips) if the library is missing."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container")
    from rtx_nerf_amd import _lib
    _lib.lib()
    torch.cuda.set_device(0)
    return torch
