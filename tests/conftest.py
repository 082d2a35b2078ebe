import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def avt():
    import avtex

    return avtex


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import avtex

    name = avtex.ops.device_check()  # raises loudly if the HIP library is missing / wrong arch
    assert name.startswith("gfx950")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _no_leaked_solver_search():
    """main.main() turns torch.backends.cudnn.benchmark on (the reference does, main.py:422); left on, every later test
    with a new convolution shape pays MIOpen's exhaustive solver search (one at-size training test: 5 minutes)."""
    import torch

    torch.backends.cudnn.benchmark = False
    yield
    torch.backends.cudnn.benchmark = False
