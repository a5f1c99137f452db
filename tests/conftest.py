import os
import sys

import pytest
import torch  # noqa: F401  -- first: torch brings its own HIP runtime, which must be initialised before libl3dpp_hip.so
#                       loads the system one (otherwise torch.cuda reports "No HIP GPUs are available" later on)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    return oracle.lib()
