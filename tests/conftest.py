import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    """||a-b|| / ||b|| in float64."""
    import torch
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
