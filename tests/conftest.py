import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    return torch.load(os.path.join(GOLDEN, "golden_unet32.pt"), weights_only=True)


@pytest.fixture(scope="session")
def ckpt():
    return torch.load(os.path.join(GOLDEN, "unet32_seed42.pth"), weights_only=True)


@pytest.fixture(scope="session")
def sd(ckpt):
    return ckpt["model_state_dict"]
