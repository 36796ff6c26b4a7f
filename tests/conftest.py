import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracles are eager ATen on chunks of ~1000 samples: with the GPU box's 128 intra-op threads they run 4 x
    # SLOWER than with 8 - 16 (measured: 88 s against ~20 s for one fp32 student gradient).  The suite runs them on 16.
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
