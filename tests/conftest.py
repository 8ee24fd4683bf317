import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~30 s")
    # CPU oracle runs: as many torch threads as this process may really use (a GPU box exposes hundreds of hardware threads behind
    # a 16-core quota; torch's default would oversubscribe 20-fold)
    import torch
    from universal_speech_enhancement_amd.testing.cpu import usable_cores
    torch.set_num_threads(usable_cores())


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
