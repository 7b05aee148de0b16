import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle runs (minutes)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle runs tiny networks on small tensors in most tests: on a many-core GPU host (256 cores) torch's default
    intra-op pool spends its time synchronising 256 threads (the sampler fuzz: 140 s at the default, 18 s at 8 threads).  Sixteen
    threads keep the full-size oracle forwards at a few seconds."""
    import torch
    n = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1, n)))
    yield
    torch.set_num_threads(n)
