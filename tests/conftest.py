import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cpu():
    """the CPU oracle (test infrastructure)"""
    import oracle
    return oracle.cpu()


@pytest.fixture(scope="session")
def ref():
    """the reference's own sources compiled for the host, or skip"""
    import oracle
    r = oracle.ref()
    if r is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    return r


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pointrcnn_amd
    pointrcnn_amd.install()
    return torch.device("cuda:0")
