import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


MLP_MODES = ("f32", "split6")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "own_arithmetic: the test sets the MLP arithmetic itself (runs once, from the fp32 default)")


def pytest_generate_tests(metafunc):
    """EVERY gpu-marked test runs twice: with the fused inference MLPs on the fp32-MFMA kernels ("f32") and on the split-bf16
    six-term kernels ("split6", what PRCNN_MLP_SPLIT=6 selects) -- the parity suite qualifies both arithmetics.  Tests that do
    not reach an inference MLP are simply run twice; tests marked own_arithmetic (they switch it themselves) and CPU tests once.
    PRCNN_TEST_MLP_MODES=f32 (or split6) restricts the run."""
    if "mlp_mode" not in metafunc.fixturenames:
        return
    gpu = metafunc.definition.get_closest_marker("gpu") is not None
    own = metafunc.definition.get_closest_marker("own_arithmetic") is not None
    modes = MLP_MODES if (gpu and not own) else ("f32",)
    want = os.environ.get("PRCNN_TEST_MLP_MODES")
    if want and gpu and not own:
        modes = tuple(m for m in MLP_MODES if m in want.split(","))
    metafunc.parametrize("mlp_mode", modes, indirect=True)


@pytest.fixture(autouse=True)
def mlp_mode(request):
    """sets pointrcnn_amd.ops.MLP_SPLIT_TERMS for the duration of the test; "f32" / "split6" """
    mode = getattr(request, "param", "f32")
    if request.node.get_closest_marker("gpu") is None:
        yield mode
        return
    from pointrcnn_amd import ops
    old = ops.MLP_SPLIT_TERMS
    ops.MLP_SPLIT_TERMS = 6 if mode == "split6" else 0
    try:
        yield mode
    finally:
        ops.MLP_SPLIT_TERMS = old


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
