import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library; GPU tests fail loudly (never skip) if it is missing."""
    import torch
    from ddnm_amd import _lib
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    return _lib.lib()


@pytest.fixture(autouse=True)
def _main_stays_in_process(monkeypatch):
    """`main.main([...])` called in-process must not re-execute itself as one rank per visible GPU on a multi-GPU host
    (ADVICE r5); the tests of the self-launch set DDNM_GPUS themselves, in the environment of the child they start."""
    if "DDNM_GPUS" not in os.environ:
        monkeypatch.setenv("DDNM_GPUS", "1")
