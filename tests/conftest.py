import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """On a GPU box, bring torch's HIP runtime up before any test loads the library: torch's
    wheel carries its own copy of the runtime, and initialising it late in a process where
    the library's has been working for a while was seen to fail ("No HIP GPUs are
    available").  A few GPU tests use torch as the allocator of caller-owned device buffers."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
