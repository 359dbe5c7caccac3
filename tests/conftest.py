import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    os.environ.setdefault("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1")      # a GPU fault should fail the test, not write a multi-GB core dump first
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except Exception:
        pass
    # The torch wheel bundles its own HIP / HSA runtime and libsvx.so links the system one: whichever is initialised FIRST in a process is the
    # one that sees the GPU.  Tests that use torch device tensors beside libsvx need torch to be first (bench.py imports torch first, too).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as om
    om.build()
    return om.Oracle()
